#!/bin/bash
# round 6, fourth GPU call: per-sample colour-map statistics at full size (report only), the
# reference's pl_shader_tests, deband decision test, 8-rank bench test, the whole suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_04
PL_PARITY_REPORT_ONLY=1 timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_metric.py tests/test_gpu_renderer.py tests/test_gpu_color.py tests/test_gpu_contrast_recovery.py -q -m gpu -s 2>&1 | grep -A1 "per sample\|passed\|failed" | cut -c1-500 > gpurun_out/${tag}_per_sample.txt
tail -40 gpurun_out/${tag}_per_sample.txt
timeout 600 oracle/_ref/ref_gpu_tests shader > gpurun_out/${tag}_ref_shader.out 2> gpurun_out/${tag}_ref_shader.err; echo "ref shader rc=$?"
tail -4 gpurun_out/${tag}_ref_shader.out | cut -c1-300; grep -A8 "FAILED" gpurun_out/${tag}_ref_shader.err | head -20 | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_ortho_deband.py tests/test_gpu_kernel_variants.py tests/test_gpu_reference_tests.py -q -m gpu 2>&1 | tail -15 | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_multigpu.py -q -m gpu 2>&1 | tail -15 | cut -c1-400
BASE=r05 NODRIVER=1 STEPS=100 bash tools/r05_ab.sh ${tag}_ab ewa_8k_to_4k_deband_tonemap
