#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/r04_ab.sh r04_05 ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap
# SQ counters of the metric's launch (dynamic instruction count) and the debug switches
export BENCH_ARGS="--bare --workload ewa_1080p_to_4k_hdr_tonemap --async-measure 0"
bash tools/pmc.sh r04_05_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A5 "k_polar_mx\|k_peak" | tee gpurun_out/r04_05_counters.txt
bash tools/pmc.sh r04_05_b SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | grep -A5 "k_polar_mx\|k_peak" | tee -a gpurun_out/r04_05_counters.txt
for dbg in 0 1 8 13; do echo -n "dbg=$dbg "; PL_HIP_PP_DEBUG=$dbg python bench.py --workload ewa_1080p_to_4k_hdr_tonemap --steps 200 --warmup 20 --async-measure 0 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], list(r['passes_us'].values()))"; done | tee -a gpurun_out/r04_05_counters.txt
timeout 600 python -m pytest tests/test_gpu_metric.py tests/test_gpu_fullsize.py tests/test_gpu_kernel_variants.py tests/test_gpu_color.py tests/test_gpu_contrast_recovery.py -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_05_tests.log
