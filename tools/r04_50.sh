#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sampling.py tests/test_gpu_renderer.py tests/test_gpu_kernel_variants.py tests/test_gpu_fullsize.py tests/test_gpu_rotation.py tests/test_gpu_mix.py tests/test_gpu_polar_mfma.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | cut -c1-250 | tail
one() { timeout 300 python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:34]: v for k, v in r['passes_us'].items()})"; }
for wl in nv12_1080p_to_4k_ewa_dither10 mix_24_to_60_ewa_1080p_to_4k; do echo -n "$wl: "; one $wl; done
echo -n "cfg3 on k_polar_pp (PL_HIP_POLAR_MFMA=0): "; PL_HIP_POLAR_MFMA=0 one ewa_lanczos_1080p_to_4k_dither10
