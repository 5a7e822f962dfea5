#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ortho_deband.py tests/test_gpu_default_kernels.py tests/test_gpu_contrast_recovery.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | cut -c1-250 | tail
one() { timeout 300 python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:40]: v for k, v in r['passes_us'].items()})"; }
for v in 1 0 1; do echo -n "high_quality_preset_1080p_to_4k PL_HIP_DEBAND_FAST=$v: "; PL_HIP_DEBAND_FAST=$v one high_quality_preset_1080p_to_4k; done 2>&1 | tee gpurun_out/r04_47_hq_preset_sdr.txt
