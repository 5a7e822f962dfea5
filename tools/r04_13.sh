#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_contrast_recovery.py tests/test_gpu_metric.py tests/test_gpu_kernel_variants.py -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04_13_tests.log
for f in 1 0 1 0; do echo -n "fused=$f "; PL_HIP_FUSED_FEATURES=$f python bench.py --workload ewa_8k_to_4k_deband_tonemap --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:24]: v for k, v in r['passes_us'].items()})"; done | tee gpurun_out/r04_13_cfg5.txt
