#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mix.py tests/test_gpu_queue.py tests/test_gpu_renderer.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | cut -c1-250 | tail
one() { timeout 300 python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:34]: v for k, v in r['passes_us'].items()})"; }
for v in 1 0 1 0; do echo -n "mix_24_to_60_ewa_1080p_to_4k PL_HIP_PASS_NATIVE=$v: "; PL_HIP_PASS_NATIVE=$v one mix_24_to_60_ewa_1080p_to_4k; done 2>&1 | tee gpurun_out/r04_52_mix_kernel.txt
