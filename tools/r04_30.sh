#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ortho_deband.py tests/test_gpu_fullsize.py tests/test_gpu_default_kernels.py -q -m gpu -k "deband or cfg5 or linear_light or hdr_downscale" 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --workload ewa_8k_to_4k_deband_tonemap --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:22]: v for k, v in r['passes_us'].items()})"; done 2>&1 | tee gpurun_out/r04_30_deband.txt
