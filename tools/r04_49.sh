#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
one() { timeout 300 python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:34]: v for k, v in r['passes_us'].items()})"; }
for r in 3 2 1 3 1; do echo -n "nv12 ewa, PL_HIP_PP_ROWS=$r: "; PL_HIP_PP_ROWS=$r one nv12_1080p_to_4k_ewa_dither10; done 2>&1 | tee gpurun_out/r04_49_pp_rows_small.txt
