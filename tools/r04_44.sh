#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
one() { timeout 300 python bench.py --workload default_preset_ewa_1080p_to_4k --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:34]: v for k, v in r['passes_us'].items()})"; }
echo -n "fused: "; one
echo -n "PL_HIP_NO_FUSION=1: "; PL_HIP_NO_FUSION=1 one
echo -n "fused: "; one
echo -n "PL_HIP_NO_FUSION=1: "; PL_HIP_NO_FUSION=1 one
