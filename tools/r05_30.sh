#!/bin/bash
cd $GRAFT_REPO_ROOT
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'])"; }
for d in 0 16 4 20 0 16; do echo -n "PP_DEBUG=$d: "; PL_HIP_PP_DEBUG=$d one ewa_lanczos_1080p_to_4k_dither10; done
