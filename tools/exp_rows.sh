#!/bin/bash
cd "$(dirname "$0")/.."
one() { python bench.py "$@" --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], r['frac'])"; }
for r in 1 2 3 4 5 6 8; do echo -n "rows=$r cfg3: "; PL_HIP_PP_ROWS=$r one; done
for r in 1 2 3 4 6 8; do echo -n "rows=$r 8k: "; PL_HIP_PP_ROWS=$r python bench.py --workload ewa_8k_to_4k_deband_tonemap --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_us'])"; done
