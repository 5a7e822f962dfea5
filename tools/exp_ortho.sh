#!/bin/bash
cd "$(dirname "$0")/.."
one() { python bench.py --workload lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['passes_us'])"; }
echo -n "generic: "; PL_HIP_ORTHO_FAST=0 one
echo -n "fast:    "; one
