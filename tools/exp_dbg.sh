#!/bin/bash
cd "$(dirname "$0")/.."
one() { python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(r['kernel_us'])"; }
export PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_b.so
for b in 0 1 128 256 384 32 33 8 24 64 4; do echo -n "debug=$b: "; PL_HIP_PP_DEBUG=$b one; done
