#!/bin/bash
# k_deband_lds tile shapes, A/B in one call: A = tree (64x32 tiles, 2 px per lane at a time, 3 workgroups per CU),
# B = 64x64 / 4 px, C = 64x64 / 2 px (2 workgroups per CU)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
one() { timeout 300 python bench.py --workload ewa_8k_to_4k_deband_tonemap --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:12]: v for k, v in r['passes_us'].items()})"; }
for v in A B C C B A; do
  case $v in A) echo -n "A: "; one ;; B) echo -n "B: "; PL_HIP_LIB=$PWD/build_ab2/libplacebo_hip_b.so one ;; C) echo -n "C: "; PL_HIP_LIB=$PWD/build_ab2/libplacebo_hip_c.so one ;; esac
done 2>&1 | tee gpurun_out/r04_35_deband_shapes.txt
