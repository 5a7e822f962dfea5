#!/bin/bash
# A/B on one box: A = build_ab/libplacebo_hip_base.so (the tree before the change), B = the in-tree library.
#   tools/r04_ab.sh <tag> [workload ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1; shift
wls=${@:-"ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap ewa_lanczos_1080p_to_4k_dither10"}
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in $wls; do
  echo "== $wl" | tee -a gpurun_out/${tag}_ab.txt
  for v in A B B A; do
    if [ $v = A ]; then echo -n "A(base): "; PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_base.so one $wl
    else echo -n "B(tree): "; one $wl; fi
  done 2>&1 | tee -a gpurun_out/${tag}_ab.txt
done
# the driver's own short command
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5:', d['value'], d['ms_per_step'])"; done | tee -a gpurun_out/${tag}_ab.txt
PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_base.so python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5 (base lib):', d['value'], d['ms_per_step'])" | tee -a gpurun_out/${tag}_ab.txt
