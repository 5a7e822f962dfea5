#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ortho_deband.py -q -m gpu -k "deband" 2>&1 | tail -3
one() { timeout 300 python bench.py --workload ewa_8k_to_4k_deband_tonemap --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:12]: v for k, v in r['passes_us'].items()})"; }
for v in A C C A; do
  case $v in A) echo -n "A(tree): "; one ;; C) echo -n "C(64x64, per-texel staging): "; PL_HIP_LIB=$PWD/build_ab2/libplacebo_hip_c.so one ;; esac
done 2>&1 | tee gpurun_out/r04_36_deband_staging.txt
