#!/bin/bash
# the map chain's transfer curves as independent chains under one switch (run_map_chain<.., WIDE>):
# A = the commit before (build_ab/libplacebo_hip_prev.so), B = tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_metric.py tests/test_gpu_default_kernels.py tests/test_gpu_polar_mfma.py -q -m gpu 2>&1 | tail -3
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap default_preset_1080p_to_4k; do
  echo "== $wl"
  for v in A B B A; do
    if [ $v = A ]; then echo -n "A(prev): "; PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_prev.so one $wl
    else echo -n "B(tree): "; one $wl; fi
  done
done 2>&1 | tee gpurun_out/r04_41_ab_wide_chain.txt
