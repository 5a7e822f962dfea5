#!/bin/bash
# round 6, third GPU call: chain-vs-interpreter diagnostic, the reference's own gpu_tests.c and
# bench.c on the backend
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_03
timeout 300 python tools/r06_03_diag.py 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 | tee gpurun_out/${tag}_diag.txt
echo "== the same with the round-5 library" | tee -a gpurun_out/${tag}_diag.txt
PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_r05.so timeout 300 python tools/r06_03_diag.py 2>&1 | tail -12 | tee -a gpurun_out/${tag}_diag.txt
for t in buffer texture planar shader scaler render ycbcr; do
  echo "== ref_gpu_tests $t" | tee -a gpurun_out/${tag}_ref_tests.txt
  timeout 600 oracle/_ref/ref_gpu_tests $t > gpurun_out/${tag}_ref_${t}.out 2> gpurun_out/${tag}_ref_${t}.err
  echo "rc=$?" | tee -a gpurun_out/${tag}_ref_tests.txt
  tail -3 gpurun_out/${tag}_ref_${t}.out | cut -c1-300 | tee -a gpurun_out/${tag}_ref_tests.txt
  grep -A8 "FAILED" gpurun_out/${tag}_ref_${t}.err | head -24 | cut -c1-300 | tee -a gpurun_out/${tag}_ref_tests.txt
done
timeout 900 oracle/_ref/ref_bench > gpurun_out/${tag}_ref_bench.txt 2> gpurun_out/${tag}_ref_bench.err
echo "ref_bench rc=$?"; cat gpurun_out/${tag}_ref_bench.txt | cut -c1-200; tail -5 gpurun_out/${tag}_ref_bench.err | cut -c1-300
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/${tag}_gputests.log
tail -12 gpurun_out/${tag}_gputests.log | cut -c1-400
