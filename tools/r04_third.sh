#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/r04_ab.sh r04_03
for f in tests/test_gpu_multigpu.py "tests/test_gpu_clear.py tests/test_gpu_c_abi.py tests/test_gpu_dither.py"; do
  timeout 600 python -m pytest $f -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\|c10d" | tail -4; echo "rc of [$f] = ${PIPESTATUS[0]}"
done 2>&1 | tee gpurun_out/r04_03_exitcheck.log
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/r04_03_gputests.log; echo "rc full = ${PIPESTATUS[0]}" >> gpurun_out/r04_03_gputests.log
tail -25 gpurun_out/r04_03_gputests.log
