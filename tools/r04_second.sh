#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_default_kernels.py tests/test_gpu_clear.py tests/test_gpu_multigpu.py tests/test_gpu_c_abi.py tests/test_gpu_dither.py::test_white_noise_plane_covers_the_padded_rows -q -s --tb=short -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids\|Gloo\|c10d" | tail -150 > gpurun_out/r04_02_newtests.log
tail -60 gpurun_out/r04_02_newtests.log
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r04_02_gputests.log
tail -12 gpurun_out/r04_02_gputests.log
