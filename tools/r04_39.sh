#!/bin/bash
# the whole GPU suite with the matrix-pipe polar kernels forced on everywhere (the suite pins
# PL_HIP_POLAR_MFMA=0 by default): which tests see a difference, and is it ever more than the kernels' +-1 code?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
PL_HIP_POLAR_MFMA=1 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | grep "^FAILED\|passed\|failed" | cut -c1-220 > gpurun_out/r04_39_gputests_mfma_forced.log
cat gpurun_out/r04_39_gputests_mfma_forced.log | tail -40
