#!/bin/bash
# the last validation of the round: the GPU suite (default and on the library's default polar kernels), smoke, the bench line and the driver's command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1
filt() { grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-220 | tail -12; }
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | filt > gpurun_out/${tag}_gputests.log; echo "rc=${PIPESTATUS[0]}" >> gpurun_out/${tag}_gputests.log; cat gpurun_out/${tag}_gputests.log
PL_HIP_POLAR_MFMA=1 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | filt > gpurun_out/${tag}_gputests_mfma_forced.log; cat gpurun_out/${tag}_gputests_mfma_forced.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.txt
bash tools/r04_collect.sh $tag bench driver
