#!/bin/bash
# round-4 final validation: the GPU suite in its four modes, smoke(), then the collection
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1
filt() { grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-220 | tail -12; }
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | filt > gpurun_out/${tag}_gputests.log; echo "rc=${PIPESTATUS[0]}" >> gpurun_out/${tag}_gputests.log; cat gpurun_out/${tag}_gputests.log
PL_HIP_POLAR_MFMA=1 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | filt > gpurun_out/${tag}_gputests_mfma_forced.log; cat gpurun_out/${tag}_gputests_mfma_forced.log
PL_HIP_MAP_CHAIN=0 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | filt > gpurun_out/${tag}_gputests_chain_off.log; cat gpurun_out/${tag}_gputests_chain_off.log
PL_HIP_ASYNC_MEASURE=0 timeout 1500 python -m pytest tests -q -m gpu 2>&1 | filt > gpurun_out/${tag}_gputests_one_stream.log; cat gpurun_out/${tag}_gputests_one_stream.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3 | tee gpurun_out/${tag}_smoke.txt
