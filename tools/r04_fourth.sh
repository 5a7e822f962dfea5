#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/r04_ramp.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_04_ramp.txt
bash tools/r04_ab.sh r04_04 ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/r04_04_gputests.log; echo "rc full = ${PIPESTATUS[0]}" >> gpurun_out/r04_04_gputests.log
tail -15 gpurun_out/r04_04_gputests.log
