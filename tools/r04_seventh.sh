#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/r04_ab.sh r04_07 ewa_1080p_to_4k_hdr_tonemap ewa_lanczos_1080p_to_4k_dither10 ewa_8k_to_4k_deband_tonemap
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/r04_07_gputests.log; echo "rc full = ${PIPESTATUS[0]}" >> gpurun_out/r04_07_gputests.log
tail -30 gpurun_out/r04_07_gputests.log
