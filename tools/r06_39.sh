#!/bin/bash
# the suite on the alternative paths: one stream, the PQ pair's closed forms, the op interpreter instead of the map chain
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for e in "PL_HIP_ASYNC_MEASURE=0" "PL_HIP_PQ_SEGMENTS=0" "PL_HIP_LOWPASS_FUSED=0 PL_HIP_CHAIN_TONE_LDS=0"; do
  echo "== $e" | tee -a gpurun_out/r06_39_alt_suites.txt
  env $e timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 | cut -c1-300 | tee -a gpurun_out/r06_39_alt_suites.txt
done
