#!/bin/bash
# L1 -> L2 read requests of the debanding pass: the gather kernel against the LDS-window kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export BENCH_ARGS="--bare --workload ewa_8k_to_4k_deband_tonemap --async-measure 0 --steps 4 --warmup 1"
(echo "== k_deband_lds (default)"; bash tools/pmc.sh r04_58a TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum 2>&1 | grep -A2 "^k_deband\|k_deband_"
echo "== k_deband_fast (PL_HIP_DEBAND_LDS=0)"; PL_HIP_DEBAND_LDS=0 bash tools/pmc.sh r04_58b TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum 2>&1 | grep -A2 "^k_deband\|k_deband_") | tee gpurun_out/r04_58_deband_l2_requests.txt
rm -rf gpurun_out/pmc_r04_58a gpurun_out/pmc_r04_58b
