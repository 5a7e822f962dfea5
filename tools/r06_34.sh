#!/bin/bash
# what k_pass_chain (cfg 4) waits for: texture addresser / L1 / L2 counters, one --pmc pass per group (each under a timeout:
# the first attempt's six-counter TA group made rocprofv3 abort and hang in its finaliser for the call's 30 minutes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export BENCH_ARGS="--bare --workload hdr10_4k_tonemap --async-measure 0"
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "TD_TD_BUSY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 90 bash tools/pmc.sh r06_34_$i $set 2>&1 | grep -A5 "k_pass_chain" | head -6
done | tee gpurun_out/r06_34_chain_counters.txt
