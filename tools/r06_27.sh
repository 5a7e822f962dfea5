#!/bin/bash
# what bounds k_pass_chain (cfg 4): wait / memory counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export BENCH_ARGS="--bare --workload hdr10_4k_tonemap --async-measure 0"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  bash tools/pmc.sh r06_27 $set 2>&1 | grep -A8 "k_pass_chain" | head -9
done | tee gpurun_out/r06_27_chain_counters.txt
