#!/bin/bash
# the texture path in the metric's launch (k_polar_mx<3,true,3,8>): the same counters as r06_34
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
export BENCH_ARGS="--bare --workload ewa_1080p_to_4k_hdr_tonemap --async-measure 0"
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum" "TD_TD_BUSY_sum TA_BUSY_avr" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 90 bash tools/pmc.sh r06_38_$i $set 2>&1 | grep -A3 "k_polar_mx" | head -4
done | tee gpurun_out/r06_38_mx_texture_path_counters.txt
