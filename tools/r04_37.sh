#!/bin/bash
# L1 / L2 traffic of the colour-map launches: are the LUT gathers pulling whole lines from L2?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
(rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | cut -c1-3000) > gpurun_out/r04_37_counters_list.txt
export BENCH_ARGS="--bare --workload ewa_1080p_to_4k_hdr_tonemap --async-measure 0"
bash tools/pmc.sh r04_37a TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum 2>&1 | grep -A3 "k_polar_mx\|k_pass_chain" | head -20
bash tools/pmc.sh r04_37b TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum 2>&1 | grep -A4 "k_polar_mx\|k_pass_chain" | head -20
export BENCH_ARGS="--bare --workload hdr10_4k_tonemap --async-measure 0"
bash tools/pmc.sh r04_37c TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum 2>&1 | grep -A3 "k_polar_mx\|k_pass_chain" | head -20
