#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PL_HIP_PASS_TRACE=1 timeout 120 python bench.py --workload mix_24_to_60_ewa_1080p_to_4k --steps 5 --warmup 1 --bare 2>&1 | grep "plh\] pass" | sort | uniq -c | sort -rn | head -8 | cut -c1-300
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mixp -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --workload mix_24_to_60_ewa_1080p_to_4k > /tmp/mixp.log 2>&1
find /tmp/mixp -name "*kernel_stats.csv" -exec head -6 {} \; | cut -c1-200
