#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nv -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --async-measure 0 --workload nv12_1080p_to_4k_ewa_dither10 > /tmp/nv.log 2>&1
find /tmp/nv -name "*kernel_stats.csv" -exec head -8 {} \; | cut -c1-200
cd $GRAFT_REPO_ROOT; PL_HIP_PASS_TRACE=1 timeout 120 python bench.py --workload nv12_1080p_to_4k_ewa_dither10 --steps 1 --warmup 1 --bare 2>&1 | grep "plh\] pass\|matrix\|polar" | sort | uniq -c | sort -rn | head -8 | cut -c1-300
