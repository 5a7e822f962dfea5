#!/bin/bash
# which HIP runtime calls does a frame make? (rocprofv3 --hip-trace --stats over the C frame loop)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=/tmp/api; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --hip-trace --stats --output-format csv -d $out -- $GRAFT_REPO_ROOT/tests/c/build/bench_frames 200 > /tmp/api.log 2>&1)
cat /tmp/api.log | tail -4
f=$(find $out -name "*hip_api_stats.csv" | head -1)
[ -z "$f" ] && f=$(find $out -name "*stats*.csv" | head -1)
echo $f; cut -c1-120 $f | head -30
