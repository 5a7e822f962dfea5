#!/bin/bash
# round-end collection: the default bench line, the same command under rocprofv3 --kernel-trace --stats
# (without the concurrent-stream / async companions, whose launches overlap), and the GPU test log.
# Outputs under gpurun_out/r02_06_*. Usage: tools/r02_final.sh [bench] [trace] [tests]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
what=${@:-"bench trace tests"}
for w in $what; do case $w in
bench)
  timeout 900 python bench.py > gpurun_out/r02_06_bench.json 2> gpurun_out/r02_06_bench.err
  tail -c 600 gpurun_out/r02_06_bench.err ;;
trace)
  out=/tmp/prof_default; rm -rf $out
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline --no-concurrent > $GRAFT_REPO_ROOT/gpurun_out/r02_06_default_bench_under_rocprof.json 2> /tmp/prof_default.err)
  find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_06_default_bench_kernel_stats.csv \;
  head -6 gpurun_out/r02_06_default_bench_kernel_stats.csv | cut -c1-160 ;;
tests)
  timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r02_06_gputests.log
  cat gpurun_out/r02_06_gputests.log ;;
esac; done
