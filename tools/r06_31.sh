#!/bin/bash
# round 6, the final tree: smoke, the whole suite, the default bench command under the kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_31
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -3 gpurun_out/${tag}_gputests.log | cut -c1-300
out=/tmp/tr31; rm -rf $out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_under_trace.json 2> /tmp/st.log)
find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_default_bench_kernel_stats.csv \;
head -8 gpurun_out/${tag}_default_bench_kernel_stats.csv | cut -c1-170
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_driver.json; python -c "import json; d=json.loads(open('gpurun_out/${tag}_driver.json').read()); print('driver command:', d['value'], d['ms_per_frame'], d['roofline']['kernel_us'], d['roofline']['frac'])"
