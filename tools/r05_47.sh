#!/bin/bash
# the round's last collection: GPU suite, smoke, the default bench line, the driver's command, the
# default command under rocprofv3, kernel stats of the workloads added this round
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r05_47
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/${tag}_gputests.log
cat gpurun_out/${tag}_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.txt
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 1500 gpurun_out/${tag}_bench.json | cut -c1-600
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_driver_cmd.jsonl
python3 -c "
import json
for l in open('gpurun_out/${tag}_driver_cmd.jsonl'):
    d=json.loads(l); print('driver cmd:', d['value'], d.get('ms_per_frame', d['ms_per_step']), d['roofline'].get('kernel_us'), d['roofline'].get('frac'))"
out=/tmp/prof_default; rm -rf $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline --no-concurrent > $GRAFT_REPO_ROOT/gpurun_out/${tag}_default_bench_under_rocprof.json 2> /tmp/prof_default.err)
find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_default_bench_kernel_stats.csv \;
head -4 gpurun_out/${tag}_default_bench_kernel_stats.csv | cut -c1-160
for wl in ewa_1080p_to_4k_hdr_tonemap_subtitles nv12_1080i_to_4k_bwdif_default_preset; do
  out=/tmp/st_$wl; rm -rf $out
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --async-measure 0 --workload $wl > /tmp/st_$wl.log 2>&1)
  find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${wl}_kernel_stats.csv \;
  head -6 gpurun_out/${tag}_${wl}_kernel_stats.csv | cut -c1-140
  python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$wl', d.get('ms_per_frame', d['ms_per_step']), {k[:34]: v for k, v in r['passes_us'].items()})" | tee -a gpurun_out/${tag}_new_workloads.txt
done
