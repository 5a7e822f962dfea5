#!/bin/bash
# round-5 collection (tools/r04_collect.sh with the HQ workloads added): the default bench line, the same command under rocprofv3 --kernel-trace --stats
# (without the concurrent-stream / async companions, whose launches overlap), per-workload kernel stats,
# and the GPU test log. Outputs under gpurun_out/<tag>_*. Usage: tools/r03_collect.sh <tag> [bench] [driver] [trace] [stats] [tests]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1; shift
what=${@:-"bench trace stats tests"}
for w in $what; do case $w in
bench)
  timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
  tail -c 300 gpurun_out/${tag}_bench.err ;;
driver)
  # the driver's own command, three times (the first one pays the cold start)
  for i in 1 2 3; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-cpu-baseline 2>/dev/null | tail -1; done > gpurun_out/${tag}_driver_cmd.jsonl
  python3 -c "
import json,sys
for l in open('gpurun_out/${tag}_driver_cmd.jsonl'):
    d=json.loads(l); print(d['value'], d.get('ms_per_frame', d['ms_per_step']), d['roofline'].get('kernel_us'))" ;;
trace)
  out=/tmp/prof_default; rm -rf $out
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --no-traffic --no-cpu-baseline --no-concurrent > $GRAFT_REPO_ROOT/gpurun_out/${tag}_default_bench_under_rocprof.json 2> /tmp/prof_default.err)
  find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_default_bench_kernel_stats.csv \;
  head -4 gpurun_out/${tag}_default_bench_kernel_stats.csv | cut -c1-140 ;;
stats)
  for wl in ewa_1080p_to_4k_hdr_tonemap ewa_lanczos_1080p_to_4k_dither10 bilinear_1080p_to_4k hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap lanczos_1080p_to_4k_dither10 ewa_lanczos_720p_to_4k_dither10 ewa_lanczos_540p_to_4k_dither10 ewa_lanczos_1440p_to_4k_dither10 ewa_720p_to_4k_hdr_tonemap ewa_lanczos_4k_to_1080p_dither10 ewa_lanczos_4k_to_1080p_linear_dither10 ewa_8k_to_4k_hdr_tonemap default_preset_1080p_to_4k default_preset_ewa_1080p_to_4k hdr10_4k_tonemap_high_quality high_quality_preset_1080p_to_4k default_preset_4k_to_1080p; do
    out=/tmp/st_$wl; rm -rf $out
    # one stream: every kernel's duration is its own (what bench.py's event times and "trace" report)
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --async-measure 0 --workload $wl > /tmp/st_$wl.log 2>&1)
    find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${wl}_kernel_stats.csv \;
    case $wl in ewa_1080p_to_4k_hdr_tonemap|hdr10_4k_tonemap)
      # the library default: the measuring pass beside the previous frame's long pass (both stretch, the frame shrinks)
      rm -rf $out
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --workload $wl > /tmp/st_$wl.log 2>&1)
      find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${wl}_async_kernel_stats.csv \; ;;
    esac
  done ;;
tests)
  timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/${tag}_gputests.log
  cat gpurun_out/${tag}_gputests.log ;;
esac; done
