#!/bin/bash
# the measuring pass beside the scaler: k_peak_tiles / k_peak_fast x priority of the measuring stream x one stream
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r05_04_peak_prio.txt
: > $out
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()}, d.get('single_stream',{}).get('ms_per_step'))"; }
for wl in ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap; do
  echo "== $wl" | tee -a $out
  for rep in 1 2; do
  for tiles in 0 1; do for prio in 0 1 -1; do
    echo -n "tiles=$tiles prio=$prio: " | tee -a $out
    PL_HIP_PEAK_TILES=$tiles PL_HIP_AUX_PRIO=$prio one $wl 2>&1 | tee -a $out
  done; done; done
done
