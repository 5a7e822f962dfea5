#!/bin/bash
# what the queue entries between two scaler launches cost: the tone curve's upload (copy + event), the frame-end event
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r05_08_queue_entries.txt
: > $out
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'])"; }
for wl in ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap; do
  echo "== $wl" | tee -a $out
  for rep in 1 2; do
  echo -n "as is:            " | tee -a $out; one $wl 2>&1 | tee -a $out
  echo -n "no tone upload:   " | tee -a $out; PL_HIP_DBG_NO_TONE_UPLOAD=1 one $wl 2>&1 | tee -a $out
  echo -n "no frame fence:   " | tee -a $out; PL_HIP_DBG_NO_FRAME_FENCE=1 one $wl 2>&1 | tee -a $out
  echo -n "neither:          " | tee -a $out; PL_HIP_DBG_NO_TONE_UPLOAD=1 PL_HIP_DBG_NO_FRAME_FENCE=1 one $wl 2>&1 | tee -a $out
  echo -n "one stream as is: " | tee -a $out; PL_HIP_ASYNC_MEASURE=0 one $wl 2>&1 | tee -a $out
  echo -n "one stream no up: " | tee -a $out; PL_HIP_ASYNC_MEASURE=0 PL_HIP_DBG_NO_TONE_UPLOAD=1 one $wl 2>&1 | tee -a $out
  done
done
