#!/bin/bash
# round 6, ninth GPU call: non-temporal stores that are really emitted (k_polar_mx / mxr / mxd /
# mxp, k_pass_chain / native / merge / mix); the round so far against the round-5 library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_09
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -6 gpurun_out/${tag}_gputests.log | cut -c1-300
BASE=r05 NODRIVER=1 STEPS=200 bash tools/r05_ab.sh ${tag}_round_ab ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap ewa_lanczos_1080p_to_4k_dither10 bilinear_1080p_to_4k ewa_8k_to_4k_deband_tonemap default_preset_1080p_to_4k default_preset_4k_to_1080p nv12_1080p_to_4k_default_preset mix_24_to_60_ewa_1080p_to_4k ewa_720p_to_4k_hdr_tonemap hdr10_4k_tonemap_high_quality
for wl in ewa_lanczos_1080p_to_4k_dither10 ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap; do
  out=/tmp/st_$wl; rm -rf $out
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --async-measure 0 --workload $wl > /tmp/st.log 2>&1)
  find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_${wl}_kernel_stats.csv \;
  head -4 gpurun_out/${tag}_${wl}_kernel_stats.csv | cut -c1-150
done
for i in 1 2; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-traffic --no-companions --no-concurrent --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver cmd:', d['value'], d.get('ms_per_frame'), d.get('one_frame_per_step'))"; done | tee gpurun_out/${tag}_driver.txt
