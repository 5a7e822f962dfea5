#!/bin/bash
# overlays: the new tests, then the metric's frame with and without two lines of subtitles
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r05_40_subtitles.txt
timeout 300 python -m pytest tests/test_gpu_overlay.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  \|Error" | cut -c1-300 | tail -12 | tee $out
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>gpurun_out/r05_40_$1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_1080p_to_4k_hdr_tonemap ewa_1080p_to_4k_hdr_tonemap_subtitles ewa_1080p_to_4k_hdr_tonemap_subtitles ewa_1080p_to_4k_hdr_tonemap; do
  echo -n "$wl: " | tee -a $out; one $wl 2>&1 | tee -a $out
done
