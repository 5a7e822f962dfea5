#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ortho_deband.py tests/test_gpu_kernel_variants.py tests/test_gpu_contrast_recovery.py tests/test_gpu_renderer.py tests/test_gpu_fullsize.py tests/test_gpu_color.py tests/test_gpu_metric.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | cut -c1-250 | tail
one() { timeout 300 python bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:44]: v for k, v in r['passes_us'].items()})"; }
(for wl in hdr10_4k_tonemap_high_quality default_preset_4k_to_1080p; do
  echo -n "$wl: "; one $wl
  echo -n "$wl, PL_HIP_PEAK_FAST=0 PL_HIP_ORTHO_FAST=0: "; PL_HIP_PEAK_FAST=0 PL_HIP_ORTHO_FAST=0 one $wl
done) 2>&1 | tee gpurun_out/r04_54_presets3.txt
