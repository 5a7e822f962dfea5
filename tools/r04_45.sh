#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_default_kernels.py tests/test_gpu_kernel_variants.py tests/test_gpu_renderer.py tests/test_gpu_metric.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | cut -c1-250 | tail
one() { timeout 300 python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:34]: v for k, v in r['passes_us'].items()})"; }
(echo "pl_render_default_params + ewa_lanczos, SDR 1080p -> 4K, 10-bit dither (bench.py --workload default_preset_ewa_1080p_to_4k)"
for v in default 0 1 default 0; do
  if [ $v = default ]; then echo -n "tree default (two passes for an upscale with non-lite pending ops): "; one default_preset_ewa_1080p_to_4k
  else echo -n "PL_HIP_NO_FUSION=$v: "; PL_HIP_NO_FUSION=$v one default_preset_ewa_1080p_to_4k; fi
done
echo -n "metric (lite pending ops: stays fused): "; one ewa_1080p_to_4k_hdr_tonemap) 2>&1 | tee gpurun_out/r04_44_default_preset_ewa_fusion.txt
