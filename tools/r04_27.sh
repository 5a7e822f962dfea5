#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_default_kernels.py -q -s -m gpu -k "linear_light or hdr_downscale" 2>&1 | grep "linear-light\|oracle (libm\|HDR downscale:\|passed\|failed\|^E  " | cut -c1-300
for wl in ewa_lanczos_4k_to_1080p_linear_dither10 ewa_8k_to_4k_hdr_tonemap; do
  for m in 1 1; do echo -n "$wl mfma=$m "; PL_HIP_POLAR_MFMA=$m timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], {k[:22]: v for k, v in r['passes_us'].items()})"; done
done 2>&1 | tee gpurun_out/r04_27_ab_mxd_linear.txt
