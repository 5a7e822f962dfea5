#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in ewa_lanczos_1440p_to_4k_dither10 ewa_lanczos_720p_to_4k_dither10 ewa_lanczos_540p_to_4k_dither10 ewa_720p_to_4k_hdr_tonemap; do
  for m in 1 0 1 0; do echo -n "$wl mxr=$m "; PL_HIP_POLAR_MXR=$m timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], list(r['passes_us'].values()))"; done
done 2>&1 | tee gpurun_out/r04_22_ab_mxr.txt
