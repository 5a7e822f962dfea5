#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for wl in ewa_1080p_to_4k_hdr_tonemap ewa_lanczos_1080p_to_4k_dither10; do
  echo "== $wl"
  for dbg in 0 16 32 48 0 16; do echo -n "dbg=$dbg "; PL_HIP_PP_DEBUG=$dbg python bench.py --workload $wl --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], list(r['passes_us'].values()))"; done
done 2>&1 | tee gpurun_out/r04_06_prio.txt
