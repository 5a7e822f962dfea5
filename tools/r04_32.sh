#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ortho_deband.py -q -m gpu -k "deband" 2>&1 | tail -3
for m in 1 1; do echo -n "lds=$m "; PL_HIP_DEBAND_LDS=$m timeout 300 python bench.py --workload ewa_8k_to_4k_deband_tonemap --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:22]: v for k, v in r['passes_us'].items()})"; done 2>&1 | tee gpurun_out/r04_32_deband_lds_np4.txt
