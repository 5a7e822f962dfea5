#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r05_48_yadif_rows.txt
timeout 300 python -m pytest tests/test_gpu_deinterlace.py -q -m gpu -x 2>&1 | grep "^FAILED\|passed\|failed\|^E  \|Error" | cut -c1-400 | tail -12 | tee $out
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), {k[:34]: v for k, v in r['passes_us'].items()})"; }
echo -n "yadif rows: " | tee -a $out; one nv12_1080i_to_4k_yadif_default_preset 2>&1 | tee -a $out
echo -n "yadif general: " | tee -a $out; PL_HIP_DEINT_ROWS=0 one nv12_1080i_to_4k_yadif_default_preset 2>&1 | tee -a $out
