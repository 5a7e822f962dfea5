#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r05_43_deint_rows.txt
timeout 600 python -m pytest tests/test_gpu_deinterlace.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  \|Error" | cut -c1-300 | tail -12 | tee $out
PL_HIP_DEINT_ROWS=0 timeout 600 python -m pytest tests/test_gpu_deinterlace.py -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  \|Error" | cut -c1-300 | tail -12 | tee -a $out
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>gpurun_out/r05_43_$1.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel'][:40], r['kernel_us'], r['frac'], {k[:34]: v for k, v in r['passes_us'].items()})"; }
for wl in nv12_1080p_to_4k_default_preset nv12_1080i_to_4k_bwdif_default_preset; do
  echo -n "$wl: " | tee -a $out; one $wl 2>&1 | tee -a $out
done
echo -n "general kernel: " | tee -a $out; PL_HIP_DEINT_ROWS=0 one nv12_1080i_to_4k_bwdif_default_preset 2>&1 | tee -a $out
