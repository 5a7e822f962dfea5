#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
one() { timeout 300 python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:40]: v for k, v in r['passes_us'].items()})"; }
for wl in high_quality_preset_1080p_to_4k nv12_1080p_to_4k_ewa_dither10 nv12_1080p_to_4k_default_preset mix_24_to_60_ewa_1080p_to_4k; do echo -n "$wl: "; one $wl; done 2>&1 | tee gpurun_out/r04_46_presets.txt
PL_HIP_PASS_TRACE=1 timeout 120 python bench.py --workload high_quality_preset_1080p_to_4k --steps 2 --warmup 1 --bare 2>&1 | grep -i "pass\|kernel" | sort | uniq -c | sort -rn | head -12 | cut -c1-250
