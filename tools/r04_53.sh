#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
one() { timeout 300 python bench.py --workload $1 --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:44]: v for k, v in r['passes_us'].items()})"; }
for wl in hdr10_4k_tonemap_high_quality default_preset_4k_to_1080p nv12_1080p_to_4k_default_preset; do echo -n "$wl: "; one $wl; done 2>&1 | tee gpurun_out/r04_53_presets2.txt
for wl in hdr10_4k_tonemap_high_quality default_preset_4k_to_1080p nv12_1080p_to_4k_default_preset; do
(cd /tmp && rm -rf /tmp/sv && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sv -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 30 --warmup 5 --async-measure 0 --workload $wl > /tmp/sv.log 2>&1; echo "== $wl"; find /tmp/sv -name "*kernel_stats.csv" -exec head -8 {} \; | cut -d, -f1,2,4 | cut -c1-120)
done 2>&1 | tee -a gpurun_out/r04_53_presets2.txt
