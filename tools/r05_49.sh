#!/bin/bash
# yadif on whole planes (k_deint_rows_yadif with the cheaper edge lanes), then -- only if its tests
# pass -- the GPU suite and the smoke test on this tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r05_49_yadif_rows.txt
timeout 300 python -m pytest tests/test_gpu_deinterlace.py -q -m gpu -x 2>&1 | grep "^FAILED\|passed\|failed\|^E  \|Error" | cut -c1-400 | tail -12 | tee $out
grep -q "failed\|FAILED\|Error" $out && exit 0
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), {k[:34]: v for k, v in r['passes_us'].items()})"; }
echo -n "yadif rows: " | tee -a $out; one nv12_1080i_to_4k_yadif_default_preset 2>&1 | tee -a $out
wl=nv12_1080i_to_4k_yadif_default_preset; o=/tmp/st_$wl; rm -rf $o
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --async-measure 0 --workload $wl > /tmp/st_$wl.log 2>&1)
find $o -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05_49_${wl}_kernel_stats.csv \;
grep deint gpurun_out/r05_49_${wl}_kernel_stats.csv | cut -c1-140 | tee -a $out
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r05_49_gputests.log
cat gpurun_out/r05_49_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r05_49_smoke.txt
