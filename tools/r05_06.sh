#!/bin/bash
# k_peak_tiles with two-level tickets: workgroups per slice, beside the scaler and alone
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r05_06_peak_groups.txt
: > $out
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py::test_peak_fast_equals_generic tests/test_gpu_metric.py tests/test_gpu_async_measure.py -q -m gpu -x 2>&1 | tail -3 | tee -a $out
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap; do
  echo "== $wl" | tee -a $out
  echo -n "old kernels: " | tee -a $out; PL_HIP_PEAK_TILES=0 one $wl 2>&1 | tee -a $out
  for g in 21 43 85 170; do
    echo -n "groups=$g: " | tee -a $out
    PL_HIP_PEAK_GROUPS=$g one $wl 2>&1 | tee -a $out
    echo -n "groups=$g one stream: " | tee -a $out
    PL_HIP_ASYNC_MEASURE=0 PL_HIP_PEAK_GROUPS=$g one $wl 2>&1 | tee -a $out
  done
  echo -n "old kernels: " | tee -a $out; PL_HIP_PEAK_TILES=0 one $wl 2>&1 | tee -a $out
  echo -n "one stream, old: " | tee -a $out; PL_HIP_ASYNC_MEASURE=0 PL_HIP_PEAK_TILES=0 one $wl 2>&1 | tee -a $out
done
cd /tmp
for g in 43 170; do
PL_HIP_ASYNC_MEASURE=0 PL_HIP_PEAK_GROUPS=$g rocprofv3 --kernel-trace --stats -d /tmp/prof_$g -o t -- python $GRAFT_REPO_ROOT/bench.py --workload ewa_1080p_to_4k_hdr_tonemap --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --bare --async-measure 0 > /dev/null 2>&1
echo "trace, one stream, groups=$g" >> $GRAFT_REPO_ROOT/$out
find /tmp/prof_$g -name "*kernel_stats.csv" | head -1 | xargs cat | head -5 >> $GRAFT_REPO_ROOT/$out
done
