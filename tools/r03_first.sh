#!/bin/bash
# round 3, first contact: the matrix-pipe polar kernel (tests, A/B timing, kernel trace) and the
# metric-pipeline parity tests. Outputs under gpurun_out/r03_01/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03_01; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py -q -s -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -60 > $O/mfma_tests.log
tail -15 $O/mfma_tests.log
for wl in ewa_lanczos_1080p_to_4k_dither10 ewa_1080p_to_4k_hdr_tonemap; do
  for m in 0 1; do
    PL_HIP_POLAR_MFMA=$m timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent > $O/${wl}_mfma$m.json 2> $O/${wl}_mfma$m.err
    python - <<PY
import json
d=json.load(open("$O/${wl}_mfma$m.json"))
print("$wl mfma=$m", d["ms_per_step"], "ms/frame", d["roofline"]["kernel"], d["roofline"]["kernel_us"], "us", d["roofline"]["passes_us"])
PY
  done
done
for rows in 2 3; do
  PL_HIP_MX_ROWS=$rows timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent > $O/cfg3_rows$rows.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/cfg3_rows$rows.json')); print('cfg3 wrows=$rows', d['ms_per_step'], d['roofline']['kernel_us'])"
done
out=/tmp/st_mx; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st_mx.log 2>&1)
find $out -name "*kernel_stats.csv" -exec cp {} $O/cfg3_mx_kernel_stats.csv \;
head -5 $O/cfg3_mx_kernel_stats.csv | cut -c1-160
timeout 1500 python -m pytest tests/test_gpu_metric.py -q -s -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -60 > $O/metric_tests.log
tail -25 $O/metric_tests.log
