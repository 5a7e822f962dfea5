cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03_02; mkdir -p $O
for wl in ewa_lanczos_1080p_to_4k_dither10 ewa_1080p_to_4k_hdr_tonemap; do
  for m in 0 1; do
    PL_HIP_POLAR_MFMA=$m timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent > $O/${wl}_mfma$m.json 2> $O/${wl}_mfma$m.err
    python - <<PY
import json
d=json.load(open("$O/${wl}_mfma$m.json"))
print("$wl mfma=$m", d["ms_per_step"], "ms/frame", d["roofline"]["kernel"][:20], d["roofline"]["kernel_us"], "us", list(d["roofline"]["passes_us"].values()))
PY
  done
done
for rows in 1 2 3; do
  PL_HIP_MX_ROWS=$rows timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent > $O/cfg3_rows$rows.json 2>/dev/null
  python -c "
import json
d=json.load(open('$O/cfg3_rows$rows.json')); print('cfg3 wrows=$rows', d['ms_per_step'], d['roofline']['kernel_us'])"
done
out=/tmp/st_mx; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st_mx.log 2>&1)
find $out -name "*kernel_stats.csv" -exec cp {} $O/cfg3_mx_kernel_stats.csv \;
head -3 $O/cfg3_mx_kernel_stats.csv | cut -c1-160
out=/tmp/st_mx2; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --workload ewa_1080p_to_4k_hdr_tonemap > /tmp/st_mx.log 2>&1)
find $out -name "*kernel_stats.csv" -exec cp {} $O/metric_mx_kernel_stats.csv \;
head -4 $O/metric_mx_kernel_stats.csv | cut -c1-160
