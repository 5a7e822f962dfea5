cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['roofline']['kernel_us'])"; }
run ewa_lanczos_1080p_to_4k_dither10 cfg3
run ewa_lanczos_1080p_to_4k_dither10 cfg3
run ewa_1080p_to_4k_hdr_tonemap metric
out=/tmp/st_mx; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st_mx.log 2>&1)
find $out -name "*kernel_stats.csv" -exec head -2 {} \; | tail -1 | cut -c1-120
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py tests/test_gpu_metric.py -q -m gpu 2>&1 | tail -3
