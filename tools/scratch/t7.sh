cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['roofline']['kernel'][:16], d['roofline']['kernel_us'], list(d['roofline']['passes_us'].values()))"; }
run bilinear_1080p_to_4k tab
PL_HIP_BILIN_TABLES=0 run bilinear_1080p_to_4k fast
out=/tmp/st_b; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 60 --warmup 10 --workload bilinear_1080p_to_4k > /tmp/st_b.log 2>&1)
find $out -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernel_variants.py tests/test_gpu_sampling.py tests/test_gpu_renderer.py tests/test_gpu_dither.py -q -m gpu -x 2>&1 | tail -5
