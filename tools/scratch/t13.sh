cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], list(d['roofline']['passes_us'].values()))"; }
run ewa_1080p_to_4k_hdr_tonemap metric
PL_HIP_PEAK_FAST=0 run ewa_1080p_to_4k_hdr_tonemap metric_old
run hdr10_4k_tonemap cfg4
PL_HIP_PEAK_FAST=0 run hdr10_4k_tonemap cfg4_old
run ewa_8k_to_4k_deband_tonemap cfg5
timeout 1500 python -m pytest tests/test_gpu_color.py tests/test_gpu_fullsize.py tests/test_gpu_metric.py tests/test_gpu_multigpu.py tests/test_gpu_async_measure.py tests/test_gpu_renderer.py -q -m gpu 2>&1 | tail -4
