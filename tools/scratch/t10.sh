cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['roofline']['kernel_us'], list(d['roofline']['passes_us'].values()))"; }
run ewa_1080p_to_4k_hdr_tonemap metric_wtc8
PL_HIP_MX_FULL4=1 run ewa_1080p_to_4k_hdr_tonemap metric_wtc4
run ewa_1080p_to_4k_hdr_tonemap metric_wtc8
PL_HIP_MX_FULL4=1 run ewa_1080p_to_4k_hdr_tonemap metric_wtc4
run ewa_lanczos_1080p_to_4k_dither10 cfg3
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py tests/test_gpu_metric.py -q -m gpu 2>&1 | tail -3
