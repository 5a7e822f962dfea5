cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-companions --no-traffic --no-concurrent $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['value'])"; }
for i in 1 2; do
run ewa_1080p_to_4k_hdr_tonemap metric_serial
run ewa_1080p_to_4k_hdr_tonemap metric_async_low "--async-measure 1"
PL_HIP_AUX_PRIORITY=0 run ewa_1080p_to_4k_hdr_tonemap metric_async_normal "--async-measure 1"
done
run hdr10_4k_tonemap cfg4_serial
run hdr10_4k_tonemap cfg4_async_low "--async-measure 1"
