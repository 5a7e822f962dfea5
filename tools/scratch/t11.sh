cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], d['roofline']['kernel_us'])"; }
for i in 1 2; do
run ewa_lanczos_1080p_to_4k_dither10 cfg3_rows2
PL_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_rows4.so run ewa_lanczos_1080p_to_4k_dither10 cfg3_rows4
run ewa_1080p_to_4k_hdr_tonemap metric_rows2
PL_HIP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_rows4.so run ewa_1080p_to_4k_hdr_tonemap metric_rows4
done
