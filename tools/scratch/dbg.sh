cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 0 1 4 8 5 12 13; do
  PL_HIP_PP_DEBUG=$d PL_HIP_POLAR_MFMA=1 timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg=$d', d['ms_per_step'], d['roofline']['kernel_us'])"
done
