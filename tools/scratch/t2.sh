cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for g in 256 512 768; do
  PL_HIP_MX_GROUPS=$g timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3 groups=$g', d['ms_per_step'], d['roofline']['kernel_us'])"
done
for d in 1 4 8 5 13; do
  PL_HIP_PP_DEBUG=$d timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg=$d', d['ms_per_step'], d['roofline']['kernel_us'])"
done
timeout 300 python bench.py --workload ewa_1080p_to_4k_hdr_tonemap --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('metric', d['ms_per_step'], d['roofline']['kernel_us'])"
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py -q -x -m gpu 2>&1 | tail -3
