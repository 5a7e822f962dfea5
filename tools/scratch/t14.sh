cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python bench.py --workload $1 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['ms_per_step'], list(d['roofline']['passes_us'].values()))"; }
run hdr10_4k_tonemap cfg4
PL_HIP_PASS_NATIVE=0 run hdr10_4k_tonemap cfg4_old
run ewa_8k_to_4k_deband_tonemap cfg5
PL_HIP_PASS_NATIVE=0 run ewa_8k_to_4k_deband_tonemap cfg5_old
run lanczos_1080p_to_4k_dither10 lanczos
PL_HIP_PASS_NATIVE=0 run lanczos_1080p_to_4k_dither10 lanczos_old
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | tail -4
