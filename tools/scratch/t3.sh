cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r03_03
PL_HIP_MX_PROF=/tmp/mxprof.bin timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3 prof', d['ms_per_step'], d['roofline']['kernel_us'])"
python tools/scratch/prof_mx.py /tmp/mxprof.bin 512
