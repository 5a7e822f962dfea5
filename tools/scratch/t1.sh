cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03_03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py -q -s -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/mfma_tests.log
grep -n "k_polar_mx vs\|10-bit dith\|HDR epi\|^FAILED\|passed\|failed" $O/mfma_tests.log | cut -c1-220
for wl in ewa_lanczos_1080p_to_4k_dither10 ewa_1080p_to_4k_hdr_tonemap; do
  for m in 0 1; do
    PL_HIP_POLAR_MFMA=$m timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2> $O/${wl}_mfma$m.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$wl mfma=$m', d['ms_per_step'], 'ms/frame', d['roofline']['kernel'][:12], d['roofline']['kernel_us'], 'us', list(d['roofline']['passes_us'].values()))"
  done
done
for g in 256 384 512 768; do
  PL_HIP_MX_GROUPS=$g timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg3 groups=$g', d['ms_per_step'], d['roofline']['kernel_us'])"
done
for d in 1 4 8 5 13; do
  PL_HIP_PP_DEBUG=$d timeout 300 python bench.py --workload ewa_lanczos_1080p_to_4k_dither10 --steps 200 --warmup 20 --no-cpu-baseline --no-companions --no-traffic --no-concurrent 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('dbg=$d', d['ms_per_step'], d['roofline']['kernel_us'])"
done
