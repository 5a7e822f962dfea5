#!/bin/bash
# round 6, thirty-sixth GPU call: k_pass_chain with the tone curve's table in LDS
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_36
timeout 1200 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_fullsize.py tests/test_gpu_contrast_recovery.py tests/test_gpu_renderer.py tests/test_gpu_edge_sizes.py tests/test_gpu_mix.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in hdr10_4k_tonemap hdr10_4k_tonemap_high_quality ewa_8k_to_4k_deband_tonemap; do
echo "== $wl: PL_HIP_CHAIN_TONE_LDS" | tee -a gpurun_out/${tag}_tone_ab.txt
for v in 0 1 1 0; do echo -n "tone_lds=$v: "; PL_HIP_CHAIN_TONE_LDS=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_tone_ab.txt
done
