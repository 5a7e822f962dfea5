#!/bin/bash
# round 6, nineteenth GPU call: the PQ linearisation of the debanding kernels from the piecewise cubics (device-memory copy)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_19
timeout 1200 python -m pytest tests/test_gpu_ortho_deband.py tests/test_gpu_fullsize.py tests/test_gpu_kernel_variants.py tests/test_gpu_default_kernels.py tests/test_gpu_edge_sizes.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 | cut -c1-300 > gpurun_out/${tag}_tests.txt
tail -12 gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 100 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_8k_to_4k_deband_tonemap hdr10_4k_tonemap_high_quality; do
echo "== $wl: PL_HIP_PQ_SEGMENTS" | tee -a gpurun_out/${tag}_seg_ab.txt
for v in 0 1 1 0; do echo -n "segments=$v: "; PL_HIP_PQ_SEGMENTS=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_seg_ab.txt
done
