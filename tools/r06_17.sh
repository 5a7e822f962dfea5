#!/bin/bash
# round 6, seventeenth GPU call: k_pass_chain_seg with the rotated loop, 1 / 2 / 8 turns per workgroup
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_17
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_fullsize.py tests/test_gpu_contrast_recovery.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap; do
echo "== $wl: PL_HIP_PQ_SEGMENTS x turns per workgroup" | tee -a gpurun_out/${tag}_seg_ab.txt
for v in "0 1" "1 1" "1 2" "1 8" "1 8" "1 2" "1 1" "0 1"; do set -- $v; echo -n "segments=$1 turns=$2: "; PL_HIP_PQ_SEGMENTS=$1 PL_HIP_CHAIN_SEG_ITERS=$2 one $wl; done 2>&1 | tee -a gpurun_out/${tag}_seg_ab.txt
done
