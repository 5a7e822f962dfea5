#!/bin/bash
# round 6, sixteenth GPU call: PQ pieces with all reads of a stage in flight; k_pass_chain_seg one / two pixels per lane
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_16
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_metric.py tests/test_gpu_fullsize.py tests/test_gpu_edge_sizes.py tests/test_gpu_contrast_recovery.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -12 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
echo "== ewa_1080p_to_4k_hdr_tonemap: PL_HIP_PQ_SEGMENTS" | tee -a gpurun_out/${tag}_seg_ab.txt
for v in 0 1 1 0; do echo -n "segments=$v: "; PL_HIP_PQ_SEGMENTS=$v one ewa_1080p_to_4k_hdr_tonemap; done 2>&1 | tee -a gpurun_out/${tag}_seg_ab.txt
for wl in hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap; do
echo "== $wl: PL_HIP_PQ_SEGMENTS x pixels per lane" | tee -a gpurun_out/${tag}_seg_ab.txt
for v in "0 1" "1 1" "1 2" "1 2" "1 1" "0 1"; do set -- $v; echo -n "segments=$1 np=$2: "; PL_HIP_PQ_SEGMENTS=$1 PL_HIP_CHAIN_SEG_NP=$2 one $wl; done 2>&1 | tee -a gpurun_out/${tag}_seg_ab.txt
done
export BENCH_ARGS="--bare --workload hdr10_4k_tonemap"
for v in 0 1; do
PL_HIP_PQ_SEGMENTS=$v bash tools/pmc.sh ${tag}_seg$v SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS 2>&1 | grep -A8 "k_pass_chain" | head -12 | tee -a gpurun_out/${tag}_seg_ab.txt
done
