#!/bin/bash
# round 6, thirteenth GPU call: the PQ pair as piecewise cubics in LDS (pqseg.hiph) in the metric's launch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_13
filt() { grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl"; }
timeout 900 python -m pytest tests/test_pqseg.py tests/test_gpu_kernel_variants.py tests/test_gpu_metric.py tests/test_gpu_default_kernels.py -q -m "gpu or not gpu" -x -s -k "pq or chain or metric or default" 2>&1 | filt | grep -v "^$" | tail -25 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
echo "== ewa_1080p_to_4k_hdr_tonemap: PL_HIP_PQ_SEGMENTS x copies" | tee gpurun_out/${tag}_seg_ab.txt
for v in "0 1" "1 1" "1 2" "1 4" "1 1" "0 1"; do set -- $v; echo -n "segments=$1 copies=$2: "; PL_HIP_PQ_SEGMENTS=$1 PL_HIP_PQ_SEG_COPIES=$2 one ewa_1080p_to_4k_hdr_tonemap; done 2>&1 | tee -a gpurun_out/${tag}_seg_ab.txt
for v in 0 1; do
  out=/tmp/sg_$v; rm -rf $out
  (cd /tmp && PL_HIP_PQ_SEGMENTS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload ewa_1080p_to_4k_hdr_tonemap > /tmp/st.log 2>&1)
  echo "trace, segments=$v: " | tee -a gpurun_out/${tag}_seg_ab.txt
  find $out -name "*kernel_stats.csv" -exec head -4 {} \; | cut -c1-150 | tee -a gpurun_out/${tag}_seg_ab.txt
done
export BENCH_ARGS="--bare --workload ewa_1080p_to_4k_hdr_tonemap"
for v in 0 1; do
PL_HIP_PQ_SEGMENTS=$v bash tools/pmc.sh ${tag}_seg$v SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES 2>&1 | grep -A8 "k_polar_mx" | head -12 | tee -a gpurun_out/${tag}_seg_ab.txt
done
