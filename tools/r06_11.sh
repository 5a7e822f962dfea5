#!/bin/bash
# round 6, eleventh GPU call: k_bilinear_strip (cfg 2) -- parity, A/B, trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_11
timeout 900 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_fullsize.py tests/test_gpu_edge_sizes.py tests/test_gpu_renderer.py -q -m gpu -k "bilinear or cfg2 or edge or renderer" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -12 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
echo "== bilinear_1080p_to_4k" | tee gpurun_out/${tag}_strip_ab.txt
for v in 0 1 1 0; do echo -n "PL_HIP_BILIN_STRIP=$v: "; PL_HIP_BILIN_STRIP=$v one bilinear_1080p_to_4k; done 2>&1 | tee -a gpurun_out/${tag}_strip_ab.txt
for v in 0 1; do
  out=/tmp/st_$v; rm -rf $out
  (cd /tmp && PL_HIP_BILIN_STRIP=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload bilinear_1080p_to_4k > /tmp/st.log 2>&1)
  echo -n "trace, strip=$v: " | tee -a gpurun_out/${tag}_strip_ab.txt
  find $out -name "*kernel_stats.csv" -exec grep "k_bilinear" {} \; | cut -c1-130 | tee -a gpurun_out/${tag}_strip_ab.txt
done
export BENCH_ARGS="--bare --workload bilinear_1080p_to_4k"
bash tools/pmc.sh ${tag}_strip_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_ANY 2>&1 | grep -A6 "k_bilinear"
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -6 gpurun_out/${tag}_gputests.log | cut -c1-300
