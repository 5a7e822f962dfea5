#!/bin/bash
# round 6, seventh GPU call: k_polar_mxp v2 (no loads in the compute phase, LDS-only barriers,
# unconditional stores, per-phase epilogue) -- parity, A/B, trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_07
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py tests/test_gpu_default_kernels.py tests/test_gpu_dither.py tests/test_gpu_edge_sizes.py -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -15 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_lanczos_1080p_to_4k_dither10 nv12_1080p_to_4k_ewa_dither10; do
  echo "== $wl" | tee -a gpurun_out/${tag}_persist_ab.txt
  for v in 0 1 1 0; do echo -n "PL_HIP_MX_PERSIST=$v: "; PL_HIP_MX_PERSIST=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_persist_ab.txt
done
out=/tmp/st_cfg3; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st_cfg3.log 2>&1)
find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_cfg3_kernel_stats.csv \;
head -3 gpurun_out/${tag}_cfg3_kernel_stats.csv | cut -c1-160
export BENCH_ARGS="--bare --workload ewa_lanczos_1080p_to_4k_dither10"
bash tools/pmc.sh ${tag}_mxp_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh ${tag}_mxp_c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY 2>&1 | grep -A6 "k_polar_mx"
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -6 gpurun_out/${tag}_gputests.log | cut -c1-300
