#!/bin/bash
# round 6, first GPU call: the matrix-pipe polar kernel with live tap rows only (54 MFMAs per wave
# tile instead of 81), the debug switches compiled out, the FAST epilogue's loads and branches;
# A/B against the round-5 library; the measuring stream under a CU mask; the whole GPU suite.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_01
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py tests/test_gpu_default_kernels.py tests/test_gpu_metric.py tests/test_gpu_dither.py tests/test_gpu_async_measure.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/${tag}_tests_mx.txt
tail -5 gpurun_out/${tag}_tests_mx.txt
BASE=r05 NODRIVER=1 STEPS=200 bash tools/r05_ab.sh ${tag}_ab ewa_lanczos_1080p_to_4k_dither10 ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap nv12_1080p_to_4k_ewa_dither10
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for n in 0 16 32 64 128 0 32 64; do echo -n "measure_cus=$n: "; PL_HIP_MEASURE_CUS=$n one ewa_1080p_to_4k_hdr_tonemap; done 2>&1 | tee gpurun_out/${tag}_cumask.txt
for n in 0 32 64; do echo -n "hdr10_4k measure_cus=$n: "; PL_HIP_MEASURE_CUS=$n one hdr10_4k_tonemap; done 2>&1 | tee -a gpurun_out/${tag}_cumask.txt
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 > gpurun_out/${tag}_gputests.log
tail -6 gpurun_out/${tag}_gputests.log
export BENCH_ARGS="--bare --workload ewa_lanczos_1080p_to_4k_dither10" PL_HIP_POLAR_MFMA=1
bash tools/pmc.sh ${tag}_mx_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh ${tag}_mx_c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh ${tag}_mx_d SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | grep -A6 "k_polar_mx"
out=/tmp/st_cfg3; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st_cfg3.log 2>&1)
find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_cfg3_kernel_stats.csv \;
head -4 gpurun_out/${tag}_cfg3_kernel_stats.csv | cut -c1-160
