#!/bin/bash
# round 6, second GPU call: the whole GPU suite on the new matrix-pipe kernel; fp contraction in
# the kernel that carries the colour-map chain (library variant built with -ffp-contract=fast for
# k_polar_mx.hip only) against the tree: time, VALU instructions, float64 distance of the metric frame
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_02
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/${tag}_gputests.log
tail -8 gpurun_out/${tag}_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.txt
BASE=contract NODRIVER=1 STEPS=200 bash tools/r05_ab.sh ${tag}_contract_ab ewa_1080p_to_4k_hdr_tonemap ewa_lanczos_1080p_to_4k_dither10
echo "== metric frame vs oracle / float64 with the contracted kernel" | tee -a gpurun_out/${tag}_contract_ab.txt
PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_contract.so timeout 600 python -m pytest tests/test_gpu_metric.py -q -m gpu -s 2>&1 | grep -i "passed\|failed\|codes\|float64\|quantile\|distance\|max\|p99" | head -30 | tee -a gpurun_out/${tag}_contract_ab.txt
echo "== the same with the tree's kernel" | tee -a gpurun_out/${tag}_contract_ab.txt
timeout 600 python -m pytest tests/test_gpu_metric.py -q -m gpu -s 2>&1 | grep -i "passed\|failed\|codes\|float64\|quantile\|distance\|max\|p99" | head -30 | tee -a gpurun_out/${tag}_contract_ab.txt
export BENCH_ARGS="--bare --workload ewa_1080p_to_4k_hdr_tonemap --async-measure 0"
bash tools/pmc.sh ${tag}_chain_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh ${tag}_chain_c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY 2>&1 | grep -A6 "k_polar_mx"
PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_contract.so bash tools/pmc.sh ${tag}_chain_contract SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A6 "k_polar_mx"
PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_r05.so bash tools/pmc.sh ${tag}_chain_r05 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A6 "k_polar_mx"
export BENCH_ARGS="--bare --workload ewa_lanczos_1080p_to_4k_dither10"
bash tools/pmc.sh ${tag}_mx_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh ${tag}_mx_b SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU 2>&1 | grep -A6 "k_polar_mx"
BASE=r05 NODRIVER=1 STEPS=200 bash tools/r05_ab.sh ${tag}_ab ewa_lanczos_1080p_to_4k_dither10
