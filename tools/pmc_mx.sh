cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--bare --workload ewa_lanczos_1080p_to_4k_dither10" PL_HIP_POLAR_MFMA=1
bash tools/pmc.sh mx_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh mx_b SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh mx_c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh mx_d SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES 2>&1 | grep -A6 "k_polar_mx"
bash tools/pmc.sh mx_e SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA 2>&1 | grep -A6 "k_polar_mx"
