cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for nt in 1 0; do
out=/tmp/st_c$nt; rm -rf $out
(cd /tmp && PL_HIP_NT_STORE=$nt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 60 --warmup 10 --workload bilinear_1080p_to_4k > /tmp/st_b.log 2>&1)
echo nt=$nt; find $out -name "*kernel_stats.csv" -exec head -2 {} \; | tail -1 | cut -c1-120
done
cd $GRAFT_REPO_ROOT
export BENCH_ARGS="--bare --workload bilinear_1080p_to_4k"
bash tools/pmc.sh bt_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES 2>&1 | grep -A6 "k_bilinear_tab"
bash tools/pmc.sh bt_b SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM 2>&1 | grep -A6 "k_bilinear_tab"
bash tools/pmc.sh bt_c FETCH_SIZE 2>&1 | grep -A2 "k_bilinear_tab"
bash tools/pmc.sh bt_d WRITE_SIZE 2>&1 | grep -A2 "k_bilinear_tab"
bash tools/pmc.sh bt_e TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum 2>&1 | grep -A5 "k_bilinear_tab"
