cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 0 1 4 5; do
out=/tmp/st_b$d; rm -rf $out
(cd /tmp && PL_HIP_PP_DEBUG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 60 --warmup 10 --workload bilinear_1080p_to_4k > /tmp/st_b.log 2>&1)
echo dbg=$d; find $out -name "*kernel_stats.csv" -exec head -2 {} \; | tail -1 | cut -c1-120
done
