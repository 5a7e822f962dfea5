#!/bin/bash
# round 6, twenty-first GPU call: k_bilinear_strip with one / four waves per workgroup
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_21
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py -q -m gpu -k "bilinear" 2>&1 | tail -2
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'])"; }
for w in 1 4; do for r in 2 4; do for i in 1 2; do echo -n "strip wpg=$w rows=$r: "; PL_HIP_BILIN_STRIP=1 PL_HIP_BILIN_STRIP_WPG=$w PL_HIP_BILIN_STRIP_ROWS=$r one bilinear_1080p_to_4k; done; done; done 2>&1 | tee gpurun_out/${tag}_strip_wpg.txt
for i in 1 2; do echo -n "k_bilinear_fast: "; one bilinear_1080p_to_4k; done | tee -a gpurun_out/${tag}_strip_wpg.txt
for w in 1 4; do
  out=/tmp/st_$w; rm -rf $out
  (cd /tmp && PL_HIP_BILIN_STRIP=1 PL_HIP_BILIN_STRIP_WPG=$w timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload bilinear_1080p_to_4k > /tmp/st.log 2>&1)
  echo -n "trace, wpg=$w: " | tee -a gpurun_out/${tag}_strip_wpg.txt
  find $out -name "*kernel_stats.csv" -exec grep "k_bilinear_strip" {} \; | cut -c1-150 | tee -a gpurun_out/${tag}_strip_wpg.txt
done
