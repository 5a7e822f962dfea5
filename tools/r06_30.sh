#!/bin/bash
# round 6, thirtieth GPU call: k_polar_mxp with a turn's four stages as a software pipeline (PL_HIP_MXP_PIPE)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_30
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py tests/test_gpu_default_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_kernel_variants.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'])"; }
for wl in ewa_lanczos_1080p_to_4k_dither10 nv12_1080p_to_4k_ewa_dither10; do
echo "== $wl: PL_HIP_MXP_PIPE" | tee -a gpurun_out/${tag}_pipe_ab.txt
for v in 0 1 1 0 1 0; do echo -n "pipe=$v: "; PL_HIP_MXP_PIPE=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_pipe_ab.txt
done
for v in 0 1; do
  out=/tmp/pp_$v; rm -rf $out
  (cd /tmp && PL_HIP_MXP_PIPE=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st.log 2>&1)
  echo -n "trace, pipe=$v: " | tee -a gpurun_out/${tag}_pipe_ab.txt
  find $out -name "*kernel_stats.csv" -exec grep "k_polar_mxp" {} \; | cut -c1-130 | tee -a gpurun_out/${tag}_pipe_ab.txt
done
