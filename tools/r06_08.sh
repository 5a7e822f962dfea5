#!/bin/bash
# round 6, eighth GPU call: how k_polar_mxp writes its target -- plain / non-temporal / system-scope
# (written through L2) stores: kernel duration in the trace and frame time
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_08
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for k in 0 1 2 0 1 2; do
  echo -n "PL_HIP_MXP_STORE=$k: " | tee -a gpurun_out/${tag}_store_kinds.txt
  PL_HIP_MXP_STORE=$k one ewa_lanczos_1080p_to_4k_dither10 | tee -a gpurun_out/${tag}_store_kinds.txt
done
for k in 0 1 2; do
  out=/tmp/st_$k; rm -rf $out
  (cd /tmp && PL_HIP_MXP_STORE=$k timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st.log 2>&1)
  echo -n "trace, store kind $k: " | tee -a gpurun_out/${tag}_store_kinds.txt
  find $out -name "*kernel_stats.csv" -exec grep k_polar_mxp {} \; | cut -c1-120 | tee -a gpurun_out/${tag}_store_kinds.txt
done
timeout 600 python -m pytest tests/test_gpu_polar_mfma.py -q -m gpu -k mxp 2>&1 | tail -2
for k in 1 2; do PL_HIP_MXP_STORE=$k timeout 600 python -m pytest tests/test_gpu_polar_mfma.py -q -m gpu -k mxp 2>&1 | tail -1; done
