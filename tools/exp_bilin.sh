#!/bin/bash
# experiment: streaming stores on/off for the two rgba16-store fast kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { python bench.py "$@" --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['kernel_us'], r['frac'])"; }
for nt in 0 1; do
  echo "== PL_HIP_NT_STORE=$nt bilinear"; PL_HIP_NT_STORE=$nt run --workload bilinear_1080p_to_4k
  echo "== PL_HIP_NT_STORE=$nt polar";    PL_HIP_NT_STORE=$nt run
done
