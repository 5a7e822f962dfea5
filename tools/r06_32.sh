#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pqseg.py tests/test_gpu_polar_mfma.py tests/test_gpu_metric.py tests/test_gpu_kernel_variants.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -25 | cut -c1-300 | tee gpurun_out/r06_32_tests.txt
