#!/bin/bash
# round 6, thirty-fifth GPU call: the tone curve's table in LDS (k_polar_mx chain epilogue)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_35
timeout 1200 python -m pytest tests/test_gpu_metric.py tests/test_gpu_kernel_variants.py tests/test_gpu_polar_mfma.py tests/test_gpu_default_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_contrast_recovery.py tests/test_gpu_edge_sizes.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | cut -c1-300 | tee gpurun_out/${tag}_tests.txt
BASE=r06b NODRIVER=1 STEPS=200 bash tools/r05_ab.sh ${tag}_ab ewa_1080p_to_4k_hdr_tonemap 2>&1 | tail -8
