#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
F="tests/test_gpu_fullsize.py tests/test_gpu_kernel_variants.py tests/test_gpu_renderer.py tests/test_gpu_rotation.py tests/test_gpu_sampling.py tests/test_gpu_c_abi.py"
echo "== PL_HIP_POLAR_MFMA=1"; PL_HIP_POLAR_MFMA=1 timeout 1200 python -m pytest $F -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | cut -c1-250 | tail -30
echo "== default (pinned to k_polar_pp)"; timeout 1200 python -m pytest $F -q -m gpu 2>&1 | grep "^FAILED\|passed\|failed" | cut -c1-250 | tail -10
