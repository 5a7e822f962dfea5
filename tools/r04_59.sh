#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_default_kernels.py tests/test_gpu_kernel_variants.py tests/test_gpu_renderer.py tests/test_gpu_metric.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | grep "^FAILED\|passed\|failed\|^E  " | cut -c1-250 | tail -5
