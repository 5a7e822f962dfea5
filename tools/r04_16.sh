#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py -q -s -m gpu -k "mxr_integer and size2" 2>&1 | grep "samples more than\|passed\|failed\|k_polar_mxr" | tee gpurun_out/r04_16_tests.log
