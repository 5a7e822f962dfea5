#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_polar_mfma.py -q -s -m gpu -k "mxr_integer and size2 and chirp" 2>&1 | grep "samples more than\|passed\|failed\|^E  " | head -12; done
