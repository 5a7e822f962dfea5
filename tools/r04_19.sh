#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/r04_mxr_dbg.py 2>&1 | grep -v amdgpu.ids | tail -25
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py -q -s -m gpu -k "mxr" 2>&1 | grep "samples more than\|passed\|failed\|k_polar_mxr\|HDR epi" | tee gpurun_out/r04_19_tests.log
