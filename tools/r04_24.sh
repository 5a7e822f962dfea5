#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_default_kernels.py tests/test_gpu_color.py -q -s -m gpu -k "linear_light or hdr_downscale or pq_pair" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | grep "linear-light\|oracle (libm\|PQ EOTF vs\|PQ OETF vs\|HDR downscale:\|passed\|failed\|^E  " | cut -c1-300 | tee gpurun_out/r04_24_tests.log
timeout 600 python -m pytest tests/test_gpu_polar_mfma.py -q -m gpu -k "mxd" 2>&1 | tail -3
