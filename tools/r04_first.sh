#!/bin/bash
# round 4, first GPU call: the new parity / boundary tests verbosely, the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_default_kernels.py tests/test_gpu_clear.py tests/test_gpu_multigpu.py tests/test_gpu_c_abi.py tests/test_gpu_dither.py::test_white_noise_plane_covers_the_padded_rows -q -s -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -150 > gpurun_out/r04_01_newtests.log
tail -40 gpurun_out/r04_01_newtests.log
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/r04_01_gputests.log
tail -8 gpurun_out/r04_01_gputests.log
timeout 600 python bench.py > gpurun_out/r04_01_bench.json 2> gpurun_out/r04_01_bench.err
tail -c 400 gpurun_out/r04_01_bench.err
cut -c1-600 gpurun_out/r04_01_bench.json
