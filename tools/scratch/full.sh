cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03_04; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gputests.log
tail -40 $O/gputests.log | cut -c1-250
