cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03_04; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" > $O/gputests.log
grep -n "^FAILED\|passed\|failed\|^E   " $O/gputests.log | cut -c1-300 | tail -40
