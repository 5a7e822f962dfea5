#!/bin/bash
# the shipped library after the last experiments were reverted: smoke, the suite, one bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -3 | cut -c1-200 | tee gpurun_out/r06_43_gputests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5:', d['value'], d['ms_per_frame'])"
python bench.py --workload hdr10_4k_tonemap --steps 100 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg 4:', d['ms_per_frame'], d['roofline']['kernel'][:40])"
