#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1
filt() { grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl\|amdgpu.ids" | grep "^FAILED\|^ERROR\|passed\|failed" | cut -c1-220 | tail -12; }
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | filt > gpurun_out/${tag}_gputests.log; echo "rc=${PIPESTATUS[0]}" >> gpurun_out/${tag}_gputests.log; cat gpurun_out/${tag}_gputests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -1
for i in 1 2; do timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_driver_full_$i.json; python3 -c "
import json; d=json.loads(open('gpurun_out/${tag}_driver_full_$i.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('valu_frac'), d['cpu_baseline']['value'])"; done
