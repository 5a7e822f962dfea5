#!/bin/bash
# round 6, final: smoke, the whole suite, the driver's command, the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_29
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -4 gpurun_out/${tag}_gputests.log | cut -c1-300
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_driver_$i.json; python -c "import sys,json; d=json.loads(open('gpurun_out/${tag}_driver_$i.json').read()); print('driver command:', d['value'], d['ms_per_frame'], d['roofline']['kernel_us'], d['roofline']['frac'], d['cpu_baseline']['value'])"; done
