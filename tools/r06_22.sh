#!/bin/bash
# round 6, twenty-second GPU call: the whole suite on the tree, the default bench line, the driver's command (with host timing)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_22
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -5 gpurun_out/${tag}_gputests.log | cut -c1-300
for i in 1 2; do PL_BENCH_DEBUG=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2> gpurun_out/${tag}_driver_dbg_$i.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5:', d['value'], d.get('ms_per_frame', d['ms_per_step']))"; grep "^bench:" gpurun_out/${tag}_driver_dbg_$i.txt | cut -c1-400; done
timeout 1500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 3000 gpurun_out/${tag}_bench.json | cut -c1-3000
