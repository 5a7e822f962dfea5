#!/bin/bash
# what makes the 20 frames x 10 behind torch.cuda.synchronize() slower than steady state: runtime knobs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions --bare 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('ms_per_frame', d['ms_per_step']))"; }
for i in 1 2; do echo -n "default: "; run; done
for i in 1 2; do echo -n "HSA_ENABLE_INTERRUPT=0: "; HSA_ENABLE_INTERRUPT=0 run; done
for i in 1 2; do echo -n "GPU_MAX_HW_QUEUES=8: "; GPU_MAX_HW_QUEUES=8 run; done
for i in 1 2; do echo -n "GPU_MAX_HW_QUEUES=2: "; GPU_MAX_HW_QUEUES=2 run; done
for i in 1 2; do echo -n "HIP_FORCE_DEV_KERNARG=1: "; HIP_FORCE_DEV_KERNARG=1 run; done
for i in 1 2; do echo -n "PL_HIP async_measure=0: "; python bench.py --gpus 1 --steps 20 --warmup 5 --async-measure 0 --no-cpu-baseline --no-traffic --no-concurrent --no-companions --bare 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('ms_per_frame', d['ms_per_step']))"; done
