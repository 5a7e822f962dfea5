#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_async_measure.py -x -q 2>&1 | tail -15
for a in 0 1 0 1; do
  timeout 300 python bench.py --bare --steps 300 --warmup 30 --async-measure $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('metric async=$a', d['ms_per_step'], d['value'])"
done
for a in 0 1; do
  timeout 300 python bench.py --bare --steps 200 --warmup 20 --workload hdr10_4k_tonemap --async-measure $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 async=$a', d['ms_per_step'], d['value'])"
  timeout 300 python bench.py --bare --steps 100 --warmup 10 --workload ewa_8k_to_4k_deband_tonemap --async-measure $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg5 async=$a', d['ms_per_step'], d['value'])"
done
