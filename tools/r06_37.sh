#!/bin/bash
# round 6, the final tree (with the tone tables in LDS): smoke, the whole suite, the default bench line, the driver's command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_37
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -3 gpurun_out/${tag}_gputests.log | cut -c1-300
timeout 1500 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; python -c "
import json
d=json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('default:', d['value'], d['ms_per_frame'], r['kernel_us'], r['frac'], r.get('valu_frac'), r.get('traffic'))
for k,v in d.get('rooflines',{}).items(): print(' ', k, v.get('ms_per_frame'), v.get('kernel_us'), v.get('frac'), (v.get('trace') or {}).get('kernel_us'), (v.get('trace') or {}).get('frac'))
print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${tag}_driver.json; python -c "import json; d=json.loads(open('gpurun_out/${tag}_driver.json').read()); print('driver command:', d['value'], d['ms_per_frame'], d['roofline']['kernel_us'], d['roofline']['frac'])"
