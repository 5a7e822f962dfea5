#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for sw in "20 5" "20 30" "60 5" "200 5" "200 30" "20 5"; do set -- $sw
PL_BENCH_DEBUG=1 python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline --no-traffic --no-concurrent --no-companions --bare 2> /tmp/dbg.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps $1 warmup $2:', d['value'], d.get('ms_per_frame', d['ms_per_step']))"; grep "^bench: steps" /tmp/dbg.txt | head -1
done 2>&1 | tee gpurun_out/r06_24_steps.txt
python - <<'PY' 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tee -a gpurun_out/r06_24_steps.txt
import os, sys, time, gc
sys.path.insert(0, os.getcwd())
import bench
import torch
torch.cuda.synchronize()
st = bench.Stream(0, "ewa_1080p_to_4k_hdr_tonemap", 10)
bench.prime(st)
for tag in ("torch imported, as main()", "again"):
    gc.collect(); gc.disable()
    for _ in range(50): st.step()
    st.g.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): st.step()
    t1 = time.perf_counter(); st.g.finish(); torch.cuda.synchronize(); t2 = time.perf_counter()
    gc.enable()
    print(tag, f"200 frames: host {1e6*(t1-t0)/200:.1f} us/frame, with finish {1e6*(t2-t0)/200:.1f}")
st.close()
PY
