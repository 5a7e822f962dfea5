#!/bin/bash
# round 6, fifth GPU call: the whole suite; cfg 5 with k_polar_mxd on its live rows; what bounds
# k_polar_mx<3,true,0,8> now (debug-switch build: no contraction / no stores / no tile loads)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_05
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -8 gpurun_out/${tag}_gputests.log | cut -c1-300
BASE=r05 NODRIVER=1 STEPS=100 bash tools/r05_ab.sh ${tag}_cfg5_ab ewa_8k_to_4k_deband_tonemap ewa_lanczos_4k_to_1080p_dither10
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
echo "== k_polar_mx floor analysis (library built with -DPLH_MX_DEBUG; PL_HIP_PP_DEBUG bits: 1 no contraction, 4 no stores, 8 no tile loads)" | tee gpurun_out/${tag}_mx_floor.txt
for wl in ewa_lanczos_1080p_to_4k_dither10 ewa_1080p_to_4k_hdr_tonemap; do
  echo "-- $wl" | tee -a gpurun_out/${tag}_mx_floor.txt
  for dbg in 0 1 4 8 5 9 12 13 0; do echo -n "debug=$dbg: "; PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_mxdebug.so PL_HIP_PP_DEBUG=$dbg one $wl; done 2>&1 | tee -a gpurun_out/${tag}_mx_floor.txt
done
timeout 900 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 2500 gpurun_out/${tag}_bench.json | cut -c1-1200
for i in 1 2 3; do timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1; done > gpurun_out/${tag}_driver_cmd.jsonl
python3 -c "
import json
for l in open('gpurun_out/${tag}_driver_cmd.jsonl'):
    d=json.loads(l); r=d['roofline']; print('driver cmd:', d['value'], d.get('ms_per_frame'), r.get('kernel_us'), r.get('frac'), r.get('timing'), d.get('one_frame_per_step'))"
