#!/bin/bash
# round 6, fifteenth GPU call: PQ pieces in k_pass_chain_seg / k_polar_mxr, the closed form's clamp fixed; suite + A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_15
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -60 > gpurun_out/${tag}_gputests.log
tail -15 gpurun_out/${tag}_gputests.log | cut -c1-300
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap; do
echo "== $wl: PL_HIP_PQ_SEGMENTS" | tee -a gpurun_out/${tag}_seg_ab.txt
for v in 0 1 1 0; do echo -n "segments=$v: "; PL_HIP_PQ_SEGMENTS=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_seg_ab.txt
done
python bench.py --list-workloads 2>/dev/null | head -30
