#!/bin/bash
# round 6, eighteenth GPU call: the tree with the PQ pieces in k_polar_mx / k_polar_mxr: the suite, the driver's
# command, A/B of the HDR workloads against the round-5 library and against PL_HIP_PQ_SEGMENTS=0, kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_18
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -6 gpurun_out/${tag}_gputests.log | cut -c1-300
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_1080p_to_4k_hdr_tonemap ewa_720p_to_4k_hdr_tonemap ewa_8k_to_4k_hdr_tonemap; do
echo "== $wl: PL_HIP_PQ_SEGMENTS" | tee -a gpurun_out/${tag}_seg_ab.txt
for v in 0 1 1 0; do echo -n "segments=$v: "; PL_HIP_PQ_SEGMENTS=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_seg_ab.txt
done
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5:', d['value'], d.get('ms_per_frame', d['ms_per_step']))"; done | tee -a gpurun_out/${tag}_seg_ab.txt
PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_r05.so python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5 (round-5 library):', d['value'], d.get('ms_per_frame', d['ms_per_step']))" | tee -a gpurun_out/${tag}_seg_ab.txt
out=/tmp/tr; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload ewa_1080p_to_4k_hdr_tonemap --async-measure 0 > /tmp/st.log 2>&1)
find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_ewa_1080p_to_4k_hdr_tonemap_kernel_stats.csv \;
head -4 gpurun_out/${tag}_ewa_1080p_to_4k_hdr_tonemap_kernel_stats.csv | cut -c1-160
