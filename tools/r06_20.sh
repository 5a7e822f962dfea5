#!/bin/bash
# round 6, twentieth GPU call: the contrast-recovery low-pass as one launch (k_lowpass2)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_20
timeout 1200 python -m pytest tests/test_gpu_contrast_recovery.py tests/test_gpu_fullsize.py tests/test_gpu_edge_sizes.py -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 | cut -c1-300 > gpurun_out/${tag}_tests.txt
tail -14 gpurun_out/${tag}_tests.txt
one() { python bench.py --workload $1 --steps 100 --warmup 20 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_8k_to_4k_deband_tonemap hdr10_4k_tonemap_high_quality; do
echo "== $wl: PL_HIP_LOWPASS_FUSED" | tee -a gpurun_out/${tag}_lowpass_ab.txt
for v in 0 1 1 0; do echo -n "fused=$v: "; PL_HIP_LOWPASS_FUSED=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_lowpass_ab.txt
done
out=/tmp/tr; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 50 --warmup 10 --workload hdr10_4k_tonemap_high_quality --async-measure 0 > /tmp/st.log 2>&1)
find $out -name "*kernel_stats.csv" -exec head -9 {} \; | cut -c1-160 | tee -a gpurun_out/${tag}_lowpass_ab.txt
