#!/bin/bash
# round 6, sixth GPU call: k_polar_mxp (persistent workgroups) -- parity against k_polar_mx, A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_06
timeout 900 python -m pytest tests/test_gpu_polar_mfma.py -q -m gpu -s -k "mxp" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 | cut -c1-300 | tee gpurun_out/${tag}_mxp_tests.txt
one() { python bench.py --workload $1 --steps 200 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in ewa_lanczos_1080p_to_4k_dither10 ewa_1080p_to_4k_hdr_tonemap nv12_1080p_to_4k_ewa_dither10 default_preset_ewa_1080p_to_4k; do
  echo "== $wl" | tee -a gpurun_out/${tag}_persist_ab.txt
  for v in 0 1 1 0; do echo -n "PL_HIP_MX_PERSIST=$v: "; PL_HIP_MX_PERSIST=$v one $wl; done 2>&1 | tee -a gpurun_out/${tag}_persist_ab.txt
done
timeout 1800 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -30 > gpurun_out/${tag}_gputests.log
tail -8 gpurun_out/${tag}_gputests.log | cut -c1-300
out=/tmp/st_cfg3; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 100 --warmup 10 --workload ewa_lanczos_1080p_to_4k_dither10 > /tmp/st_cfg3.log 2>&1)
find $out -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_cfg3_kernel_stats.csv \;
head -4 gpurun_out/${tag}_cfg3_kernel_stats.csv | cut -c1-160
