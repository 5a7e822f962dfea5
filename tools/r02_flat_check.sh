#!/bin/bash
# after moving k_bilinear_fast / k_nearest_fast / k_ortho_fast from flat_ to global_ loads: parity tests and
# kernel durations (rocprofv3 trace) of the workloads that use them
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for w in bilinear_1080p_to_4k lanczos_1080p_to_4k_dither10 nv12_1080p_to_4k_default_preset; do
  out=/tmp/fc_$w; rm -rf $out
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --workload $w --steps 60 --warmup 10 > /tmp/fc_$w.log 2>&1)
  echo "== $w: $(tail -1 /tmp/fc_$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
  cut -c1-110 $(find $out -name "*kernel_stats.csv") | head -5
done
