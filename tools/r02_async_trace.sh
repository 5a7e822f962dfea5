#!/bin/bash
# kernel timeline (start / end per launch) with async_measure on: metric workload and hdr10_4k_tonemap
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in ${WLS:-ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/async_trace_$w
  rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 8 --async-measure 1 --workload $w > $out/log.txt 2>&1)
  f=$(find $out -name '*kernel_trace.csv' | head -1)
  python - "$f" <<'PY' > $out/timeline.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-t0)/1e3:12.1f} {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} q{r.get("Queue_Id","?")} {r["Kernel_Name"][:40]}')
PY
  find $out -name '*.csv' -delete
done
