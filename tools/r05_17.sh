#!/bin/bash
# the driver's short run (--steps 20 --warmup 5) on one time axis: what do the first timed frames after the sync cost?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r05_17_trace
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 20 --warmup 5 --workload ewa_1080p_to_4k_hdr_tonemap > $out/log.txt 2>&1)
python - $out <<'PY' > $out/timeline.txt
import csv, sys, glob, os
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:30]))
ev.sort()
# the timed region = the last 20 k_polar_mx launches; print from 8 launches before it
ks = [i for i, e in enumerate(ev) if e[2].startswith("void k_polar_mx")]
first = ks[-20]
start = ev[first][0]
prev_end = None
for s, e, n in ev[ks[-27]:]:
    gap = "" if prev_end is None else f"gap {max(0, s - prev_end) / 1e3:7.1f}"
    print(f"{(s - start) / 1e3:10.1f} {(e - start) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {n:32s} {gap}")
    if n.startswith("void k_polar_mx"):
        prev_end = e
PY
find $out -name '*.csv' -delete
cat $out/timeline.txt
