#!/bin/bash
# host API calls and kernels on one time axis: when is the next scaler launch issued?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r05_09_trace
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 40 --warmup 8 --async-measure 1 --workload ewa_1080p_to_4k_hdr_tonemap > $out/log.txt 2>&1)
ls -R $out | head -20
python - $out <<'PY' > $out/timeline.txt
import csv, sys, glob, os
d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:36]))
for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "  api " + r["Function"]))
ev.sort()
t0 = ev[0][0]
# the last ~6 frames
ks = [e for e in ev if e[2].startswith("K void k_polar_mx")]
start = ks[-7][0]
for s, e, n in ev:
    if s >= start:
        print(f"{(s - start) / 1e3:10.1f} {(e - start) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {n}")
PY
find $out -name '*.csv' -delete
head -150 $out/timeline.txt
