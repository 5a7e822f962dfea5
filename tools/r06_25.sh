#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
out=/tmp/tr25; rm -rf $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 60 --warmup 5 --bare --no-cpu-baseline --no-traffic --no-concurrent --no-companions > /tmp/st.log 2>&1)
f=$(find $out -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY' | tee gpurun_out/r06_25_after_sync.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_polar_mx" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
dur = [e - s for s, e in zip(st, en)]
gaps = [i for i in range(1, len(rows)) if st[i] - en[i - 1] > 150000]
print("k_polar_mx launches", len(rows), "gaps > 150 us before launch index", gaps[-6:])
for g in gaps[-3:]:
    seg = dur[g:g + 700]
    print("after the gap at", g, "(idle %.0f us):" % ((st[g] - en[g - 1]) / 1e3),
          "kernel us per 50 launches:", " ".join("%.1f" % (sum(seg[i:i + 50]) / max(1, len(seg[i:i + 50])) / 1e3) for i in range(0, len(seg), 50)))
    per = [(st[i + 1] - st[i]) / 1e3 for i in range(g, min(g + 700, len(rows) - 1))]
    print("   start-to-start us per 50 launches:", " ".join("%.1f" % (sum(per[i:i + 50]) / max(1, len(per[i:i + 50]))) for i in range(0, len(per), 50)))
PY
