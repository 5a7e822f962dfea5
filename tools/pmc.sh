#!/bin/bash
# usage: tools/pmc.sh <tag> <counters...> -- runs bench under rocprofv3 --pmc, writes gpurun_out/pmc_<tag>.csv
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp
rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > $out/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$out/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-28s n=%3d mean=%.4g" % (c, len(v), sum(v)/len(v)))
PY
