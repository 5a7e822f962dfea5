#!/bin/bash
# usage: tools/profile.sh <workload> -- kernel-trace stats + HBM traffic counters for one bench workload.
# Writes gpurun_out/prof_<workload>/{stats.csv,fetch.txt,write.txt}
w=$1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$w
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --workload $w > $out/bench_stats.log 2>&1
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --workload $w > $out/bench_$c.log 2>&1
  python - <<PY > $out/$c.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for fn in glob.glob("$out/$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("%-72s n=%3d mean=%.6g" % (k, len(v), sum(v)/len(v)))
PY
done
rm -rf $out/stats $out/FETCH_SIZE $out/WRITE_SIZE
head -12 $out/stats.csv; cat $out/FETCH_SIZE.txt $out/WRITE_SIZE.txt; tail -1 $out/bench_stats.log
