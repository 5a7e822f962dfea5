#!/bin/bash
# usage (on the GPU box, from the repo root): tools/r02_profile.sh <tag> [workloads...]
# For every workload: rocprofv3 --kernel-trace --stats (kernel_stats.csv) and SQ / memory
# counters in separate --pmc passes, summarised into gpurun_out/<tag>_<workload>_{stats.csv,pmc.txt}.
tag=$1; shift
wl=${@:-"ewa_1080p_to_4k_hdr_tonemap ewa_lanczos_1080p_to_4k_dither10 bilinear_1080p_to_4k hdr10_4k_tonemap ewa_8k_to_4k_deband_tonemap"}
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $root/gpurun_out
cd /tmp
for w in $wl; do
  out=/tmp/prof_$w; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $root/bench.py --bare --steps 40 --warmup 10 --workload $w > $out/stats.log 2>&1
  find $out/stats -name "*kernel_stats.csv" -exec cp {} $root/gpurun_out/${tag}_${w}_kernel_stats.csv \;
  : > $root/gpurun_out/${tag}_${w}_pmc.txt
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
             "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVES" \
             "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
    d=$out/pmc_$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $root/bench.py --bare --steps 6 --warmup 2 --workload $w > $d.log 2>&1
    python - "$d" >> $root/gpurun_out/${tag}_${w}_pmc.txt <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    for c, v in sorted(d.items()):
        print("%-74s %-22s n=%3d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))
PY
  done
  head -8 $root/gpurun_out/${tag}_${w}_kernel_stats.csv
done
