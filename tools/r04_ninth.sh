#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_fullsize.py -q -m gpu -x -k "bilinear or cfg2 or tables" 2>&1 | tail -4 | tee gpurun_out/r04_09_tests.log
for t in 1 0 1 0; do echo -n "tables=$t "; PL_HIP_BILIN_TABLES=$t python bench.py --workload bilinear_1080p_to_4k --steps 400 --warmup 40 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], r['frac'])"; done | tee gpurun_out/r04_09_ab.txt
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bl_prof -- python $GRAFT_REPO_ROOT/bench.py --bare --steps 60 --warmup 10 --workload bilinear_1080p_to_4k > /dev/null 2>&1; find /tmp/bl_prof -name "*kernel_stats.csv" -exec head -3 {} \; | cut -c1-160 | tee -a $GRAFT_REPO_ROOT/gpurun_out/r04_09_ab.txt
export BENCH_ARGS="--bare --workload bilinear_1080p_to_4k"
cd $GRAFT_REPO_ROOT; bash tools/pmc.sh r04_09 SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU 2>&1 | grep -A5 "k_bilinear" | tee -a gpurun_out/r04_09_ab.txt
