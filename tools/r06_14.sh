#!/bin/bash
# round 6, fourteenth GPU call: the PQ pieces against float64 (per-sample statistics, both settings), the suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=r06_14
for v in 0 1; do
echo "==== PL_HIP_PQ_SEGMENTS=$v" >> gpurun_out/${tag}_per_sample.txt
PL_HIP_PQ_SEGMENTS=$v PL_PARITY_REPORT_ONLY=1 timeout 1200 python -m pytest tests/test_gpu_metric.py tests/test_gpu_fullsize.py -q -m gpu -s -k "metric or cfg3 or cfg4 or ewa" 2>&1 | grep -A1 "per sample\|passed\|failed" | cut -c1-500 >> gpurun_out/${tag}_per_sample.txt
done
cat gpurun_out/${tag}_per_sample.txt | grep -v "^--" | tail -50
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -40 > gpurun_out/${tag}_gputests.log
tail -25 gpurun_out/${tag}_gputests.log | cut -c1-300
