#!/bin/bash
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel'], r['kernel_us'], r['frac'], r['passes_us'])"; }
run
run --workload ewa_8k_to_4k_deband_tonemap
run --workload ewa_1080p_to_4k_hdr_tonemap
