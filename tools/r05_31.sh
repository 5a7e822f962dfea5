#!/bin/bash
cd $GRAFT_REPO_ROOT
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'])"; }
for wl in ewa_lanczos_1080p_to_4k_dither10 bilinear_1080p_to_4k ewa_1080p_to_4k_hdr_tonemap lanczos_1080p_to_4k_dither10; do
for nt in 1 0 1 0; do echo -n "$wl NT_STORE=$nt: "; PL_HIP_NT_STORE=$nt one $wl; done; done
