#!/bin/bash
# the decode pass into the f16 intermediate (k_pass_chain<.., F16DST>): one or two pixels per lane
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_kernel_variants.py tests/test_gpu_renderer.py tests/test_gpu_edge_sizes.py -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in default_preset_4k_to_1080p default_preset_1080p_to_4k default_preset_ewa_1080p_to_4k; do
for np in 1 2 2 1; do echo -n "$wl NP=$np: "; PL_HIP_CHAIN_F16_NP=$np one $wl; done; done
