#!/bin/bash
# (the PL_HIP_ORTHO_DEBUG switches this script drove were temporary and are no longer in k_ortho_fast: kept as the record of how profiles/r05_summary.md got its numbers)
# where the default preset's separable passes spend their time: debug switches (temporary) of k_ortho_fast
cd $GRAFT_REPO_ROOT
one() { python bench.py --workload $1 --steps 300 --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for d in 0 1 2 4 3 7 0; do echo -n "debug=$d: "; PL_HIP_ORTHO_DEBUG=$d one default_preset_1080p_to_4k; done
