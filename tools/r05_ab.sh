#!/bin/bash
# A/B on one box: A = build_ab/libplacebo_hip_${BASE:-r04}.so, B = the in-tree library.
#   [BASE=r04] [TESTS="tests/x.py ..."] tools/r05_ab.sh <tag> [workload ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
tag=$1; shift
base=build_ab/libplacebo_hip_${BASE:-r04}.so
out=gpurun_out/${tag}.txt
: > $out
if [ -n "$TESTS" ]; then
  timeout 900 python -m pytest $TESTS -q -m gpu -x 2>&1 | grep "^FAILED\|passed\|failed\|^E  \|Error" | cut -c1-300 | tail -12 | tee -a $out
fi
wls=${@:-"ewa_1080p_to_4k_hdr_tonemap hdr10_4k_tonemap"}
one() { python bench.py --workload $1 --steps ${STEPS:-300} --warmup 30 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d.get('ms_per_frame', d['ms_per_step']), r['kernel_us'], {k[:28]: v for k, v in r['passes_us'].items()})"; }
for wl in $wls; do
  echo "== $wl" | tee -a $out
  for v in A B B A; do
    if [ $v = A ]; then echo -n "A($BASE): "; PL_HIP_LIB=$PWD/$base one $wl
    else echo -n "B(tree): "; one $wl; fi
  done 2>&1 | tee -a $out
done
if [ -z "$NODRIVER" ]; then
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5:', d['value'], d.get('ms_per_frame', d['ms_per_step']))"; done | tee -a $out
PL_HIP_LIB=$PWD/$base python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-concurrent --no-companions 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver-style 20/5 (base lib):', d['value'], d.get('ms_per_frame', d['ms_per_step']))" | tee -a $out
fi
