#!/bin/bash
# A/B kernel experiments: build a second copy of the library with extra HIP flags
#   tools/ab.sh build '<-D flags>'     (here, cross-compiles; the .so travels with gpurun)
#   tools/ab.sh run <bench args...>    (on the GPU box: A = in-tree library, B = variant)
cd "$(dirname "$0")/.."
case "$1" in
build)
    rm -rf build_ab && mkdir -p build_ab/obj
    make -C libplacebo_amd/csrc -j16 OUT=$PWD/build_ab/libplacebo_hip_b.so BUILD=$PWD/build_ab/obj \
        EXTRA_HIPFLAGS="$2" 2>&1 | grep -E "error|warning" -A5
    ls -la build_ab/*.so ;;
run)
    shift
    one() { python bench.py "$@" --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['kernel_us'], r['frac'], r['passes_us'])"; }
    # (A B B A A B: consecutive processes on a box can alternate between two timing modes)
    for v in A B B A A B; do
        if [ $v = A ]; then echo -n "A: "; one "$@"
        else echo -n "B: "; PL_HIP_LIB=$PWD/build_ab/libplacebo_hip_b.so one "$@"; fi
    done ;;
esac
