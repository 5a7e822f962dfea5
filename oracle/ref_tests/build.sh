#!/bin/bash
# TEST INFRASTRUCTURE ONLY -- builds oracle/_ref/ref_gpu_tests and oracle/_ref/ref_bench: the
# reference's own src/tests/gpu_tests.c and bench.c, compiled from where they lie under
# /root/reference against the REFERENCE's headers and linked to libplacebo_amd/libplacebo_hip.so.
# (Run by oracle/build_ref.sh; the binaries travel to the GPU box with the snapshot, the reference
# tree does not exist there.)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
REF="${PL_REFERENCE:-/root/reference}"
OUT="$ROOT/oracle/_ref"
LIB="$ROOT/libplacebo_amd/libplacebo_hip.so"
if [ ! -d "$REF/src/tests" ]; then
    echo "ref_tests/build.sh: $REF not present -- keeping prebuilt binaries if any" >&2
    exit 0
fi
if [ ! -f "$LIB" ] || [ ! -f "$OUT/obj/pl_string.o" ]; then
    echo "ref_tests/build.sh: build the library and oracle/_ref first" >&2
    exit 0
fi
python3 "$HERE/cut_reference_tests.py" "$REF" "$OUT/gen"
CFLAGS="-std=c11 -O1 -g -D_GNU_SOURCE -DPL_STATIC -DPL_HAVE_PTHREAD -DPTHREAD_HAS_SETCLOCK -w \
 -I$OUT/gen -I$REF/src/include -I$REF/src -I$REF/src/tests -I$ROOT/include"
# (-I$ROOT/include last: only <libplacebo/hip.h> is taken from there, every other public header
# resolves to the reference's own)
gcc $CFLAGS -c "$HERE/ref_gpu_tests_main.c" -o "$OUT/obj/ref_gpu_tests_main.o"
gcc $CFLAGS -c "$HERE/ref_bench_main.c" -o "$OUT/obj/ref_bench_main.o"
# utils.h / common.h of the reference use a few of its internal helpers (pl_str_hash, pl_alloc):
# the objects oracle/build_ref.sh compiled from the reference's sources
INT="$OUT/obj/pl_alloc.o $OUT/obj/pl_string.o $OUT/obj/format.o $OUT/obj/convert.o"
for t in ref_gpu_tests ref_bench; do
    g++ -o "$OUT/$t" "$OUT/obj/${t}_main.o" $INT -L"$ROOT/libplacebo_amd" -l:libplacebo_hip.so \
        -Wl,-rpath,'$ORIGIN/../../libplacebo_amd' -lm -lpthread
done
echo "built $OUT/ref_gpu_tests, $OUT/ref_bench"
