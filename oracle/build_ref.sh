#!/bin/bash
# TEST INFRASTRUCTURE ONLY — builds the *real* reference CPU half (Tier-0 maths)
# from the sources where they lie under /root/reference into oracle/_ref/.
# Nothing is copied into the repo: generated headers + objects + libplref.so all
# land in oracle/_ref/ (git-ignored, travels to the GPU box with gpurun).
#
# Recipe follows SURVEY.md §8(c): hand-written config.h / config_internal.h /
# version.h, the reference's own flags (meson.build:406-410), no meson.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${PL_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
    echo "build_ref: $REF not present (GPU box?) — keeping prebuilt $OUT if any" >&2
    exit 0
fi
mkdir -p "$OUT/gen/libplacebo" "$OUT/obj"
# config.h from the reference template (substitution output only)
sed -e 's/@majorver@/7/' -e 's/@apiver@/365/' \
    -e 's/@extra_defs@/#undef PL_HAVE_VULKAN\n#undef PL_HAVE_OPENGL\n#undef PL_HAVE_D3D11\n#undef PL_HAVE_LCMS\n#undef PL_HAVE_SHADERC\n#undef PL_HAVE_GLSLANG\n#undef PL_HAVE_XXHASH\n#define PL_HAVE_DOVI 1\n#undef PL_HAVE_LIBDOVI/' \
    "$REF/src/include/libplacebo/config.h.in" > "$OUT/gen/libplacebo/config.h"
cat > "$OUT/gen/config_internal.h" <<'EOT'
#pragma once
#define BUILD_API_VER 365
#define BUILD_FIX_VER 0
#define PL_DEBUG_ABORT 0
#define PL_HAVE_EXECINFO 1
EOT
echo '#define BUILD_VERSION "v7.365.0"' > "$OUT/gen/version.h"

CFLAGS="-std=c11 -O2 -fPIC -D_GNU_SOURCE -DPL_STATIC -DPL_HAVE_PTHREAD -DPTHREAD_HAS_SETCLOCK \
 -fno-math-errno -fno-signed-zeros -fno-trapping-math -w \
 -I$OUT/gen -I$REF/src/include -I$REF/src"
SRCS="filters tone_mapping gamut_mapping colorspace dither common log pl_alloc pl_string cache format"
OBJS=""
for s in $SRCS; do
    gcc $CFLAGS -c "$REF/src/$s.c" -o "$OUT/obj/$s.o"
    OBJS="$OBJS $OUT/obj/$s.o"
done
# pure helpers of the plane-upload utility (GPU entry points stubbed by ref_shim.c)
gcc $CFLAGS -c "$REF/src/utils/upload.c" -o "$OUT/obj/utils_upload.o"
OBJS="$OBJS $OUT/obj/utils_upload.o"
# the frame queue is host-only logic (tests/test_frame_queue.py replays traces through it)
gcc $CFLAGS -c "$REF/src/utils/frame_queue.c" -o "$OUT/obj/utils_frame_queue.o"
OBJS="$OBJS $OUT/obj/utils_frame_queue.o"
g++ -std=c++20 -O2 -fPIC -w -DPL_STATIC -I$OUT/gen -I$REF/src/include -I$REF/src \
    -c "$REF/src/convert.cc" -o "$OUT/obj/convert.o"
# ref_shim.c is OUR glue (exposes a few internals as plain C-ABI for ctypes)
gcc $CFLAGS -c "$HERE/ref_shim.c" -o "$OUT/obj/ref_shim.o"
# cpu_baseline.c is OUR per-pixel driver around the reference's CPU functions (bench.py's
# cpu_baseline leg, kind "reference"); built with the reference's own flags + OpenMP
gcc $CFLAGS -fopenmp -c "$HERE/cpu_baseline.c" -o "$OUT/obj/cpu_baseline.o"
g++ -shared -fopenmp -Wl,--no-undefined -Wl,-Bsymbolic -o "$OUT/libplref.so" $OBJS "$OUT/obj/convert.o" "$OUT/obj/ref_shim.o" "$OUT/obj/cpu_baseline.o" -lm -lpthread
# the reference's src/gpu.c on its own: its shader-variable constructors and std140 / std430 layout
# functions are pure (tests/test_host.py holds the product's to them); libplref.so cannot take it,
# its shim stands in for half of that file's entry points
gcc $CFLAGS -c "$REF/src/gpu.c" -o "$OUT/obj/gpu.o"
gcc $CFLAGS -c "$HERE/ref_gpu_shim.c" -o "$OUT/obj/ref_gpu_shim.o"
gcc -shared -Wl,--no-undefined -Wl,-Bsymbolic -o "$OUT/libplref_gpu.so" "$OUT/obj/gpu.o" "$OUT/obj/ref_gpu_shim.o" \
    "$OUT/obj/common.o" "$OUT/obj/log.o" "$OUT/obj/pl_alloc.o" "$OUT/obj/pl_string.o" "$OUT/obj/format.o" "$OUT/obj/convert.o" -lstdc++ -lm -lpthread
echo "built $OUT/libplref.so, $OUT/libplref_gpu.so"
# the reference's own GPU tests and benchmark against the HIP backend (needs the library: built first
# by __graft_entry__.build())
"$HERE/ref_tests/build.sh"
