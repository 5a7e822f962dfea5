#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.

Prepares the reference's own GPU test and benchmark sources (/root/reference/src/tests/gpu_tests.c,
bench.c) for compilation against this backend -- IN PLACE: the reference files are read where they
lie, the output goes to oracle/_ref/gen/ (git-ignored, like every other product of oracle/build_ref.sh)
and is compiled into oracle/_ref/ref_gpu_tests / ref_bench by oracle/ref_tests/build.sh. Nothing of
the reference's text is kept in the repository.

What this script does is CUT, by pattern, the passages that cannot run on a backend without a GLSL
compiler or that belong to stages SURVEY.md section 2 marks out of scope, and in three places put a
line of OUR glue where a cut removed a declaration later code needs. Every cut is listed with its
reason in CUTS below and written to oracle/_ref/gen/<name>_cuts.txt, which the test that runs the
binary prints (tests/test_gpu_reference_tests.py). Everything else -- every REQUIRE, every
tolerance, every parameter sweep -- is the reference's text, compiled verbatim.
"""
import os
import re
import sys

# (function the passage lies in, first line pattern, last line pattern, replacement, reason)
# Patterns are matched against whole lines (re.search) in file order, the first match after the
# previous cut's end; both the first and the last line are removed.
GPU_TESTS_CUTS = [
    ("pl_shader_tests", r'^    const char \*vert_shader =$', r'^        "}";$',
     None, "GLSL source of a raster pass (vertex shader): pl_pass_create with real GLSL is refused "
           "(no compiler: SURVEY 8b, INTEGRATION.md section 2)"),
    ("pl_shader_tests", r'^    const char \*frag_shader =$', r'^        "}";$',
     None, "GLSL source of the same raster pass (fragment shader)"),
    ("pl_shader_tests", r'^    pl_fmt vert_fmt;$', r'^    pl_timer_destroy\(gpu, &timer\);$',
     "    ref_tests_draw_gradient(gpu, fbo);   /* OUR glue: the raster pass' output, uploaded */\n",
     "the raster pass itself (vertex attributes, pl_pass_create / pl_pass_run with GLSL). Replaced by an "
     "upload of the gradient it draws -- ((x + .5) / W, (y + .5) / H, 0, 1) -- so that the reference's "
     "TEST_FBO_PATTERN and everything that samples `src` afterwards run on real data"),
    ("pl_shader_tests", r'^    if \(sizeof\(vertices\) <= gpu->limits.max_vbo_size\) \{$',
     r'^    TEST_FBO_PATTERN\(1e-6, "%s", "using custom vertices"\);$',
     "    pl_dispatch dp = pl_dispatch_create(gpu->log, gpu);   /* OUR glue: declarations the cut removed */\n"
     "    pl_shader sh;\n",
     "vertex buffers, index buffers, pl_shader_custom (GLSL body) + pl_dispatch_vertex: raster / GLSL"),
    ("pl_shader_tests", r'^    if \(fbo->params.storable\) \{$', r'^    \}$',
     None, "pl_tex_blit_compute: an internal helper of the reference (src/gpu/utils.c:852) that is itself a "
           "GLSL compute pass; the public pl_tex_blit is what pl_texture_tests exercises"),
    ("pl_shader_tests", r'^        // For testing, force the use of CS if possible$', r'^        \}$',
     None, "writes the reference's PRIVATE struct pl_shader_t (sh->type = SH_COMPUTE, group_size): this "
           "backend's pl_shader has another layout, and every pass is a compute launch anyway"),
    ("pl_shader_tests", r'^    // Test film grain synthesis$', r'^    pl_shader_obj_destroy\(&grain\);$',
     None, "pl_shader_film_grain: film grain synthesis is out of scope (SURVEY section 2)"),
    ("pl_shader_tests", r'^    // Test custom shaders$', r'^    \}\)\);$',
     None, "pl_shader_custom with a GLSL body: no compiler"),
    ("pl_render_tests", r'^    // Test film grain synthesis$',
     r'^    image.film_grain = \(struct pl_film_grain_data\) \{0\};$',
     None, "pl_frame.film_grain: out of scope; the renderer raises PL_RENDER_ERR_FILM_GRAIN as the "
           "reference does without compute shaders"),
    ("pl_render_tests", r'^    // Test mpv-style custom shaders$', r'^    \}$',
     None, "mpv user shaders (GLSL hooks), fragment flavour: no compiler"),
    ("pl_render_tests", r'^    if \(gpu->glsl.compute && gpu->limits.max_ssbo_size\) \{$', r'^    \}$',
     None, "mpv user shaders, compute flavour"),
    ("pl_ycbcr_tests", r'^        \.num_hooks = 1,$', r'^        \}\},$',
     "        0   /* OUR glue: no hook */\n",
     "a no-op C hook at PL_HOOK_CHROMA_INPUT whose only purpose is to force the reference's chroma-merge "
     "path: hooks are refused here, and the planes are merged by k_pass_merge in any case. The round "
     "trip and its 150-LSB16 bound run as written"),
]

BENCH_CUTS = [
    (None, r'^#include <libplacebo/vulkan.h>$', r'^#include <libplacebo/vulkan.h>$',
     "#include <libplacebo/hip.h>   /* OUR glue: the backend header (pl_vulkan -> pl_hip in the wrapper) */\n",
     "the Vulkan backend header (needs vulkan/vulkan.h, absent here)"),
    (None, r'^static void bench_av1_grain\(', r'^static void bench_reshape_poly\(',
     "static void bench_reshape_poly(pl_shader sh, pl_shader_obj *state, pl_tex src)\n",
     "the three film-grain benchmarks (pl_shader_film_grain: out of scope)"),
    (None, r'^    benchmark\(vk->gpu, "av1_grain",', r'^    benchmark\(vk->gpu, "h274_grain",',
     None, "their three calls"),
]


def cut(text, cuts, name, out_dir):
    lines = text.split("\n")
    out, log = [], []
    pos = 0
    for func, first, last, repl, reason in cuts:
        a = next((i for i in range(pos, len(lines)) if re.search(first, lines[i])), None)
        if a is None:
            sys.exit(f"{name}: pattern {first!r} not found (the reference changed?)")
        b = next((i for i in range(a, len(lines)) if re.search(last, lines[i])), None)
        if b is None:
            sys.exit(f"{name}: end pattern {last!r} not found")
        out.extend(lines[pos:a])
        # keep the line numbers of the reference (diagnostics print __LINE__): pad the cut
        pad = (b - a + 1)
        if repl:
            rl = repl.rstrip("\n").split("\n")
            out.extend(rl)
            pad -= len(rl)
        out.extend([""] * max(pad, 0))
        log.append(f"lines {a + 1}-{b + 1}" + (f" ({func})" if func else "") + f": {reason}")
        pos = b + 1
    out.extend(lines[pos:])
    with open(os.path.join(out_dir, name + "_hip.c"), "w") as f:
        f.write("\n".join(out))
    with open(os.path.join(out_dir, name + "_cuts.txt"), "w") as f:
        f.write(f"{name}.c as compiled against the HIP backend: {len(cuts)} passages cut "
                f"({sum(1 for c in cuts if c[3])} with a line of glue in their place)\n")
        f.write("\n".join("  - " + l for l in log) + "\n")


def main():
    ref, out_dir = sys.argv[1], sys.argv[2]
    os.makedirs(out_dir, exist_ok=True)
    cut(open(os.path.join(ref, "src/tests/gpu_tests.c")).read(), GPU_TESTS_CUTS, "gpu_tests", out_dir)
    cut(open(os.path.join(ref, "src/tests/bench.c")).read(), BENCH_CUTS, "bench", out_dir)


if __name__ == "__main__":
    main()
