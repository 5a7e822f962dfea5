/*
 * TEST INFRASTRUCTURE ONLY -- the reference's own GPU tests (src/tests/gpu_tests.c) against the
 * HIP backend.
 *
 * The reference runs this file once per backend from a ten-line wrapper (src/tests/vulkan.c,
 * opengl_surfaceless.c, d3d11.c: create the backend, call gpu_shader_tests(gpu)). This is that
 * wrapper for pl_hip. The test source is compiled from where it lies under /root/reference, through
 * oracle/ref_tests/cut_reference_tests.py, which removes the passages that need a GLSL compiler or
 * an out-of-scope stage (every cut is listed in oracle/_ref/gen/gpu_tests_cuts.txt and printed by
 * tests/test_gpu_reference_tests.py); the public headers it is compiled against are the
 * REFERENCE's (-I/root/reference/src/include), the library it is linked to is libplacebo_hip.so.
 *
 *   ref_gpu_tests [buffer|texture|planar|shader|scaler|render|ycbcr ...]   (default: all, in the
 *   order of gpu_shader_tests, src/tests/gpu_tests.c:1837-1848)
 */
#include "gpu_tests.h"      /* the reference's: utils.h (REQUIRE ...), <libplacebo/gpu.h> */
#include <libplacebo/hip.h>

/* What the cut raster pass of pl_shader_tests draws into `fbo` (src/tests/gpu_tests.c:388-427: a
 * triangle strip over the whole target with colours (0,0,0) (1,0,0) (0,1,0) (1,1,0) at its corners
 * = the gradient its own TEST_FBO_PATTERN then checks): uploaded instead. */
static void ref_tests_draw_gradient(pl_gpu gpu, pl_tex fbo)
{
    const int w = fbo->params.w, h = fbo->params.h;
    float *px = malloc(sizeof(float) * 4 * w * h);
    REQUIRE(px);
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float *c = &px[4 * (y * w + x)];
            c[0] = (x + 0.5f) / w;
            c[1] = (y + 0.5f) / h;
            c[2] = 0.0f;
            c[3] = 1.0f;
        }
    }
    REQUIRE(fbo->params.format->type == PL_FMT_FLOAT && fbo->params.format->texel_size == 16);
    /* (the test's fbo is a render / blit target, not host_writable: through a texture created
     * with the data, then the public pl_tex_blit) */
    pl_tex tmp = pl_tex_create(gpu, pl_tex_params(
        .format = fbo->params.format, .w = w, .h = h, .blit_src = true, .initial_data = px,
    ));
    REQUIRE(tmp);
    pl_tex_blit(gpu, pl_tex_blit_params(.src = tmp, .dst = fbo));
    pl_tex_destroy(gpu, &tmp);
    free(px);
}

#include "gpu_tests_hip.c"  /* oracle/_ref/gen: the reference's gpu_tests.c minus the listed cuts */

int main(int argc, char **argv)
{
    pl_log log = pl_test_logger();
    pl_hip hip = pl_hip_create(log, NULL);
    if (!hip)
        return SKIP;
    pl_gpu gpu = hip->gpu;
    static const struct { const char *name; void (*fn)(pl_gpu); } tests[] = {
        { "buffer", pl_buffer_tests }, { "texture", pl_texture_tests }, { "planar", pl_planar_tests },
        { "shader", pl_shader_tests }, { "scaler", pl_scaler_tests }, { "render", pl_render_tests },
        { "ycbcr", pl_ycbcr_tests },
    };
    int ran = 0;
    for (int t = 0; t < (int) (sizeof(tests) / sizeof(tests[0])); t++) {
        bool want = argc < 2;
        for (int a = 1; a < argc; a++)
            want |= !strcmp(argv[a], tests[t].name);
        if (!want)
            continue;
        srand(1);
        tests[t].fn(gpu);
        REQUIRE(!pl_gpu_is_failed(gpu));
        printf("=== %s: done\n", tests[t].name);
        ran++;
    }
    pl_hip_destroy(&hip);
    pl_log_destroy(&log);
    printf("=== ref_gpu_tests: %d test function(s) ran, every REQUIRE held\n", ran);
    return ran ? 0 : 2;
}
