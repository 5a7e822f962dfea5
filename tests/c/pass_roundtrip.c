/*
 * Test infrastructure: pl_pass on this backend (gpu.h: a pass is created from the text
 * pl_shader_finalize returns for a recorded shader). A shader -- EWA upscale + ordered dither --
 * is recorded twice with the same parameters; one copy is run through pl_dispatch_finish, the
 * other is finalized, turned into a pl_pass (after which the shader is freed), and the pass is
 * run three times: whole target, a sub-rectangle given as scissors, and into a second target.
 * All of it must be byte-identical to the dispatched frame. A compute-type pass writes through a
 * storage image descriptor. GLSL text, stale text and mismatched targets must be refused.
 * Exit status 0 and a final "ok" line = every check held.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/dispatch.h>
#include <libplacebo/hip.h>
#include <libplacebo/shaders/dithering.h>
#include <libplacebo/shaders/sampling.h>

static int errors_logged;

static void count_errors(void *priv, enum pl_log_level level, const char *msg)
{
    (void) priv; (void) msg;
    if (level <= PL_LOG_ERR)
        errors_logged++;
}

#define CHECK(x) do { if (!(x)) { printf("FAILED line %d: %s\n", __LINE__, #x); return 1; } } while (0)

enum { SW = 64, SH = 40, DW = 128, DH = 80 };

static bool record(pl_shader sh, pl_tex src, pl_shader_obj *lut)
{
    if (!pl_shader_sample_polar(sh, pl_sample_src(.tex = src, .new_w = DW, .new_h = DH),
                                pl_sample_filter_params(.filter = pl_filter_ewa_lanczos, .lut = lut)))
        return false;
    pl_shader_dither(sh, 8, NULL, pl_dither_params(.method = PL_DITHER_ORDERED_FIXED));
    return !pl_shader_is_failed(sh);
}

int main(void)
{
    pl_log log = pl_log_create(PL_API_VER, pl_log_params(.log_cb = count_errors,
                                                         .log_level = PL_LOG_ERR));
    pl_hip hip = pl_hip_create(log, pl_hip_params(.device = 0));
    CHECK(hip);
    pl_gpu gpu = hip->gpu;
    pl_fmt rgba16 = pl_find_named_fmt(gpu, "rgba16"), rgba8 = pl_find_named_fmt(gpu, "rgba8");
    CHECK(rgba16 && rgba8);

    static uint16_t img[SH][SW][4];
    for (int y = 0; y < SH; y++)
        for (int x = 0; x < SW; x++) {
            img[y][x][0] = (uint16_t) lrint(65535 * (0.5 + 0.5 * sin(0.37 * x + 0.11 * y)));
            img[y][x][1] = (uint16_t) lrint(65535 * (0.5 + 0.5 * sin(0.05 * x * y)));
            img[y][x][2] = (uint16_t) ((x * 1021 + y * 4093) & 0xffff);
            img[y][x][3] = 65535;
        }
    pl_tex src = pl_tex_create(gpu, pl_tex_params(.w = SW, .h = SH, .format = rgba16, .sampleable = true,
                                                  .host_writable = true, .initial_data = img));
#define TARGET() pl_tex_create(gpu, pl_tex_params(.w = DW, .h = DH, .format = rgba8, .renderable = true, \
                                                  .storable = true, .host_readable = true, .blit_dst = true))
    pl_tex ref = TARGET(), a = TARGET(), b = TARGET(), c = TARGET(), d = TARGET();
    CHECK(src && ref && a && b && c && d);
    static uint8_t want[DH][DW][4], got[DH][DW][4];

    // the reference frame: through the dispatch
    pl_dispatch dp = pl_dispatch_create(log, gpu);
    pl_shader_obj lut_dp = NULL, lut = NULL;
    pl_shader sh = pl_dispatch_begin(dp);
    CHECK(record(sh, src, &lut_dp));
    CHECK(pl_dispatch_finish(dp, pl_dispatch_params(.shader = &sh, .target = ref)));
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = ref, .ptr = want)));

    // the same recording as a pass
    sh = pl_shader_alloc(log, pl_shader_params(.gpu = gpu));
    CHECK(record(sh, src, &lut));
    const struct pl_shader_res *res = pl_shader_finalize(sh);
    CHECK(res && strstr(res->glsl, "#pl_hip_pass "));
    pl_pass pass = pl_pass_create(gpu, pl_pass_params(.type = PL_PASS_RASTER, .glsl_shader = res->glsl,
                                                      .target_format = rgba8));
    CHECK(pass && pass->params.type == PL_PASS_RASTER && pass->params.target_format == rgba8);
    char *stale = strdup(res->glsl);
    pl_shader_free(&sh);            // the pass holds what it needs
    pl_shader_obj_destroy(&lut);    // ... including its own reference on the filter tables
    CHECK(errors_logged == 0);

    pl_pass_run(gpu, pl_pass_run_params(.pass = pass, .target = a));
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = a, .ptr = got)));
    CHECK(!memcmp(got, want, sizeof(want)));
    // a second target, and again the first: a pass is reusable
    pl_pass_run(gpu, pl_pass_run_params(.pass = pass, .target = b,
                                        .viewport = { 0, 0, DW, DH }, .scissors = { 0, 0, DW, DH }));
    pl_pass_run(gpu, pl_pass_run_params(.pass = pass, .target = a));
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = b, .ptr = got)));
    CHECK(!memcmp(got, want, sizeof(want)));
    CHECK(errors_logged == 0 && !pl_gpu_is_failed(gpu));

    // what must be refused
    int before = errors_logged;
    CHECK(!pl_pass_create(gpu, pl_pass_params(.type = PL_PASS_RASTER, .target_format = rgba8,
                                              .glsl_shader = "void main() { gl_FragColor = vec4(1.0); }")));
    CHECK(!pl_pass_create(gpu, pl_pass_params(.type = PL_PASS_RASTER, .target_format = rgba8,
                                              .glsl_shader = stale)));     // its shader is gone
    CHECK(errors_logged == before + 2);
    before = errors_logged;
    pl_tex wrong = pl_tex_create(gpu, pl_tex_params(.w = DW, .h = DH, .format = rgba16, .renderable = true,
                                                    .storable = true));
    pl_pass_run(gpu, pl_pass_run_params(.pass = pass, .target = wrong));                    // format
    pl_pass_run(gpu, pl_pass_run_params(.pass = pass, .target = c, .scissors = { 0, 0, 32, 32 })); // size
    CHECK(errors_logged == before + 2 && !pl_gpu_is_failed(gpu));

    // compute flavour: the output is the storage image bound as the only descriptor
    sh = pl_shader_alloc(log, pl_shader_params(.gpu = gpu));
    CHECK(record(sh, src, &lut));
    res = pl_shader_finalize(sh);
    CHECK(res);
    pl_pass cpass = pl_pass_create(gpu, pl_pass_params(.type = PL_PASS_COMPUTE, .glsl_shader = res->glsl,
        .num_descriptors = 1, .descriptors = &(struct pl_desc) { .name = "out_image",
            .type = PL_DESC_STORAGE_IMG, .access = PL_DESC_ACCESS_WRITEONLY }));
    CHECK(cpass);
    pl_shader_free(&sh);
    pl_pass_run(gpu, pl_pass_run_params(.pass = cpass, .compute_groups = { 4, 10, 1 },
                                        .desc_bindings = &(struct pl_desc_binding) { .object = d }));
    CHECK(pl_tex_download(gpu, pl_tex_transfer_params(.tex = d, .ptr = got)));
    CHECK(!memcmp(got, want, sizeof(want)));

    pl_pass_destroy(gpu, &cpass);
    pl_pass_destroy(gpu, &pass);
    CHECK(!pass && !cpass);
    free(stale);
    pl_shader_obj_destroy(&lut);
    pl_shader_obj_destroy(&lut_dp);
    pl_dispatch_destroy(&dp);
    pl_tex_destroy(gpu, &wrong);
    pl_tex_destroy(gpu, &src); pl_tex_destroy(gpu, &ref); pl_tex_destroy(gpu, &a);
    pl_tex_destroy(gpu, &b); pl_tex_destroy(gpu, &c); pl_tex_destroy(gpu, &d);
    pl_hip_destroy(&hip);
    pl_log_destroy(&log);
    puts("ok");
    return 0;
}
