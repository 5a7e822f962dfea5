/*
 * The stages of pl_render_image added in round 5, from plain C against the public headers: an
 * interlaced frame deinterlaced with its neighbours, subtitles as an overlay, an affine
 * distortion, a frame blended into what the target held, a Dolby Vision frame.
 *
 * Compiled twice by tests/c/Makefile like render_frame.c: against include/ and against the
 * REFERENCE's headers (pl_overlay, pl_deinterlace_params, pl_distort_params, pl_blend_params,
 * pl_dovi_metadata laid out by libplacebo's own declarations), both linked against
 * libplacebo_hip.so. tests/test_gpu_c_abi.py runs both and requires identical bytes, and checks
 * the frames whose content follows from the inputs without arithmetic (bob doubles rows, a
 * half turn moves texels, an opaque overlay replaces pixels).
 *
 * usage: frame_stages <out.raw>     (six 64 x 48 rgba16 frames, one after the other)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/hip.h>
#include <libplacebo/renderer.h>
#include <libplacebo/shaders/deinterlacing.h>
#include <libplacebo/shaders/sampling.h>

#define W 64
#define H 48

static void die(const char *what)
{
    fprintf(stderr, "frame_stages: %s\n", what);
    exit(1);
}

static pl_tex make_tex(pl_gpu gpu, pl_fmt fmt, int w, int h, const void *data)
{
    pl_tex tex = pl_tex_create(gpu, pl_tex_params(
        .w = w, .h = h, .format = fmt,
        .sampleable = true, .renderable = true, .storable = true,
        .host_writable = true, .host_readable = true, .blit_dst = true,
    ));
    if (!tex)
        die("pl_tex_create");
    if (data && !pl_tex_upload(gpu, pl_tex_transfer_params(.tex = tex, .ptr = (void *) data)))
        die("pl_tex_upload");
    return tex;
}

// frame t of a moving pattern; every row says which field it belongs to in its blue channel
static void pattern(uint16_t *px, int t)
{
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            uint16_t *p = &px[4 * (y * W + x)];
            p[0] = (uint16_t) (1000 * ((x + 3 * t) % 64));
            p[1] = (uint16_t) (1300 * (y % 48));
            p[2] = (uint16_t) (y % 2 ? 60000 : 5000);
            p[3] = 65535;
        }
    }
}

static struct pl_frame rgb_frame(pl_tex tex)
{
    return (struct pl_frame) {
        .num_planes = 1,
        .planes = {{ .texture = tex, .components = 3, .component_mapping = {0, 1, 2} }},
        .repr = { .sys = PL_COLOR_SYSTEM_RGB, .levels = PL_COLOR_LEVELS_FULL },
        .color = { .primaries = PL_COLOR_PRIM_BT_709, .transfer = PL_COLOR_TRC_SRGB },
    };
}

int main(int argc, char **argv)
{
    if (argc < 2)
        die("usage: frame_stages <out.raw>");
    pl_log log = pl_log_create(PL_API_VER, pl_log_params(
        .log_cb = pl_log_simple,
        .log_level = PL_LOG_WARN,
    ));
    pl_hip hip = pl_hip_create(log, pl_hip_params(.device = 0));
    if (!hip)
        die("no HIP device");
    pl_gpu gpu = hip->gpu;
    pl_fmt rgba16 = pl_find_named_fmt(gpu, "rgba16"), r8 = pl_find_named_fmt(gpu, "r8");
    if (!rgba16 || !r8)
        die("formats");

    static uint16_t px[3][W * H * 4];
    pl_tex src[3];
    for (int t = 0; t < 3; t++) {
        pattern(px[t], t);
        src[t] = make_tex(gpu, rgba16, W, H, px[t]);
    }
    pl_tex dst = make_tex(gpu, rgba16, W, H, NULL);
    pl_renderer rr = pl_renderer_create(log, gpu);
    if (!rr)
        die("pl_renderer_create");
    FILE *out = fopen(argv[1], "wb");
    if (!out)
        die("cannot open the output file");
    static uint16_t result[W * H * 4];
#define EMIT(what) do {                                                                           \
        if (pl_renderer_get_errors(rr).errors)                                                    \
            die(what ": renderer raised an error");                                               \
        if (!pl_tex_download(gpu, pl_tex_transfer_params(.tex = dst, .ptr = result)))             \
            die(what ": download");                                                               \
        fwrite(result, sizeof(result), 1, out);                                                   \
    } while (0)

    struct pl_render_params params = pl_render_fast_params;
    params.dither_params = NULL;
    struct pl_frame target = rgb_frame(dst);
    target.planes[0].components = 4;
    target.planes[0].component_mapping[3] = 3;
    target.repr.alpha = PL_ALPHA_INDEPENDENT;

    /* 1: the middle frame, top field shown, the other field doubled from it (bob) */
    struct pl_frame prev = rgb_frame(src[0]), next = rgb_frame(src[2]), image = rgb_frame(src[1]);
    image.field = PL_FIELD_TOP;
    image.first_field = PL_FIELD_TOP;
    image.prev = &prev;
    image.next = &next;
    struct pl_deinterlace_params deint = { .algo = PL_DEINTERLACE_BOB };
    params.deinterlace_params = &deint;
    if (!pl_render_image(rr, &image, &target, &params))
        die("deinterlace (bob)");
    EMIT("bob");

    /* 2: the same through bwdif with both neighbours */
    deint.algo = PL_DEINTERLACE_BWDIF;
    if (!pl_render_image(rr, &image, &target, &params))
        die("deinterlace (bwdif)");
    EMIT("bwdif");
    params.deinterlace_params = NULL;
    image.field = PL_FIELD_NONE;
    image.prev = image.next = NULL;

    /* 3: an opaque 16 x 8 box (one part of a glyph atlas whose texels are all 255) over the
     * progressive frame, in target coordinates */
    static uint8_t atlas[16 * 16];
    memset(atlas, 255, sizeof(atlas));
    pl_tex glyphs = make_tex(gpu, r8, 16, 16, atlas);
    const struct pl_overlay_part part = {
        .src = { 0, 0, 16, 8 },
        .dst = { 8, 4, 24, 12 },
        .color = { 1.0f, 0.5f, 0.25f, 1.0f },
    };
    const struct pl_overlay osd = {
        .tex = glyphs,
        .mode = PL_OVERLAY_MONOCHROME,
        .coords = PL_OVERLAY_COORDS_DST_FRAME,
        .repr = { .sys = PL_COLOR_SYSTEM_RGB, .levels = PL_COLOR_LEVELS_FULL,
                  .alpha = PL_ALPHA_INDEPENDENT },
        .color = { .primaries = PL_COLOR_PRIM_BT_709, .transfer = PL_COLOR_TRC_SRGB },
        .parts = &part,
        .num_parts = 1,
    };
    target.overlays = &osd;
    target.num_overlays = 1;
    if (!pl_render_image(rr, &image, &target, &params))
        die("overlay");
    EMIT("overlay");
    target.overlays = NULL;
    target.num_overlays = 0;

    /* 4: a half turn (every texel lands on a texel) through distort_params */
    struct pl_distort_params distort = pl_distort_default_params;
    distort.transform.mat = (pl_matrix2x2) {{{ -1, 0 }, { 0, -1 }}};
    params.distort_params = &distort;
    if (!pl_render_image(rr, &image, &target, &params))
        die("distort");
    EMIT("distort");
    params.distort_params = NULL;

    /* 5: a frame with 50 % alpha blended over what the target holds (the half turn) */
    static uint16_t half[W * H * 4];
    memcpy(half, px[1], sizeof(half));
    for (int i = 0; i < W * H; i++)
        half[4 * i + 3] = 32768;
    pl_tex translucent = make_tex(gpu, rgba16, W, H, half);
    struct pl_frame layer = rgb_frame(translucent);
    layer.planes[0].components = 4;
    layer.planes[0].component_mapping[3] = 3;
    layer.repr.alpha = PL_ALPHA_INDEPENDENT;
    params.blend_params = &pl_alpha_overlay;
    params.background_transparency = 1.0f;
    params.skip_target_clearing = true;
    if (!pl_render_image(rr, &layer, &target, &params))
        die("blend");
    EMIT("blend");
    params.blend_params = NULL;
    params.background_transparency = 0.0f;
    params.skip_target_clearing = false;

    /* 6: the frame declared as Dolby Vision with curves and matrices that change nothing
     * (identity polynomials, unit matrices, the decoder's LMS -> RGB undone by `linear`) */
    static struct pl_dovi_metadata dovi;
    memset(&dovi, 0, sizeof(dovi));
    for (int c = 0; c < 3; c++) {
        dovi.comp[c].num_pivots = 2;
        dovi.comp[c].pivots[1] = 1.0f;
        dovi.comp[c].poly_coeffs[0][1] = 1.0f;
        dovi.nonlinear.m[c][c] = 1.0f;
    }
    dovi.linear = (pl_matrix3x3) {{                 /* the inverse of the decoder's fixed matrix */
        { 0.44082, 0.53537, 0.02381 },
        { 0.16199, 0.75865, 0.07936 },
        { 0.00000, 0.02578, 0.97407 },
    }};
    struct pl_frame dv = rgb_frame(src[1]);
    dv.repr.sys = PL_COLOR_SYSTEM_DOLBYVISION;
    dv.repr.dovi = &dovi;
    dv.color = (struct pl_color_space) { .primaries = PL_COLOR_PRIM_BT_2020, .transfer = PL_COLOR_TRC_PQ };
    struct pl_frame hdr_target = target;
    hdr_target.color = dv.color;
    if (!pl_render_image(rr, &dv, &hdr_target, &params))
        die("dolby vision");
    EMIT("dolby vision");

    fclose(out);
    pl_renderer_destroy(&rr);
    pl_tex_destroy(gpu, &translucent);
    pl_tex_destroy(gpu, &glyphs);
    pl_tex_destroy(gpu, &dst);
    for (int t = 0; t < 3; t++)
        pl_tex_destroy(gpu, &src[t]);
    pl_hip_destroy(&hip);
    pl_log_destroy(&log);
    printf("frame_stages: 6 frames written\n");
    return 0;
}
