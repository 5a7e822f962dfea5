/*
 * A plain C program against the public headers: upload one RGBA16 frame, upscale it 2x with
 * EWA Lanczos through pl_render_image, download, write the raw result.
 *
 * It is compiled twice by tests/c/Makefile:
 *   render_frame_ours : against include/ (this repository's headers)
 *   render_frame_ref  : against the REFERENCE's headers (/root/reference/src/include, with
 *                       only <libplacebo/hip.h> taken from include/) -- i.e. the structs are
 *                       laid out by libplacebo's own declarations -- and linked against the
 *                       same libplacebo_hip.so.
 * tests/test_gpu_c_abi.py runs both on the GPU and requires identical output that also matches
 * the oracle. This is the drop-in property of SURVEY.md 8(b) exercised from C, not ctypes.
 *
 * Written the way a player (mpv's vo_gpu_next, FFmpeg's vf_libplacebo) uses the API: the scaler is
 * selected BY NAME through the preset tables (pl_find_filter_preset, filters.h:316-329) and targets
 * are cleared with pl_frame_clear_rgba / pl_frame_clear_tiles (renderer.h:672-690) -- the entry
 * points VERDICT r03 found missing from the library. The frame is followed in the output file by a
 * 32 x 16 frame cleared to one colour and the same frame cleared to tiles.
 *
 * usage: render_frame <out.raw> [src_w src_h]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/hip.h>
#include <libplacebo/renderer.h>

static void die(const char *what)
{
    fprintf(stderr, "render_frame: %s\n", what);
    exit(1);
}

int main(int argc, char **argv)
{
    if (argc < 2)
        die("usage: render_frame <out.raw> [src_w src_h]");
    const int sw = argc > 3 ? atoi(argv[2]) : 96, sh = argc > 3 ? atoi(argv[3]) : 64;
    const int dw = 2 * sw, dh = 2 * sh;

    pl_log log = pl_log_create(PL_API_VER, pl_log_params(
        .log_cb = pl_log_simple,
        .log_level = PL_LOG_WARN,
    ));
    pl_hip hip = pl_hip_create(log, pl_hip_params(.device = 0));
    if (!hip)
        die("no HIP device");
    pl_gpu gpu = hip->gpu;

    // the reference's bench pattern (src/tests/bench.c:32-51), quantised to 16 bit
    uint16_t *pixels = malloc((size_t) sw * sh * 4 * sizeof(uint16_t));
    const double xc = (sw - 1) / 2.0, yc = (sh - 1) / 2.0, phi = 1.6180339887498948;
    const double fr = 0.1 * M_PI * 0.5 / sqrt(xc * xc + yc * yc), fg = fr / phi, fb = fg / phi;
    for (int y = 0; y < sh; y++) {
        for (int x = 0; x < sw; x++) {
            const double r2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
            uint16_t *px = &pixels[4 * ((size_t) y * sw + x)];
            px[0] = lrint(65535.0 * (0.5 * sin(fr * r2) + 0.5));
            px[1] = lrint(65535.0 * (0.5 * sin(fg * r2) + 0.5));
            px[2] = lrint(65535.0 * (0.5 * sin(fb * r2) + 0.5));
            px[3] = 65535;
        }
    }

    pl_fmt fmt = pl_find_named_fmt(gpu, "rgba16");
    if (!fmt)
        die("no rgba16 format");
    pl_tex src = pl_tex_create(gpu, pl_tex_params(
        .w = sw, .h = sh, .format = fmt,
        .sampleable = true, .host_writable = true,
        .initial_data = pixels,
    ));
    pl_tex dst = pl_tex_create(gpu, pl_tex_params(
        .w = dw, .h = dh, .format = fmt,
        .renderable = true, .storable = true, .host_readable = true,
    ));
    if (!src || !dst)
        die("texture creation failed");

    struct pl_frame image = {
        .num_planes = 1,
        .planes = {{
            .texture = src,
            .components = 3,
            .component_mapping = {0, 1, 2},
        }},
        .repr = {
            .sys = PL_COLOR_SYSTEM_RGB,
            .levels = PL_COLOR_LEVELS_FULL,
            .bits = { .sample_depth = 16, .color_depth = 16 },
        },
        .color = pl_color_space_srgb,
    };
    struct pl_frame target = image;
    target.planes[0].texture = dst;
    target.planes[0].components = 4;
    target.planes[0].component_mapping[3] = 3;

    pl_renderer rr = pl_renderer_create(log, gpu);
    struct pl_render_params params = pl_render_fast_params;
    const struct pl_filter_preset *preset = pl_find_filter_preset("ewa_lanczos");
    if (!preset || preset->filter != &pl_filter_ewa_lanczos || !preset->description)
        die("pl_find_filter_preset(\"ewa_lanczos\")");
    if (pl_find_filter_preset("no such filter") || !pl_find_filter_preset("none") ||
        pl_find_filter_preset("none")->filter || pl_find_filter_preset("triangle")->filter != &pl_filter_bilinear)
        die("pl_find_filter_preset: none / aliases / unknown names");
    const struct pl_filter_function_preset *fpreset = pl_find_filter_function_preset("jinc");
    if (!fpreset || fpreset->function != &pl_filter_function_jinc || pl_find_filter_function_preset("nope"))
        die("pl_find_filter_function_preset");
    int counted = 0;
    while (pl_filter_presets[counted].name)
        counted++;
    if (counted != pl_num_filter_presets || pl_filter_function_presets[pl_num_filter_function_presets].name)
        die("preset tables are not {0}-terminated at their advertised length");
    // the pre-v6 lists and tables option parsers still walk (renderer.h:855-880, colorspace.h:60)
    if (strcmp(pl_scale_filters[0].name, "none") || strcmp(pl_scale_filters[1].name, "oversample") ||
        pl_scale_filters[pl_num_scale_filters].name || pl_frame_mixers[pl_num_frame_mixers].name ||
        pl_frame_mixers[2].filter != &pl_filter_oversample)
        die("pl_scale_filters / pl_frame_mixers");
    if (strcmp(pl_color_transfer_names[PL_COLOR_TRC_PQ], pl_color_transfer_name(PL_COLOR_TRC_PQ)) ||
        !pl_color_system_names[PL_COLOR_SYSTEM_BT_709] || !pl_color_primaries_names[PL_COLOR_PRIM_BT_2020])
        die("colour name tables");
    // ICC: this build, like the reference built here, has no lcms2 (src/shaders/icc.c:802-836)
    if (pl_icc_open(NULL, &(struct pl_icc_profile) {0}, &pl_icc_default_params) ||
        pl_icc_default_params.intent != PL_INTENT_RELATIVE_COLORIMETRIC)
        die("pl_icc_open / pl_icc_default_params");
    // (deprecated) the gpu's cache through the renderer: no pl_cache was set, so there is nothing
    // to save -- and nothing to crash on
    if (pl_renderer_save(rr, NULL) != 0)
        die("pl_renderer_save without a cache");
    params.upscaler = preset->filter;
    // (the whole target is drawn over: what the clear leaves must not show)
    pl_frame_clear_rgba(gpu, &target, (const float[4]) {1.0f, 0.0f, 1.0f, 1.0f});
    if (!pl_render_image(rr, &image, &target, &params))
        die("pl_render_image failed");
    if (pl_renderer_get_errors(rr).errors)
        die("renderer reported errors");

    uint16_t *out = malloc((size_t) dw * dh * 4 * sizeof(uint16_t));
    if (!pl_tex_download(gpu, pl_tex_transfer_params(.tex = dst, .ptr = out)))
        die("download failed");

    FILE *f = fopen(argv[1], "wb");
    if (!f)
        die("cannot open the output file");
    fwrite(out, sizeof(uint16_t), (size_t) dw * dh * 4, f);

    // a small frame of its own, cleared to a colour and then to tiles
    enum { CW = 32, CH = 16 };
    pl_tex small = pl_tex_create(gpu, pl_tex_params(
        .w = CW, .h = CH, .format = fmt,
        .renderable = true, .storable = true, .blit_dst = true, .host_readable = true,
    ));
    if (!small)
        die("texture creation failed");
    struct pl_frame cleared = target;
    cleared.planes[0].texture = small;
    uint16_t texels[CW * CH * 4];
    pl_frame_clear_rgba(gpu, &cleared, (const float[4]) {0.25f, 0.5f, 0.75f, 0.5f});
    if (!pl_tex_download(gpu, pl_tex_transfer_params(.tex = small, .ptr = texels)))
        die("download failed");
    fwrite(texels, sizeof(uint16_t), CW * CH * 4, f);
    pl_frame_clear_tiles(gpu, &cleared, pl_render_default_params.tile_colors, 4);
    if (!pl_tex_download(gpu, pl_tex_transfer_params(.tex = small, .ptr = texels)))
        die("download failed");
    fwrite(texels, sizeof(uint16_t), CW * CH * 4, f);
    fclose(f);
    pl_tex_destroy(gpu, &small);

    uint64_t sum = 0;
    for (size_t i = 0; i < (size_t) dw * dh * 4; i++)
        sum += out[i];
    printf("%dx%d -> %dx%d sum=%llu sizeof(pl_frame)=%zu sizeof(pl_render_params)=%zu\n",
           sw, sh, dw, dh, (unsigned long long) sum, sizeof(struct pl_frame),
           sizeof(struct pl_render_params));

    free(out);
    free(pixels);
    pl_renderer_destroy(&rr);
    pl_tex_destroy(gpu, &src);
    pl_tex_destroy(gpu, &dst);
    pl_hip_destroy(&hip);
    pl_log_destroy(&log);
    return 0;
}
