/*
 * Frame times of pl_render_image from a plain C caller (no Python between the frames): what an
 * application linking the library sees. bench.py drives the same entry point through ctypes, which
 * costs a few microseconds per call -- visible on the passes that take 20 us.
 *
 * usage: bench_frames [frames]     (1080p -> 4K, sources and targets resident, 8 rotating targets)
 * prints one line per workload: name, microseconds per frame, output Mpixels/s, and the time the
 * calls themselves took (equal to the frame time = the host is the bottleneck)
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <libplacebo/hip.h>
#include <libplacebo/renderer.h>
#include <libplacebo/shaders/dithering.h>

#define SW 1920
#define SH 1080
#define DW 3840
#define DH 2160
#define POOL 8

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static void die(const char *what)
{
    fprintf(stderr, "bench_frames: %s\n", what);
    exit(1);
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 300;
    pl_log log = pl_log_create(PL_API_VER, pl_log_params(.log_cb = pl_log_simple, .log_level = PL_LOG_WARN));
    pl_hip hip = pl_hip_create(log, pl_hip_params(.device = 0));
    if (!hip)
        die("no HIP device");
    pl_gpu gpu = hip->gpu;
    pl_fmt fmt = pl_find_named_fmt(gpu, "rgba16");

    // the reference's bench pattern (src/tests/bench.c:32-51), quantised to 16 bit
    uint16_t *pixels = malloc((size_t) SW * SH * 4 * sizeof(uint16_t));
    const double xc = (SW - 1) / 2.0, yc = (SH - 1) / 2.0, phi = 1.6180339887498948;
    const double fr = 0.1 * M_PI * 0.5 / sqrt(xc * xc + yc * yc), fg = fr / phi, fb = fg / phi;
    for (int y = 0; y < SH; y++) {
        for (int x = 0; x < SW; x++) {
            const double r2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
            uint16_t *px = &pixels[4 * ((size_t) y * SW + x)];
            px[0] = lrint(65535.0 * (0.5 * sin(fr * r2) + 0.5));
            px[1] = lrint(65535.0 * (0.5 * sin(fg * r2) + 0.5));
            px[2] = lrint(65535.0 * (0.5 * sin(fb * r2) + 0.5));
            px[3] = 65535;
        }
    }
    pl_tex src[POOL], dst[POOL];
    for (int i = 0; i < POOL; i++) {
        src[i] = pl_tex_create(gpu, pl_tex_params(.w = SW, .h = SH, .format = fmt, .sampleable = true,
                                                  .host_writable = true, .initial_data = pixels));
        dst[i] = pl_tex_create(gpu, pl_tex_params(.w = DW, .h = DH, .format = fmt, .renderable = true,
                                                  .storable = true, .host_readable = true));
        if (!src[i] || !dst[i])
            die("texture creation failed");
    }

    const struct pl_color_repr full16 = {
        .sys = PL_COLOR_SYSTEM_RGB, .levels = PL_COLOR_LEVELS_FULL,
        .bits = { .sample_depth = 16, .color_depth = 16 },
    };
    struct pl_color_repr ten_bit = full16;
    ten_bit.bits.color_depth = 10;
    ten_bit.bits.bit_shift = 6;
    struct pl_color_space hdr10 = pl_color_space_hdr10;
    hdr10.hdr.max_luma = 1000.0f;
    const struct pl_color_space bt1886 = { .primaries = PL_COLOR_PRIM_BT_709, .transfer = PL_COLOR_TRC_BT_1886 };

    struct pl_dither_params dither = pl_dither_default_params;
    struct pl_peak_detect_params peak = pl_peak_detect_default_params;
    peak.percentile = 99.995f;

    struct pl_render_params bilinear = pl_render_fast_params;
    struct pl_render_params ewa = pl_render_fast_params;
    ewa.upscaler = &pl_filter_ewa_lanczos;
    ewa.dither_params = &dither;
    ewa.disable_dither_gamma_correction = true;
    struct pl_render_params metric = pl_render_default_params;
    metric.upscaler = &pl_filter_ewa_lanczos;
    metric.dither_params = &dither;
    metric.peak_detect_params = &peak;

    const struct {
        const char *name;
        const struct pl_render_params *params;
        struct pl_color_space in, out;
        struct pl_color_repr out_repr;
    } work[] = {
        { "bilinear_1080p_to_4k", &bilinear, pl_color_space_srgb, pl_color_space_srgb, full16 },
        { "ewa_lanczos_1080p_to_4k_dither10", &ewa, pl_color_space_srgb, pl_color_space_srgb, ten_bit },
        { "ewa_1080p_to_4k_hdr_tonemap", &metric, hdr10, bt1886, ten_bit },
    };

    for (size_t w = 0; w < sizeof(work) / sizeof(work[0]); w++) {
        pl_renderer rr = pl_renderer_create(log, gpu);
        struct pl_frame image = {
            .num_planes = 1,
            .planes = {{ .components = 3, .component_mapping = {0, 1, 2} }},
            .repr = full16,
            .color = work[w].in,
        };
        struct pl_frame target = {
            .num_planes = 1,
            .planes = {{ .components = 4, .component_mapping = {0, 1, 2, 3} }},
            .repr = work[w].out_repr,
            .color = work[w].out,
        };
        double t0 = 0;
        for (int f = -30; f < frames; f++) {
            if (f == 0) {
                pl_gpu_finish(gpu);
                t0 = now_us();
            }
            image.planes[0].texture = src[(f + 30) % POOL];
            target.planes[0].texture = dst[(f + 30) % POOL];
            if (!pl_render_image(rr, &image, &target, work[w].params))
                die("pl_render_image failed");
        }
        const double host_us = (now_us() - t0) / frames;     // the calls alone: what the host spends per frame
        pl_gpu_finish(gpu);
        const double us = (now_us() - t0) / frames;
        if (pl_renderer_get_errors(rr).errors)
            die("renderer reported errors");
        printf("%-36s %8.2f us/frame %10.1f Mpx/s   (host: %.2f us per call)\n", work[w].name, us,
               (double) DW * DH / us, host_us);
        pl_renderer_destroy(&rr);
    }

    for (int i = 0; i < POOL; i++) {
        pl_tex_destroy(gpu, &src[i]);
        pl_tex_destroy(gpu, &dst[i]);
    }
    free(pixels);
    pl_hip_destroy(&hip);
    pl_log_destroy(&log);
    return 0;
}
