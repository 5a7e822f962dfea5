/*
 * TEST / BENCH INFRASTRUCTURE ONLY — the CPU baseline of BASELINE.md section 2.
 *
 * This file is OUR glue: per-pixel loops that drive the REFERENCE's own CPU code
 * (filters.c, tone_mapping.c, gamut_mapping.c, colorspace.c, dither.c, compiled as they lie
 * under /root/reference into oracle/_ref/libplref.so by oracle/build_ref.sh) the way the
 * GPU shaders of src/shaders/sampling.c:587-912 and src/shaders/colorspace.c:1612-2024 use the
 * tables those functions produce. It is compiled against the reference's headers and linked
 * into libplref.so; bench.py times it (`cpu_baseline.kind = "reference"`), rows distributed
 * over `threads` host threads. It is never part of the product.
 *
 * BASELINE.json configs[0]: plcb_ewa_r32f(direct = 1 / 0)  (pl_filter_sample per tap vs LUT)
 *          configs[2]:      plcb_ewa_rgb_dither
 *          configs[3]:      plcb_tone_map
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <omp.h>

#include <libplacebo/colorspace.h>
#include <libplacebo/dither.h>
#include <libplacebo/filters.h>
#include <libplacebo/gamut_mapping.h>
#include <libplacebo/tone_mapping.h>

#define EXPORT __attribute__((visibility("default")))

static inline int clampi(int v, int lo, int hi)
{
    return v < lo ? lo : v > hi ? hi : v;
}

static inline float lut_lerp(const float *lut, int n, float x)
{
    const float pos = fminf(fmaxf(x, 0.0f), 1.0f) * (n - 1);
    const int i = (int) pos;
    const float f = pos - i;
    return lut[i] * (1.0f - f) + lut[i < n - 1 ? i + 1 : i] * f;
}

/* EWA-Lanczos resample of one float channel. direct: weight of every tap =
 * pl_filter_sample(&pl_filter_ewa_lanczos, d) (double, j1); else the 256-entry LUT of
 * pl_filter_generate with linear interpolation (what the GPU path does). Returns the average
 * number of taps inside the radius per output pixel. */
EXPORT double plcb_ewa_r32f(const float *src, int sw, int sh, float *dst, int dw, int dh,
                            int direct, int threads)
{
    struct pl_filter_params fp = {
        .config = pl_filter_ewa_lanczos,
        .lut_entries = 256,
        .cutoff = 1e-3,
    };
    pl_filter filt = pl_filter_generate(NULL, &fp);
    if (!filt)
        return -1;
    const float radius = filt->radius;
    const int bound = ceilf(radius);
    double taps = 0;

    #pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : taps)
    for (int oy = 0; oy < dh; oy++) {
        for (int ox = 0; ox < dw; ox++) {
            const float tx = (ox + 0.5f) / dw * sw - 0.5f, ty = (oy + 0.5f) / dh * sh - 0.5f;
            const float fx = tx - floorf(tx), fy = ty - floorf(ty);
            const int bx = floorf(tx), by = floorf(ty);
            float acc = 0, wsum = 0;
            for (int y = 1 - bound; y <= bound; y++) {
                for (int x = 1 - bound; x <= bound; x++) {
                    const float d = hypotf(x - fx, y - fy);
                    if (d >= radius)
                        continue;
                    const float w = direct ? (float) pl_filter_sample(&fp.config, d)
                                           : lut_lerp(filt->weights, 256, d / radius);
                    const float c = src[(size_t) clampi(by + y, 0, sh - 1) * sw +
                                        clampi(bx + x, 0, sw - 1)];
                    acc += w * c;
                    wsum += w;
                    taps++;
                }
            }
            dst[(size_t) oy * dw + ox] = acc / wsum;
        }
    }
    pl_filter_free(&filt);
    return taps / ((double) dw * dh);
}

/* configs[2]: LUT EWA on three channels of an RGBA float image + blue-noise dither to `depth`
 * bits with M = pl_generate_blue_noise(64). */
EXPORT int plcb_ewa_rgb_dither(const float *src, int sw, int sh, float *dst, int dw, int dh,
                               int depth, int threads)
{
    struct pl_filter_params fp = {
        .config = pl_filter_ewa_lanczos,
        .lut_entries = 256,
        .cutoff = 1e-3,
    };
    pl_filter filt = pl_filter_generate(NULL, &fp);
    float *noise = malloc(64 * 64 * sizeof(float));
    if (!filt || !noise)
        return -1;
    pl_generate_blue_noise(noise, 64);
    const float radius = filt->radius, scale = (float) ((1 << depth) - 1);
    const int bound = ceilf(radius);

    #pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int oy = 0; oy < dh; oy++) {
        for (int ox = 0; ox < dw; ox++) {
            const float tx = (ox + 0.5f) / dw * sw - 0.5f, ty = (oy + 0.5f) / dh * sh - 0.5f;
            const float fx = tx - floorf(tx), fy = ty - floorf(ty);
            const int bx = floorf(tx), by = floorf(ty);
            float acc[3] = {0}, wsum = 0;
            for (int y = 1 - bound; y <= bound; y++) {
                for (int x = 1 - bound; x <= bound; x++) {
                    const float d = hypotf(x - fx, y - fy);
                    if (d >= radius)
                        continue;
                    const float w = lut_lerp(filt->weights, 256, d / radius);
                    const float *c = src + 4 * ((size_t) clampi(by + y, 0, sh - 1) * sw +
                                                clampi(bx + x, 0, sw - 1));
                    for (int k = 0; k < 3; k++)
                        acc[k] += w * c[k];
                    wsum += w;
                }
            }
            float *o = dst + 4 * ((size_t) oy * dw + ox);
            const float bias = noise[(oy & 63) * 64 + (ox & 63)];
            for (int k = 0; k < 3; k++)
                o[k] = floorf(scale * (acc[k] / wsum) + bias) / scale;
            o[3] = 1.0f;
        }
    }
    free(noise);
    pl_filter_free(&filt);
    return 0;
}

/* configs[3]: BT.2020 PQ -> BT.709 BT.1886 per pixel with the reference's own tables:
 * pl_color_linearize -> rgb2lms -> PQ -> IPT -> pl_tone_map_generate LUT -> chroma scaling ->
 * pl_gamut_map_generate 3-D LUT (trilinear) -> inverse -> pl_color_delinearize.
 * `img` is RGBA float, PQ-coded, modified in place. Returns the seconds spent generating the
 * two LUTs (reported separately) through *lut_seconds. */
EXPORT int plcb_tone_map(float *img, size_t npix, float src_peak_nits, int threads,
                         double *lut_seconds)
{
    struct pl_color_space src = pl_color_space_hdr10, dst = pl_color_space_bt709;
    src.hdr.max_luma = src_peak_nits;
    pl_color_space_infer_map(&src, &dst);

    const double t0 = omp_get_wtime();
    struct pl_tone_map_params tp = {
        .function = &pl_tone_map_spline,
        .constants = { PL_TONE_MAP_CONSTANTS },
        .input_scaling = PL_HDR_PQ,
        .output_scaling = PL_HDR_PQ,
        .lut_size = 256,
        .hdr = src.hdr,
    };
    pl_color_space_nominal_luma_ex(pl_nominal_luma_params(
        .color = &src, .metadata = PL_HDR_METADATA_ANY, .scaling = PL_HDR_PQ,
        .out_min = &tp.input_min, .out_max = &tp.input_max, .out_avg = &tp.input_avg));
    pl_color_space_nominal_luma_ex(pl_nominal_luma_params(
        .color = &dst, .metadata = PL_HDR_METADATA_HDR10, .scaling = PL_HDR_PQ,
        .out_min = &tp.output_min, .out_max = &tp.output_max));
    pl_tone_map_params_infer(&tp);
    float tone_lut[256];
    pl_tone_map_generate(tone_lut, &tp);

    enum { NI = 48, NC = 32, NH = 256 };
    struct pl_gamut_map_params gp = {
        .function = &pl_gamut_map_perceptual,
        .constants = { PL_GAMUT_MAP_CONSTANTS },
        .input_gamut = src.hdr.prim,
        .output_gamut = dst.hdr.prim,
        .min_luma = tp.output_min,
        .max_luma = tp.output_max,
        .lut_size_I = NI, .lut_size_C = NC, .lut_size_h = NH,
        .lut_stride = 3,
    };
    float *glut = malloc(sizeof(float) * 3 * NI * NC * NH);
    if (!glut)
        return -1;
    pl_gamut_map_generate(glut, &gp);
    if (lut_seconds)
        *lut_seconds = omp_get_wtime() - t0;

    const pl_matrix3x3 rgb2lms = pl_ipt_rgb2lms(pl_raw_primaries_get(src.primaries));
    const pl_matrix3x3 lms2rgb = pl_ipt_lms2rgb(pl_raw_primaries_get(dst.primaries));
    const float in_rng = tp.input_max - tp.input_min, out_rng = gp.max_luma - gp.min_luma;

    #pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + 4 * i;
        pl_color_linearize(&src, c);                        // PQ -> linear, 1.0 = 203 nits
        pl_matrix3x3_apply(&rgb2lms, c);
        for (int k = 0; k < 3; k++)                         // LMS (nits) -> PQ
            c[k] = pl_hdr_rescale(PL_HDR_NORM, PL_HDR_PQ, fmaxf(c[k], 0.0f));
        pl_matrix3x3_apply(&pl_ipt_lms2ipt, c);
        // tone map the intensity, scale the chroma with it
        const float i_orig = fmaxf(c[0], 1e-6f);
        c[0] = lut_lerp(tone_lut, 256, (c[0] - tp.input_min) / in_rng);
        const float k = fminf(i_orig / fmaxf(c[0], 1e-6f), c[0] / i_orig);
        c[1] *= k;
        c[2] *= k;
        // gamut map through the (I, C, h) lattice
        const float fi = fminf(fmaxf((c[0] - gp.min_luma) / out_rng, 0.0f), 1.0f) * (NI - 1);
        const float fc = fminf(fmaxf(2.0f * hypotf(c[1], c[2]), 0.0f), 1.0f) * (NC - 1);
        const float fh = (atan2f(c[2], c[1]) / (2 * (float) M_PI) + 0.5f) * (NH - 1);
        const int i0 = fi, c0 = fc, h0 = fh;
        const int i1 = i0 < NI - 1 ? i0 + 1 : i0, c1 = c0 < NC - 1 ? c0 + 1 : c0,
                  h1 = h0 < NH - 1 ? h0 + 1 : h0;
        const float wi = fi - i0, wc = fc - c0, wh = fh - h0;
        float out[3];
        for (int ch = 0; ch < 3; ch++) {
            #define G(h, cc, ii) glut[3 * (((size_t) (h) * NC + (cc)) * NI + (ii)) + ch]
            const float a = G(h0, c0, i0) * (1 - wi) + G(h0, c0, i1) * wi;
            const float b = G(h0, c1, i0) * (1 - wi) + G(h0, c1, i1) * wi;
            const float d = G(h1, c0, i0) * (1 - wi) + G(h1, c0, i1) * wi;
            const float e = G(h1, c1, i0) * (1 - wi) + G(h1, c1, i1) * wi;
            #undef G
            out[ch] = (a * (1 - wc) + b * wc) * (1 - wh) + (d * (1 - wc) + e * wc) * wh;
        }
        memcpy(c, out, sizeof(out));
        pl_matrix3x3_apply(&pl_ipt_ipt2lms, c);
        for (int k2 = 0; k2 < 3; k2++)                      // PQ -> LMS (1.0 = 203 nits)
            c[k2] = pl_hdr_rescale(PL_HDR_PQ, PL_HDR_NORM, fmaxf(c[k2], 0.0f));
        pl_matrix3x3_apply(&lms2rgb, c);
        pl_color_delinearize(&dst, c);
    }
    free(glut);
    return 0;
}

EXPORT int plcb_max_threads(void)
{
    return omp_get_max_threads();
}
