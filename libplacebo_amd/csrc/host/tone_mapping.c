/*
 * libplacebo-hip — Tier-0 host maths: tone-mapping curves -> 1-D LUT.
 *
 * Fresh implementation of the behaviour of the reference's src/tone_mapping.c:
 *   parameter inference / rescaling   tone_mapping.c:25-129
 *   pl_tone_map_generate / _sample    :147-178
 *   knee selection (ST 2094-10 style) :228-267
 *   curves: st2094-40 :299, st2094-10 :420, bt2390 :462, bt2446a :507-542,
 *           spline :552, reinhard :613, mobius :638, hable :667, gamma :693,
 *           linear :716
 * The 256-entry LUT is regenerated per frame when peak detection is active
 * (2.8 us); all arithmetic is float in the reference's order so the LUT is
 * bit-identical (tests/test_tier0_ref.py).
 */
#include <math.h>
#include <string.h>

#include <libplacebo/tone_mapping.h>

#include "host_common.h"

#define MIXF(a, b, x) ((x) * (b) + (1 - (x)) * (a))
#define CLAMPF(x, lo, hi) fminf(fmaxf(x, lo), hi)

static inline float smoothstepf(float edge0, float edge1, float x)
{
    if (edge0 == edge1)
        return x >= edge0;
    x = (x - edge0) / (edge1 - edge0);
    x = PL_CLAMP(x, 0.0f, 1.0f);
    return x * x * (3.0f - 2.0f * x);
}

static void sanitize_constants(struct pl_tone_map_constants *c)
{
    const float eps = 1e-6f;
    c->knee_adaptation   = CLAMPF(c->knee_adaptation, 0.0f, 1.0f);
    c->knee_minimum      = CLAMPF(c->knee_minimum, eps, 0.5f - eps);
    c->knee_maximum      = CLAMPF(c->knee_maximum, 0.5f + eps, 1.0f - eps);
    c->knee_default      = CLAMPF(c->knee_default, c->knee_minimum, c->knee_maximum);
    c->knee_offset       = CLAMPF(c->knee_offset, 0.5f, 2.0f);
    c->slope_tuning      = CLAMPF(c->slope_tuning, 0.0f, 10.0f);
    c->slope_offset      = CLAMPF(c->slope_offset, 0.0f, 1.0f);
    c->spline_contrast   = CLAMPF(c->spline_contrast, 0.0f, 1.5f);
    c->reinhard_contrast = CLAMPF(c->reinhard_contrast, eps, 1.0f - eps);
    c->linear_knee       = CLAMPF(c->linear_knee, eps, 1.0f - eps);
    c->exposure          = CLAMPF(c->exposure, eps, 10.0f);
}

bool pl_tone_map_params_equal(const struct pl_tone_map_params *a,
                              const struct pl_tone_map_params *b)
{
    return a->function == b->function && a->param == b->param &&
           a->input_scaling == b->input_scaling && a->output_scaling == b->output_scaling &&
           a->lut_size == b->lut_size &&
           a->input_min == b->input_min && a->input_max == b->input_max &&
           a->input_avg == b->input_avg &&
           a->output_min == b->output_min && a->output_max == b->output_max &&
           !memcmp(&a->constants, &b->constants, sizeof(a->constants)) &&
           pl_hdr_metadata_equal(&a->hdr, &b->hdr);
}

bool pl_tone_map_params_noop(const struct pl_tone_map_params *p)
{
    const float in_min  = pl_hdr_rescale(p->input_scaling,  PL_HDR_NITS, p->input_min);
    const float in_max  = pl_hdr_rescale(p->input_scaling,  PL_HDR_NITS, p->input_max);
    const float out_min = pl_hdr_rescale(p->output_scaling, PL_HDR_NITS, p->output_min);
    const float out_max = pl_hdr_rescale(p->output_scaling, PL_HDR_NITS, p->output_max);
    const bool can_inverse = p->function->map_inverse;

    return fabs(in_min - out_min) < 1e-4 &&             // same black point
           in_max < out_max + 1e-2 &&                   // no range reduction
           (out_max < in_max + 1e-2 || !can_inverse);   // no expansion
}

void pl_tone_map_params_infer(struct pl_tone_map_params *par)
{
    if (!par->function)
        par->function = &pl_tone_map_clip;

    if (par->param) {
        // legacy single-parameter API
        const struct pl_tone_map_function *f = par->function;
        if (f == &pl_tone_map_st2094_40 || f == &pl_tone_map_st2094_10)
            par->constants.knee_adaptation = par->param;
        if (f == &pl_tone_map_bt2390)
            par->constants.knee_offset = par->param;
        if (f == &pl_tone_map_spline)
            par->constants.spline_contrast = par->param;
        if (f == &pl_tone_map_reinhard)
            par->constants.reinhard_contrast = par->param;
        if (f == &pl_tone_map_mobius || f == &pl_tone_map_gamma)
            par->constants.linear_knee = par->param;
        if (f == &pl_tone_map_linear || f == &pl_tone_map_linear_light)
            par->constants.exposure = par->param;
    }

    sanitize_constants(&par->constants);

    // the source peak is never assumed to be below min(100 nits, target peak)
    float sdr = pl_hdr_rescale(par->output_scaling, par->input_scaling, par->output_max);
    sdr = fminf(sdr, pl_hdr_rescale(PL_HDR_NITS, par->input_scaling, 100));
    par->input_max = fmaxf(par->input_max, sdr);

    // curves without an inverse cannot expand the range
    if (!par->function->map_inverse) {
        const float in_max = pl_hdr_rescale(par->input_scaling, par->output_scaling,
                                            par->input_max);
        par->output_max = fminf(par->output_max, in_max);
    }
}

// inferred parameters, expressed in the curve's own scaling
static struct pl_tone_map_params curve_params(const struct pl_tone_map_params *params)
{
    struct pl_tone_map_params p = *params;
    pl_tone_map_params_infer(&p);

    const enum pl_hdr_scaling s = p.function->scaling;
    p.input_scaling = p.output_scaling = s;
    p.input_min  = pl_hdr_rescale(params->input_scaling,  s, p.input_min);
    p.input_max  = pl_hdr_rescale(params->input_scaling,  s, p.input_max);
    p.input_avg  = pl_hdr_rescale(params->input_scaling,  s, p.input_avg);
    p.output_min = pl_hdr_rescale(params->output_scaling, s, p.output_min);
    p.output_max = pl_hdr_rescale(params->output_scaling, s, p.output_max);
    return p;
}

static void run_curve(float *lut, const struct pl_tone_map_params *p)
{
    if (p->output_max > p->input_max + 1e-4) {
        p->function->map_inverse(lut, p);
    } else {
        p->function->map(lut, p);
    }
}

void pl_tone_map_generate(float *out, const struct pl_tone_map_params *params)
{
    const struct pl_tone_map_params p = curve_params(params);
    const size_t n = params->lut_size;

    // sample points evenly spaced in the *caller's* input scaling
    for (size_t i = 0; i < n; i++) {
        float x = (float) i / (n - 1);
        x = MIXF(params->input_min, params->input_max, x);
        out[i] = pl_hdr_rescale(params->input_scaling, p.function->scaling, x);
    }

    run_curve(out, &p);

    for (size_t i = 0; i < n; i++) {
        const float x = PL_CLAMP(out[i], p.output_min, p.output_max);
        out[i] = pl_hdr_rescale(p.function->scaling, params->output_scaling, x);
    }
}

float pl_tone_map_sample(float x, const struct pl_tone_map_params *params)
{
    struct pl_tone_map_params p = curve_params(params);
    p.lut_size = 1;

    x = PL_CLAMP(x, params->input_min, params->input_max);
    x = pl_hdr_rescale(params->input_scaling, p.function->scaling, x);
    run_curve(&x, &p);
    x = PL_CLAMP(x, p.output_min, p.output_max);
    return pl_hdr_rescale(p.function->scaling, params->output_scaling, x);
}

/* ------------------------------------------------------------------------ */
/* helpers shared by the curves                                              */

#define LUT_LOOP(lut, p, x) \
    for (float *it_ = (lut), *end_ = (lut) + (p)->lut_size, x; \
         it_ < end_ && (x = *it_, 1); *it_++ = x)

// input-absolute -> input-relative
static inline float rel_in(float x, const struct pl_tone_map_params *p)
{
    return (x - p->input_min) / (p->input_max - p->input_min);
}

// input-absolute -> output-relative
static inline float rel_out(float x, const struct pl_tone_map_params *p)
{
    return (x - p->input_min) / (p->output_max - p->output_min);
}

// output-relative -> output-absolute
static inline float abs_out(float x, const struct pl_tone_map_params *p)
{
    return x * (p->output_max - p->output_min) + p->output_min;
}

static inline float bt1886_eotf(float x, float min, float max)
{
    const float lb = powf(min, 1/2.4f);
    const float lw = powf(max, 1/2.4f);
    return powf((lw - lb) * x + lb, 2.4f);
}

static inline float bt1886_oetf(float x, float min, float max)
{
    const float lb = powf(min, 1/2.4f);
    const float lw = powf(max, 1/2.4f);
    return (powf(x, 1/2.4f) - lb) / (lw - lb);
}

// Knee (pivot) selection in PQ space from the scene average, pulled towards
// the 1:1 line by `knee_adaptation` (ST 2094-10 inspired)
static void pick_knee(float *out_src_knee, float *out_dst_knee,
                      const struct pl_tone_map_params *p)
{
    const float src_min = pl_hdr_rescale(p->input_scaling,  PL_HDR_PQ, p->input_min);
    const float src_max = pl_hdr_rescale(p->input_scaling,  PL_HDR_PQ, p->input_max);
    const float src_avg = pl_hdr_rescale(p->input_scaling,  PL_HDR_PQ, p->input_avg);
    const float dst_min = pl_hdr_rescale(p->output_scaling, PL_HDR_PQ, p->output_min);
    const float dst_max = pl_hdr_rescale(p->output_scaling, PL_HDR_PQ, p->output_max);

    const float min_knee = p->constants.knee_minimum;
    const float max_knee = p->constants.knee_maximum;
    const float def_knee = p->constants.knee_default;
    const float src_knee_min = MIXF(src_min, src_max, min_knee);
    const float src_knee_max = MIXF(src_min, src_max, max_knee);
    const float dst_knee_min = MIXF(dst_min, dst_max, min_knee);
    const float dst_knee_max = MIXF(dst_min, dst_max, max_knee);

    float src_knee = PL_DEF(src_avg, MIXF(src_min, src_max, def_knee));
    src_knee = CLAMPF(src_knee, src_knee_min, src_knee_max);

    // where the same relative position lands in the target range
    const float target = (src_knee - src_min) / (src_max - src_min);
    const float adapted = MIXF(dst_min, dst_max, target);

    // adapt more strongly the closer the knee is to its allowed extremes
    const float tuning = 1.0f - smoothstepf(max_knee, def_knee, target) *
                                smoothstepf(min_knee, def_knee, target);
    const float adaptation = MIXF(p->constants.knee_adaptation, 1.0f, tuning);
    float dst_knee = MIXF(src_knee, adapted, adaptation);
    dst_knee = CLAMPF(dst_knee, dst_knee_min, dst_knee_max);

    *out_src_knee = pl_hdr_rescale(PL_HDR_PQ, p->input_scaling, src_knee);
    *out_dst_knee = pl_hdr_rescale(PL_HDR_PQ, p->output_scaling, dst_knee);
}

/* ------------------------------------------------------------------------ */
/* curves                                                                    */

static void map_identity(float *lut, const struct pl_tone_map_params *p)
{
    (void) lut; (void) p;
}

static const uint16_t pascal[17][17] = {
    {1},
    {1,1},
    {1,2,1},
    {1,3,3,1},
    {1,4,6,4,1},
    {1,5,10,10,5,1},
    {1,6,15,20,15,6,1},
    {1,7,21,35,35,21,7,1},
    {1,8,28,56,70,56,28,8,1},
    {1,9,36,84,126,126,84,36,9,1},
    {1,10,45,120,210,252,210,120,45,10,1},
    {1,11,55,165,330,462,462,330,165,55,11,1},
    {1,12,66,220,495,792,924,792,495,220,66,12,1},
    {1,13,78,286,715,1287,1716,1716,1287,715,286,78,13,1},
    {1,14,91,364,1001,2002,3003,3432,3003,2002,1001,364,91,14,1},
    {1,15,105,455,1365,3003,5005,6435,6435,5005,3003,1365,455,105,15,1},
    {1,16,120,560,1820,4368,8008,11440,12870,11440,8008,4368,1820,560,120,16,1},
};

// first Bezier anchor that matches the slope of the linear segment at the knee
static inline float bezier_intercept(uint8_t N, float Kx, float Ky)
{
    if (Kx <= 0 || Ky >= 1)
        return 1.0f / N;
    const float slope = Ky / Kx * (1 - Kx) / (1 - Ky);
    return fminf(slope / N, 1.0f);
}

static void map_st2094_40(float *lut, const struct pl_tone_map_params *p)
{
    const float D = p->output_max;
    float P[17], Kx, Ky, T;
    uint8_t N;

    if (p->hdr.ootf.num_anchors) {
        // curve from HDR10+ metadata
        Kx = PL_CLAMP(p->hdr.ootf.knee_x, 0, 1);
        Ky = PL_CLAMP(p->hdr.ootf.knee_y, 0, 1);
        T = PL_CLAMP(p->hdr.ootf.target_luma, p->input_min, p->input_max);
        N = p->hdr.ootf.num_anchors + 1;
        memcpy(P + 1, p->hdr.ootf.anchors, (N - 1) * sizeof(*P));
        P[0] = 0.0f;
        P[N] = 1.0f;
    } else {
        // no metadata: brightness matching through the picked knee
        float src_knee, dst_knee;
        pick_knee(&src_knee, &dst_knee, p);
        Kx = src_knee / p->input_max;
        Ky = dst_knee / p->output_max;

        const float slope = Ky / Kx * (1 - Kx) / (1 - Ky);
        N = PL_CLAMP((int) ceilf(slope), 2, (int) PL_ARRAY_SIZE(P) - 1);
        P[0] = 0.0f;
        P[1] = bezier_intercept(N, Kx, Ky);
        for (int i = 2; i <= N; i++)
            P[i] = 1.0f;
        T = D;
    }

    if (D < T) {
        // display darker than the curve's target: brighten
        const float Dmin = 0.0f, u = fmaxf(0.0f, (D - Dmin) / (T - Dmin));
        Kx *= u;
        Ky *= u;

        const float beta = N * Kx / (1 - Kx);
        const float Kxy = fminf(Kx * p->input_max / D, beta / (beta + 1));
        Ky = MIXF(Kxy, Ky, u);

        for (int k = 2; k <= N; k++)
            P[k] = MIXF(1.0f, P[k], u);
        P[1] = MIXF(bezier_intercept(N, Kx, Ky), P[1], u);
    } else if (D > T) {
        // display brighter than the target: linearise
        const float w = powf(1 - (D - T) / (p->input_max - T), 1.4f);
        Ky *= T / D;

        const float Kxy = Kx * D / p->input_max;
        Ky = MIXF(Kxy, Ky, w);

        for (int k = 2; k < N; k++) {
            const float anchor_lin = (float) k / N;
            P[k] = MIXF(anchor_lin, P[k], w);
        }
        P[1] = MIXF(bezier_intercept(N, Kx, Ky), P[1], w);
    }

    LUT_LOOP(lut, p, x) {
        x = bt1886_oetf(x, p->input_min, p->input_max);
        x = bt1886_eotf(x, 0.0f, 1.0f);

        if (x <= Kx && Kx) {
            x *= Ky / Kx; // linear segment
        } else {
            const float t = (x - Kx) / (1 - Kx);
            x = 0; // Bernstein sum
            for (uint8_t k = 0; k <= N; k++)
                x += pascal[N][k] * powf(t, k) * powf(1 - t, N - k) * P[k];
            x = Ky + (1 - Ky) * x;
        }

        x = bt1886_oetf(x, 0.0f, 1.0f);
        x = bt1886_eotf(x, p->output_min, p->output_max);
    }
}

static void map_st2094_10(float *lut, const struct pl_tone_map_params *p)
{
    float src_knee, dst_knee;
    pick_knee(&src_knee, &dst_knee, p);

    // rational curve through (x1,y1), (x2,y2), (x3,y3)
    const float x1 = p->input_min,  x3 = p->input_max,  x2 = src_knee;
    const float y1 = p->output_min, y3 = p->output_max, y2 = dst_knee;

    const pl_matrix3x3 cmat = {{
        { x2*x3*(y2 - y3), x1*x3*(y3 - y1), x1*x2*(y1 - y2) },
        { x3*y3 - x2*y2,   x1*y1 - x3*y3,   x2*y2 - x1*y1   },
        { x3 - x2,         x1 - x3,         x2 - x1         },
    }};

    float coeffs[3] = { y1, y2, y3 };
    pl_matrix3x3_apply(&cmat, coeffs);

    const float k = 1.0 / (x3*y3*(x1 - x2) + x2*y2*(x3 - x1) + x1*y1*(x2 - x3));
    const float c1 = k * coeffs[0];
    const float c2 = k * coeffs[1];
    const float c3 = k * coeffs[2];

    LUT_LOOP(lut, p, x)
        x = (c1 + c2 * x) / (1 + c3 * x);
}

static void map_bt2390(float *lut, const struct pl_tone_map_params *p)
{
    const float minLum = rel_in(p->output_min, p);
    const float maxLum = rel_in(p->output_max, p);
    const float offset = p->constants.knee_offset;
    const float ks = (1 + offset) * maxLum - offset;
    const float bp = minLum > 0 ? fminf(1 / minLum, 4) : 4;
    const float gain_inv = 1 + minLum / maxLum * powf(1 - maxLum, bp);
    const float gain = maxLum < 1 ? 1 / gain_inv : 1;

    LUT_LOOP(lut, p, x) {
        x = rel_in(x, p);

        if (ks < 1) {
            // hermite roll-off above the knee
            const float tb = (x - ks) / (1 - ks);
            const float tb2 = tb * tb;
            const float tb3 = tb2 * tb;
            const float pb = (2 * tb3 - 3 * tb2 + 1) * ks +
                             (tb3 - 2 * tb2 + tb) * (1 - ks) +
                             (-2 * tb3 + 3 * tb2) * maxLum;
            x = x < ks ? x : pb;
        }

        if (x < 1) {
            // black point lift
            x += minLum * powf(1 - x, bp);
            x = gain * (x - minLum) + minLum;
        }

        x = x * (p->input_max - p->input_min) + p->input_min;
    }
}

static void map_bt2446a(float *lut, const struct pl_tone_map_params *p)
{
    const float phdr = 1 + 32 * powf(p->input_max / 10000, 1/2.4f);
    const float psdr = 1 + 32 * powf(p->output_max / 10000, 1/2.4f);

    LUT_LOOP(lut, p, x) {
        x = powf(rel_in(x, p), 1/2.4f);
        x = logf(1 + (phdr - 1) * x) / logf(phdr);

        if (x <= 0.7399f) {
            x = 1.0770f * x;
        } else if (x < 0.9909f) {
            x = (-1.1510f * x + 2.7811f) * x - 0.6302f;
        } else {
            x = 0.5f * x + 0.5f;
        }

        x = (powf(psdr, x) - 1) / (psdr - 1);
        x = bt1886_eotf(x, p->output_min, p->output_max);
    }
}

static void map_bt2446a_inv(float *lut, const struct pl_tone_map_params *p)
{
    LUT_LOOP(lut, p, x) {
        x = bt1886_oetf(x, p->input_min, p->input_max);
        x *= 255.0;
        if (x > 70) {
            x = powf(x, (2.8305e-6f * x - 7.4622e-4f) * x + 1.2528f);
        } else {
            x = powf(x, (1.8712e-5f * x - 2.7334e-3f) * x + 1.3141f);
        }
        x = powf(x / 1000, 2.4f);
        x = abs_out(x, p);
    }
}

static void map_spline(float *lut, const struct pl_tone_map_params *p)
{
    float src_pivot, dst_pivot;
    pick_knee(&src_pivot, &dst_pivot, p);

    // slope of the straight line black -> pivot ...
    float slope = (dst_pivot - p->output_min) / (src_pivot - p->input_min);

    // ... softened: exponent shrinks towards 0 (slope -> 1) for small peak
    // differences, grows towards 1 (linear) for large ones
    float ratio = p->input_max / p->output_max - 1.0f;
    ratio = CLAMPF(p->constants.slope_tuning * ratio, p->constants.slope_offset,
                   1.0f + p->constants.slope_offset);
    slope = powf(slope, (1.0f - p->constants.spline_contrast) * ratio);

    // coordinates relative to the pivot
    const float in_min = p->input_min - src_pivot;
    const float in_max = p->input_max - src_pivot;
    const float out_min = p->output_min - dst_pivot;
    const float out_max = p->output_max - dst_pivot;

    // below: quadratic P with P(in_min) = out_min, P(0) = 0, P'(0) = slope
    const float Pa = (out_min - slope * in_min) / (in_min * in_min);
    const float Pb = slope;

    // above: cubic Q with Q(in_max) = out_max, Q''(in_max) = 0, Q(0) = 0, Q'(0) = slope
    const float t = 2 * in_max * in_max;
    const float Qa = (slope * in_max - out_max) / (in_max * t);
    const float Qb = -3 * (slope * in_max - out_max) / t;
    const float Qc = slope;

    LUT_LOOP(lut, p, x) {
        x -= src_pivot;
        x = x > 0 ? ((Qa * x + Qb) * x + Qc) * x : (Pa * x + Pb) * x;
        x += dst_pivot;
    }
}

static void map_reinhard(float *lut, const struct pl_tone_map_params *p)
{
    const float peak = rel_out(p->input_max, p),
                contrast = p->constants.reinhard_contrast,
                offset = (1.0 - contrast) / contrast,
                scale = (peak + offset) / peak;

    LUT_LOOP(lut, p, x) {
        x = rel_out(x, p);
        x = x / (x + offset);
        x *= scale;
        x = abs_out(x, p);
    }
}

static void map_mobius(float *lut, const struct pl_tone_map_params *p)
{
    const float peak = rel_out(p->input_max, p),
                j = p->constants.linear_knee;

    // M(x) = scale * (x+a)/(x+b) with M(j) = j, M'(j) = 1, M(peak) = 1
    const float a = -j*j * (peak - 1.0f) / (j*j - 2.0f * j + peak);
    const float b = (j*j - 2.0f * j * peak + peak) / fmaxf(1e-6f, peak - 1.0f);
    const float scale = (b*b + 2.0f * b*j + j*j) / (b - a);

    LUT_LOOP(lut, p, x) {
        x = rel_out(x, p);
        x = x <= j ? x : scale * (x + a) / (x + b);
        x = abs_out(x, p);
    }
}

static inline float hable_curve(float x)
{
    const float A = 0.15, B = 0.50, C = 0.10, D = 0.20, E = 0.02, F = 0.30;
    return ((x * (A*x + C*B) + D*E) / (x * (A*x + B) + D*F)) - E/F;
}

static void map_hable(float *lut, const struct pl_tone_map_params *p)
{
    const float peak = p->input_max / p->output_max,
                scale = 1.0f / hable_curve(peak);

    LUT_LOOP(lut, p, x) {
        x = bt1886_oetf(x, p->input_min, p->input_max);
        x = bt1886_eotf(x, 0, peak);
        x = scale * hable_curve(x);
        x = bt1886_oetf(x, 0, 1);
        x = bt1886_eotf(x, p->output_min, p->output_max);
    }
}

static void map_gamma(float *lut, const struct pl_tone_map_params *p)
{
    const float peak = rel_out(p->input_max, p),
                cutoff = p->constants.linear_knee,
                gamma = logf(cutoff) / logf(cutoff / peak);

    LUT_LOOP(lut, p, x) {
        x = rel_out(x, p);
        x = x > cutoff ? powf(x / peak, gamma) : x;
        x = abs_out(x, p);
    }
}

static void map_linear(float *lut, const struct pl_tone_map_params *p)
{
    const float gain = p->constants.exposure;

    LUT_LOOP(lut, p, x) {
        x = rel_in(x, p);
        x *= gain;
        x = abs_out(x, p);
    }
}

/* ------------------------------------------------------------------------ */

// The trailing four values are the legacy single-parameter description (deprecated since
// v6.311; the curves themselves read pl_tone_map_constants). Kept for programs that list them.
#define CURVE(sym, nm, desc, scal, fwd, inv, ...) \
    const struct pl_tone_map_function sym = { .name = nm, .description = desc, \
        .scaling = scal, .map = fwd, .map_inverse = inv, __VA_ARGS__ }
#define LEGACY(what, lo, def, hi) \
    .param_desc = what, .param_min = lo, .param_def = def, .param_max = hi

CURVE(pl_tone_map_clip,      "clip",      "No tone mapping (clip)",        PL_HDR_NORM, map_identity, map_identity);
CURVE(pl_tone_map_st2094_40, "st2094-40", "SMPTE ST 2094-40 Annex B",      PL_HDR_NITS, map_st2094_40, NULL,
      LEGACY("Knee point target", 0.00f, 0.70f, 1.00f));
CURVE(pl_tone_map_st2094_10, "st2094-10", "SMPTE ST 2094-10 Annex B.2",    PL_HDR_NITS, map_st2094_10, NULL,
      LEGACY("Knee point target", 0.00f, 0.70f, 1.00f));
CURVE(pl_tone_map_bt2390,    "bt2390",    "ITU-R BT.2390 EETF",            PL_HDR_PQ,   map_bt2390, NULL,
      LEGACY("Knee offset", 0.50, 1.00, 2.00));
CURVE(pl_tone_map_bt2446a,   "bt2446a",   "ITU-R BT.2446 Method A",        PL_HDR_NITS, map_bt2446a, map_bt2446a_inv);
CURVE(pl_tone_map_spline,    "spline",    "Single-pivot polynomial spline", PL_HDR_PQ,  map_spline, map_spline,
      LEGACY("Contrast", 0.00f, 0.50f, 1.50f));
CURVE(pl_tone_map_reinhard,  "reinhard",  "Reinhard",                      PL_HDR_NORM, map_reinhard, NULL,
      LEGACY("Contrast", 0.001, 0.50, 0.99));
CURVE(pl_tone_map_mobius,    "mobius",    "Mobius",                        PL_HDR_NORM, map_mobius, NULL,
      LEGACY("Knee point", 0.00, 0.30, 0.99));
CURVE(pl_tone_map_hable,     "hable",     "Filmic tone-mapping (Hable)",   PL_HDR_NORM, map_hable, NULL);
CURVE(pl_tone_map_gamma,     "gamma",     "Gamma function with knee",      PL_HDR_NORM, map_gamma, NULL,
      LEGACY("Knee point", 0.001, 0.30, 1.00));
CURVE(pl_tone_map_linear,    "linear",    "Perceptually linear stretch",   PL_HDR_PQ,   map_linear, map_linear,
      LEGACY("Exposure", 0.001, 1.00, 10.0));
CURVE(pl_tone_map_linear_light, "linearlight", "Linear light stretch",     PL_HDR_NORM, map_linear, map_linear,
      LEGACY("Exposure", 0.001, 1.00, 10.0));

const struct pl_tone_map_function * const pl_tone_map_functions[] = {
    &pl_tone_map_clip,
    &pl_tone_map_st2094_40,
    &pl_tone_map_st2094_10,
    &pl_tone_map_bt2390,
    &pl_tone_map_bt2446a,
    &pl_tone_map_spline,
    &pl_tone_map_reinhard,
    &pl_tone_map_mobius,
    &pl_tone_map_hable,
    &pl_tone_map_gamma,
    &pl_tone_map_linear,
    &pl_tone_map_linear_light,
    NULL
};

const int pl_num_tone_map_functions = PL_ARRAY_SIZE(pl_tone_map_functions) - 1;

const struct pl_tone_map_function *pl_find_tone_map_function(const char *name)
{
    for (int i = 0; name && i < pl_num_tone_map_functions; i++) {
        if (!strcmp(name, pl_tone_map_functions[i]->name))
            return pl_tone_map_functions[i];
    }
    return NULL;
}
