/*
 * libplacebo-hip — Tier-0 host maths: gamut mapping in IPTPQc4 -> 3-D LUT.
 *
 * Fresh implementation of the behaviour of the reference's src/gamut_mapping.c:
 *   lattice generation + worker split   gamut_mapping.c:372-443
 *   IPT <-> RGB through a tabulated PQ   :242-338, in-gamut test :340-370
 *   boundary search (bisection / golden section with a 1-entry peak cache)
 *                                         :487-545
 *   mappers: perceptual :711, softclip :748 (+ hue-shift spline :613-709),
 *            relative :816, desaturate :833, saturation :849, absolute :866,
 *            highlight :889, linear :910, darken :933, clip :968
 *
 * Bit-exactness notes (the 48x32x256 LUT is compared entry by entry with the
 * reference in tests/test_tier0_ref.py):
 *   - the per-hue peak cache makes results depend on evaluation order, so the
 *     lattice is processed in the same chunks (ceil(h/32) hue slices per
 *     worker, fresh cache per worker) and the same I-fastest order;
 *   - the PQ EOTF goes through the same 1024-entry table (pq_eotf_table.inc).
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/gamut_mapping.h>

#include "host_common.h"
#include "colorspace_priv.h"

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

static void sanitize_constants(struct pl_gamut_map_constants *c)
{
    c->perceptual_deadzone = CLAMPF(c->perceptual_deadzone, 0.0f, 1.0f);
    c->perceptual_strength = CLAMPF(c->perceptual_strength, 0.0f, 1.0f);
    c->colorimetric_gamma  = CLAMPF(c->colorimetric_gamma, 0.0f, 10.0f);
    c->softclip_knee       = CLAMPF(c->softclip_knee, 0.0f, 1.0f);
    c->softclip_desat      = CLAMPF(c->softclip_desat, 0.0f, 1.0f);
}

bool pl_gamut_map_params_equal(const struct pl_gamut_map_params *a,
                               const struct pl_gamut_map_params *b)
{
    return a->function == b->function &&
           a->min_luma == b->min_luma && a->max_luma == b->max_luma &&
           a->lut_size_I == b->lut_size_I && a->lut_size_C == b->lut_size_C &&
           a->lut_size_h == b->lut_size_h && a->lut_stride == b->lut_stride &&
           !memcmp(&a->constants, &b->constants, sizeof(a->constants)) &&
           pl_raw_primaries_equal(&a->input_gamut, &b->input_gamut) &&
           pl_raw_primaries_equal(&a->output_gamut, &b->output_gamut);
}

static const struct pl_gamut_map_function *fn_of(const struct pl_gamut_map_params *p)
{
    return p->function ? p->function : &pl_gamut_map_clip;
}

static void map_clip(float *lut, const struct pl_gamut_map_params *p);

bool pl_gamut_map_params_noop(const struct pl_gamut_map_params *params)
{
    if (fn_of(params)->map == &map_clip)
        return true;

    const struct pl_raw_primaries src = params->input_gamut, dst = params->output_gamut;
    if (!pl_primaries_compatible(&dst, &src))
        return true;

    bool need_map = !pl_primaries_superset(&dst, &src);
    need_map |= !pl_cie_xy_equal(&src.white, &dst.white);
    if (fn_of(params)->bidirectional)
        need_map |= !pl_raw_primaries_equal(&dst, &src);
    return !need_map;
}

/* ------------------------------------------------------------------------ */
/* colour model                                                              */

struct rgb { float R, G, B; };
struct ipt { float I, P, T; };
struct ich { float I, C, h; };

static inline struct ich to_ich(struct ipt c)
{
    return (struct ich) { c.I, sqrtf(c.P * c.P + c.T * c.T), atan2f(c.T, c.P) };
}

static inline struct ipt to_ipt(struct ich c)
{
    return (struct ipt) { c.I, c.C * cosf(c.h), c.C * sinf(c.h) };
}

enum { PQ_TABLE = 1024 };
static const uint32_t pq_bits[PQ_TABLE + 1] = {
#include "pq_eotf_table.inc"
};

static inline float pq_at(int i)
{
    const union { uint32_t u; float f; } c = { pq_bits[i] };
    return c.f;
}

static inline float pq_eotf(float x)
{
    const float pos = fminf(fmaxf(x, 0.0f), 1.0f) * (PQ_TABLE - 1);
    const int i = floorf(pos);
    const float fr = pos - i;
    return MIXF(pq_at(i), pq_at(i + 1), fr);
}

static inline float pq_oetf(float x)
{
    x = powf(fmaxf(x, 0.0f), PQ_M1);
    x = (PQ_C1 + PQ_C2 * x) / (1.0f + PQ_C3 * x);
    return powf(x, PQ_M2);
}

// A gamut in LMS terms + its legal range, plus a one-entry per-hue peak cache
struct gamut {
    pl_matrix3x3 lms2rgb, rgb2lms;
    float min_luma, max_luma;   // PQ
    float min_rgb, max_rgb;     // linear, 1.0 = 10000 cd/m^2
    struct ich *peak_cache;
};

struct peak_caches {
    struct ich src, dst;
};

static void setup_gamuts(struct gamut *dst, struct gamut *src, struct peak_caches *cache,
                         const struct pl_gamut_map_params *p)
{
    const float epsilon = 1e-6;
    memset(cache, 0, sizeof(*cache));
    const struct gamut base = {
        .min_luma = p->min_luma,
        .max_luma = p->max_luma,
        .min_rgb  = pq_eotf(p->min_luma) - epsilon,
        .max_rgb  = pq_eotf(p->max_luma) + epsilon,
    };

    if (dst) {
        *dst = base;
        dst->lms2rgb = dst->rgb2lms = pl_ipt_rgb2lms(&p->output_gamut);
        dst->peak_cache = &cache->dst;
        pl_matrix3x3_invert(&dst->lms2rgb);
    }
    if (src) {
        *src = base;
        src->lms2rgb = src->rgb2lms = pl_ipt_rgb2lms(&p->input_gamut);
        src->peak_cache = &cache->src;
        pl_matrix3x3_invert(&src->lms2rgb);
    }
}

static inline struct ipt rgb_to_ipt(struct rgb c, const struct gamut *g)
{
    const float (*m)[3] = g->rgb2lms.m;
    const float L = m[0][0] * c.R + m[0][1] * c.G + m[0][2] * c.B;
    const float M = m[1][0] * c.R + m[1][1] * c.G + m[1][2] * c.B;
    const float S = m[2][0] * c.R + m[2][1] * c.G + m[2][2] * c.B;
    const float Lp = pq_oetf(L), Mp = pq_oetf(M), Sp = pq_oetf(S);
    return (struct ipt) {
        .I = 0.4000f * Lp + 0.4000f * Mp + 0.2000f * Sp,
        .P = 4.4550f * Lp - 4.8510f * Mp + 0.3960f * Sp,
        .T = 0.8056f * Lp + 0.3572f * Mp - 1.1628f * Sp,
    };
}

static inline void ipt_to_lms_pq(struct ipt c, float *Lp, float *Mp, float *Sp)
{
    *Lp = c.I + 0.0975689f * c.P + 0.205226f * c.T;
    *Mp = c.I - 0.1138760f * c.P + 0.133217f * c.T;
    *Sp = c.I + 0.0326151f * c.P - 0.676887f * c.T;
}

static inline struct rgb lms_to_rgb(float L, float M, float S, const struct gamut *g)
{
    const float (*m)[3] = g->lms2rgb.m;
    return (struct rgb) {
        .R = m[0][0] * L + m[0][1] * M + m[0][2] * S,
        .G = m[1][0] * L + m[1][1] * M + m[1][2] * S,
        .B = m[2][0] * L + m[2][1] * M + m[2][2] * S,
    };
}

static inline struct rgb ipt_to_rgb(struct ipt c, const struct gamut *g)
{
    float Lp, Mp, Sp;
    ipt_to_lms_pq(c, &Lp, &Mp, &Sp);
    return lms_to_rgb(pq_eotf(Lp), pq_eotf(Mp), pq_eotf(Sp), g);
}

static inline bool in_gamut(struct ipt c, const struct gamut *g)
{
    float Lp, Mp, Sp;
    ipt_to_lms_pq(c, &Lp, &Mp, &Sp);
    if (Lp < g->min_luma || Lp > g->max_luma ||
        Mp < g->min_luma || Mp > g->max_luma ||
        Sp < g->min_luma || Sp > g->max_luma)
        return false; // outside the legal LMS range

    const struct rgb v = lms_to_rgb(pq_eotf(Lp), pq_eotf(Mp), pq_eotf(Sp), g);
    return v.R >= g->min_rgb && v.R <= g->max_rgb &&
           v.G >= g->min_rgb && v.G <= g->max_rgb &&
           v.B >= g->min_rgb && v.B <= g->max_rgb;
}

/* ------------------------------------------------------------------------ */
/* boundary searches                                                         */

static const float max_delta = 5e-5f;

// Largest in-gamut chroma at (I, h), bisecting within [Cmin, Cmax]
static inline struct ich boundary_chroma(float I, float h, float Cmin, float Cmax,
                                         const struct gamut *g)
{
    if (I <= g->min_luma)
        return (struct ich) { .I = g->min_luma, .C = 0, .h = h };
    if (I >= g->max_luma)
        return (struct ich) { .I = g->max_luma, .C = 0, .h = h };

    const float tol = I * max_delta;
    struct ich res = { .I = I, .C = (Cmin + Cmax) / 2, .h = h };
    do {
        if (in_gamut(to_ipt(res), g)) {
            Cmin = res.C;
        } else {
            Cmax = res.C;
        }
        res.C = (Cmin + Cmax) / 2;
    } while (Cmax - Cmin > tol);
    return res;
}

// The cusp: most saturated in-gamut colour of a hue (golden-section over I)
static inline struct ich cusp(float hue, const struct gamut *g)
{
    if (g->peak_cache->I && fabsf(g->peak_cache->h - hue) < 1e-3)
        return *g->peak_cache;

    static const float invphi = 0.6180339887498948f;
    static const float invphi2 = 0.38196601125010515f;

    struct ich lo = { .I = g->min_luma, .h = hue };
    struct ich hi = { .I = g->max_luma, .h = hue };
    float de = hi.I - lo.I;
    struct ich a = { .I = lo.I + invphi2 * de };
    struct ich b = { .I = lo.I + invphi  * de };
    a = boundary_chroma(a.I, hue, 0.0f, 0.5f, g);
    b = boundary_chroma(b.I, hue, 0.0f, 0.5f, g);

    while (de > max_delta) {
        de *= invphi;
        if (a.C > b.C) {
            hi = b;
            b = a;
            a.I = lo.I + invphi2 * de;
            a = boundary_chroma(a.I, hue, lo.C - max_delta, 0.5f, g);
        } else {
            lo = a;
            a = b;
            b.I = lo.I + invphi * de;
            b = boundary_chroma(b.I, hue, hi.C - max_delta, 0.5f, g);
        }
    }

    const struct ich peak = a.C > b.C ? a : b;
    *g->peak_cache = peak;
    return peak;
}

// MIX(base, c, x) along an exponential in I (x > 1 extrapolates)
static inline struct ich toward_base(struct ich c, float x, float gamma, float base)
{
    return (struct ich) {
        .I = base + (c.I - base) * powf(x, gamma),
        .C = c.C * x,
        .h = c.h,
    };
}

// Clip towards the cusp along an exponential curve (gamma 0 = pure desaturation)
static inline struct ipt clip_along_gamma(struct ipt c, float gamma, const struct gamut *g)
{
    if (c.I <= g->min_luma)
        return (struct ipt) { .I = g->min_luma };
    if (in_gamut(c, g))
        return c;

    const struct ich ich = to_ich(c);
    if (!gamma)
        return to_ipt(boundary_chroma(ich.I, ich.h, 0.0f, ich.C, g));

    const float tol = fmaxf(ich.I * max_delta, 1e-7f);
    const struct ich peak = cusp(ich.h, g);

    // soften gamma near black / the achromatic axis, boost it above the peak
    const float Irel = fmaxf((ich.I - g->min_luma) / (peak.I - g->min_luma), 0.0f);
    gamma = gamma * powf(Irel, 3) * fminf(ich.C / peak.C, 1.0f);

    float lo = 0.0f, hi = 1.0f, x = 0.5f;
    do {
        const struct ich test = toward_base(ich, x, gamma, peak.I);
        if (in_gamut(to_ipt(test), g)) {
            lo = x;
        } else {
            hi = x;
        }
        x = (lo + hi) / 2.0f;
    } while (hi - lo > tol);

    return to_ipt(toward_base(ich, x, gamma, peak.I));
}

// Mobius soft knee on value/target, identity below the knee
static float soft_knee(float value, float source, float target,
                       const struct pl_gamut_map_constants *c)
{
    if (!target)
        return 0.0f;
    const float peak = source / target;
    const float x = fminf(value / target, peak);
    const float j = c->softclip_knee;
    if (x <= j || peak <= 1.0)
        return value;
    const float a = -j*j * (peak - 1.0f) / (j*j - 2.0f * j + peak);
    const float b = (j*j - 2.0f * j * peak + peak) / fmaxf(1e-6f, peak - 1.0f);
    const float scale = (b*b + 2.0f * b*j + j*j) / (b - a);
    return scale * (x + a) / (x + b) * target;
}

/* ------------------------------------------------------------------------ */
/* LUT iteration                                                             */

#define LUT_FLOATS(p) ((size_t) (p)->lut_size_I * (p)->lut_size_C * (p)->lut_size_h * (p)->lut_stride)
#define EACH_IPT(lut, p, c)                                                      \
    for (struct ipt *i_ = (struct ipt *) (lut),                                  \
                    *end_ = (struct ipt *) ((lut) + LUT_FLOATS(p)), c;           \
         i_ < end_ && (c = *i_, 1);                                              \
         *i_ = c, i_ = (struct ipt *) ((float *) i_ + (p)->lut_stride))

/* ------------------------------------------------------------------------ */
/* mappers                                                                   */

static void map_clip(float *lut, const struct pl_gamut_map_params *p)
{
    (void) lut; (void) p;
}

static void map_perceptual(float *lut, const struct pl_gamut_map_params *p)
{
    const struct pl_gamut_map_constants *c = &p->constants;
    struct peak_caches cache;
    struct gamut dst, src;
    setup_gamuts(&dst, &src, &cache, p);

    EACH_IPT(lut, p, v) {
        const struct ich ich = to_ich(v);
        const struct ich src_peak = cusp(ich.h, &src);
        const struct ich dst_peak = cusp(ich.h, &dst);
        const struct ipt mapped = rgb_to_ipt(ipt_to_rgb(v, &src), &dst);

        // leave the interior of the gamut alone
        const float maxC = fmaxf(src_peak.C, dst_peak.C);
        float k = smoothstepf(c->perceptual_deadzone, 1.0f, ich.C / maxC);
        k *= c->perceptual_strength;
        v.I = MIXF(v.I, mapped.I, k);
        v.P = MIXF(v.P, mapped.P, k);
        v.T = MIXF(v.T, mapped.T, k);

        struct rgb rgb = ipt_to_rgb(v, &dst);
        const float maxRGB = fmaxf(rgb.R, fmaxf(rgb.G, rgb.B));
        rgb.R = fmaxf(soft_knee(rgb.R, maxRGB, dst.max_rgb, c), dst.min_rgb);
        rgb.G = fmaxf(soft_knee(rgb.G, maxRGB, dst.max_rgb, c), dst.min_rgb);
        rgb.B = fmaxf(soft_knee(rgb.B, maxRGB, dst.max_rgb, c), dst.min_rgb);
        v = rgb_to_ipt(rgb, &dst);
    }
}

/* hue-shift spline through the 12 primary/secondary/tertiary reference hues */

enum { HS = 12, HN = HS + 2 };

struct hue_node { float hue, delta; };

struct hue_spline {
    float dh[HN], slope[HN], K[HN];
    float prev_hue, prev_shift;
    struct hue_node node[HN];
};

static int cmp_node(const void *a, const void *b)
{
    const float fa = *(const float *) a, fb = *(const float *) b;
    return (fa > fb) - (fa < fb);
}

static float wrap_pi(float h)
{
    if (h > M_PI)
        return h - 2 * M_PI;
    if (h < -M_PI)
        return h + 2 * M_PI;
    return h;
}

static void hue_spline_init(struct hue_spline *s, const struct gamut *src, const struct gamut *dst)
{
    const float O = pq_eotf(src->min_luma), X = pq_eotf(src->max_luma);
    const float M = (O + X) / 2.0f;
    const struct rgb ref[HS] = {
        {X, O, O}, {O, X, O}, {O, O, X},
        {O, X, X}, {X, O, X}, {X, X, O},
        {O, X, M}, {X, O, M}, {X, M, O},
        {O, M, X}, {M, O, X}, {M, X, O},
    };

    memset(s, 0, sizeof(*s));
    for (int i = 0; i < HS; i++) {
        const struct ich a = to_ich(rgb_to_ipt(ref[i], src));
        const struct ich b = to_ich(rgb_to_ipt(ref[i], dst));
        s->node[i + 1].hue = a.h;
        s->node[i + 1].delta = wrap_pi(b.h - a.h);
    }

    // sort by hue, then wrap one node around each end
    qsort(s->node + 1, HS, sizeof(s->node[0]), cmp_node);
    s->node[0]      = s->node[HS];
    s->node[HS + 1] = s->node[1];
    s->node[0].hue      -= 2 * M_PI;
    s->node[HS + 1].hue += 2 * M_PI;

    // natural cubic spline: tridiagonal solve for the second derivatives K
    float tmp[HN][HN] = {0};
    for (int i = HN - 1; i > 0; i--) {
        s->dh[i - 1] = s->node[i].hue - s->node[i - 1].hue;
        s->slope[i] = (s->node[i].delta - s->node[i - 1].delta) / s->dh[i - 1];
    }
    for (int i = 1; i < HN - 1; i++) {
        tmp[i][i] = 2 * (s->dh[i - 1] + s->dh[i]);
        if (i != 1)
            tmp[i][i - 1] = tmp[i - 1][i] = s->dh[i - 1];
        tmp[i][HN - 1] = 6 * (s->slope[i + 1] - s->slope[i]);
    }
    for (int i = 1; i < HN - 2; i++) {
        const float q = tmp[i + 1][i] / tmp[i][i];
        for (int j = 1; j <= HN - 1; j++)
            tmp[i + 1][j] -= q * tmp[i][j];
    }
    for (int i = HN - 2; i > 0; i--) {
        float sum = 0.0f;
        for (int j = i; j <= HN - 2; j++)
            sum += tmp[i][j] * s->K[j];
        s->K[i] = (tmp[i][HN - 1] - sum) / tmp[i][i];
    }

    s->prev_hue = -10.0f;
}

static struct ich hue_spline_apply(struct hue_spline *s, struct ich c)
{
    if (!(fabsf(c.h - s->prev_hue) < 1e-6f)) {
        for (int i = 0; i < HN - 1; i++) {
            if (s->node[i + 1].hue > c.h) {
                const float a = (s->K[i + 1] - s->K[i]) / (6 * s->dh[i]);
                const float b = s->K[i] / 2;
                const float cc = s->slope[i + 1] -
                                 (2 * s->dh[i] * s->K[i] + s->K[i + 1] * s->dh[i]) / 6;
                const float d = s->node[i].delta;
                const float x = c.h - s->node[i].hue;
                const float delta = ((a * x + b) * x + cc) * x + d;
                s->prev_shift = c.h + delta;
                s->prev_hue = c.h;
                break;
            }
        }
    }
    return (struct ich) { c.I, c.C, s->prev_shift };
}

static void map_softclip(float *lut, const struct pl_gamut_map_params *p)
{
    const struct pl_gamut_map_constants *c = &p->constants;

    // two cache sets: the hue shift invalidates the pre-shift peaks
    struct peak_caches cache_pre, cache_post;
    struct gamut dst_pre, src_pre, src_post, dst_post;
    struct hue_spline spline;
    setup_gamuts(&dst_pre, &src_pre, &cache_pre, p);
    setup_gamuts(&dst_post, &src_post, &cache_post, p);
    hue_spline_init(&spline, &src_pre, &dst_pre);

    EACH_IPT(lut, p, v) {
        const struct gamut *src = &src_pre, *dst = &dst_pre;

        if (v.I <= dst->min_luma) {
            v.P = v.T = 0.0f;
            continue;
        }

        struct ich ich = to_ich(v);
        if (ich.C <= 1e-2f)
            continue; // achromatic: nothing to do

        float margin = 1.0f;
        const struct ich shifted = hue_spline_apply(&spline, ich);
        if (fabsf(shifted.h - ich.h) >= 1e-3f) {
            const struct ich src_border = boundary_chroma(ich.I, ich.h, 0.0f, 0.5f, src);
            const struct ich dst_border = boundary_chroma(ich.I, ich.h, 0.0f, 0.5f, dst);
            const float k = smoothstepf(dst_border.C * c->softclip_knee, src_border.C, ich.C);
            ich.h = MIXF(ich.h, shifted.h, k);
            src = &src_post;
            dst = &dst_post;

            // the hue leaf changes size under the shift: rescale the margin
            const struct ich shift_border = boundary_chroma(ich.I, ich.h, 0.0f, 0.5f, src);
            margin *= fmaxf(1.0f, src_border.C / shift_border.C);
        }

        // soft-clip chroma between the source and target cusps
        const struct ich source = cusp(ich.h, src);
        const struct ich target = cusp(ich.h, dst);
        const struct ich border = boundary_chroma(ich.I, ich.h, 0.0f, target.C, dst);
        const float chromaticity = MIXF(target.C, border.C, c->softclip_desat);
        ich.C = soft_knee(ich.C, margin * source.C, chromaticity, c);

        // then soft-clip the RGB result against the saturated colour
        const struct ich saturated = { ich.I, chromaticity, ich.h };
        const struct rgb peak = ipt_to_rgb(to_ipt(saturated), dst);
        struct rgb rgb = ipt_to_rgb(to_ipt(ich), dst);
        rgb.R = fmaxf(soft_knee(rgb.R, peak.R, dst->max_rgb, c), dst->min_rgb);
        rgb.G = fmaxf(soft_knee(rgb.G, peak.G, dst->max_rgb, c), dst->min_rgb);
        rgb.B = fmaxf(soft_knee(rgb.B, peak.B, dst->max_rgb, c), dst->min_rgb);
        v = rgb_to_ipt(rgb, dst);
    }
}

static void map_relative(float *lut, const struct pl_gamut_map_params *p)
{
    struct peak_caches cache;
    struct gamut dst;
    setup_gamuts(&dst, NULL, &cache, p);

    EACH_IPT(lut, p, v)
        v = clip_along_gamma(v, p->constants.colorimetric_gamma, &dst);
}

static void map_desaturate(float *lut, const struct pl_gamut_map_params *p)
{
    struct peak_caches cache;
    struct gamut dst;
    setup_gamuts(&dst, NULL, &cache, p);

    EACH_IPT(lut, p, v)
        v = clip_along_gamma(v, 0.0f, &dst);
}

static void map_saturation(float *lut, const struct pl_gamut_map_params *p)
{
    struct peak_caches cache;
    struct gamut dst, src;
    setup_gamuts(&dst, &src, &cache, p);

    EACH_IPT(lut, p, v)
        v = rgb_to_ipt(ipt_to_rgb(v, &src), &dst);
}

static void map_absolute(float *lut, const struct pl_gamut_map_params *p)
{
    struct peak_caches cache;
    struct gamut dst;
    setup_gamuts(&dst, NULL, &cache, p);
    const pl_matrix3x3 m = pl_get_adaptation_matrix(p->output_gamut.white, p->input_gamut.white);

    EACH_IPT(lut, p, v) {
        struct rgb rgb = ipt_to_rgb(v, &dst);
        pl_matrix3x3_apply(&m, (float *) &rgb);
        v = rgb_to_ipt(rgb, &dst);
        v = clip_along_gamma(v, p->constants.colorimetric_gamma, &dst);
    }
}

static void map_highlight(float *lut, const struct pl_gamut_map_params *p)
{
    struct peak_caches cache;
    struct gamut dst;
    setup_gamuts(&dst, NULL, &cache, p);

    EACH_IPT(lut, p, v) {
        if (!in_gamut(v, &dst)) {
            v.I = fminf(v.I + 0.1f, 1.0f);
            v.P = CLAMPF(-1.2f * v.P, -0.5f, 0.5f);
            v.T = CLAMPF(-1.2f * v.T, -0.5f, 0.5f);
        }
    }
}

static void map_linear(float *lut, const struct pl_gamut_map_params *p)
{
    struct peak_caches cache;
    struct gamut dst, src;
    setup_gamuts(&dst, &src, &cache, p);

    float gain = 1.0f;
    for (float hue = -M_PI; hue < M_PI; hue += 0.1f)
        gain = fminf(gain, cusp(hue, &dst).C / cusp(hue, &src).C);

    EACH_IPT(lut, p, v) {
        struct ich ich = to_ich(v);
        ich.C *= gain;
        v = to_ipt(ich);
    }
}

static void map_darken(float *lut, const struct pl_gamut_map_params *p)
{
    struct peak_caches cache;
    struct gamut dst, src;
    setup_gamuts(&dst, &src, &cache, p);

    static const struct rgb corners[6] = {
        {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
        {0, 1, 1}, {1, 0, 1}, {1, 1, 0},
    };

    float gain = 1.0f;
    for (int i = 0; i < 6; i++) {
        const struct rgb q = ipt_to_rgb(rgb_to_ipt(corners[i], &src), &dst);
        const float maxRGB = PL_MAX(PL_MAX(q.R, q.G), q.B);
        gain = fminf(gain, 1.0 / maxRGB);
    }

    EACH_IPT(lut, p, v) {
        struct rgb rgb = ipt_to_rgb(v, &dst);
        rgb.R *= gain;
        rgb.G *= gain;
        rgb.B *= gain;
        v = rgb_to_ipt(rgb, &dst);
        v = clip_along_gamma(v, p->constants.colorimetric_gamma, &dst);
    }
}

/* ------------------------------------------------------------------------ */
/* LUT generation                                                            */

struct chunk {
    const struct pl_gamut_map_params *params;
    float *out;
    int start, count; // hue slices
};

static void *run_chunk(void *priv)
{
    const struct chunk *ck = priv;
    const struct pl_gamut_map_params *p = ck->params;

    // lattice points, I fastest
    float *in = ck->out;
    for (int h = ck->start; h < ck->start + ck->count; h++) {
        for (int C = 0; C < p->lut_size_C; C++) {
            for (int I = 0; I < p->lut_size_I; I++) {
                const float Ix = (float) I / (p->lut_size_I - 1);
                const float Cx = (float) C / (p->lut_size_C - 1);
                const float hx = (float) h / (p->lut_size_h - 1);
                const struct ipt v = to_ipt((struct ich) {
                    .I = MIXF(p->min_luma, p->max_luma, Ix),
                    .C = MIXF(0.0f, 0.5f, Cx),
                    .h = MIXF(-M_PI, M_PI, hx),
                });
                in[0] = v.I;
                in[1] = v.P;
                in[2] = v.T;
                in += p->lut_stride;
            }
        }
    }

    struct pl_gamut_map_params fixed = *p;
    sanitize_constants(&fixed.constants);
    fixed.lut_size_h = ck->count;
    fn_of(p)->map(ck->out, &fixed);
    return NULL;
}

void pl_gamut_map_generate(float *out, const struct pl_gamut_map_params *params)
{
    // Same partition as the reference: at most 32 workers of ceil(h/32) hue
    // slices each. The partition is part of the result (per-worker caches).
    enum { MAX_WORKERS = 32 };
    struct chunk chunks[MAX_WORKERS];
    pthread_t threads[MAX_WORKERS];
    bool started[MAX_WORKERS] = {0};

    const int per = (params->lut_size_h + MAX_WORKERS - 1) / MAX_WORKERS;
    const int num = (params->lut_size_h + per - 1) / per;
    for (int i = 0; i < num; i++) {
        const int start = i * per;
        const int count = PL_MIN(per, params->lut_size_h - start);
        chunks[i] = (struct chunk) { params, out, start, count };
        out += (size_t) count * params->lut_size_C * params->lut_size_I * params->lut_stride;
    }

    for (int i = 0; i < num; i++) {
        started[i] = pthread_create(&threads[i], NULL, run_chunk, &chunks[i]) == 0;
        if (!started[i])
            run_chunk(&chunks[i]);
    }
    for (int i = 0; i < num; i++) {
        if (started[i])
            pthread_join(threads[i], NULL);
    }
}

void pl_gamut_map_sample(float x[3], const struct pl_gamut_map_params *params)
{
    struct pl_gamut_map_params fixed = *params;
    sanitize_constants(&fixed.constants);
    fixed.lut_size_I = fixed.lut_size_C = fixed.lut_size_h = 1;
    fixed.lut_stride = 3;
    fn_of(params)->map(x, &fixed);
}

/* ------------------------------------------------------------------------ */

#define MAPPER(sym, nm, desc, fn, bidi) \
    const struct pl_gamut_map_function sym = { .name = nm, .description = desc, \
        .map = fn, .bidirectional = bidi }

MAPPER(pl_gamut_map_clip,       "clip",       "No gamut mapping (hard clip)",  map_clip, false);
MAPPER(pl_gamut_map_perceptual, "perceptual", "Perceptual mapping",            map_perceptual, true);
MAPPER(pl_gamut_map_softclip,   "softclip",   "Soft clipping",                 map_softclip, false);
MAPPER(pl_gamut_map_relative,   "relative",   "Colorimetric clip",             map_relative, false);
MAPPER(pl_gamut_map_saturation, "saturation", "Saturation mapping",            map_saturation, true);
MAPPER(pl_gamut_map_absolute,   "absolute",   "Absolute colorimetric clip",    map_absolute, false);
MAPPER(pl_gamut_map_desaturate, "desaturate", "Desaturating clip",             map_desaturate, false);
MAPPER(pl_gamut_map_darken,     "darken",     "Darken and clip",               map_darken, false);
MAPPER(pl_gamut_map_highlight,  "highlight",  "Highlight out-of-gamut pixels", map_highlight, false);
MAPPER(pl_gamut_map_linear,     "linear",     "Linear desaturate",             map_linear, false);

const struct pl_gamut_map_function * const pl_gamut_map_functions[] = {
    &pl_gamut_map_clip,
    &pl_gamut_map_perceptual,
    &pl_gamut_map_softclip,
    &pl_gamut_map_relative,
    &pl_gamut_map_saturation,
    &pl_gamut_map_absolute,
    &pl_gamut_map_desaturate,
    &pl_gamut_map_darken,
    &pl_gamut_map_highlight,
    &pl_gamut_map_linear,
    NULL
};

const int pl_num_gamut_map_functions = PL_ARRAY_SIZE(pl_gamut_map_functions) - 1;

const struct pl_gamut_map_function *pl_find_gamut_map_function(const char *name)
{
    for (int i = 0; name && i < pl_num_gamut_map_functions; i++) {
        if (!strcmp(name, pl_gamut_map_functions[i]->name))
            return pl_gamut_map_functions[i];
    }
    return NULL;
}
