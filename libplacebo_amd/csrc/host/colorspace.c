/*
 * libplacebo-hip — Tier-0 host maths: colour representation, colour spaces,
 * primaries / matrices, CPU transfer functions.
 *
 * Fresh implementation of the behaviour of the reference's src/colorspace.c:
 *   repr normalisation / decode matrix   colorspace.c:190-216, 1692-1896
 *   HDR rescaling, nominal luma, inference  :367-418, 790-972
 *   CPU transfer functions                  :565-740
 *   primaries, RGB<->XYZ, CAT16, IPT        :1102-1396, 1543-1670
 * The produced floats become kernel constants, so operation order and
 * precision (float vs double) follow the reference expression by expression
 * (pinned against oracle/_ref in tests/test_tier0_ref.py).
 */
#include <math.h>
#include <string.h>

#include <libplacebo/colorspace.h>

#include "host_common.h"
#include "colorspace_priv.h"
#include "cache_priv.h"

#define MAX3(a, b, c) PL_MAX(PL_MAX(a, b), c)
#define MIXF(a, b, t) ((1 - (t)) * (a) + (t) * (b))

/* ------------------------------------------------------------------------ */
/* colour systems / representation                                           */

// Small classifications of the enums are written as membership in a bit set.
#define MEMBER(e) (UINT64_C(1) << (e))
static inline bool among(int value, uint64_t set)
{
    return value >= 0 && value < 64 && ((set >> value) & 1);
}

bool pl_color_system_is_ycbcr_like(enum pl_color_system sys)
{
    // every known system except the tristimulus ones carries one luma-like and two
    // difference-like channels
    const bool known = sys > PL_COLOR_SYSTEM_UNKNOWN && sys < PL_COLOR_SYSTEM_COUNT;
    return known && !among(sys, MEMBER(PL_COLOR_SYSTEM_RGB) | MEMBER(PL_COLOR_SYSTEM_XYZ));
}

bool pl_color_system_is_linear(enum pl_color_system sys)
{
    // systems whose decoding is more than a matrix: constant luminance, ICtCp, Dolby Vision
    // reshaping, XYZ's own gamma
    return !among(sys, MEMBER(PL_COLOR_SYSTEM_BT_2020_C) | MEMBER(PL_COLOR_SYSTEM_BT_2100_PQ) |
                       MEMBER(PL_COLOR_SYSTEM_BT_2100_HLG) | MEMBER(PL_COLOR_SYSTEM_DOLBYVISION) |
                       MEMBER(PL_COLOR_SYSTEM_XYZ));
}

const char *const pl_color_system_names[PL_COLOR_SYSTEM_COUNT] = {
    [PL_COLOR_SYSTEM_UNKNOWN]       = "Auto (unknown)",
    [PL_COLOR_SYSTEM_BT_601]        = "ITU-R Rec. BT.601 (SD)",
    [PL_COLOR_SYSTEM_BT_709]        = "ITU-R Rec. BT.709 (HD)",
    [PL_COLOR_SYSTEM_SMPTE_240M]    = "SMPTE-240M",
    [PL_COLOR_SYSTEM_BT_2020_NC]    = "ITU-R Rec. BT.2020 (non-constant luminance)",
    [PL_COLOR_SYSTEM_BT_2020_C]     = "ITU-R Rec. BT.2020 (constant luminance)",
    [PL_COLOR_SYSTEM_BT_2100_PQ]    = "ITU-R Rec. BT.2100 ICtCp PQ variant",
    [PL_COLOR_SYSTEM_BT_2100_HLG]   = "ITU-R Rec. BT.2100 ICtCp HLG variant",
    [PL_COLOR_SYSTEM_DOLBYVISION]   = "Dolby Vision (invalid for output)",
    [PL_COLOR_SYSTEM_YCGCO]         = "YCgCo (derived from RGB)",
    [PL_COLOR_SYSTEM_YCGCO_RE]      = "YCgCo-R, even addition of bits",
    [PL_COLOR_SYSTEM_YCGCO_RO]      = "YCgCo-R, odd addition of bits",
    [PL_COLOR_SYSTEM_RGB]           = "Red, Green and Blue",
    [PL_COLOR_SYSTEM_XYZ]           = "Digital Cinema Distribution Master (XYZ)",
};

const char *pl_color_system_name(enum pl_color_system sys)
{
    return sys >= 0 && sys < PL_COLOR_SYSTEM_COUNT ? pl_color_system_names[sys] : "?";
}

enum pl_color_system pl_color_system_guess_ycbcr(int width, int height)
{
    return width >= 1280 || height > 576 ? PL_COLOR_SYSTEM_BT_709 : PL_COLOR_SYSTEM_BT_601;
}

bool pl_bit_encoding_equal(const struct pl_bit_encoding *b1, const struct pl_bit_encoding *b2)
{
    return b1->sample_depth == b2->sample_depth && b1->color_depth == b2->color_depth &&
           b1->bit_shift == b2->bit_shift;
}

const struct pl_color_repr pl_color_repr_unknown = {0};
const struct pl_color_repr pl_color_repr_rgb   = { PL_COLOR_SYSTEM_RGB,        PL_COLOR_LEVELS_FULL };
const struct pl_color_repr pl_color_repr_sdtv  = { PL_COLOR_SYSTEM_BT_601,     PL_COLOR_LEVELS_LIMITED };
const struct pl_color_repr pl_color_repr_hdtv  = { PL_COLOR_SYSTEM_BT_709,     PL_COLOR_LEVELS_LIMITED };
const struct pl_color_repr pl_color_repr_uhdtv = { PL_COLOR_SYSTEM_BT_2020_NC, PL_COLOR_LEVELS_LIMITED };
const struct pl_color_repr pl_color_repr_jpeg  = { PL_COLOR_SYSTEM_BT_601,     PL_COLOR_LEVELS_FULL };

bool pl_color_repr_equal(const struct pl_color_repr *c1, const struct pl_color_repr *c2)
{
    return c1->sys == c2->sys && c1->levels == c2->levels && c1->alpha == c2->alpha &&
           c1->dovi == c2->dovi && pl_bit_encoding_equal(&c1->bits, &c2->bits);
}

void pl_color_repr_merge(struct pl_color_repr *orig, const struct pl_color_repr *update)
{
    orig->sys    = PL_DEF(orig->sys, update->sys);
    orig->levels = PL_DEF(orig->levels, update->levels);
    orig->alpha  = PL_DEF(orig->alpha, update->alpha);
    orig->dovi   = PL_DEF(orig->dovi, update->dovi);
    orig->bits.sample_depth = PL_DEF(orig->bits.sample_depth, update->bits.sample_depth);
    orig->bits.color_depth  = PL_DEF(orig->bits.color_depth, update->bits.color_depth);
    orig->bits.bit_shift    = PL_DEF(orig->bits.bit_shift, update->bits.bit_shift);
}

enum pl_color_levels pl_color_levels_guess(const struct pl_color_repr *repr)
{
    // an explicit tag wins, except that Dolby Vision is full range by definition
    const bool full_by_nature = repr->sys == PL_COLOR_SYSTEM_DOLBYVISION ||
                                !pl_color_system_is_ycbcr_like(repr->sys);
    if (repr->levels && repr->sys != PL_COLOR_SYSTEM_DOLBYVISION)
        return repr->levels;
    return full_by_nature ? PL_COLOR_LEVELS_FULL : PL_COLOR_LEVELS_LIMITED;
}

float pl_color_repr_normalize(struct pl_color_repr *repr)
{
    struct pl_bit_encoding *bits = &repr->bits;
    // depth of the stored samples / of the colour values in them; either stands in for the
    // other when unset, 8 bits when both are
    int stored = bits->sample_depth ? bits->sample_depth : bits->color_depth;
    int coded  = bits->color_depth ? bits->color_depth : bits->sample_depth;
    if (!stored)
        stored = coded = 8;
    const bool limited = pl_color_levels_guess(repr) == PL_COLOR_LEVELS_LIMITED;

    // a representational left shift is undone by a power of two (exact in float) ...
    float scale = ldexpf(1.0f, -bits->bit_shift);
    if (limited) {
        // ... so is limited range, which pads with zero bits
        scale = ldexpf(scale, stored - coded);
    } else {
        // full range spans all codes at either depth: (2^stored - 1) / (2^coded - 1), in double
        const double top_stored = (double) ((1LL << stored) - 1), top_coded = (double) ((1LL << coded) - 1);
        scale = (float) (scale * (top_stored / top_coded));
    }

    bits->bit_shift = 0;
    bits->color_depth = bits->sample_depth;
    return scale;
}

/* ------------------------------------------------------------------------ */
/* primaries / transfer                                                      */

bool pl_color_primaries_is_wide_gamut(enum pl_color_primaries prim)
{
    // the standard-gamut containers (and "unknown", which is inferred as one of them)
    return !among(prim, MEMBER(PL_COLOR_PRIM_UNKNOWN) | MEMBER(PL_COLOR_PRIM_BT_601_525) |
                        MEMBER(PL_COLOR_PRIM_BT_601_625) | MEMBER(PL_COLOR_PRIM_BT_709) |
                        MEMBER(PL_COLOR_PRIM_BT_470M) | MEMBER(PL_COLOR_PRIM_EBU_3213));
}

const char *const pl_color_primaries_names[PL_COLOR_PRIM_COUNT] = {
    [PL_COLOR_PRIM_UNKNOWN]     = "Auto (unknown)",
    [PL_COLOR_PRIM_BT_601_525]  = "ITU-R Rec. BT.601 (525-line = NTSC, SMPTE-C)",
    [PL_COLOR_PRIM_BT_601_625]  = "ITU-R Rec. BT.601 (625-line = PAL, SECAM)",
    [PL_COLOR_PRIM_BT_709]      = "ITU-R Rec. BT.709 (HD), also sRGB",
    [PL_COLOR_PRIM_BT_470M]     = "ITU-R Rec. BT.470 M",
    [PL_COLOR_PRIM_EBU_3213]    = "EBU Tech. 3213-E / JEDEC P22 phosphors",
    [PL_COLOR_PRIM_BT_2020]     = "ITU-R Rec. BT.2020 (Ultra HD)",
    [PL_COLOR_PRIM_APPLE]       = "Apple RGB",
    [PL_COLOR_PRIM_ADOBE]       = "Adobe RGB (1998)",
    [PL_COLOR_PRIM_PRO_PHOTO]   = "ProPhoto RGB (ROMM)",
    [PL_COLOR_PRIM_CIE_1931]    = "CIE 1931 RGB primaries",
    [PL_COLOR_PRIM_DCI_P3]      = "DCI-P3 (Digital Cinema)",
    [PL_COLOR_PRIM_DISPLAY_P3]  = "DCI-P3 (Digital Cinema) with D65 white point",
    [PL_COLOR_PRIM_V_GAMUT]     = "Panasonic V-Gamut (VARICAM)",
    [PL_COLOR_PRIM_S_GAMUT]     = "Sony S-Gamut",
    [PL_COLOR_PRIM_FILM_C]      = "Traditional film primaries with Illuminant C",
    [PL_COLOR_PRIM_ACES_AP0]    = "ACES Primaries #0",
    [PL_COLOR_PRIM_ACES_AP1]    = "ACES Primaries #1",
};

const char *pl_color_primaries_name(enum pl_color_primaries prim)
{
    return prim >= 0 && prim < PL_COLOR_PRIM_COUNT ? pl_color_primaries_names[prim] : "?";
}

enum pl_color_primaries pl_color_primaries_guess(int width, int height)
{
    // standard-definition rasters, by their line count; anything else (and anything at least
    // 1280 wide or taller than PAL) is assumed to be BT.709
    static const struct { int lines; enum pl_color_primaries prim; } sd[] = {
        { 576, PL_COLOR_PRIM_BT_601_625 },
        { 480, PL_COLOR_PRIM_BT_601_525 },
        { 486, PL_COLOR_PRIM_BT_601_525 },
    };
    const bool hd = width >= 1280 || height > 576;
    for (size_t i = 0; !hd && i < PL_ARRAY_SIZE(sd); i++) {
        if (height == sd[i].lines)
            return sd[i].prim;
    }
    return PL_COLOR_PRIM_BT_709;
}

const char *const pl_color_transfer_names[PL_COLOR_TRC_COUNT] = {
    [PL_COLOR_TRC_UNKNOWN]      = "Auto (unknown SDR)",
    [PL_COLOR_TRC_BT_1886]      = "ITU-R Rec. BT.1886 (CRT emulation + OOTF)",
    [PL_COLOR_TRC_SRGB]         = "IEC 61966-2-4 sRGB (CRT emulation)",
    [PL_COLOR_TRC_LINEAR]       = "Linear light content",
    [PL_COLOR_TRC_GAMMA18]      = "Pure power gamma 1.8",
    [PL_COLOR_TRC_GAMMA20]      = "Pure power gamma 2.0",
    [PL_COLOR_TRC_GAMMA22]      = "Pure power gamma 2.2",
    [PL_COLOR_TRC_GAMMA24]      = "Pure power gamma 2.4",
    [PL_COLOR_TRC_GAMMA26]      = "Pure power gamma 2.6",
    [PL_COLOR_TRC_GAMMA28]      = "Pure power gamma 2.8",
    [PL_COLOR_TRC_PRO_PHOTO]    = "ProPhoto RGB (ROMM)",
    [PL_COLOR_TRC_ST428]        = "Digital Cinema Distribution Master (XYZ)",
    [PL_COLOR_TRC_PQ]           = "ITU-R BT.2100 PQ (perceptual quantizer), aka SMPTE ST2048",
    [PL_COLOR_TRC_HLG]          = "ITU-R BT.2100 HLG (hybrid log-gamma), aka ARIB STD-B67",
    [PL_COLOR_TRC_V_LOG]        = "Panasonic V-Log (VARICAM)",
    [PL_COLOR_TRC_S_LOG1]       = "Sony S-Log1",
    [PL_COLOR_TRC_S_LOG2]       = "Sony S-Log2",
    [PL_COLOR_TRC_SCRGB]        = "IEC 61966-2-2 scRGB (extended linear BT.709)",
};

const char *pl_color_transfer_name(enum pl_color_transfer trc)
{
    return trc >= 0 && trc < PL_COLOR_TRC_COUNT ? pl_color_transfer_names[trc] : "?";
}

float pl_color_transfer_nominal_peak(enum pl_color_transfer trc)
{
    switch (trc) {
    case PL_COLOR_TRC_SCRGB:
    case PL_COLOR_TRC_PQ:       return 10000.0 / PL_COLOR_SDR_WHITE;
    case PL_COLOR_TRC_HLG:      return 12.0 / 3.17955; // 75% HLG, scene-referred
    case PL_COLOR_TRC_V_LOG:    return 46.0855;
    case PL_COLOR_TRC_S_LOG1:   return 6.52;
    case PL_COLOR_TRC_S_LOG2:   return 9.212;
    default:                    return 1.0;
    }
}

const struct pl_hdr_metadata pl_hdr_metadata_empty = {0};
const struct pl_hdr_metadata pl_hdr_metadata_hdr10 = {
    .prim = {
        .red   = {0.708,    0.292},
        .green = {0.170,    0.797},
        .blue  = {0.131,    0.046},
        .white = {0.31271,  0.32902},
    },
    .min_luma = 0,
    .max_luma = 10000,
    .max_cll  = 10000,
};

float pl_hdr_rescale(enum pl_hdr_scaling from, enum pl_hdr_scaling to, float x)
{
    if (from == to || !x)
        return x;
    x = fmaxf(x, 0.0f);

    // -> PL_HDR_NORM
    switch (from) {
    case PL_HDR_PQ:
        x = powf(x, 1.0f / PQ_M2);
        x = fmaxf(x - PQ_C1, 0.0f) / (PQ_C2 - PQ_C3 * x);
        x = powf(x, 1.0f / PQ_M1);
        x *= 10000.0f;
        x /= PL_COLOR_SDR_WHITE;
        break;
    case PL_HDR_NITS:
        x /= PL_COLOR_SDR_WHITE;
        break;
    case PL_HDR_SQRT:
        x *= x;
        break;
    default:
        break;
    }

    // PL_HDR_NORM -> target
    switch (to) {
    case PL_HDR_SQRT:
        return sqrtf(x);
    case PL_HDR_NITS:
        return x * PL_COLOR_SDR_WHITE;
    case PL_HDR_PQ:
        x *= PL_COLOR_SDR_WHITE / 10000.0f;
        x = powf(x, PQ_M1);
        x = (PQ_C1 + PQ_C2 * x) / (1.0f + PQ_C3 * x);
        return powf(x, PQ_M2);
    default:
        return x;
    }
}

static bool bezier_equal(const struct pl_hdr_bezier *a, const struct pl_hdr_bezier *b)
{
    return a->target_luma == b->target_luma && a->knee_x == b->knee_x &&
           a->knee_y == b->knee_y && a->num_anchors == b->num_anchors &&
           !memcmp(a->anchors, b->anchors, sizeof(a->anchors[0]) * a->num_anchors);
}

bool pl_hdr_metadata_equal(const struct pl_hdr_metadata *a, const struct pl_hdr_metadata *b)
{
    return pl_raw_primaries_equal(&a->prim, &b->prim) &&
           a->min_luma == b->min_luma && a->max_luma == b->max_luma &&
           a->max_cll == b->max_cll && a->max_fall == b->max_fall &&
           a->scene_max[0] == b->scene_max[0] && a->scene_max[1] == b->scene_max[1] &&
           a->scene_max[2] == b->scene_max[2] && a->scene_avg == b->scene_avg &&
           bezier_equal(&a->ootf, &b->ootf) &&
           a->max_pq_y == b->max_pq_y && a->avg_pq_y == b->avg_pq_y;
}

void pl_hdr_metadata_merge(struct pl_hdr_metadata *orig, const struct pl_hdr_metadata *update)
{
    pl_raw_primaries_merge(&orig->prim, &update->prim);
    orig->min_luma = PL_DEF(orig->min_luma, update->min_luma);
    orig->max_luma = PL_DEF(orig->max_luma, update->max_luma);
    orig->max_cll  = PL_DEF(orig->max_cll, update->max_cll);
    orig->max_fall = PL_DEF(orig->max_fall, update->max_fall);
    if (!orig->scene_max[1])
        memcpy(orig->scene_max, update->scene_max, sizeof(orig->scene_max));
    orig->scene_avg = PL_DEF(orig->scene_avg, update->scene_avg);
    if (!orig->ootf.target_luma)
        orig->ootf = update->ootf;
    orig->max_pq_y = PL_DEF(orig->max_pq_y, update->max_pq_y);
    orig->avg_pq_y = PL_DEF(orig->avg_pq_y, update->avg_pq_y);
}

bool pl_hdr_metadata_contains(const struct pl_hdr_metadata *data, enum pl_hdr_metadata_type type)
{
    const bool hdr10 = data->max_luma;
    const bool hdr10plus = data->scene_avg &&
        (data->scene_max[0] || data->scene_max[1] || data->scene_max[2]);
    const bool cie_y = data->max_pq_y && data->avg_pq_y;

    switch (type) {
    case PL_HDR_METADATA_NONE:      return true;
    case PL_HDR_METADATA_ANY:       return hdr10 || hdr10plus || cie_y;
    case PL_HDR_METADATA_HDR10:     return hdr10;
    case PL_HDR_METADATA_HDR10PLUS: return hdr10plus;
    case PL_HDR_METADATA_CIE_Y:     return cie_y;
    default:                        return false;
    }
}

const struct pl_color_space pl_color_space_unknown = {0};
const struct pl_color_space pl_color_space_srgb = { PL_COLOR_PRIM_BT_709, PL_COLOR_TRC_SRGB };
const struct pl_color_space pl_color_space_bt709 = { PL_COLOR_PRIM_BT_709, PL_COLOR_TRC_BT_1886 };
const struct pl_color_space pl_color_space_hdr10 = { PL_COLOR_PRIM_BT_2020, PL_COLOR_TRC_PQ };
const struct pl_color_space pl_color_space_bt2020_hlg = { PL_COLOR_PRIM_BT_2020, PL_COLOR_TRC_HLG };
const struct pl_color_space pl_color_space_monitor = { PL_COLOR_PRIM_BT_709, PL_COLOR_TRC_UNKNOWN };

bool pl_color_space_is_hdr(const struct pl_color_space *csp)
{
    return csp->hdr.max_luma > PL_COLOR_SDR_WHITE || pl_color_transfer_is_hdr(csp->transfer);
}

bool pl_color_space_is_black_scaled(const struct pl_color_space *csp)
{
    // curves with an absolute black point of their own are not rescaled to the display's
    return !among(csp->transfer, MEMBER(PL_COLOR_TRC_BT_1886) | MEMBER(PL_COLOR_TRC_PQ) |
                                 MEMBER(PL_COLOR_TRC_SCRGB) | MEMBER(PL_COLOR_TRC_V_LOG) |
                                 MEMBER(PL_COLOR_TRC_S_LOG1) | MEMBER(PL_COLOR_TRC_S_LOG2));
}

/* ------------------------------------------------------------------------ */
/* CPU transfer functions                                                    */

static void nominal_norm(const struct pl_color_space *csp, float *min, float *max)
{
    pl_color_space_nominal_luma_ex(pl_nominal_luma_params(
        .color = csp, .metadata = PL_HDR_METADATA_HDR10, .scaling = PL_HDR_NORM,
        .out_min = min, .out_max = max,
    ));
}

void plh_bt1886_params(float csp_min, float csp_max, float *a, float *b)
{
    const float lb = powf(csp_min, 1 / 2.4f);
    const float lw = powf(csp_max, 1 / 2.4f);
    *a = powf(lw - lb, 2.4f);
    *b = lb / (lw - lb);
}

void plh_hlg_params(float csp_min, float csp_max, float *y, float *b)
{
    *y = 1.2f * powf(1.111f, log2f(csp_max / HLG_REF));
    *b = sqrtf(3 * powf(csp_min / csp_max, 1 / *y));
}

#define EACH(expr) do {                         \
        for (int i_ = 0; i_ < 3; i_++) {        \
            const float X = color[i_];          \
            color[i_] = (expr);                 \
        }                                       \
    } while (0)

void pl_color_linearize(const struct pl_color_space *csp, float color[3])
{
    const enum pl_color_transfer trc = csp->transfer;
    if (trc == PL_COLOR_TRC_LINEAR)
        return;

    float csp_min, csp_max;
    nominal_norm(csp, &csp_min, &csp_max);
    if (trc != PL_COLOR_TRC_SCRGB)
        EACH(fmaxf(X, 0));

    switch (trc) {
    case PL_COLOR_TRC_SRGB:
        EACH(X > 0.04045f ? powf((X + 0.055f) / 1.055f, 2.4f) : X / 12.92f);
        break;
    case PL_COLOR_TRC_BT_1886: {
        float a, b;
        plh_bt1886_params(csp_min, csp_max, &a, &b);
        EACH(a * powf(X + b, 2.4f));
        return;
    }
    case PL_COLOR_TRC_GAMMA18: EACH(powf(X, 1.8f)); break;
    case PL_COLOR_TRC_GAMMA20: EACH(powf(X, 2.0f)); break;
    case PL_COLOR_TRC_UNKNOWN:
    case PL_COLOR_TRC_GAMMA22: EACH(powf(X, 2.2f)); break;
    case PL_COLOR_TRC_GAMMA24: EACH(powf(X, 2.4f)); break;
    case PL_COLOR_TRC_GAMMA26: EACH(powf(X, 2.6f)); break;
    case PL_COLOR_TRC_GAMMA28: EACH(powf(X, 2.8f)); break;
    case PL_COLOR_TRC_PRO_PHOTO:
        EACH(X > 0.03125f ? powf(X, 1.8f) : X / 16);
        break;
    case PL_COLOR_TRC_ST428:
        EACH(52.37f / 48 * powf(X, 2.6f));
        break;
    case PL_COLOR_TRC_PQ:
        EACH(powf(X, 1 / PQ_M2));
        EACH(fmaxf(X - PQ_C1, 0) / (PQ_C2 - PQ_C3 * X));
        EACH(10000 / PL_COLOR_SDR_WHITE * powf(X, 1 / PQ_M1));
        break;
    case PL_COLOR_TRC_HLG: {
        float y, b;
        plh_hlg_params(csp_min, csp_max, &y, &b);
        const pl_matrix3x3 rgb2xyz = pl_get_rgb2xyz_matrix(pl_raw_primaries_get(csp->primaries));
        const float *coef = rgb2xyz.m[1];
        EACH((1 - b) * X + b);
        EACH(X > 0.5f ? expf((X - HLG_C) / HLG_A) + HLG_B : 4 * X * X);
        float luma = coef[0] * color[0] + coef[1] * color[1] + coef[2] * color[2];
        luma = powf(fmaxf(luma / 12, 0), y - 1);
        EACH(luma * X / 12);
        break;
    }
    case PL_COLOR_TRC_V_LOG:
        EACH(X >= 0.181f ? powf(10, (X - VLOG_D) / VLOG_C) - VLOG_B : (X - 0.125f) / 5.6f);
        break;
    case PL_COLOR_TRC_S_LOG1:
        EACH(powf(10, (X - SLOG_C) / SLOG_A) - SLOG_B);
        break;
    case PL_COLOR_TRC_S_LOG2:
        EACH(X >= SLOG_Q ? (powf(10, (X - SLOG_C) / SLOG_A) - SLOG_B) / SLOG_K2
                         : (X - SLOG_Q) / SLOG_P);
        break;
    case PL_COLOR_TRC_SCRGB:
        EACH(X * (PL_COLOR_SCRGB_WHITE / PL_COLOR_SDR_WHITE));
        return;
    default:
        return;
    }

    // scale_out
    if (pl_color_space_is_black_scaled(csp) && trc != PL_COLOR_TRC_HLG)
        EACH((csp_max - csp_min) * X + csp_min);
}

void pl_color_delinearize(const struct pl_color_space *csp, float color[3])
{
    const enum pl_color_transfer trc = csp->transfer;
    if (trc == PL_COLOR_TRC_LINEAR)
        return;

    float csp_min, csp_max;
    nominal_norm(csp, &csp_min, &csp_max);
    if (pl_color_space_is_black_scaled(csp) && trc != PL_COLOR_TRC_HLG)
        EACH((X - csp_min) / (csp_max - csp_min));
    if (trc != PL_COLOR_TRC_SCRGB)
        EACH(fmaxf(X, 0));

    switch (trc) {
    case PL_COLOR_TRC_SRGB:
        EACH(X >= 0.0031308f ? 1.055f * powf(X, 1 / 2.4f) - 0.055f : 12.92f * X);
        return;
    case PL_COLOR_TRC_BT_1886: {
        float a, b;
        plh_bt1886_params(csp_min, csp_max, &a, &b);
        EACH(powf(X / a, 1 / 2.4f) - b);
        return;
    }
    case PL_COLOR_TRC_GAMMA18: EACH(powf(X, 1 / 1.8f)); return;
    case PL_COLOR_TRC_GAMMA20: EACH(powf(X, 1 / 2.0f)); return;
    case PL_COLOR_TRC_UNKNOWN:
    case PL_COLOR_TRC_GAMMA22: EACH(powf(X, 1 / 2.2f)); return;
    case PL_COLOR_TRC_GAMMA24: EACH(powf(X, 1 / 2.4f)); return;
    case PL_COLOR_TRC_GAMMA26: EACH(powf(X, 1 / 2.6f)); return;
    case PL_COLOR_TRC_GAMMA28: EACH(powf(X, 1 / 2.8f)); return;
    case PL_COLOR_TRC_ST428:
        EACH(powf(X * 48 / 52.37f, 1 / 2.6f));
        return;
    case PL_COLOR_TRC_PRO_PHOTO:
        EACH(X >= 0.001953f ? powf(X, 1 / 1.8f) : 16 * X);
        return;
    case PL_COLOR_TRC_PQ:
        EACH(powf(X * PL_COLOR_SDR_WHITE / 10000, PQ_M1));
        EACH(powf((PQ_C1 + PQ_C2 * X) / (1 + PQ_C3 * X), PQ_M2));
        return;
    case PL_COLOR_TRC_HLG: {
        float y, b;
        plh_hlg_params(csp_min, csp_max, &y, &b);
        const pl_matrix3x3 rgb2xyz = pl_get_rgb2xyz_matrix(pl_raw_primaries_get(csp->primaries));
        const float *coef = rgb2xyz.m[1];
        float luma = coef[0] * color[0] + coef[1] * color[1] + coef[2] * color[2];
        luma = fmaxf(1e-6f, powf(luma / csp_max, (1 - y) / y));
        EACH(12 / csp_max * luma * X);
        EACH(X > 1 ? HLG_A * logf(X - HLG_B) + HLG_C : 0.5f * sqrtf(X));
        EACH((X - b) / (1 - b));
        return;
    }
    case PL_COLOR_TRC_V_LOG:
        EACH(X >= 0.01f ? VLOG_C * log10f(X + VLOG_B) + VLOG_D : 5.6f * X + 0.125f);
        return;
    case PL_COLOR_TRC_S_LOG1:
        EACH(SLOG_A * log10f(X + SLOG_B) + SLOG_C);
        return;
    case PL_COLOR_TRC_S_LOG2:
        EACH(X >= 0 ? SLOG_A * log10f(SLOG_B * X + SLOG_C) : SLOG_P * X + SLOG_Q);
        return;
    case PL_COLOR_TRC_SCRGB:
        EACH(X * (PL_COLOR_SDR_WHITE / PL_COLOR_SCRGB_WHITE));
        return;
    default:
        return;
    }
}

/* ------------------------------------------------------------------------ */
/* colour space inference                                                    */

void pl_color_space_merge(struct pl_color_space *orig, const struct pl_color_space *update)
{
    orig->primaries = PL_DEF(orig->primaries, update->primaries);
    orig->transfer = PL_DEF(orig->transfer, update->transfer);
    pl_hdr_metadata_merge(&orig->hdr, &update->hdr);
}

bool pl_color_space_equal(const struct pl_color_space *c1, const struct pl_color_space *c2)
{
    return c1->primaries == c2->primaries && c1->transfer == c2->transfer &&
           pl_hdr_metadata_equal(&c1->hdr, &c2->hdr);
}

// HDR10+ MaxSCL/avg -> luminance estimate, weighting by how monochromatic
// the brightest component is
static void luma_from_maxrgb(const struct pl_color_space *csp, enum pl_hdr_scaling scaling,
                             float *out_max, float *out_avg)
{
    const float maxscl = MAX3(csp->hdr.scene_max[0], csp->hdr.scene_max[1], csp->hdr.scene_max[2]);
    if (!maxscl)
        return;

    struct pl_raw_primaries prim = csp->hdr.prim;
    pl_raw_primaries_merge(&prim, pl_raw_primaries_get(csp->primaries));
    const pl_matrix3x3 rgb2xyz = pl_get_rgb2xyz_matrix(&prim);

    const float max_luma = rgb2xyz.m[1][0] * csp->hdr.scene_max[0] +
                           rgb2xyz.m[1][1] * csp->hdr.scene_max[1] +
                           rgb2xyz.m[1][2] * csp->hdr.scene_max[2];
    const float coef = max_luma / maxscl;
    *out_max = pl_hdr_rescale(PL_HDR_NITS, scaling, max_luma);
    *out_avg = pl_hdr_rescale(PL_HDR_NITS, scaling, coef * csp->hdr.scene_avg);
}

// The nominal range of a colour space is assembled from up to three kinds of metadata, in
// increasing order of authority, then sanitised and completed from the transfer function.
struct luma_range { float min, max, avg; };

static bool md_selected(enum pl_hdr_metadata_type asked, enum pl_hdr_metadata_type kind)
{
    return asked == PL_HDR_METADATA_ANY || asked == kind;
}

static void range_from_metadata(const struct pl_nominal_luma_params *params, struct luma_range *r)
{
    const struct pl_hdr_metadata *hdr = &params->color->hdr;
    const enum pl_hdr_scaling to = params->scaling;
    if (params->metadata == PL_HDR_METADATA_NONE)
        return;

    // static mastering display metadata (any selection includes it); MaxCLL stands in for
    // a missing peak
    r->min = pl_hdr_rescale(PL_HDR_NITS, to, hdr->min_luma);
    r->max = pl_hdr_rescale(PL_HDR_NITS, to, hdr->max_luma);
    if (!r->max && hdr->max_cll)
        r->max = pl_hdr_rescale(PL_HDR_NITS, to, hdr->max_cll);

    if (md_selected(params->metadata, PL_HDR_METADATA_HDR10PLUS) &&
        pl_hdr_metadata_contains(hdr, PL_HDR_METADATA_HDR10PLUS))
        luma_from_maxrgb(params->color, to, &r->max, &r->avg);

    if (md_selected(params->metadata, PL_HDR_METADATA_CIE_Y) &&
        pl_hdr_metadata_contains(hdr, PL_HDR_METADATA_CIE_Y)) {
        r->max = pl_hdr_rescale(PL_HDR_PQ, to, hdr->max_pq_y);
        r->avg = pl_hdr_rescale(PL_HDR_PQ, to, hdr->avg_pq_y);
    }
}

void pl_color_space_nominal_luma_ex(const struct pl_nominal_luma_params *params)
{
    if (!params || (!params->out_min && !params->out_max && !params->out_avg))
        return;

    const enum pl_color_transfer trc = params->color->transfer;
    const enum pl_hdr_scaling to = params->scaling;
    struct luma_range r = {0};
    range_from_metadata(params, &r);

    // whatever was tagged must lie inside what PQ can express, and be a range
    const float floor_ = pl_hdr_rescale(PL_HDR_NITS, to, PL_COLOR_HDR_BLACK);
    const float ceil_  = pl_hdr_rescale(PL_HDR_PQ, to, 1.0f);
    if (r.max)
        r.max = PL_CLAMP(r.max, floor_, ceil_);
    if (r.min)
        r.min = PL_CLAMP(r.min, floor_, ceil_);
    const bool inverted = r.max && r.min >= r.max;
    if (inverted || r.min >= ceil_)
        r.min = r.max = 0;

    // the rest follows from the transfer function
    if (!r.max) {
        r.max = trc == PL_COLOR_TRC_HLG
              ? pl_hdr_rescale(PL_HDR_NITS, to, PL_COLOR_HLG_PEAK)
              : pl_hdr_rescale(PL_HDR_NORM, to, pl_color_transfer_nominal_peak(trc));
    }
    if (!r.min) {
        if (pl_color_transfer_is_hdr(trc)) {
            r.min = floor_;
        } else {
            // SDR: the nominal contrast below the peak
            const float peak_nits = pl_hdr_rescale(to, PL_HDR_NITS, r.max);
            r.min = pl_hdr_rescale(PL_HDR_NITS, to, peak_nits / PL_COLOR_SDR_CONTRAST);
        }
    }
    if (r.avg)
        r.avg = PL_CLAMP(r.avg, r.min, r.max);

    if (params->out_min) *params->out_min = r.min;
    if (params->out_max) *params->out_max = r.max;
    if (params->out_avg) *params->out_avg = r.avg;
}

void pl_color_space_infer(struct pl_color_space *space)
{
    space->primaries = PL_DEF(space->primaries, PL_COLOR_PRIM_BT_709);
    space->transfer = PL_DEF(space->transfer, PL_COLOR_TRC_BT_1886);

    pl_color_space_nominal_luma_ex(pl_nominal_luma_params(
        .color      = space,
        .metadata   = PL_HDR_METADATA_HDR10,
        .scaling    = PL_HDR_NITS,
        .out_max    = &space->hdr.max_luma,
        .out_min    = space->hdr.min_luma ? NULL : &space->hdr.min_luma, // keep a tagged minimum
    ));

    if (!pl_primaries_valid(&space->hdr.prim))
        space->hdr.prim = *pl_raw_primaries_get(space->primaries);
}

// Transfer function of a display space left untagged, given the space it is paired with:
// close-to-2.2 curves are adopted as they are (no needless small adaptation), HDR and log
// curves pair with BT.1886 (which models an SDR display's contrast), everything else with a
// pure power curve (no black crush).
static enum pl_color_transfer companion_transfer(enum pl_color_transfer ref)
{
    static const enum pl_color_transfer pairs[][2] = {
        { PL_COLOR_TRC_BT_1886,   PL_COLOR_TRC_BT_1886 },
        { PL_COLOR_TRC_SRGB,      PL_COLOR_TRC_SRGB },
        { PL_COLOR_TRC_GAMMA22,   PL_COLOR_TRC_GAMMA22 },
        { PL_COLOR_TRC_PQ,        PL_COLOR_TRC_BT_1886 },
        { PL_COLOR_TRC_HLG,       PL_COLOR_TRC_BT_1886 },
        { PL_COLOR_TRC_V_LOG,     PL_COLOR_TRC_BT_1886 },
        { PL_COLOR_TRC_S_LOG1,    PL_COLOR_TRC_BT_1886 },
        { PL_COLOR_TRC_S_LOG2,    PL_COLOR_TRC_BT_1886 },
        { PL_COLOR_TRC_PRO_PHOTO, PL_COLOR_TRC_SRGB },
    };
    for (size_t i = 0; i < PL_ARRAY_SIZE(pairs); i++) {
        if (pairs[i][0] == ref)
            return pairs[i][1];
    }
    return PL_COLOR_TRC_GAMMA22;
}

static void infer_both_ref(struct pl_color_space *space, struct pl_color_space *ref)
{
    pl_color_space_infer(ref);
    if (!space->primaries) {
        // a wide-gamut partner does not make an untagged space wide-gamut
        const bool wide = pl_color_primaries_is_wide_gamut(ref->primaries);
        space->primaries = wide ? PL_COLOR_PRIM_BT_709 : ref->primaries;
    }
    if (!space->transfer)
        space->transfer = companion_transfer(ref->transfer);
    pl_color_space_infer(space);
}

void pl_color_space_infer_ref(struct pl_color_space *space, const struct pl_color_space *refp)
{
    struct pl_color_space ref = *refp;
    infer_both_ref(space, &ref);
}

void pl_color_space_infer_map(struct pl_color_space *src, struct pl_color_space *dst)
{
    const bool unknown_src_contrast = !src->hdr.min_luma;
    const bool unknown_dst_contrast = !dst->hdr.min_luma;

    infer_both_ref(dst, src);

    // an untagged, black-scaled source adopts the target's black point ...
    const bool dynamic_src_contrast = pl_color_space_is_black_scaled(src) ||
                                      src->transfer == PL_COLOR_TRC_BT_1886;
    if (unknown_src_contrast && dynamic_src_contrast)
        src->hdr.min_luma = dst->hdr.min_luma;

    // ... and vice versa between two SDR curves
    if (unknown_dst_contrast && !pl_color_space_is_hdr(src) && !pl_color_space_is_hdr(dst))
        dst->hdr.min_luma = src->hdr.min_luma;

    // HLG is display-referred to the output peak
    if (src->transfer == PL_COLOR_TRC_HLG && pl_color_space_is_hdr(dst))
        src->hdr.max_luma = dst->hdr.max_luma;
}

const struct pl_color_adjustment pl_color_adjustment_neutral = { PL_COLOR_ADJUSTMENT_NEUTRAL };

void pl_chroma_location_offset(enum pl_chroma_location loc, float *x, float *y)
{
    *x = *y = 0;
    loc = PL_DEF(loc, PL_CHROMA_LEFT);
    if (loc == PL_CHROMA_LEFT || loc == PL_CHROMA_TOP_LEFT || loc == PL_CHROMA_BOTTOM_LEFT)
        *x = -0.5;
    if (loc == PL_CHROMA_TOP_LEFT || loc == PL_CHROMA_TOP_CENTER)
        *y = -0.5;
    if (loc == PL_CHROMA_BOTTOM_LEFT || loc == PL_CHROMA_BOTTOM_CENTER)
        *y = 0.5;
}

/* ------------------------------------------------------------------------ */
/* white points, primaries, matrices                                         */

struct pl_cie_xy pl_daylight_from_temp(float temp)
{
    temp = PL_CLAMP(temp, 1000, 25000);
    const double ti = 1000.0 / temp, ti2 = ti * ti, ti3 = ti2 * ti;
    const double x = temp <= 7000
        ? -4.6070 * ti3 + 2.9678 * ti2 + 0.09911 * ti + 0.244063
        : -2.0064 * ti3 + 1.9018 * ti2 + 0.24748 * ti + 0.237040;
    return (struct pl_cie_xy) { .x = x, .y = -3 * (x * x) + 2.87 * x - 0.275 };
}

struct pl_cie_xy pl_blackbody_from_temp(float temp)
{
    temp = PL_CLAMP(temp, 1667, 25000);
    const double ti = 1000.0 / temp, ti2 = ti * ti, ti3 = ti2 * ti;
    const double x = temp <= 4000
        ? -0.2661239 * ti3 - 0.2343580 * ti2 + 0.8776956 * ti + 0.179910
        : -3.0258469 * ti3 + 2.1070379 * ti2 + 0.2226347 * ti + 0.240390;
    const double x2 = x * x, x3 = x2 * x;
    double y;
    if (temp <= 2222) {
        y = -1.1063814 * x3 - 1.34811020 * x2 + 2.18555832 * x - 0.20219683;
    } else if (temp <= 4000) {
        y = -0.9549476 * x3 - 1.37418593 * x2 + 2.09137015 * x - 0.16748867;
    } else {
        y =  3.0817580 * x3 - 5.87338670 * x2 + 3.75112997 * x - 0.37001483;
    }
    return (struct pl_cie_xy) { x, y };
}

struct pl_cie_xy pl_white_from_temp(float temp)
{
    const struct pl_cie_xy a = pl_blackbody_from_temp(temp);
    const struct pl_cie_xy b = pl_daylight_from_temp(temp);
    float f = (temp - 2500) / (4000 - 2500);
    f = PL_CLAMP(f, 0.0f, 1.0f);
    return (struct pl_cie_xy) { .x = MIXF(a.x, b.x, f), .y = MIXF(a.y, b.y, f) };
}

bool pl_raw_primaries_equal(const struct pl_raw_primaries *a, const struct pl_raw_primaries *b)
{
    return pl_cie_xy_equal(&a->red, &b->red) && pl_cie_xy_equal(&a->green, &b->green) &&
           pl_cie_xy_equal(&a->blue, &b->blue) && pl_cie_xy_equal(&a->white, &b->white);
}

bool pl_raw_primaries_similar(const struct pl_raw_primaries *a, const struct pl_raw_primaries *b)
{
    const float delta = fabsf(a->red.x   - b->red.x)   + fabsf(a->red.y   - b->red.y)   +
                        fabsf(a->green.x - b->green.x) + fabsf(a->green.y - b->green.y) +
                        fabsf(a->blue.x  - b->blue.x)  + fabsf(a->blue.y  - b->blue.y)  +
                        fabsf(a->white.x - b->white.x) + fabsf(a->white.y - b->white.y);
    return delta < 0.001;
}

void pl_raw_primaries_merge(struct pl_raw_primaries *orig, const struct pl_raw_primaries *update)
{
    float *pa = (float *) orig;
    const float *pb = (const float *) update;
    for (int i = 0; i < 8; i++)
        pa[i] = PL_DEF(pa[i], pb[i]);
}

#define W_D50 {0.3457, 0.3585}
#define W_D65 {0.3127, 0.3290}
#define W_C   {0.3100, 0.3160}
#define W_E   {1.0/3.0, 1.0/3.0}
#define W_DCI {0.3140, 0.3510}
#define W_ACES {0.32168, 0.33767}
#define PRIM(rx, ry, gx, gy, bx, by, wp) { {rx, ry}, {gx, gy}, {bx, by}, wp }

const struct pl_raw_primaries *pl_raw_primaries_get(enum pl_color_primaries prim)
{
    // ITU-R BT.470-6 / BT.601-7 / BT.709-5 / BT.2020-0, SMPTE RP 431-2, vendor manuals
    static const struct pl_raw_primaries table[PL_COLOR_PRIM_COUNT] = {
        [PL_COLOR_PRIM_BT_470M]    = PRIM(0.670, 0.330, 0.210, 0.710, 0.140, 0.080, W_C),
        [PL_COLOR_PRIM_BT_601_525] = PRIM(0.630, 0.340, 0.310, 0.595, 0.155, 0.070, W_D65),
        [PL_COLOR_PRIM_BT_601_625] = PRIM(0.640, 0.330, 0.290, 0.600, 0.150, 0.060, W_D65),
        [PL_COLOR_PRIM_BT_709]     = PRIM(0.640, 0.330, 0.300, 0.600, 0.150, 0.060, W_D65),
        [PL_COLOR_PRIM_BT_2020]    = PRIM(0.708, 0.292, 0.170, 0.797, 0.131, 0.046, W_D65),
        [PL_COLOR_PRIM_APPLE]      = PRIM(0.625, 0.340, 0.280, 0.595, 0.115, 0.070, W_D65),
        [PL_COLOR_PRIM_ADOBE]      = PRIM(0.640, 0.330, 0.210, 0.710, 0.150, 0.060, W_D65),
        [PL_COLOR_PRIM_PRO_PHOTO]  = PRIM(0.7347, 0.2653, 0.1596, 0.8404, 0.0366, 0.0001, W_D50),
        [PL_COLOR_PRIM_CIE_1931]   = PRIM(0.7347, 0.2653, 0.2738, 0.7174, 0.1666, 0.0089, W_E),
        [PL_COLOR_PRIM_DCI_P3]     = PRIM(0.680, 0.320, 0.265, 0.690, 0.150, 0.060, W_DCI),
        [PL_COLOR_PRIM_DISPLAY_P3] = PRIM(0.680, 0.320, 0.265, 0.690, 0.150, 0.060, W_D65),
        [PL_COLOR_PRIM_V_GAMUT]    = PRIM(0.730, 0.280, 0.165, 0.840, 0.100, -0.03, W_D65),
        [PL_COLOR_PRIM_S_GAMUT]    = PRIM(0.730, 0.280, 0.140, 0.855, 0.100, -0.05, W_D65),
        [PL_COLOR_PRIM_FILM_C]     = PRIM(0.681, 0.319, 0.243, 0.692, 0.145, 0.049, W_C),
        [PL_COLOR_PRIM_EBU_3213]   = PRIM(0.630, 0.340, 0.295, 0.605, 0.155, 0.077, W_D65),
        [PL_COLOR_PRIM_ACES_AP0]   = PRIM(0.7347, 0.2653, 0.0000, 1.0000, 0.0001, -0.0770, W_ACES),
        [PL_COLOR_PRIM_ACES_AP1]   = PRIM(0.713, 0.293, 0.165, 0.830, 0.128, 0.044, W_ACES),
    };

    if (!prim)
        prim = PL_COLOR_PRIM_BT_709;
    return &table[prim];
}

// RGB -> XYZ from chromaticities (Lindbloom): scale each primary's XYZ column
// so that RGB = (1,1,1) maps to the white point
pl_matrix3x3 pl_get_rgb2xyz_matrix(const struct pl_raw_primaries *prim)
{
    pl_matrix3x3 out = {{{0}}};
    const float X[4] = { pl_cie_X(prim->red), pl_cie_X(prim->green),
                         pl_cie_X(prim->blue), pl_cie_X(prim->white) };
    const float Z[4] = { pl_cie_Z(prim->red), pl_cie_Z(prim->green),
                         pl_cie_Z(prim->blue), pl_cie_Z(prim->white) };

    for (int i = 0; i < 3; i++) {
        out.m[0][i] = X[i];
        out.m[1][i] = 1;
        out.m[2][i] = Z[i];
    }
    pl_matrix3x3_invert(&out);

    float S[3];
    for (int i = 0; i < 3; i++)
        S[i] = out.m[i][0] * X[3] + out.m[i][1] * 1 + out.m[i][2] * Z[3];

    for (int i = 0; i < 3; i++) {
        out.m[0][i] = S[i] * X[i];
        out.m[1][i] = S[i] * 1;
        out.m[2][i] = S[i] * Z[i];
    }
    return out;
}

pl_matrix3x3 pl_get_xyz2rgb_matrix(const struct pl_raw_primaries *prim)
{
    pl_matrix3x3 out = pl_get_rgb2xyz_matrix(prim);
    pl_matrix3x3_invert(&out);
    return out;
}

// CAT16 one-step von Kries adaptation
static const pl_matrix3x3 m_cat16 = {{
    {  0.401288, 0.650173, -0.051461 },
    { -0.250268, 1.204414,  0.045854 },
    { -0.002079, 0.048952,  0.953127 },
}};

// mat := mat * (XYZ_dest <- XYZ_src)
static void chromatic_adaptation(struct pl_cie_xy src, struct pl_cie_xy dest, pl_matrix3x3 *mat)
{
    if (fabs(src.x - dest.x) < 1e-6 && fabs(src.y - dest.y) < 1e-6)
        return; // same white point

    float C[3][2];
    for (int i = 0; i < 3; i++) {
        C[i][0] = m_cat16.m[i][0] * pl_cie_X(src)  + m_cat16.m[i][1] * 1 +
                  m_cat16.m[i][2] * pl_cie_Z(src);
        C[i][1] = m_cat16.m[i][0] * pl_cie_X(dest) + m_cat16.m[i][1] * 1 +
                  m_cat16.m[i][2] * pl_cie_Z(dest);
    }

    pl_matrix3x3 tmp = {0};
    for (int i = 0; i < 3; i++)
        tmp.m[i][i] = C[i][1] / C[i][0];
    pl_matrix3x3_mul(&tmp, &m_cat16);

    pl_matrix3x3 ma_inv = m_cat16;
    pl_matrix3x3_invert(&ma_inv);
    pl_matrix3x3_mul(mat, &ma_inv);
    pl_matrix3x3_mul(mat, &tmp);
}

pl_matrix3x3 pl_get_adaptation_matrix(struct pl_cie_xy src, struct pl_cie_xy dst)
{
    struct pl_raw_primaries csp = *pl_raw_primaries_get(PL_COLOR_PRIM_BT_709);
    csp.white = src;

    pl_matrix3x3 rgb2xyz = pl_get_rgb2xyz_matrix(&csp);
    pl_matrix3x3 xyz2rgb = rgb2xyz;
    pl_matrix3x3_invert(&xyz2rgb);

    chromatic_adaptation(src, dst, &xyz2rgb);
    pl_matrix3x3_mul(&xyz2rgb, &rgb2xyz);
    return xyz2rgb;
}

pl_matrix3x3 pl_ipt_rgb2lms(const struct pl_raw_primaries *prim)
{
    static const pl_matrix3x3 hpe = {{ // Hunt-Pointer-Estevez XYZ->LMS (D65)
        {  0.40024f, 0.70760f, -0.08081f },
        { -0.22630f, 1.16532f,  0.04570f },
        {  0.00000f, 0.00000f,  0.91822f },
    }};

    const float c = 0.04; // 4% crosstalk
    pl_matrix3x3 m = {{
        { 1 - 2*c,       c,       c },
        {       c, 1 - 2*c,       c },
        {       c,       c, 1 - 2*c },
    }};
    pl_matrix3x3_mul(&m, &hpe);

    static const struct pl_cie_xy d65 = W_D65;
    chromatic_adaptation(prim->white, d65, &m);

    const pl_matrix3x3 rgb2xyz = pl_get_rgb2xyz_matrix(prim);
    pl_matrix3x3_mul(&m, &rgb2xyz);
    return m;
}

pl_matrix3x3 pl_ipt_lms2rgb(const struct pl_raw_primaries *prim)
{
    pl_matrix3x3 m = pl_ipt_rgb2lms(prim);
    pl_matrix3x3_invert(&m);
    return m;
}

// Ebner & Fairchild IPT (1998), and its numerical inverse
const pl_matrix3x3 pl_ipt_lms2ipt = {{
    { 0.4000,  0.4000,  0.2000 },
    { 4.4550, -4.8510,  0.3960 },
    { 0.8056,  0.3572, -1.1628 },
}};

const pl_matrix3x3 pl_ipt_ipt2lms = {{
    { 1.0,  0.0975689,  0.205226 },
    { 1.0, -0.1138760,  0.133217 },
    { 1.0,  0.0326151, -0.676887 },
}};

/* ---- colour blindness simulation (pl_get_cone_matrix, colorspace.c:1397-1540) ------------- */

const struct pl_cone_params pl_vision_normal        = { PL_CONE_NONE, 1.0 };
const struct pl_cone_params pl_vision_protanomaly   = { PL_CONE_L,    0.5 };
const struct pl_cone_params pl_vision_protanopia    = { PL_CONE_L,    0.0 };
const struct pl_cone_params pl_vision_deuteranomaly = { PL_CONE_M,    0.5 };
const struct pl_cone_params pl_vision_deuteranopia  = { PL_CONE_M,    0.0 };
const struct pl_cone_params pl_vision_tritanomaly   = { PL_CONE_S,    0.5 };
const struct pl_cone_params pl_vision_tritanopia    = { PL_CONE_S,    0.0 };
const struct pl_cone_params pl_vision_monochromacy  = { PL_CONE_LM,   0.0 };
const struct pl_cone_params pl_vision_achromatopsia = { PL_CONE_LMS,  0.0 };

// Coefficient of cone `j` when cone `i` is rebuilt from cones j and k such that the colours
// `p` (a primary) and `w` (white) keep their response:  (p_i - p_k w_i / w_k) / (p_j - p_k w_j / w_k)
static float cone_coeff(const float p[3], const float w[3], int i, int j, int k)
{
    return (p[i] - p[k] * w[i] / w[k]) / (p[j] - p[k] * w[j] / w[k]);
}

pl_matrix3x3 pl_get_cone_matrix(const struct pl_cone_params *params,
                                const struct pl_raw_primaries *prim)
{
    if (params->cones == PL_CONE_NONE)
        return pl_matrix3x3_identity;

    // LMS <- RGB (CAT16 cone space)
    pl_matrix3x3 rgb2lms = m_cat16;
    const pl_matrix3x3 rgb2xyz = pl_get_rgb2xyz_matrix(prim);
    pl_matrix3x3_mul(&rgb2lms, &rgb2xyz);

    // cone responses to red, blue and white
    float red[3] = { 1.0, 0.0, 0.0 }, blue[3] = { 0.0, 0.0, 1.0 }, white[3] = { 1.0, 1.0, 1.0 };
    pl_matrix3x3_apply(&rgb2lms, red);
    pl_matrix3x3_apply(&rgb2lms, blue);
    pl_matrix3x3_apply(&rgb2lms, white);

    const float c = params->strength;
    pl_matrix3x3 distort = pl_matrix3x3_identity;
    const int cones = params->cones;
    if (cones == PL_CONE_L || cones == PL_CONE_M || cones == PL_CONE_S) {
        // one cone is (partly) rebuilt from the other two; neutral and the opposing primary
        // (blue for L / M, red for S) are preserved
        const int i = cones == PL_CONE_L ? 0 : cones == PL_CONE_M ? 1 : 2;
        const int j = i == 0 ? 1 : 0, k = i == 2 ? 1 : 2;
        const float *p = i == 2 ? red : blue;
        const float a = cone_coeff(p, white, i, j, k), b = cone_coeff(p, white, i, k, j);
        distort.m[i][i] = c;
        distort.m[i][j] = (1.0 - c) * a;
        distort.m[i][k] = (1.0 - c) * b;
    } else if (cones == PL_CONE_LMS) {
        // rods only: roughly a mix of L and M
        const float w[3] = { 0.3605, 0.6415, -0.002 };
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
                distort.m[i][j] = (1.0 - c) * w[j] * white[i] / white[j];
                if (i == j)
                    distort.m[i][j] += c;
            }
        }
    } else {
        // two cones missing: both follow the remaining one `r`, neutral is preserved
        const int r = cones == PL_CONE_LM ? 2 : cones == PL_CONE_MS ? 0 : 1;
        for (int i = 0; i < 3; i++) {
            if (i == r)
                continue;
            distort.m[i][i] = c;
            distort.m[i][r] = (1.0 - c) * (white[i] / white[r]);
        }
    }

    // RGB <- LMS * distort * LMS <- RGB
    pl_matrix3x3 out = rgb2lms;
    pl_matrix3x3_invert(&out);
    pl_matrix3x3_mul(&out, &distort);
    pl_matrix3x3_mul(&out, &rgb2lms);
    return out;
}

pl_matrix3x3 pl_get_color_mapping_matrix(const struct pl_raw_primaries *src,
                                         const struct pl_raw_primaries *dst,
                                         enum pl_rendering_intent intent)
{
    if (intent == PL_INTENT_SATURATION)
        return pl_matrix3x3_identity; // primaries map to primaries

    // RGBd<-RGBs = RGBd<-XYZd * XYZd<-XYZs * XYZs<-RGBs
    pl_matrix3x3 xyz2rgb_d = pl_get_xyz2rgb_matrix(dst);
    if (intent != PL_INTENT_ABSOLUTE_COLORIMETRIC)
        chromatic_adaptation(src->white, dst->white, &xyz2rgb_d);

    const pl_matrix3x3 rgb2xyz_s = pl_get_rgb2xyz_matrix(src);
    pl_matrix3x3_mul(&xyz2rgb_d, &rgb2xyz_s);
    return xyz2rgb_d;
}

// signed area test of p against the line a->b
static float side(const struct pl_cie_xy p, const struct pl_cie_xy a, const struct pl_cie_xy b)
{
    return (p.x - b.x) * (a.y - b.y) - (a.x - b.x) * (p.y - b.y);
}

static bool point_in_gamut(struct pl_cie_xy point, const struct pl_raw_primaries *prim)
{
    const float d1 = side(point, prim->red, prim->green),
                d2 = side(point, prim->green, prim->blue),
                d3 = side(point, prim->blue, prim->red);
    const bool has_neg = d1 < -1e-6f || d2 < -1e-6f || d3 < -1e-6f,
               has_pos = d1 >  1e-6f || d2 >  1e-6f || d3 >  1e-6f;
    return !(has_neg && has_pos);
}

bool pl_primaries_superset(const struct pl_raw_primaries *a, const struct pl_raw_primaries *b)
{
    return point_in_gamut(b->red, a) && point_in_gamut(b->green, a) && point_in_gamut(b->blue, a);
}

bool pl_primaries_valid(const struct pl_raw_primaries *prim)
{
    const float area = (prim->blue.x - prim->green.x) * (prim->red.y  - prim->green.y)
                     - (prim->red.x  - prim->green.x) * (prim->blue.y - prim->green.y);
    return fabs(area) > 1e-6 && point_in_gamut(prim->white, prim);
}

static inline float xy_dist2(struct pl_cie_xy a, struct pl_cie_xy b)
{
    const float dx = a.x - b.x, dy = a.y - b.y;
    return dx * dx + dy * dy;
}

bool pl_primaries_compatible(const struct pl_raw_primaries *a, const struct pl_raw_primaries *b)
{
    const float RR = xy_dist2(a->red, b->red),   RG = xy_dist2(a->red, b->green),
                RB = xy_dist2(a->red, b->blue),  GR = xy_dist2(a->green, b->red),
                GG = xy_dist2(a->green, b->green), GB = xy_dist2(a->green, b->blue),
                BR = xy_dist2(a->blue, b->red),  BG = xy_dist2(a->blue, b->green),
                BB = xy_dist2(a->blue, b->blue);
    return RR < RG && RR < RB && GG < GR && GG < GB && BB < BR && BB < BG;
}

static struct pl_cie_xy line_intersection(struct pl_cie_xy a, struct pl_cie_xy b,
                                          struct pl_cie_xy c, struct pl_cie_xy d)
{
    const float det = (a.x - b.x) * (c.y - d.y) - (a.y - b.y) * (c.x - d.x);
    const float t = ((a.x - c.x) * (c.y - d.y) - (a.y - c.y) * (c.x - d.x)) / det;
    return (struct pl_cie_xy) {
        .x = t ? a.x + t * (b.x - a.x) : 0.0f,
        .y = t ? a.y + t * (b.y - a.y) : 0.0f,
    };
}

// x, y, z clockwise; a, b, c the enclosing gamut: clip vertex y
static struct pl_cie_xy clip_vertex(struct pl_cie_xy x, struct pl_cie_xy y, struct pl_cie_xy z,
                                    struct pl_cie_xy a, struct pl_cie_xy b, struct pl_cie_xy c)
{
    const float d1 = side(y, a, b);
    const float d2 = side(y, b, c);
    if (d1 <= 0.0f && d2 <= 0.0f)
        return y;
    if (d1 > 0.0f && d2 > 0.0f)
        return b;
    if (d1 > 0.0f)
        return line_intersection(a, b, y, z);
    return line_intersection(x, y, b, c);
}

struct pl_raw_primaries pl_primaries_clip(const struct pl_raw_primaries *src,
                                          const struct pl_raw_primaries *dst)
{
    return (struct pl_raw_primaries) {
        .red   = clip_vertex(src->green, src->red, src->blue, dst->green, dst->red, dst->blue),
        .green = clip_vertex(src->blue, src->green, src->red, dst->blue, dst->green, dst->red),
        .blue  = clip_vertex(src->red, src->blue, src->green, dst->red, dst->blue, dst->green),
        .white = src->white,
    };
}

/* ------------------------------------------------------------------------ */
/* YCbCr-like decoding matrices                                              */

// Y'CbCr -> R'G'B' for luma weights (lr, lg, lb): Y -> (1,1,1), Cb/Cr
// orthogonal to the luma vector, scaled to cover the RGB cube
static pl_matrix3x3 ycbcr_matrix(float lr, float lg, float lb)
{
    return (pl_matrix3x3) {{
        {1, 0,                    2 * (1-lr)          },
        {1, -2 * (1-lb) * lb/lg, -2 * (1-lr) * lr/lg  },
        {1,  2 * (1-lb),          0                   },
    }};
}

static void hue_sat(pl_matrix3x3 *m, const struct pl_color_adjustment *params)
{
    // rotate / scale the chroma subvector
    const float huecos = params->saturation * cos(params->hue);
    const float huesin = params->saturation * sin(params->hue);
    for (int i = 0; i < 3; i++) {
        const float u = m->m[i][1], v = m->m[i][2];
        m->m[i][1] = huecos * u - huesin * v;
        m->m[i][2] = huesin * u + huecos * v;
    }
}

pl_transform3x3 pl_color_repr_decode(struct pl_color_repr *repr,
                                     const struct pl_color_adjustment *params)
{
    params = PL_DEF(params, &pl_color_adjustment_neutral);

    pl_matrix3x3 m = pl_matrix3x3_identity;
    switch (repr->sys) {
    case PL_COLOR_SYSTEM_BT_709:     m = ycbcr_matrix(0.2126, 0.7152, 0.0722); break;
    case PL_COLOR_SYSTEM_BT_601:     m = ycbcr_matrix(0.2990, 0.5870, 0.1140); break;
    case PL_COLOR_SYSTEM_SMPTE_240M: m = ycbcr_matrix(0.2122, 0.7013, 0.0865); break;
    case PL_COLOR_SYSTEM_BT_2020_NC: m = ycbcr_matrix(0.2627, 0.6780, 0.0593); break;
    case PL_COLOR_SYSTEM_BT_2020_C:
        // component shuffle only; chroma stays in [-0.5, 0.5] for the
        // non-linear constant-luminance stage
        m = (pl_matrix3x3) {{ {0, 0, 1}, {1, 0, 0}, {0, 1, 0} }};
        break;
    case PL_COLOR_SYSTEM_BT_2100_PQ: {
        // ICtCp -> L'M'S' (inverse of the spec matrix, ITU-T H-Suppl. 18)
        static const float lm_t = 0.008609, lm_p = 0.111029625;
        m = (pl_matrix3x3) {{
            {1.0,  lm_t,  lm_p},
            {1.0, -lm_t, -lm_p},
            {1.0, 0.560031, -0.320627},
        }};
        break;
    }
    case PL_COLOR_SYSTEM_BT_2100_HLG: {
        static const float lm_t = 0.01571858011, lm_p = 0.2095810681;
        m = (pl_matrix3x3) {{
            {1.0,  lm_t,  lm_p},
            {1.0, -lm_t, -lm_p},
            {1.0, 1.02127108, -0.605274491},
        }};
        break;
    }
    case PL_COLOR_SYSTEM_DOLBYVISION:
        if (repr->dovi)
            m = repr->dovi->nonlinear;
        break;
    case PL_COLOR_SYSTEM_YCGCO:
        m = (pl_matrix3x3) {{ {1, -1, 1}, {1, 1, 0}, {1, -1, -1} }};
        break;
    case PL_COLOR_SYSTEM_YCGCO_RE:
    case PL_COLOR_SYSTEM_YCGCO_RO:
        m = (pl_matrix3x3) {{ {1, -0.5, 0.5}, {1, 0.5, 0}, {1, -0.5, -0.5} }};
        break;
    case PL_COLOR_SYSTEM_XYZ:
        // assume DCI-P3 primaries for DCDM content
        m = pl_get_xyz2rgb_matrix(pl_raw_primaries_get(PL_COLOR_PRIM_DCI_P3));
        break;
    default:
        break; // RGB / unknown: identity
    }

    if (pl_color_system_is_ycbcr_like(repr->sys)) {
        hue_sat(&m, params);
    } else if (params->saturation != 1.0 || params->hue != 0.0) {
        // emulate hue/saturation on RGB through a BT.709 YCbCr round trip
        pl_matrix3x3 yuv2rgb = ycbcr_matrix(0.2126, 0.7152, 0.0722);
        pl_matrix3x3 rgb2yuv = yuv2rgb;
        pl_matrix3x3_invert(&rgb2yuv);
        hue_sat(&yuv2rgb, params);
        pl_matrix3x3_rmul(&rgb2yuv, &m);
        pl_matrix3x3_rmul(&yuv2rgb, &m);
    }

    if (params->temperature) {
        const struct pl_cie_xy src = pl_white_from_temp(6500);
        const struct pl_cie_xy dst = pl_white_from_temp(6500 + 3500 * params->temperature);
        const pl_matrix3x3 adapt = pl_get_adaptation_matrix(src, dst);
        pl_matrix3x3_rmul(&adapt, &m);
    }

    pl_transform3x3 out = { .mat = m };
    const int bit_depth = PL_DEF(repr->bits.sample_depth, PL_DEF(repr->bits.color_depth, 8));

    // code value ranges, as fractions of the sampled [0,1] range
    const double scale = (1LL << bit_depth) / ((1LL << bit_depth) - 1.0);
    double ymax = 1.0, ymin = 0.0, cmax = 1.0;
    double cmid = 128 / 256. * scale; // *not* exactly 0.5
    if (pl_color_levels_guess(repr) == PL_COLOR_LEVELS_LIMITED) {
        ymax = 235 / 256. * scale;
        ymin =  16 / 256. * scale;
        cmax = 240 / 256. * scale;
    }

    double ymul = 1.0 / (ymax - ymin);
    double cmul = 0.5 / (cmax - cmid);

    if (repr->sys == PL_COLOR_SYSTEM_YCGCO_RE || repr->sys == PL_COLOR_SYSTEM_YCGCO_RO) {
        const int additional_bits = repr->sys == PL_COLOR_SYSTEM_YCGCO_RE ? 2 : 1;
        const double max_y = (1LL << (bit_depth - additional_bits)) - 1;
        const double max_c = (1LL << (bit_depth)) - 1;
        ymul = cmul = max_c / max_y;
        ymin = 0;
        cmid = (1 << (bit_depth - 1)) / max_c;
    }

    double mul[3]   = { ymul, ymul, ymul };
    double black[3] = { ymin, ymin, ymin };
    if (repr->sys == PL_COLOR_SYSTEM_DOLBYVISION && repr->dovi) {
        // the stream's matrix already expands the levels; its offsets still apply (:1857-1864)
        for (int i = 0; i < 3; i++) {
            mul[i] = 1.0;
            black[i] = repr->dovi->nonlinear_offset[i] * scale;
        }
    } else if (pl_color_system_is_ycbcr_like(repr->sys)) {
        mul[1]   = mul[2]   = cmul;
        black[1] = black[2] = cmid;
    }

    // contrast = gain, brightness = constant lift
    for (int i = 0; i < 3; i++) {
        mul[i] *= params->contrast;
        out.c[i] += params->brightness;
    }

    // fold the range expansion into the matrix, keeping black -> RGB 0
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            out.mat.m[i][j] *= mul[j];
            out.c[i] -= out.mat.m[i][j] * black[j];
        }
    }

    pl_matrix3x3_scale(&out.mat, pl_color_repr_normalize(repr));

    repr->sys    = PL_COLOR_SYSTEM_RGB;
    repr->levels = PL_COLOR_LEVELS_FULL;
    return out;
}

/* ---- ICC profile descriptions (carried through pl_frame, never interpreted here) ---- */

bool pl_icc_profile_equal(const struct pl_icc_profile *a, const struct pl_icc_profile *b)
{
    if (a->len != b->len)
        return false;
    return !a->len || a->signature == b->signature;    // no profile at all == no profile
}

void pl_icc_profile_compute_signature(struct pl_icc_profile *profile)
{
    // (an empty profile gets the hash of nothing, not 0: src/colorspace.c:1910-1916 assigns 0 and
    // then goes on to overwrite it)
    profile->signature = plh_mem_hash(profile->data, profile->len);
}
