/*
 * libplacebo-hip — colour stages: host half of src/shaders/colorspace.c.
 *
 * Every function records the op(s) whose device code lives in
 * csrc/hip/{transfer,colormap}.hiph / k_peak.hip, computing the constants the
 * way the reference embeds them into GLSL:
 *   SH_FLOAT(x)  -> the float itself
 *   "%f" printf  -> plh_fmtf(x): 6-decimal rounding (PQ / HLG / log constants!)
 *   GLSL constant expressions (1.0/2.4, vec3(1.0/C)) -> folded in fp32
 *
 *   pl_shader_set_alpha            colorspace.c:26-49
 *   pl_shader_decode/encode_color  :275-573
 *   pl_shader_linearize/delinearize:589-847
 *   pl_shader_sigmoidize/un        :851-894
 *   pl_shader_detect_peak + host reduction  :1020-1381
 *   pl_shader_color_map_ex         :1612-2024
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <libplacebo/shaders/colorspace.h>

#include "shaders_priv.h"
#include "cache_priv.h"
#include "colorspace_priv.h"

#define MIXF(a, b, x) ((x) * (b) + (1 - (x)) * (a))

const struct pl_sigmoid_params pl_sigmoid_default_params = { PL_SIGMOID_DEFAULTS };
const struct pl_peak_detect_params pl_peak_detect_default_params = { PL_PEAK_DETECT_DEFAULTS };
const struct pl_peak_detect_params pl_peak_detect_high_quality_params = { PL_PEAK_DETECT_HQ_DEFAULTS };
const struct pl_color_map_params pl_color_map_default_params = { PL_COLOR_MAP_DEFAULTS };
const struct pl_color_map_params pl_color_map_high_quality_params = { PL_COLOR_MAP_HQ_DEFAULTS };

static struct plh_op *simple_op(pl_shader sh, int kind, const char *listing)
{
    struct plh_op *op = sh_op(sh, kind);
    if (op)
        sh_listf(sh, "%s\n", listing);
    return op;
}

/* ------------------------------------------------------------------------ */

void pl_shader_set_alpha(pl_shader sh, struct pl_color_repr *repr, enum pl_alpha_mode mode)
{
    const bool src_has_alpha = repr->alpha == PL_ALPHA_INDEPENDENT ||
                               repr->alpha == PL_ALPHA_PREMULTIPLIED;
    const bool dst_not_premul = mode == PL_ALPHA_INDEPENDENT || mode == PL_ALPHA_NONE;

    if (repr->alpha == PL_ALPHA_PREMULTIPLIED && dst_not_premul) {
        simple_op(sh, PLH_OP_UNPREMULTIPLY, "unpremultiply()");
        repr->alpha = PL_ALPHA_INDEPENDENT;
    }
    if (repr->alpha == PL_ALPHA_INDEPENDENT && mode == PL_ALPHA_PREMULTIPLIED) {
        simple_op(sh, PLH_OP_PREMULTIPLY, "premultiply()");
        repr->alpha = PL_ALPHA_PREMULTIPLIED;
    }
    if (src_has_alpha && mode == PL_ALPHA_NONE) {
        simple_op(sh, PLH_OP_ALPHA_ONE, "alpha_one()");
        repr->alpha = PL_ALPHA_NONE;
    }
}

static void op_affine(pl_shader sh, const pl_transform3x3 *tr, const char *what)
{
    struct plh_op *op = sh_op(sh, PLH_OP_AFFINE);
    if (!op)
        return;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            op->f[3 * i + j] = tr->mat.m[i][j];
        op->f[9 + i] = tr->c[i];
    }
    sh_listf(sh, "affine(%s: [%g %g %g; %g %g %g; %g %g %g] + [%g %g %g])\n", what,
             op->f[0], op->f[1], op->f[2], op->f[3], op->f[4], op->f[5],
             op->f[6], op->f[7], op->f[8], op->f[9], op->f[10], op->f[11]);
}

static void ictcp_constants(struct plh_op *op, bool hlg)
{
    op->i0 = hlg;
    if (!hlg) {
        const float m1 = plh_fmtf(PQ_M1), m2 = plh_fmtf(PQ_M2);
        op->f[0] = 1.0f / m2; op->f[1] = plh_fmtf(PQ_C1); op->f[2] = plh_fmtf(PQ_C2);
        op->f[3] = plh_fmtf(PQ_C3); op->f[4] = 1.0f / m1;
        op->f[5] = m1; op->f[6] = op->f[1]; op->f[7] = op->f[2]; op->f[8] = op->f[3];
        op->f[9] = m2;
    } else {
        const float A = plh_fmtf(HLG_A), B = plh_fmtf(HLG_B), C = plh_fmtf(HLG_C);
        op->f[5] = C; op->f[6] = 1.0f / A; op->f[7] = B;
        op->f[8] = A; op->f[9] = B; op->f[10] = C;
    }
}

// Dolby Vision reshaping (reference colorspace.c:106-271): the curves of the three components,
// packed the way the reference packs its uniforms (coefficients per piece, MMR rows per piece and
// order) into one table the op reads (colormap.hiph: op_dovi_reshape)
void pl_shader_dovi_reshape(pl_shader sh, const struct pl_dovi_metadata *data)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0) || !data)
        return;
    pl_gpu gpu = SH_GPU(sh);
    if (!gpu) {
        SH_FAIL(sh, "pl_shader_dovi_reshape requires a GPU (the curves are a device table)");
        return;
    }

    struct plh_dovi_comp table[3];
    memset(table, 0, sizeof(table));
    bool any = false;
    for (int c = 0; c < 3; c++) {
        const struct pl_reshape_data *comp = &data->comp[c];
        struct plh_dovi_comp *out = &table[c];
        if (!comp->num_pivots)
            continue;
        if (comp->num_pivots < 2 || comp->num_pivots > 9) {
            SH_FAIL(sh, "pl_shader_dovi_reshape: component %d has %d pivots (2 .. 9 allowed)",
                    c, comp->num_pivots);
            return;
        }
        any = true;
        out->num_pivots = comp->num_pivots;
        out->mmr_single = true;
        out->min_order = 3;
        out->max_order = 1;
        int mmr_idx = 0;
        for (int i = 0; i < comp->num_pivots - 1; i++) {
            if (comp->method[i] == 0) {
                out->has_poly = true;
                out->coeffs[i][3] = 0.0f;  // order 0 = polynomial
                for (int k = 0; k < 3; k++)
                    out->coeffs[i][k] = comp->poly_coeffs[i][k];
            } else if (comp->method[i] == 1 && comp->mmr_order[i] >= 1 && comp->mmr_order[i] <= 3) {
                out->min_order = PL_MIN(out->min_order, comp->mmr_order[i]);
                out->max_order = PL_MAX(out->max_order, comp->mmr_order[i]);
                out->mmr_single = !out->has_mmr;
                out->has_mmr = true;
                out->coeffs[i][3] = comp->mmr_order[i];
                out->coeffs[i][0] = comp->mmr_constant[i];
                out->coeffs[i][1] = mmr_idx;
                for (int j = 0; j < comp->mmr_order[i]; j++) {
                    const float *w = comp->mmr_coeffs[i][j];
                    float *rows = &out->mmr[mmr_idx][0];
                    rows[0] = w[0]; rows[1] = w[1]; rows[2] = w[2]; rows[3] = 0.0f;
                    rows[4] = w[3]; rows[5] = w[4]; rows[6] = w[5]; rows[7] = w[6];
                    mmr_idx += 2;
                }
            } else {
                SH_FAIL(sh, "pl_shader_dovi_reshape: invalid method / order in component %d", c);
                return;
            }
        }
        // the inner pivots, then a quasi-infinite sentinel (:188-196)
        for (int i = 0; i < 8; i++)
            out->pivots[i] = i < comp->num_pivots - 2 ? comp->pivots[i + 1] : 1e9f;
        out->lo = comp->pivots[0];
        out->hi = comp->pivots[comp->num_pivots - 1];
    }

    sh_describef(sh, "reshaping");
    if (!any) {
        // (the reference still clamps nothing and leaves the colour alone)
        return;
    }
    // (one table per shader: a second reshaping in the same shader has no meaning)
    const void *dev = sh->scratch ? NULL : plh_gpu_upload_scratch(gpu, table, sizeof(table));
    struct plh_op *op = dev ? sh_op(sh, PLH_OP_DOVI_RESHAPE) : NULL;
    if (dev) {
        sh->scratch = dev;      // (given back by sh_release)
        sh->scratch_gpu = gpu;
    }
    if (!op) {
        SH_FAIL(sh, "pl_shader_dovi_reshape: could not upload the reshaping curves (too many shaders "
                "with reshaping curves recorded and not yet dispatched?)");
        return;
    }
    op->ptr = dev;
    sh_listf(sh, "dovi_reshape(pivots %d %d %d)\n", table[0].num_pivots, table[1].num_pivots,
             table[2].num_pivots);
}

void pl_shader_decode_color(pl_shader sh, struct pl_color_repr *repr,
                            const struct pl_color_adjustment *params)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    if (repr->sys == PL_COLOR_SYSTEM_DOLBYVISION && !repr->dovi) {
        SH_FAIL(sh, "PL_COLOR_SYSTEM_DOLBYVISION requires pl_color_repr.dovi");
        return;
    }

    sh_describef(sh, "color decoding");
    const struct pl_dovi_metadata *dovi = repr->dovi;
    if (repr->sys == PL_COLOR_SYSTEM_DOLBYVISION) {
        // (:285-292) the integer scale first, then the curves work on normalised values
        const float scale = pl_color_repr_normalize(repr);
        struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
        if (!op)
            return;
        op->f[0] = op->f[1] = op->f[2] = scale;
        op->f[3] = 1.0f;
        sh_listf(sh, "scale(%g, rgb only)\n", scale);
        pl_shader_dovi_reshape(sh, dovi);
    }
    const enum pl_color_system orig_sys = repr->sys;
    const pl_transform3x3 tr = pl_color_repr_decode(repr, params);
    if (memcmp(&tr, &pl_transform3x3_identity, sizeof(tr)))
        op_affine(sh, &tr, "decode");

    switch (orig_sys) {
    case PL_COLOR_SYSTEM_BT_2020_C:
        simple_op(sh, PLH_OP_BT2020C_DEC, "bt2020c_decode()");
        break;
    case PL_COLOR_SYSTEM_BT_2100_PQ:
    case PL_COLOR_SYSTEM_BT_2100_HLG: {
        struct plh_op *op = simple_op(sh, PLH_OP_ICTCP_DEC, "ictcp_decode()");
        if (op)
            ictcp_constants(op, orig_sys == PL_COLOR_SYSTEM_BT_2100_HLG);
        break;
    }
    case PL_COLOR_SYSTEM_DOLBYVISION: {
        // (:392-420) Dolby Vision decodes to BT.2020-referred HPE LMS: the inverse of that matrix,
        // times the stream's own, between a PQ EOTF and a PQ OETF
        pl_matrix3x3 lms2rgb = {{
            { 3.06441879, -2.16597676,  0.10155818},
            {-0.65612108,  1.78554118, -0.12943749},
            { 0.01736321, -0.04725154,  1.03004253},
        }};
        pl_matrix3x3_mul(&lms2rgb, &dovi->linear);
        struct plh_op *op = simple_op(sh, PLH_OP_DOVI_LMS, "dovi_lms()");
        if (op) {
            for (int i = 0; i < 3; i++) {
                for (int j = 0; j < 3; j++)
                    op->f[3 * i + j] = lms2rgb.m[i][j];
            }
            const float m1 = plh_fmtf(PQ_M1), m2 = plh_fmtf(PQ_M2);
            op->f[9] = 1.0f / m2; op->f[10] = plh_fmtf(PQ_C1); op->f[11] = plh_fmtf(PQ_C2);
            op->f[12] = plh_fmtf(PQ_C3); op->f[13] = 1.0f / m1;
            op->f[14] = m1; op->f[15] = m2;
        }
        break;
    }
    default:
        break;
    }

    if (params && params->gamma == 0) {
        struct plh_op *op = simple_op(sh, PLH_OP_GAMMA, "gamma(0)");
        if (op)
            op->f[0] = 0.0f;
    } else if (params && params->gamma != 1) {
        struct plh_op *op = simple_op(sh, PLH_OP_GAMMA, "gamma()");
        if (op)
            op->f[0] = 1 / params->gamma;
    }

    pl_shader_set_alpha(sh, repr, PL_ALPHA_INDEPENDENT);
}

static bool transform_is_identity(const pl_transform3x3 *t)
{
    for (int i = 0; i < 3; i++) {
        if (t->c[i] != 0.0f)
            return false;
        for (int j = 0; j < 3; j++) {
            if (t->mat.m[i][j] != (i == j ? 1.0f : 0.0f))
                return false;
        }
    }
    return true;
}

void pl_shader_encode_color(pl_shader sh, const struct pl_color_repr *repr)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    if (repr->sys == PL_COLOR_SYSTEM_DOLBYVISION) {
        SH_FAIL(sh, "Cannot un-apply dolbyvision yet (no inverse reshaping)!");
        return;
    }

    sh_describef(sh, "color encoding");
    if (repr->alpha == PL_ALPHA_PREMULTIPLIED)
        simple_op(sh, PLH_OP_PREMULTIPLY, "premultiply()");

    switch (repr->sys) {
    case PL_COLOR_SYSTEM_BT_2020_C:
        simple_op(sh, PLH_OP_BT2020C_ENC, "bt2020c_encode()");
        break;
    case PL_COLOR_SYSTEM_BT_2100_PQ:
    case PL_COLOR_SYSTEM_BT_2100_HLG: {
        struct plh_op *op = simple_op(sh, PLH_OP_ICTCP_ENC, "ictcp_encode()");
        if (op)
            ictcp_constants(op, repr->sys == PL_COLOR_SYSTEM_BT_2100_HLG);
        break;
    }
    default:
        break;
    }

    // code values = M^-1 (colour - c): recorded unless it is the identity. (Full-range RGB at
    // equal sample and colour depth decodes with an exactly-unit matrix; applying it would not
    // change a bit, so nothing is recorded for it.)
    struct pl_color_repr copy = *repr;
    pl_transform3x3 tr = pl_color_repr_decode(&copy, NULL);
    if (!transform_is_identity(&tr)) {
        pl_transform3x3_invert(&tr);
        op_affine(sh, &tr, "encode");
    }
}

/* ------------------------------------------------------------------------ */
/* transfer functions                                                        */

static void nominal_norm(const struct pl_color_space *csp, float *min, float *max)
{
    pl_color_space_nominal_luma_ex(pl_nominal_luma_params(
        .color = csp, .metadata = PL_HDR_METADATA_HDR10, .scaling = PL_HDR_NORM,
        .out_min = min, .out_max = max,
    ));
}

static void luma_coeffs(const struct pl_color_space *csp, float out[3])
{
    const pl_matrix3x3 rgb2xyz = pl_get_rgb2xyz_matrix(pl_raw_primaries_get(csp->primaries));
    out[0] = rgb2xyz.m[1][0];
    out[1] = rgb2xyz.m[1][1];
    out[2] = rgb2xyz.m[1][2];
}

static float gamma_of(enum pl_color_transfer trc)
{
    switch (trc) {
    case PL_COLOR_TRC_GAMMA18: return 1.8f;
    case PL_COLOR_TRC_GAMMA20: return 2.0f;
    case PL_COLOR_TRC_GAMMA24: return 2.4f;
    case PL_COLOR_TRC_GAMMA26: return 2.6f;
    case PL_COLOR_TRC_GAMMA28: return 2.8f;
    default:                   return 2.2f; // UNKNOWN, GAMMA22
    }
}

// Fills op->i0/i1/f[] for a linearize stage (shared with peak detection)
void plh_fill_linearize(struct plh_op *op, const struct pl_color_space *csp)
{
    const enum pl_color_transfer trc = csp->transfer;
    float csp_min, csp_max;
    nominal_norm(csp, &csp_min, &csp_max);

    float *f = op->f;
    op->i0 = trc;
    op->i1 = trc != PL_COLOR_TRC_SCRGB ? PLH_TRC_CLAMP0 : 0;
    bool scale_out = true;

    switch (trc) {
    case PL_COLOR_TRC_BT_1886:
        plh_bt1886_params(csp_min, csp_max, &f[2], &f[3]);
        scale_out = false;
        break;
    case PL_COLOR_TRC_UNKNOWN:
    case PL_COLOR_TRC_GAMMA18: case PL_COLOR_TRC_GAMMA20: case PL_COLOR_TRC_GAMMA22:
    case PL_COLOR_TRC_GAMMA24: case PL_COLOR_TRC_GAMMA26: case PL_COLOR_TRC_GAMMA28:
        f[2] = gamma_of(trc);
        break;
    case PL_COLOR_TRC_PQ: {
        const float m2 = plh_fmtf(PQ_M2), m1 = plh_fmtf(PQ_M1);
        f[2] = 1.0f / m2;
        f[3] = plh_fmtf(PQ_C1); f[4] = plh_fmtf(PQ_C2); f[5] = plh_fmtf(PQ_C3);
        f[6] = 1.0f / m1;
        f[7] = plh_fmtf(10000.0 / PL_COLOR_SDR_WHITE);
        scale_out = false;
        break;
    }
    case PL_COLOR_TRC_HLG: {
        float y, b;
        plh_hlg_params(csp_min, csp_max, &y, &b);
        f[2] = 1 - b; f[3] = b;
        f[4] = plh_fmtf(HLG_C); f[5] = 1.0f / plh_fmtf(HLG_A); f[6] = plh_fmtf(HLG_B);
        f[7] = csp_max;
        luma_coeffs(csp, &f[8]);
        f[11] = y - 1;
        scale_out = false;
        break;
    }
    case PL_COLOR_TRC_V_LOG:
        f[2] = plh_fmtf(VLOG_D); f[3] = 1.0f / plh_fmtf(VLOG_C); f[4] = plh_fmtf(VLOG_B);
        scale_out = false;
        break;
    case PL_COLOR_TRC_S_LOG1:
        f[2] = plh_fmtf(SLOG_C); f[3] = 1.0f / plh_fmtf(SLOG_A); f[4] = plh_fmtf(SLOG_B);
        scale_out = false;
        break;
    case PL_COLOR_TRC_S_LOG2:
        f[2] = plh_fmtf(SLOG_Q); f[3] = 1.0f / plh_fmtf(SLOG_P); f[4] = plh_fmtf(SLOG_C);
        f[5] = 1.0f / plh_fmtf(SLOG_A); f[6] = plh_fmtf(SLOG_B);
        f[7] = 1.0f / plh_fmtf(SLOG_K2); f[8] = plh_fmtf(SLOG_Q);
        scale_out = false;
        break;
    case PL_COLOR_TRC_SCRGB:
        f[2] = plh_fmtf(PL_COLOR_SCRGB_WHITE / PL_COLOR_SDR_WHITE);
        scale_out = false;
        break;
    default: // SRGB, PRO_PHOTO, ST428: literal constants in the kernel
        break;
    }

    if (scale_out && (csp_max != 1 || csp_min != 0)) {
        op->i1 |= PLH_TRC_RESCALE;
        f[0] = csp_max - csp_min;
        f[1] = csp_min;
    }
}

// pl_shader_cone_distort (colorspace.c:2040-2064): three recorded ops, the matrix is one AFFINE
void pl_shader_cone_distort(pl_shader sh, struct pl_color_space csp,
                            const struct pl_cone_params *params)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    if (!params || !params->cones)
        return;

    pl_color_space_infer(&csp);
    pl_shader_linearize(sh, &csp);
    const pl_transform3x3 tr = {
        .mat = pl_get_cone_matrix(params, pl_raw_primaries_get(csp.primaries)),
    };
    op_affine(sh, &tr, "cone distortion");
    pl_shader_delinearize(sh, &csp);
}

// pl_shader_extract_features (colorspace.c:1383-1404)
void pl_shader_extract_features(pl_shader sh, struct pl_color_space csp)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    sh_describef(sh, "feature extraction");
    pl_shader_linearize(sh, &csp);
    struct plh_op *op = sh_op(sh, PLH_OP_FEATURES);
    if (!op)
        return;
    // "vec3 lms = %f * mat3 * color.rgb": scalar * matrix first (left to right), in fp32
    const pl_matrix3x3 rgb2lms = pl_ipt_rgb2lms(pl_raw_primaries_get(csp.primaries));
    const float k = plh_fmtf(PL_COLOR_SDR_WHITE / 10000);
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            op->f[3 * i + j] = k * rgb2lms.m[i][j];
    }
    op->f[9] = plh_fmtf(PQ_M1); op->f[10] = plh_fmtf(PQ_C1); op->f[11] = plh_fmtf(PQ_C2);
    op->f[12] = plh_fmtf(PQ_C3); op->f[13] = plh_fmtf(PQ_M2);
    sh_listf(sh, "extract_features()\n");
}

void pl_shader_linearize(pl_shader sh, const struct pl_color_space *csp)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    if (csp->transfer == PL_COLOR_TRC_LINEAR)
        return;

    struct plh_op *op = sh_op(sh, PLH_OP_LINEARIZE);
    if (!op)
        return;
    plh_fill_linearize(op, csp);
    sh_listf(sh, "linearize(%s)\n", pl_color_transfer_name(csp->transfer));
}

void pl_shader_delinearize(pl_shader sh, const struct pl_color_space *csp)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    const enum pl_color_transfer trc = csp->transfer;
    if (trc == PL_COLOR_TRC_LINEAR)
        return;

    float csp_min, csp_max;
    nominal_norm(csp, &csp_min, &csp_max);

    struct plh_op *op = sh_op(sh, PLH_OP_DELINEARIZE);
    if (!op)
        return;
    float *f = op->f;
    op->i0 = trc;
    op->i1 = trc != PL_COLOR_TRC_SCRGB ? PLH_TRC_CLAMP0 : 0;
    if (pl_color_space_is_black_scaled(csp) && trc != PL_COLOR_TRC_HLG &&
        (csp_max != 1 || csp_min != 0)) {
        op->i1 |= PLH_TRC_RESCALE;
        f[0] = 1 / (csp_max - csp_min);
        f[1] = -csp_min / (csp_max - csp_min);
    }

    switch (trc) {
    case PL_COLOR_TRC_BT_1886: {
        float a, b;
        plh_bt1886_params(csp_min, csp_max, &a, &b);
        f[2] = 1.0 / a;
        f[3] = b;
        break;
    }
    case PL_COLOR_TRC_UNKNOWN:
    case PL_COLOR_TRC_GAMMA18: case PL_COLOR_TRC_GAMMA20: case PL_COLOR_TRC_GAMMA22:
    case PL_COLOR_TRC_GAMMA24: case PL_COLOR_TRC_GAMMA26: case PL_COLOR_TRC_GAMMA28:
        f[2] = 1.0f / gamma_of(trc); // GLSL constant expression 1.0/2.2 etc.
        break;
    case PL_COLOR_TRC_PQ:
        f[2] = 1.0f / plh_fmtf(10000 / PL_COLOR_SDR_WHITE);
        f[3] = plh_fmtf(PQ_M1); f[4] = plh_fmtf(PQ_C1); f[5] = plh_fmtf(PQ_C2);
        f[6] = plh_fmtf(PQ_C3); f[7] = plh_fmtf(PQ_M2);
        break;
    case PL_COLOR_TRC_HLG: {
        float y, b;
        plh_hlg_params(csp_min, csp_max, &y, &b);
        f[2] = 1.0f / csp_max;
        luma_coeffs(csp, &f[3]);
        f[6] = (1 - y) / y;
        f[7] = plh_fmtf(HLG_A); f[8] = plh_fmtf(HLG_B); f[9] = plh_fmtf(HLG_C);
        f[10] = 1 / (1 - b);
        f[11] = -b / (1 - b);
        break;
    }
    case PL_COLOR_TRC_V_LOG:
        f[2] = plh_fmtf(VLOG_C / M_LN10); f[3] = plh_fmtf(VLOG_B); f[4] = plh_fmtf(VLOG_D);
        break;
    case PL_COLOR_TRC_S_LOG1:
        f[2] = plh_fmtf(SLOG_A / M_LN10); f[3] = plh_fmtf(SLOG_B); f[4] = plh_fmtf(SLOG_C);
        break;
    case PL_COLOR_TRC_S_LOG2:
        f[2] = plh_fmtf(SLOG_P); f[3] = plh_fmtf(SLOG_Q); f[4] = plh_fmtf(SLOG_A / M_LN10);
        f[5] = plh_fmtf(SLOG_K2); f[6] = plh_fmtf(SLOG_B); f[7] = plh_fmtf(SLOG_C);
        break;
    case PL_COLOR_TRC_SCRGB:
        f[2] = plh_fmtf(PL_COLOR_SDR_WHITE / PL_COLOR_SCRGB_WHITE);
        break;
    default:
        break;
    }
    sh_listf(sh, "delinearize(%s)\n", pl_color_transfer_name(trc));
}

void pl_shader_sigmoidize(pl_shader sh, const struct pl_sigmoid_params *params)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    params = PL_DEF(params, &pl_sigmoid_default_params);
    const float center = PL_DEF(params->center, pl_sigmoid_default_params.center);
    const float slope  = PL_DEF(params->slope, pl_sigmoid_default_params.slope);

    // the curve must pass through (0,0) and (1,1)
    const float offset = 1.0 / (1 + expf(slope * center));
    const float scale  = 1.0 / (1 + expf(slope * (center - 1))) - offset;

    struct plh_op *op = sh_op(sh, PLH_OP_SIGMOIDIZE);
    if (!op)
        return;
    op->f[0] = center;
    op->f[1] = 1.0 / slope;
    op->f[2] = scale;
    op->f[3] = offset;
    sh_listf(sh, "sigmoidize(center=%g, slope=%g)\n", center, slope);
}

void pl_shader_unsigmoidize(pl_shader sh, const struct pl_sigmoid_params *params)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;
    params = PL_DEF(params, &pl_sigmoid_default_params);
    const float center = PL_DEF(params->center, pl_sigmoid_default_params.center);
    const float slope  = PL_DEF(params->slope, pl_sigmoid_default_params.slope);
    const float offset = 1.0 / (1 + expf(slope * center));
    const float scale  = 1.0 / (1 + expf(slope * (center - 1))) - offset;

    struct plh_op *op = sh_op(sh, PLH_OP_UNSIGMOIDIZE);
    if (!op)
        return;
    op->f[0] = 1.0 / scale;
    op->f[1] = slope;
    op->f[2] = center;
    op->f[3] = offset / scale;
    sh_listf(sh, "unsigmoidize(center=%g, slope=%g)\n", center, slope);
}

/* ------------------------------------------------------------------------ */
/* peak detection                                                            */

enum {
    PEAK_WORDS = 816,   // sizeof(struct peak_buf_data) / 4
    SLICES    = 12,
    PQ_BITS   = 14,
    PQ_MAX    = (1 << PQ_BITS) - 1,
    HIST_BITS = 7,
    HIST_BIAS = 1 << (HIST_BITS - 1),
    HIST_BINS = (1 << HIST_BITS) - HIST_BIAS,
};

#define HIST_PQ(bin) (((bin) + HIST_BIAS) << (PQ_BITS - HIST_BITS))

struct peak_buf_data {
    unsigned frame_wg_count[SLICES];
    unsigned frame_wg_active[SLICES];
    unsigned frame_sum_pq[SLICES];
    unsigned frame_max_pq[SLICES];
    unsigned frame_hist[SLICES][HIST_BINS];
};

struct sh_color_map_obj {
    struct {
        struct pl_tone_map_params params;
        pl_buf lut;
        int lut_size;
    } tone;

    struct {
        struct pl_gamut_map_params params;
        pl_buf lut;
        bool valid;
    } gamut;

    // Scene brightness measurement. The device side lives for as long as the object does:
    // one result buffer (rewritten whole by every measuring pass), its pinned host mirror,
    // the constant block and the kernel's scratch copies -- a frame costs no allocation and
    // exactly one wait (the read of its result), none when the result may lag a frame.
    struct {
        struct pl_peak_detect_params params;
        pl_buf buf;             // 816 words, written by the measuring kernel's last workgroup (k_peak.hip)
        struct peak_buf_data *mirror;   // pinned, device-visible: the kernel's mailbox, + 1 word
        uint32_t ticket;        // the value the pass in flight publishes behind its result
        plh_event written;      // recorded by the dispatch behind the measuring pass
        bool awaiting;          // a measuring pass was recorded and its result not yet taken
        bool launched;          // ... and that pass has been dispatched (`written` is live)
        int on;                 // ... on this stream (gpu_priv.h: 0 main, 1 measurement)
        uint64_t seq;           // ... as that stream's launch number `seq`
        uint64_t tables_seq;    // main-stream stamp of the last upload into consts / scratch
        pl_buf consts;          // constant block of the detect stage
        float consts_now[16];   // what `consts` holds
        pl_buf scratch;         // zeroed accumulators of the measuring kernels (k_peak.hip): left zeroed by each
        float avg_pq, max_pq;   // the filtered state
    } peak;
};

static void sh_color_map_uninit(pl_gpu gpu, void *ptr)
{
    struct sh_color_map_obj *obj = ptr;
    pl_buf_destroy(gpu, &obj->tone.lut);
    pl_buf_destroy(gpu, &obj->gamut.lut);
    pl_buf_destroy(gpu, &obj->peak.buf);
    pl_buf_destroy(gpu, &obj->peak.consts);
    pl_buf_destroy(gpu, &obj->peak.scratch);
    plh_event_destroy(obj->peak.written);
    plh_host_free(obj->peak.mirror);
    memset(obj, 0, sizeof(*obj));
}

static bool peak_params_eq(const struct pl_peak_detect_params *a,
                           const struct pl_peak_detect_params *b)
{
    // allow_delayed does not change the measurement
    return a->smoothing_period == b->smoothing_period &&
           a->scene_threshold_low == b->scene_threshold_low &&
           a->scene_threshold_high == b->scene_threshold_high &&
           a->percentile == b->percentile;
}

static inline float smoothstepf(float edge0, float edge1, float x)
{
    if (edge0 == edge1)
        return x >= edge0;
    x = (x - edge0) / (edge1 - edge0);
    x = PL_CLAMP(x, 0.0f, 1.0f);
    return x * x * (3.0f - 2.0f * x);
}

// ---- host side of the measurement: fetch -> totals -> one sample -> temporal filter ----------
// (what the reference does in update_peak_buf / measure_peak, src/shaders/colorspace.c:
// 1003-1150; the arithmetic of every step is the reference's, float for float)

// The kernel spreads its atomics over SLICES partial results; everything below works on sums.
struct peak_totals {
    uint64_t groups, lit_groups;    // workgroups that ran / that saw a non-black pixel
    uint64_t sum_pq;                // sum over lit groups of their mean PQ code
    unsigned max_pq;
    uint64_t hist[HIST_BINS], hist_n;
};

static void peak_totals_of(const struct peak_buf_data *raw, struct peak_totals *t)
{
    memset(t, 0, sizeof(*t));
    for (int k = 0; k < SLICES; k++) {
        t->groups     += raw->frame_wg_count[k];
        t->lit_groups += raw->frame_wg_active[k];
        t->sum_pq     += raw->frame_sum_pq[k];
        t->max_pq      = PL_MAX(t->max_pq, raw->frame_max_pq[k]);
        for (int b = 0; b < HIST_BINS; b++)
            t->hist[b] += raw->frame_hist[k][b];
    }
    for (int b = 0; b < HIST_BINS; b++)
        t->hist_n += t->hist[b];
}

// PQ level below which `percentile` percent of the histogrammed pixels lie; the frame maximum
// where the histogram cannot tell (none collected, 0 / 100 %, or the rank is the last pixel).
static float peak_percentile(const struct peak_totals *t, float percentile)
{
    const float top = (float) t->max_pq / PQ_MAX;
    if (percentile <= 0 || percentile >= 100 || !t->hist_n)
        return top;
    const uint64_t rank = ceilf(percentile / 100.0f * t->hist_n);
    if (rank >= t->hist_n)
        return top;

    uint64_t below = 0;     // pixels in the bins before `b`
    int b = 0;
    while (b < HIST_BINS - 1 && below + t->hist[b] < rank)
        below += t->hist[b++];
    // pixels assumed evenly spread inside the bin, between the last pixel of the bin before
    // and the first pixel of the bin after; the topmost occupied bin ends at the maximum
    const uint64_t above = below + t->hist[b] + 1;
    const float lo = (float) HIST_PQ(b) / PQ_MAX;
    const float hi = above > t->hist_n ? top : (float) HIST_PQ(b + 1) / PQ_MAX;
    const float where = (float) (rank - below) / (above - below);
    return MIXF(lo, hi, where);
}

struct peak_sample { float avg_pq, max_pq, lit; };

static struct peak_sample peak_sample_of(const struct peak_totals *t, float percentile)
{
    if (!t->lit_groups)    // solid black frame
        return (struct peak_sample) { PL_COLOR_HDR_BLACK, PL_COLOR_HDR_BLACK, 0.0f };
    return (struct peak_sample) {
        .avg_pq = (float) t->sum_pq / (t->lit_groups * PQ_MAX),
        .max_pq = peak_percentile(t, percentile),
        .lit    = (float) t->lit_groups / t->groups,
    };
}

// first-order low-pass with a scene-change bypass, on avg and max alike
static void peak_filter(float *avg_pq, float *max_pq, struct peak_sample in,
                        const struct pl_peak_detect_params *params)
{
    if (!*avg_pq) {
        *avg_pq = in.avg_pq;    // first sample: adopt
        *max_pq = in.max_pq;
    } else {
        // a change below one PQ code is measurement jitter
        const float lsb = 1.0f / PQ_MAX;
        if (fabsf(in.avg_pq - *avg_pq) < lsb)
            in.avg_pq = *avg_pq;
        if (fabsf(in.max_pq - *max_pq) < lsb)
            in.max_pq = *max_pq;
    }

    const float gain = params->smoothing_period ? 1.0f - expf(-1.0f / params->smoothing_period)
                                                : 1.0f;
    *avg_pq += gain * (in.avg_pq - *avg_pq);
    *max_pq += gain * (in.max_pq - *max_pq);

    if (params->scene_threshold_low > 0 && params->scene_threshold_high > 0) {
        // thresholds are in units of 1 % PQ; the jump is weighted by how much of the frame is lit
        const float unit = 1e-2f;
        const float jump = in.lit * fabsf(in.avg_pq - *avg_pq);
        const float snap = smoothstepf(params->scene_threshold_low * unit,
                                       params->scene_threshold_high * unit, jump);
        *avg_pq = MIXF(*avg_pq, in.avg_pq, snap);
        *max_pq = MIXF(*max_pq, in.max_pq, snap);
    }
}

// (profiling hook, tools/r05_22.py: nanoseconds the host has spent waiting for measurements since
// the last call -- what separates "the host is the bottleneck" from "the host waits for the GPU")
static long plh_dbg_wait_ns;
PL_API long plh_test_peak_wait_ns(void);
long plh_test_peak_wait_ns(void) { const long v = plh_dbg_wait_ns; plh_dbg_wait_ns = 0; return v; }

// Take the outstanding measurement, if there is one and it may be taken now.
// `must`: the caller is about to reuse the buffer -- wait for the pass rather than give up.
static void peak_collect(pl_gpu gpu, struct sh_color_map_obj *obj, bool must)
{
    const struct pl_peak_detect_params *params = &obj->peak.params;
    if (!obj->peak.awaiting)
        return;
    if (!obj->peak.launched) {
        // recorded, never dispatched: either abandoned (`must`), or the caller asks for the
        // result from inside the very shader that measures
        if (must) {
            obj->peak.awaiting = false;
        } else if (!params->allow_delayed) {
            pl_msg(gpu->log, PL_LOG_WARN, "Peak detection usage error: attempted detecting "
                   "peak and using detected peak in the same shader program, but "
                   "`params->allow_delayed` is false! Ignoring, but expect incorrect output.");
        }
        return;
    }
    if (!must && params->allow_delayed && plh_event_query(obj->peak.written) == 0)
        return;     // still rendering: this frame goes with the previous result

    obj->peak.awaiting = obj->peak.launched = false;
    plh_stream stream = plh_gpu_stream_n(gpu, obj->peak.on);
    bool have = false;
    if (!plh_gpu_has_peak_exchange(gpu)) {
        // the fold kernel wrote the result into the pinned mirror and then its ticket: poll
        // that word (a few microseconds after the kernel retires, against ~20 for a stream
        // wait plus a copy). Bounded: a lost kernel must not hang the caller.
        volatile const uint32_t *seen = (volatile const uint32_t *) obj->peak.mirror + PEAK_WORDS;
        struct timespec w0, w1;
        clock_gettime(CLOCK_MONOTONIC, &w0);
        for (long spin = 0; spin < 200000000L && !have; spin++) {
            have = __atomic_load_n(seen, __ATOMIC_ACQUIRE) == obj->peak.ticket;
            if (!have && (spin & 1023) == 1023 && plh_event_query(obj->peak.written) != 0)
                break;      // the stream is past the pass (or failed): look once more below
        }
        if (!have)
            have = __atomic_load_n(seen, __ATOMIC_ACQUIRE) == obj->peak.ticket;
        clock_gettime(CLOCK_MONOTONIC, &w1);
        plh_dbg_wait_ns += (w1.tv_sec - w0.tv_sec) * 1000000000L + (w1.tv_nsec - w0.tv_nsec);
    } else {
        // ranks rendering one scene fold their measurements together first (hip.h); the
        // exchange works on the device buffer, which is then copied back
        plh_gpu_peak_exchange(gpu, pl_hip_buf_ptr(obj->peak.buf), sizeof(struct peak_buf_data));
    }
    if (!have && (plh_copy2d_d2h(stream, obj->peak.mirror, sizeof(*obj->peak.mirror),
                                 pl_hip_buf_ptr(obj->peak.buf), sizeof(*obj->peak.mirror),
                                 sizeof(*obj->peak.mirror), 1) || plh_stream_sync(stream))) {
        pl_msg(gpu->log, PL_LOG_ERR, "Failed reading peak detection buffer!");
        return;
    }

    plh_gpu_reached(gpu, obj->peak.on, obj->peak.seq);
    struct peak_totals totals;
    peak_totals_of(obj->peak.mirror, &totals);
    if (!totals.groups)
        return;     // an empty launch measured nothing
    peak_filter(&obj->peak.avg_pq, &obj->peak.max_pq,
                peak_sample_of(&totals, params->percentile), params);
}

// dispatch.c, behind the launch of a pass that carries a measurement
plh_event plh_peak_written_event(pl_shader_obj state)
{
    struct sh_color_map_obj *obj = state ? state->priv : NULL;
    return obj ? obj->peak.written : NULL;
}

void plh_peak_pass_launched(pl_gpu gpu, pl_shader_obj state, int on, uint64_t seq, bool recorded)
{
    struct sh_color_map_obj *obj = state->priv;
    obj->peak.on = on;
    obj->peak.seq = seq;
    if (recorded || !plh_event_record(obj->peak.written, plh_gpu_stream_n(gpu, on)))
        obj->peak.launched = true;
}

bool pl_shader_detect_peak(pl_shader sh, struct pl_color_space csp, pl_shader_obj *state,
                           const struct pl_peak_detect_params *params)
{
    params = PL_DEF(params, &pl_peak_detect_default_params);
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return false;

    pl_gpu gpu = SH_GPU(sh);
    if (!gpu) {
        pl_msg(sh->log, PL_LOG_ERR, "HDR peak detection requires a GPU");
        return false;
    }
    if (sh->pass.s.type == PLH_SAMPLE_POLAR || sh->pass.s.type == PLH_SAMPLE_ORTHO ||
        sh->pass.s.type == PLH_SAMPLE_DEBAND || sh->pass.s.type == PLH_SAMPLE_DEINTERLACE ||
        sh->pass.s.type == PLH_SAMPLE_DISTORT) {
        // those samplers own the workgroup shape (or, the distortion, a kernel the measuring
        // kernels do not carry: they would measure and store (0, 0, 0, 1) without a word);
        // measure in a separate pass
        pl_msg(sh->log, PL_LOG_ERR, "pl_shader_detect_peak cannot be merged into a "
               "polar/ortho/deband/deinterlace/distort pass on the HIP backend (materialise it first)");
        return false;
    }

    const bool use_histogram = params->percentile > 0 && params->percentile < 100;
    size_t shmem_req = 3 * sizeof(uint32_t);
    if (use_histogram)
        shmem_req += sizeof(uint32_t[HIST_BINS]);
    if (!sh_try_compute(sh, 16, 16, true, shmem_req))
        return false;

    struct sh_color_map_obj *obj = SH_OBJ(sh, state, PL_SHADER_OBJ_COLOR_MAP,
                                          struct sh_color_map_obj, sh_color_map_uninit);
    if (!obj)
        return false;

    if (peak_params_eq(&obj->peak.params, params)) {
        peak_collect(gpu, obj, true);   // the previous frame's result, before its buffer is reused
    } else {
        pl_reset_detected_peak(*state);
    }

    if (!obj->peak.buf) {
        obj->peak.buf = pl_buf_create(gpu, pl_buf_params(
            .size = sizeof(struct peak_buf_data), .host_readable = true, .storable = true));
        obj->peak.mirror = plh_host_alloc_coherent(sizeof(struct peak_buf_data) + sizeof(uint32_t));
        if (obj->peak.mirror)
            memset(obj->peak.mirror, 0, sizeof(struct peak_buf_data) + sizeof(uint32_t));
        if (!obj->peak.buf || !obj->peak.mirror || plh_event_create(&obj->peak.written)) {
            SH_FAIL(sh, "Failed creating peak detection SSBO!");
            return false;
        }
    }
    obj->peak.params = *params;
    obj->peak.awaiting = true;
    obj->peak.launched = false;

    struct plh_op *op = sh_op(sh, PLH_OP_PEAK_DETECT);
    if (!op)
        return false;

    // the measurement linearizes a *copy* of the colour
    pl_color_space_infer(&csp);
    if (csp.transfer != PL_COLOR_TRC_LINEAR) {
        plh_fill_linearize(op, &csp);
    } else {
        op->i0 = PL_COLOR_TRC_LINEAR;
    }
    op->i2 = use_histogram;

    float consts[16] = {0};
    luma_coeffs(&csp, &consts[0]);
    consts[3] = PL_COLOR_SDR_WHITE / 10000.0;
    consts[4] = PQ_M1; consts[5] = PQ_C1; consts[6] = PQ_C2; consts[7] = PQ_C3; consts[8] = PQ_M2;
    consts[9] = fmaxf(params->black_cutoff, 0.0f) * 1e-2f;
    if (!obj->peak.consts) {
        obj->peak.consts = pl_buf_create(gpu, pl_buf_params(.size = sizeof(consts),
                                                            .storable = true));
        if (!obj->peak.consts)
            return false;
    }
    if (memcmp(consts, obj->peak.consts_now, sizeof(consts))) {
        plh_buf_write(gpu, obj->peak.consts, 0, consts, sizeof(consts));
        memcpy(obj->peak.consts_now, consts, sizeof(consts));
        obj->peak.tables_seq = plh_gpu_stamp(gpu, 0);
    }
    op->ptr2 = pl_hip_buf_ptr(obj->peak.consts);
    if (!obj->peak.scratch) {
        // (k_peak_tiles uses the first 4.3 K words: padded accumulators + tickets; k_peak_fast /
        // k_pass_peak all of the copies, + the block counter of k_peak_fold behind them)
        const size_t size = (size_t) PLH_PEAK_COPIES * sizeof(struct peak_buf_data) + 64;
        void *zeros = calloc(1, size);
        obj->peak.scratch = zeros ? pl_buf_create(gpu, pl_buf_params(
            .size = size, .storable = true, .initial_data = zeros)) : NULL;
        free(zeros);
        if (!obj->peak.scratch)
            return false;
        obj->peak.tables_seq = plh_gpu_stamp(gpu, 0);
    }
    sh->aux_after = obj->peak.tables_seq;
    sh->pass.peak_buf = pl_hip_buf_ptr(obj->peak.buf);
    sh->pass.peak_scratch = pl_hip_buf_ptr(obj->peak.scratch);
    obj->peak.ticket = obj->peak.ticket + 1 ? obj->peak.ticket + 1 : 1;    // never 0
    sh->pass.peak_mailbox = obj->peak.mirror;
    sh->pass.peak_ticket = obj->peak.ticket;
    sh->detect_peak = true;
    sh->peak_state = *state;    // (held below: outlives the shader's dispatch)
    sh_hold(sh, *state);

    sh_describef(sh, "peak detection");
    sh_listf(sh, "detect_peak(trc=%s, histogram=%d, cutoff=%g)\n",
             pl_color_transfer_name(csp.transfer), (int) use_histogram, consts[9]);
    return true;
}

bool pl_get_detected_hdr_metadata(const pl_shader_obj state, struct pl_hdr_metadata *out)
{
    if (!state || state->type != PL_SHADER_OBJ_COLOR_MAP)
        return false;

    struct sh_color_map_obj *obj = state->priv;
    peak_collect(state->gpu, obj, false);
    if (!obj->peak.avg_pq)
        return false;

    out->max_pq_y = obj->peak.max_pq;
    out->avg_pq_y = obj->peak.avg_pq;
    return true;
}

void pl_reset_detected_peak(pl_shader_obj state)
{
    if (!state || state->type != PL_SHADER_OBJ_COLOR_MAP)
        return;

    struct sh_color_map_obj *obj = state->priv;
    // the filter state and the request go, the device objects stay
    memset(&obj->peak.params, 0, sizeof(obj->peak.params));
    obj->peak.awaiting = obj->peak.launched = false;
    obj->peak.avg_pq = obj->peak.max_pq = 0.0f;
}

void *pl_hip_peak_buffer(const pl_shader_obj state, size_t *out_size)
{
    if (!state || state->type != PL_SHADER_OBJ_COLOR_MAP)
        return NULL;
    struct sh_color_map_obj *obj = state->priv;
    if (!obj->peak.buf || !obj->peak.awaiting)
        return NULL;
    if (out_size)
        *out_size = sizeof(struct peak_buf_data);
    return pl_hip_buf_ptr(obj->peak.buf);
}

/* ------------------------------------------------------------------------ */
/* colour mapping                                                            */

static void mat_to_f(float *f, const pl_matrix3x3 *m)
{
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            f[3 * i + j] = m->m[i][j];
    }
}

// 48x32x256 IPT LUT -> rgba16 unorm with a +32767 chroma bias (fill_gamut_lut,
// colorspace.c:1589-1610)
static void fill_gamut_lut(void *data, void *priv)
{
    const struct pl_gamut_map_params *gamut = priv;
    const size_t n = (size_t) gamut->lut_size_I * gamut->lut_size_C * gamut->lut_size_h;
    float *tmp = malloc(n * 3 * sizeof(float));
    if (!tmp)
        abort();
    pl_gamut_map_generate(tmp, gamut);
    const float *in = tmp;
    uint16_t *out = data;
    for (size_t i = 0; i < n; i++) {
        out[0] = roundf(in[0] * UINT16_MAX);
        out[1] = roundf(in[1] * UINT16_MAX + (UINT16_MAX >> 1));
        out[2] = roundf(in[2] * UINT16_MAX + (UINT16_MAX >> 1));
        out[3] = 0;
        in += 3;
        out += 4;
    }
    free(tmp);
}

// everything the table depends on except its size, which the cache protocol checks
// (gamut_map_signature, colorspace.c:991-1001)
static uint64_t gamut_lut_signature(const struct pl_gamut_map_params *par)
{
    uint64_t sig = PLH_CACHE_KEY_GAMUT_LUT;
    plh_hash_merge(&sig, plh_mem_hash(par->function->name, strlen(par->function->name)));
    plh_hash_merge(&sig, plh_mem_hash(&par->input_gamut, sizeof(par->input_gamut)));
    plh_hash_merge(&sig, plh_mem_hash(&par->output_gamut, sizeof(par->output_gamut)));
    plh_hash_merge(&sig, plh_mem_hash(&par->min_luma, sizeof(par->min_luma)));
    plh_hash_merge(&sig, plh_mem_hash(&par->max_luma, sizeof(par->max_luma)));
    plh_hash_merge(&sig, plh_mem_hash(&par->constants, sizeof(par->constants)));
    return sig;
}

static pl_buf make_gamut_lut(pl_shader sh, const struct pl_gamut_map_params *gamut)
{
    pl_gpu gpu = SH_GPU(sh);
    const size_t n = (size_t) gamut->lut_size_I * gamut->lut_size_C * gamut->lut_size_h;
    const size_t size = n * 4 * sizeof(uint16_t);
    uint16_t *packed = malloc(size);
    if (!packed)
        return NULL;
    const bool hit = plh_cache_memoize(plh_gpu_cache(gpu), gamut_lut_signature(gamut), packed,
                                       size, fill_gamut_lut, (void *) gamut);
    pl_msg(sh->log, PL_LOG_DEBUG, hit ? "Re-using cached gamut LUT (%s)" : "Generated gamut LUT (%s)",
           gamut->function->name);
    pl_buf buf = pl_buf_create(gpu, pl_buf_params(.size = size, .storable = true,
                                                  .initial_data = packed));
    free(packed);
    return buf;
}

void pl_shader_color_map_ex(pl_shader sh, const struct pl_color_map_params *params,
                            const struct pl_color_map_args *args)
{
    if (!sh_require(sh, PL_SHADER_SIG_COLOR, 0, 0))
        return;

    pl_gpu gpu = SH_GPU(sh);
    struct pl_color_space measured = args->src;
    struct sh_color_map_obj *obj = NULL;
    if (args->state) {
        pl_get_detected_hdr_metadata(*args->state, &measured.hdr);
        obj = SH_OBJ(sh, args->state, PL_SHADER_OBJ_COLOR_MAP, struct sh_color_map_obj,
                     sh_color_map_uninit);
        if (!obj)
            return;
    }

    // every decision of the request, as one value (colormap_plan.c)
    struct plh_colormap_plan plan;
    plh_colormap_resolve(&plan, params, &measured, &args->dst, args->state != NULL);
    params = PL_DEF(params, &pl_color_map_default_params);
    const struct pl_color_space src = plan.src, dst = plan.dst;
    const struct pl_tone_map_params tone = plan.tone;
    const struct pl_gamut_map_params gamut = plan.gamut;
    const bool can_fast = plan.closed_form, need_tone_map = plan.need_tone;
    const bool need_gamut_map = plan.need_gamut;
    if (plan.identity) {
        if (args->prelinearized)
            pl_shader_delinearize(sh, &dst);
        return;
    }

    if (!args->prelinearized)
        pl_shader_linearize(sh, &src);

    pl_matrix3x3 rgb2lms = pl_ipt_rgb2lms(pl_raw_primaries_get(src.primaries));
    pl_matrix3x3 lms2rgb = pl_ipt_lms2rgb(pl_raw_primaries_get(dst.primaries));
    if (plan.fold_saturation) {
        // LMS -> source gamut -> (same coordinates read as) target gamut -> LMS, then out
        const pl_matrix3x3 lms2src = pl_ipt_lms2rgb(&gamut.input_gamut);
        const pl_matrix3x3 dst2lms = pl_ipt_rgb2lms(&gamut.output_gamut);
        sh_describef(sh, "gamut map (saturation)");
        pl_matrix3x3_mul(&lms2rgb, &dst2lms);
        pl_matrix3x3_mul(&lms2rgb, &lms2src);
    }

    if (!need_tone_map && !need_gamut_map) {
        // nothing non-linear left: RGB -> LMS -> RGB is a single 3x3
        if (src.primaries != dst.primaries) {
            sh_describef(sh, "colorspace conversion");
            pl_matrix3x3_mul(&lms2rgb, &rgb2lms);
            const pl_transform3x3 tr = { .mat = lms2rgb };
            op_affine(sh, &tr, "primaries");
        }
        goto done;
    }

    // ---- full path through IPT ------------------------------------------------------
    struct plh_op *op = sh_op(sh, PLH_OP_RGB2IPT);
    if (!op)
        return;
    mat_to_f(op->f, &rgb2lms);
    op->f[9]  = plh_fmtf(PL_COLOR_SDR_WHITE / 10000);
    op->f[10] = plh_fmtf(PQ_M1); op->f[11] = plh_fmtf(PQ_C1); op->f[12] = plh_fmtf(PQ_C2);
    op->f[13] = plh_fmtf(PQ_C3); op->f[14] = plh_fmtf(PQ_M2);
    sh_listf(sh, "rgb2ipt()\n");

    if (need_tone_map) {
        const struct pl_tone_map_function *fun = tone.function;
        sh_describef(sh, "%s tone map (%.0f -> %.0f)", fun->name,
                     pl_hdr_rescale(PL_HDR_PQ, PL_HDR_NITS, tone.input_max),
                     pl_hdr_rescale(PL_HDR_PQ, PL_HDR_NITS, tone.output_max));

        op = sh_op(sh, PLH_OP_TONE_MAP);
        if (!op)
            return;
        if (fun == &pl_tone_map_clip && can_fast) {
            op->i0 = 0;
            op->f[0] = tone.input_min;
            op->f[1] = tone.input_max;
        } else if (fun == &pl_tone_map_linear && can_fast) {
            const float gain = tone.constants.exposure;
            const float scale = tone.input_max - tone.input_min;
            op->i0 = 1;
            op->f[0] = gain / scale;
            op->f[1] = -gain / scale * tone.input_min;
            op->f[2] = tone.output_max - tone.output_min;
            op->f[3] = tone.output_min;
        } else {
            if (!obj) {
                SH_FAIL(sh, "Tone-mapping LUT requires a state object");
                return;
            }
            const bool update = !obj->tone.lut || obj->tone.lut_size != (int) tone.lut_size ||
                                !pl_tone_map_params_equal(&tone, &obj->tone.params);
            if (update) {
                float *lut = malloc(tone.lut_size * sizeof(float));
                if (!lut)
                    return;
                pl_tone_map_generate(lut, &tone);
                if (!obj->tone.lut || obj->tone.lut_size != (int) tone.lut_size) {
                    pl_buf_destroy(gpu, &obj->tone.lut);
                    obj->tone.lut = pl_buf_create(gpu, pl_buf_params(
                        .size = tone.lut_size * sizeof(float), .storable = true,
                        .initial_data = lut));
                } else {
                    plh_buf_write(gpu, obj->tone.lut, 0, lut, tone.lut_size * sizeof(float));
                }
                free(lut);
                obj->tone.lut_size = tone.lut_size;
            }
            obj->tone.params = tone;
            if (!obj->tone.lut) {
                SH_FAIL(sh, "Failed generating tone-mapping LUT!");
                return;
            }
            const float lut_range = tone.input_max - tone.input_min;
            op->i0 = 2;
            op->i1 = tone.lut_size;
            op->f[0] = 1.0f / lut_range;
            op->f[1] = -tone.input_min / lut_range;
            // (size - 1, size - 2 as floats: the kernels have no scalar int -> float conversion)
            op->f[8] = (float) (tone.lut_size - 1);
            op->f[9] = (float) (tone.lut_size - 2);
            op->ptr = pl_hip_buf_ptr(obj->tone.lut);
        }
        // contrast recovery (:1879-1921): detail = highres - bicubic(feature map)
        const bool need_recovery = tone.input_max >= tone.output_max;
        if (need_recovery && params->contrast_recovery && args->feature_map) {
            pl_tex fm = args->feature_map;
            struct plh_view v;
            plh_tex_view(fm, &v);
            if (v.fmt != PLH_FMT_R16F || v.w > 0xffff || v.h > 0xffff) {
                SH_FAIL(sh, "Contrast recovery needs an r16hf feature map (got '%s')",
                        fm->params.format->name);
                return;
            }
            op->ptr2 = v.ptr;
            op->i2 = v.w | (v.h << 16);
            memcpy(&op->f[4], &v.pitch, sizeof(float));
            op->f[5] = params->contrast_recovery;
            op->f[6] = tone.output_min;
            op->f[7] = tone.output_max;
            sh_listf(sh, "contrast_recovery(%g, feature map %dx%d)\n", params->contrast_recovery,
                     v.w, v.h);
        }
        sh_listf(sh, "tone_map(%s, mode=%d, in=[%g,%g] avg=%g, out=[%g,%g])\n", fun->name,
                 op->i0, tone.input_min, tone.input_max, tone.input_avg, tone.output_min,
                 tone.output_max);
    }

    if (need_gamut_map) {
        if (!obj) {
            SH_FAIL(sh, "Gamut-mapping LUT requires a state object");
            return;
        }
        sh_describef(sh, "gamut map (%s)", gamut.function->name);
        if (!obj->gamut.valid || !obj->gamut.lut ||
            !pl_gamut_map_params_equal(&gamut, &obj->gamut.params)) {
            pl_buf_destroy(gpu, &obj->gamut.lut);
            obj->gamut.lut = make_gamut_lut(sh, &gamut);
            obj->gamut.params = gamut;
            obj->gamut.valid = !!obj->gamut.lut;
        }
        if (!obj->gamut.lut) {
            SH_FAIL(sh, "Failed generating gamut-mapping LUT!");
            return;
        }

        op = sh_op(sh, PLH_OP_GAMUT_LUT);
        if (!op)
            return;
        const float lut_range = gamut.max_luma - gamut.min_luma;
        op->i0 = gamut.lut_size_I;
        op->i1 = gamut.lut_size_C;
        op->i2 = gamut.lut_size_h;
        op->f[0] = 1.0f / lut_range;
        op->f[1] = -gamut.min_luma / lut_range;
        op->f[2] = plh_fmtf(0.5f / M_PI);
        op->f[3] = params->lut3d_tricubic ? 1.0f : 0.0f;
        // size - 1 and size - 2 of the three axes as floats (cmfast.hiph: gamut_lookup)
        op->f[4] = gamut.lut_size_I - 1; op->f[5] = gamut.lut_size_C - 1; op->f[6] = gamut.lut_size_h - 1;
        op->f[7] = gamut.lut_size_I - 2; op->f[8] = gamut.lut_size_C - 2; op->f[9] = gamut.lut_size_h - 2;
        op->f[10] = gamut.lut_size_I; op->f[11] = gamut.lut_size_C;
        op->ptr = pl_hip_buf_ptr(obj->gamut.lut);
        sh_listf(sh, "gamut_lut(%s, %dx%dx%d%s)\n", gamut.function->name, op->i0, op->i1, op->i2,
                 params->lut3d_tricubic ? ", tricubic" : "");
    }

    op = sh_op(sh, PLH_OP_IPT2RGB);
    if (!op)
        return;
    mat_to_f(op->f, &lms2rgb);
    op->f[9]  = 1.0f / plh_fmtf(PQ_M2);
    op->f[10] = plh_fmtf(PQ_C1); op->f[11] = plh_fmtf(PQ_C2); op->f[12] = plh_fmtf(PQ_C3);
    op->f[13] = 1.0f / plh_fmtf(PQ_M1);
    op->f[14] = plh_fmtf(10000 / PL_COLOR_SDR_WHITE);
    sh_listf(sh, "ipt2rgb()\n");
    if (args->state)
        sh_hold(sh, *args->state);

done:
    pl_shader_delinearize(sh, &dst);
}

void pl_shader_color_map(pl_shader sh, const struct pl_color_map_params *params,
                         struct pl_color_space src, struct pl_color_space dst,
                         pl_shader_obj *state, bool prelinearized)
{
    pl_shader_color_map_ex(sh, params, pl_color_map_args(
        .src = src, .dst = dst, .prelinearized = prelinearized, .state = state,
    ));
}
