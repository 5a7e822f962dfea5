/*
 * libplacebo-hip — the renderer's planner (see render_plan.h).
 *
 * Every function is a pure decision over descriptions. The behaviour each one has to reproduce
 * is the reference's (file:line cited at the function); float expressions that feed sampling
 * coordinates keep the reference's operation order, because those values end up in kernels
 * whose output is compared bit for bit.
 */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "render_plan.h"
#include "host_common.h"

/* ======================================================================================== */
/* geometry: reference src/renderer.c:3068-3159 (fix_refs_and_rects)                          */

static inline bool rect_unset(const pl_rect2df *rc)
{
    return (!rc->x0 && !rc->x1) || (!rc->y0 && !rc->y1);
}

// One axis of the fit. (s0, s1) = image crop, (d0, d1) = target crop, both possibly flipped;
// `limit` = target extent. On return the source interval is ascending and shifted / shrunk to
// what the rounded, clipped target interval shows; the target interval is integral and carries
// the net flip.
static void fit_axis(float *s0, float *s1, float *d0, float *d1, float limit)
{
    const bool flip = (*s0 > *s1) != (*d0 > *d1);
    const float slo = PL_MIN(*s0, *s1), shi = PL_MAX(*s0, *s1);
    const float dlo = PL_MIN(*d0, *d1), dhi = PL_MAX(*d0, *d1);

    const float rlo = roundf(PL_CLAMP(dlo, 0.0, limit));
    const float rhi = roundf(PL_CLAMP(dhi, 0.0, limit));

    // source texels per target pixel, from the unrounded rects
    const float step = (shi - slo) / (dhi - dlo);
    *s0 = slo + (rlo - dlo) * step;
    *s1 = slo + (rhi - dlo) * step;
    *d0 = flip ? rhi : rlo;
    *d1 = flip ? rlo : rhi;
}

struct rp_geometry rp_fit_rects(pl_rect2df src, int iw, int ih, pl_rotation image_rot,
                                pl_rect2df dst, int tw, int th, pl_rotation target_rot)
{
    struct rp_geometry geo = {0};
    if (rect_unset(&dst)) {
        dst.x1 = tw;
        dst.y1 = th;
    }
    if (rect_unset(&src)) {
        src.x1 = iw;
        src.y1 = ih;
    }

    // The image is processed in its own orientation: counter-rotate the target rect into it;
    // the output stage transposes / flips the stores (:3113-3117)
    geo.rotation = pl_rotation_normalize(image_rot - target_rot);
    pl_rect2df_rotate(&dst, -geo.rotation);
    const bool quarter = geo.rotation % PL_ROTATION_180 == PL_ROTATION_90;
    const float lim_x = quarter ? th : tw, lim_y = quarter ? tw : th;

    fit_axis(&src.x0, &src.x1, &dst.x0, &dst.x1, lim_x);
    fit_axis(&src.y0, &src.y1, &dst.y0, &dst.y1, lim_y);

    geo.src = src;
    geo.dstf = dst;
    geo.dst = (pl_rect2d) { dst.x0, dst.y0, dst.x1, dst.y1 };
    return geo;
}

// Without an image (:3084-3101): only the target rect is rounded, in the target's own rotation
struct rp_geometry rp_fit_target(pl_rect2df dst, int tw, int th, pl_rotation target_rot)
{
    struct rp_geometry geo = {0};
    if (rect_unset(&dst)) {
        dst.x1 = tw;
        dst.y1 = th;
    }
    geo.rotation = pl_rotation_normalize(-target_rot);
    pl_rect2df_rotate(&dst, -geo.rotation);
    const bool quarter = geo.rotation % PL_ROTATION_180 == PL_ROTATION_90;
    const float lim_x = quarter ? th : tw, lim_y = quarter ? tw : th;
    dst = (pl_rect2df) {
        .x0 = roundf(PL_CLAMP(dst.x0, 0.0, lim_x)), .y0 = roundf(PL_CLAMP(dst.y0, 0.0, lim_y)),
        .x1 = roundf(PL_CLAMP(dst.x1, 0.0, lim_x)), .y1 = roundf(PL_CLAMP(dst.y1, 0.0, lim_y)),
    };
    geo.dstf = dst;
    geo.dst = (pl_rect2d) { dst.x0, dst.y0, dst.x1, dst.y1 };
    return geo;
}

/* ======================================================================================== */
/* planes and frames: :287-335 (detect_plane_type), :3048-3066, :3161-3293 (fix_frame)        */

enum rp_plane_role rp_plane_role(const struct pl_plane *plane, const struct pl_color_repr *repr)
{
    if (!pl_color_system_is_ycbcr_like(repr->sys)) {
        const bool lone_alpha = plane->components == 1 &&
                                plane->component_mapping[0] == PL_CHANNEL_A;
        if (lone_alpha)
            return RP_PLANE_ALPHA;
        return repr->sys == PL_COLOR_SYSTEM_XYZ ? RP_PLANE_XYZ : RP_PLANE_RGB;
    }

    // YCbCr-like: the most significant channel a plane carries names it
    enum rp_plane_role role = RP_PLANE_UNUSED;
    for (int c = 0; c < plane->components; c++) {
        enum rp_plane_role r;
        switch (plane->component_mapping[c]) {
        case PL_CHANNEL_Y:  r = RP_PLANE_LUMA; break;
        case PL_CHANNEL_CB:
        case PL_CHANNEL_CR: r = RP_PLANE_CHROMA; break;
        case PL_CHANNEL_A:  r = RP_PLANE_ALPHA; break;
        default:            r = RP_PLANE_UNUSED; break;
        }
        role = PL_MAX(role, r);
    }
    return role;
}

int rp_reference_plane(const struct pl_frame *frame)
{
    for (int i = 0; i < frame->num_planes; i++) {
        const enum rp_plane_role r = rp_plane_role(&frame->planes[i], &frame->repr);
        if (r == RP_PLANE_LUMA || r == RP_PLANE_RGB || r == RP_PLANE_XYZ)
            return i;
    }
    return 0;
}

void rp_complete_frame(struct pl_frame *frame)
{
    pl_tex tex = frame->planes[rp_reference_plane(frame)].texture;

    // XYZ decodes to linear DCI-P3 (pl_color_repr_decode does the conversion)
    if (frame->repr.sys == PL_COLOR_SYSTEM_XYZ) {
        frame->color.primaries = PL_COLOR_PRIM_DCI_P3;
        frame->color.transfer = PL_COLOR_TRC_ST428;
    }
    if (tex && frame->color.primaries == PL_COLOR_PRIM_UNKNOWN)
        frame->color.primaries = pl_color_primaries_guess(tex->params.w, tex->params.h);

    // no plane carries alpha -> the frame has none, whatever the repr says
    bool alpha = false;
    for (int p = 0; p < frame->num_planes && !alpha; p++) {
        const struct pl_plane *pl = &frame->planes[p];
        for (int c = 0; c < pl->components; c++)
            alpha |= pl->component_mapping[c] == PL_CHANNEL_A;
    }
    if (!alpha)
        frame->repr.alpha = PL_ALPHA_NONE;

    // an integer format knows how many bits are sampled
    struct pl_bit_encoding *b = &frame->repr.bits;
    if (b->sample_depth || !tex || tex->params.format->type != PL_FMT_UNORM)
        return;
    const int sampled = tex->params.format->component_depth[0];
    int used = b->color_depth ? b->color_depth : sampled;
    if (used > sampled)
        used = sampled;
    b->sample_depth = sampled;
    b->color_depth = used;
    b->bit_shift += sampled - used;
}

void rp_complete_frames(struct pl_frame *image, struct pl_frame *target)
{
    rp_complete_frame(image);
    pl_color_space_infer_map(&image->color, &target->color);
    rp_complete_frame(target);  // after the inference: a guess must not override it
    if (image->repr.alpha == PL_ALPHA_UNKNOWN)
        image->repr.alpha = PL_ALPHA_INDEPENDENT;
    if (target->repr.alpha == PL_ALPHA_UNKNOWN)
        target->repr.alpha = PL_ALPHA_PREMULTIPLIED;
}

const char *rp_frame_problem(const struct pl_frame *f, bool is_target)
{
    if (f->num_planes < 1 || f->num_planes > PL_MAX_PLANES)
        return "invalid number of planes";
    for (int i = 0; i < f->num_planes; i++) {
        const struct pl_plane *p = &f->planes[i];
        if (!p->texture)
            return "a plane has no texture";
        if (p->components < 1 || p->components > 4)
            return "a plane has an invalid number of components";
        if (is_target && !p->texture->params.storable)
            return "target textures must be storable (every pass is a compute pass)";
        if (!is_target && !p->texture->params.sampleable)
            return "image textures must be sampleable";
    }
    const struct pl_plane *ref = &f->planes[rp_reference_plane(f)];
    if (ref->shift_x || ref->shift_y)
        return "the reference plane must have no shift";
    return NULL;
}

// :1447-1468 (guess_frame_lut_type)
enum pl_lut_type rp_frame_lut_type(const struct pl_frame *frame, bool reversed)
{
    const struct pl_custom_lut *lut = frame->lut;
    if (!lut)
        return PL_LUT_UNKNOWN;
    if (frame->lut_type)
        return frame->lut_type;

    const enum pl_color_system from = reversed ? lut->repr_out.sys : lut->repr_in.sys;
    const enum pl_color_system to   = reversed ? lut->repr_in.sys : lut->repr_out.sys;
    if (from == PL_COLOR_SYSTEM_RGB && to == PL_COLOR_SYSTEM_RGB)
        return PL_LUT_NORMALIZED;
    if (from == frame->repr.sys && to == PL_COLOR_SYSTEM_RGB)
        return PL_LUT_CONVERSION;
    return PL_LUT_NATIVE;   // cannot tell: the least surprising placement
}

/* ======================================================================================== */
/* image layout: :1553-1830 (pass_read_image, up to the plane shaders)                        */

// planes are subsampled by whole factors; a fractional size means the plane was rounded up
static float subsampling(int plane_size, int ref_size)
{
    const float r = (float) plane_size / ref_size;
    return r >= 1 ? roundf(r) : 1.0 / roundf(1.0 / r);
}

void rp_layout_image(const struct pl_frame *image, struct rp_image_layout *lay)
{
    memset(lay, 0, sizeof(*lay));
    lay->ref = rp_reference_plane(image);
    pl_tex ref_tex = image->planes[lay->ref].texture;

    // what an unwritten channel holds: black luma / grey chroma in the native encoding
    const int bits = image->repr.bits.sample_depth;
    const float code_scale = bits ? (1llu << bits) / ((1llu << bits) - 1.0f) : 1.0f;
    if (pl_color_levels_guess(&image->repr) == PL_COLOR_LEVELS_LIMITED)
        lay->neutral_luma = 16 / 256.0f * code_scale;
    lay->neutral_chroma = pl_color_system_is_ycbcr_like(image->repr.sys) ? 0.5f * code_scale
                                                                         : lay->neutral_luma;

    for (int i = 0; i < image->num_planes; i++) {
        struct rp_plane_layout *pl = &lay->planes[i];
        pl->plane = image->planes[i];
        pl->role = rp_plane_role(&pl->plane, &image->repr);

        // an alpha mode of NONE hides the alpha channel wherever it lives
        if (image->repr.alpha == PL_ALPHA_NONE) {
            if (pl->role == RP_PLANE_ALPHA) {
                pl->role = RP_PLANE_UNUSED;
                continue;
            }
            for (int c = 0; c < pl->plane.components; c++) {
                if (pl->plane.component_mapping[c] == PL_CHANNEL_A)
                    pl->plane.component_mapping[c] = PL_CHANNEL_NONE;
            }
        }
        if (!pl->role)
            continue;

        const float kx = subsampling(pl->plane.texture->params.w, ref_tex->params.w),
                    ky = subsampling(pl->plane.texture->params.h, ref_tex->params.h);
        const float sx = pl->plane.shift_x, sy = pl->plane.shift_y;
        pl->rect.x0 = (image->crop.x0 - sx) * kx;
        pl->rect.y0 = (image->crop.y0 - sy) * ky;
        pl->rect.x1 = (image->crop.x1 - sx) * kx;
        pl->rect.y1 = (image->crop.y1 - sy) * ky;
        pl->logical_w = ref_tex->params.w * kx;
        pl->logical_h = ref_tex->params.h * ky;

        int n = 0;
        for (int c = 0; c < pl->plane.components; c++) {
            const int ch = pl->plane.component_mapping[c];
            if (ch == PL_CHANNEL_Y)
                pl->neutral[n++] = lay->neutral_luma;
            else if (ch == PL_CHANNEL_U || ch == PL_CHANNEL_V)
                pl->neutral[n++] = lay->neutral_chroma;
        }
    }

    // The image proper lives on whole texels of the reference plane: drop the sub-texel offset
    // (towards zero) and the sub-texel size mismatch, and remember both (:1810-1828)
    const pl_rect2df exact = lay->planes[lay->ref].rect;
    lay->grid.x0 = truncf(exact.x0);
    lay->grid.y0 = truncf(exact.y0);
    lay->grid.x1 = lay->grid.x0 + roundf(pl_rect_w(exact));
    lay->grid.y1 = lay->grid.y0 + roundf(pl_rect_h(exact));
    lay->off_x = exact.x0 - lay->grid.x0;
    lay->off_y = exact.y0 - lay->grid.y0;
    lay->stretch_x = pl_rect_w(lay->grid) / pl_rect_w(exact);
    lay->stretch_y = pl_rect_h(lay->grid) / pl_rect_h(exact);
}

struct pl_sample_src rp_plane_request(const struct rp_image_layout *lay, int i)
{
    const struct rp_plane_layout *pl = &lay->planes[i];
    const pl_rect2df exact = lay->planes[lay->ref].rect;

    // this plane's texels per reference texel, and where the snapped grid starts in it
    const float kx = pl_rect_w(pl->rect) / pl_rect_w(exact),
                ky = pl_rect_h(pl->rect) / pl_rect_h(exact);
    const float x0 = pl->rect.x0 - kx * lay->off_x,
                y0 = pl->rect.y0 - ky * lay->off_y;

    struct pl_sample_src req = {
        .components   = pl->plane.components,
        .address_mode = pl->plane.address_mode,
        .new_w        = pl_rect_w(lay->grid),
        .new_h        = pl_rect_h(lay->grid),
        .rect = {
            .x0 = x0,
            .y0 = y0,
            .x1 = x0 + lay->stretch_x * pl_rect_w(pl->rect),
            .y1 = y0 + lay->stretch_y * pl_rect_h(pl->rect),
        },
    };
    if (pl->plane.flipped) {
        req.rect.y0 = pl->logical_h - req.rect.y0;
        req.rect.y1 = pl->logical_h - req.rect.y1;
    }
    return req;
}

bool rp_plane_request_is_identity(const struct pl_sample_src *req)
{
    return req->rect.x0 == 0 && req->rect.y0 == 0 &&
           req->rect.x1 == req->new_w && req->rect.y1 == req->new_h;
}

/* ======================================================================================== */
/* scalers: :597-682 (sample_src_info)                                                        */

static enum rp_direction axis_direction(float out, float in)
{
    const float ratio = out / fabsf(in);
    if (ratio < 1.0 - 1e-6)
        return RP_DIR_DOWN;
    if (ratio > 1.0 + 1e-6)
        return RP_DIR_UP;
    return RP_DIR_NONE;
}

struct rp_scaler rp_pick_scaler(const struct rp_caps *caps, const struct pl_render_params *params,
                                enum rp_usage usage, const struct pl_sample_src *req,
                                pl_fmt src_format)
{
    struct rp_scaler sc = { .kind = RP_SCALER_BUILTIN };
    sc.axis[0] = axis_direction(req->new_w, pl_rect_w(req->rect));
    sc.axis[1] = axis_direction(req->new_h, pl_rect_h(req->rect));
    if (params->correct_subpixel_offsets) {
        // a pure sub-texel shift is resampled with the upscaler
        if (!sc.axis[0] && fabsf(req->rect.x0) > 1e-6f)
            sc.axis[0] = RP_DIR_UP;
        if (!sc.axis[1] && fabsf(req->rect.y0) > 1e-6f)
            sc.axis[1] = RP_DIR_UP;
    }
    sc.dir = PL_MAX(sc.axis[0], sc.axis[1]);    // (DOWN > UP > NONE)

    if (sc.dir == RP_DIR_NONE) {
        sc.kind = RP_SCALER_NEAREST;
        return sc;
    }

    const bool plane = usage == RP_USE_PLANE;
    if (sc.dir == RP_DIR_UP) {
        sc.filter = plane && params->plane_upscaler ? params->plane_upscaler : params->upscaler;
    } else if (usage == RP_USE_LOWPASS) {
        sc.filter = &pl_filter_bicubic;     // fixed (:630)
    } else {
        sc.filter = plane && params->plane_downscaler ? params->plane_downscaler
                                                      : params->downscaler;
    }

    if (caps->sampling_broken || !sc.filter)
        return sc;  // built-in
    if (sc.filter->kernel == &pl_filter_function_oversample) {
        sc.kind = RP_SCALER_OVERSAMPLE;
        return sc;
    }

    sc.kind = RP_SCALER_FILTER;
    // Filters with a closed form that a bilinear fetch can evaluate are replaced by it -- when
    // upscaling, or when the caller renounces anti-aliasing
    pl_fmt fmt = src_format ? src_format : caps->fbo[4];
    const bool linear = fmt && (fmt->caps & PL_FMT_CAP_LINEAR);
    const bool eligible = (sc.dir == RP_DIR_UP || params->skip_anti_aliasing) &&
                          !params->disable_builtin_scalers;
    if (eligible) {
        static const struct { const struct pl_filter_config *cfg; enum rp_scaler_kind kind; }
        shortcuts[] = {
            { &pl_filter_bicubic,  RP_SCALER_BICUBIC },
            { &pl_filter_hermite,  RP_SCALER_HERMITE },
            { &pl_filter_gaussian, RP_SCALER_GAUSSIAN },
            { &pl_filter_bilinear, RP_SCALER_BUILTIN },
        };
        for (size_t i = 0; linear && i < PL_ARRAY_SIZE(shortcuts); i++) {
            if (pl_filter_config_eq(sc.filter, shortcuts[i].cfg))
                sc.kind = shortcuts[i].kind;
        }
        if (pl_filter_config_eq(sc.filter, &pl_filter_nearest))
            sc.kind = linear ? RP_SCALER_NEAREST : RP_SCALER_BUILTIN;
    }

    // a real filter needs an intermediate image
    if (sc.kind == RP_SCALER_FILTER && !caps->fbo[4])
        sc.kind = RP_SCALER_BUILTIN;
    return sc;
}

/* ======================================================================================== */
/* main scaling stage: :1964-2087 (pass_scale_main)                                           */

struct rp_scale_stage rp_plan_scale(const struct rp_caps *caps, const struct pl_render_params *params,
                                    const struct pl_sample_src *req, pl_fmt src_format,
                                    const struct pl_color_space *img_color, int comps,
                                    bool fixed_size_input)
{
    struct rp_scale_stage st = { .out_w = req->new_w, .out_h = req->new_h };
    st.scaler = rp_pick_scaler(caps, params, RP_USE_MAIN, req, src_format);
    const enum rp_direction dir = st.scaler.dir;

    // An upscale shrinks the measurement (fewer pixels before it), anything else grows or
    // keeps it: measure on the smaller side
    st.peak_before = dir == RP_DIR_UP;

    if (dir == RP_DIR_NONE && !fixed_size_input) {
        st.skip = true;
        return st;
    }
    if (st.scaler.kind == RP_SCALER_BUILTIN && !fixed_size_input) {
        st.defer = true;
        return st;
    }

    st.sigmoid = dir == RP_DIR_UP && params->sigmoid_params;
    st.linear = dir == RP_DIR_DOWN;

    pl_fmt fbo = caps->fbo[comps];
    if (params->disable_linear_scaling || fbo->component_depth[0] < 16)
        st.sigmoid = st.linear = false;
    if (pl_color_space_is_hdr(img_color)) {
        st.sigmoid = false;                     // the sigmoid clips to [0, 1]
        if (fbo->type != PL_FMT_FLOAT)
            st.linear = false;                  // linear HDR needs the float range
    }
    st.restore_transfer = !st.linear && !st.sigmoid &&
                          img_color->transfer == PL_COLOR_TRC_LINEAR;
    return st;
}

/* ======================================================================================== */
/* HDR peak: :1183-1250 (hdr_update_peak), conditions only                                    */

const char *rp_peak_skip_reason(const struct rp_caps *caps, const struct pl_render_params *params,
                                const struct pl_color_space *image, const struct pl_color_space *img,
                                const struct pl_color_space *target)
{
    if (!params->peak_detect_params)
        return "not requested";
    if (!pl_color_space_is_hdr(image))
        return "image is not HDR";
    if (caps->peak_broken)
        return "disabled after an earlier failure";
    if (caps->fbo[4] && !(caps->fbo[4]->caps & PL_FMT_CAP_STORABLE))
        return "intermediate format is not storable";

    float ceiling = pl_color_transfer_nominal_peak(image->transfer) * PL_COLOR_SDR_WHITE;
    if (image->transfer == PL_COLOR_TRC_HLG)
        ceiling = img->hdr.max_luma;
    if (ceiling <= target->hdr.max_luma + 1e-6)
        return "the target covers the image's range";
    if (img->hdr.avg_pq_y)
        return "dynamic metadata already present";

    const struct pl_color_map_params *cm = params->color_map_params;
    const enum pl_hdr_metadata_type wanted = cm ? cm->metadata : PL_HDR_METADATA_ANY;
    if (wanted != PL_HDR_METADATA_ANY && wanted != PL_HDR_METADATA_CIE_Y)
        return "the tone mapper is told to use other metadata";
    if (cm && cm->tone_mapping_function == &pl_tone_map_st2094_40 && img->hdr.ootf.num_anchors)
        return "HDR10+ OOTF in use";
    if (params->lut && params->lut_type == PL_LUT_CONVERSION)
        return "a conversion LUT does the tone mapping";
    return NULL;
}

/* ======================================================================================== */
/* contrast recovery: :2089-2154 (get_feature_map), conditions + geometry                     */

bool rp_fuse_into_polar(enum rp_direction dir, bool pending_lite, float antiring, int force)
{
    if (force == 1 || dir == RP_DIR_NONE)
        return false;
    if (antiring > 0)
        return false;   // the anti-ringing variant has no fused form
    if (force == 0)
        return true;
    return dir == RP_DIR_DOWN || pending_lite;
}

bool rp_wants_feature_map(const struct rp_caps *caps, const struct pl_render_params *params,
                          const struct pl_color_space *img, const struct pl_color_space *target,
                          int out_w, int out_h, int *map_w, int *map_h)
{
    const struct pl_color_map_params *cm = params->color_map_params ? params->color_map_params
                                                                    : &pl_color_map_default_params;
    if (!cm->contrast_recovery || cm->contrast_smoothness <= 1)
        return false;
    if (!caps->fbo[4] || !caps->fbo[1])
        return false;
    if (!pl_color_space_is_hdr(img) || img->hdr.max_luma <= target->hdr.max_luma + 1e-6)
        return false;   // nothing gets compressed
    if (caps->sampling_broken || caps->contrast_broken)
        return false;
    if (params->lut && params->lut_type == PL_LUT_CONVERSION)
        return false;
    *map_w = ceilf(out_w / cm->contrast_smoothness);
    *map_h = ceilf(out_h / cm->contrast_smoothness);
    return true;
}

/* ======================================================================================== */
/* output: :2586-2964 (pass_output_target)                                                    */

void rp_plan_output(const struct pl_render_params *params, const struct pl_frame *target,
                    const struct rp_geometry *geo, int img_comps, enum pl_alpha_mode img_alpha,
                    struct rp_output_stage *out)
{
    memset(out, 0, sizeof(*out));
    // (a blended output has an alpha whatever the target stores: the blend unit reads it, :2713)
    const bool target_alpha = target->repr.alpha != PL_ALPHA_NONE || params->blend_params;

    // background / border modes, with the pre-v7.346 switches folded in (:2498, :2707)
    out->background = params->background;
    out->border = params->border;
    if (params->blend_against_tiles)
        out->background = PL_CLEAR_TILES;
    else if (params->skip_target_clearing)
        out->background = PL_CLEAR_SKIP;
    if (params->skip_target_clearing)
        out->border = PL_CLEAR_SKIP;
    // a blurred background / border is not implemented here: it shows the background colour
    // (what the reference does when its blur pass is unavailable, :2500, PL_RENDER_ERR_BLUR)
    if (out->background == PL_CLEAR_BLUR)
        out->background = PL_CLEAR_COLOR;
    if (out->border == PL_CLEAR_BLUR)
        out->border = PL_CLEAR_COLOR;
    // fully transparent background on a target with alpha: nothing to blend against
    if (params->background_transparency >= 1.0 && target_alpha)
        out->background = PL_CLEAR_SKIP;

    int comps = img_comps;
    enum pl_alpha_mode alpha = img_alpha;
    if (comps == 4 && (out->background != PL_CLEAR_SKIP || !target_alpha)) {
        out->premultiply = true;
        alpha = PL_ALPHA_PREMULTIPLIED;
        if (out->background == PL_CLEAR_COLOR) {
            out->blend = true;
            if (!params->background_transparency || !target_alpha) {
                out->drop_alpha = true;
                alpha = PL_ALPHA_NONE;
                comps = 3;
            }
        } else if (out->background == PL_CLEAR_TILES) {
            // :2734-2756: blended against the tile pattern, alpha becomes 1
            out->blend = true;
            out->drop_alpha = true;
            alpha = PL_ALPHA_NONE;
            comps = 3;
        }
    }

    out->repr = target->repr;
    out->scale = pl_color_repr_normalize(&out->repr);
    if (alpha == out->repr.alpha || comps < 4) {
        out->repr.alpha = PL_ALPHA_NONE;    // nothing to convert
    } else {
        out->unpremultiply = true;
    }

    out->target_lut = rp_frame_lut_type(target, true);
    out->encode = out->target_lut != PL_LUT_CONVERSION;
    out->delinearize_xyz = out->encode && out->repr.sys == PL_COLOR_SYSTEM_XYZ;

    out->dst = geo->dst;
    out->transposed = geo->rotation % PL_ROTATION_180 == PL_ROTATION_90;
    if (out->transposed) {
        out->dst = (pl_rect2d) { geo->dst.y0, geo->dst.x0, geo->dst.y1, geo->dst.x1 };
    }
    const bool flip_x = out->dst.x1 < out->dst.x0, flip_y = out->dst.y1 < out->dst.y0;

    out->clear_border = pl_frame_is_cropped(target) && out->border != PL_CLEAR_SKIP;

    const int depth = target->repr.bits.color_depth;
    out->dither_depth = depth && (depth < 16 || params->force_dither) ? depth : 0;

    // per plane: the target rect in the plane's own texels (:2835-2864)
    pl_tex ref = target->planes[rp_reference_plane(target)].texture;
    out->num_planes = target->num_planes;
    for (int i = 0; i < target->num_planes; i++) {
        const struct pl_plane *pl = &target->planes[i];
        struct rp_output_plane *op = &out->planes[i];
        op->ratio_x = subsampling(pl->texture->params.w, ref->params.w);
        op->ratio_y = subsampling(pl->texture->params.h, ref->params.h);

        op->exact = (pl_rect2df) {
            (out->dst.x0 - pl->shift_x) * op->ratio_x, (out->dst.y0 - pl->shift_y) * op->ratio_y,
            (out->dst.x1 - pl->shift_x) * op->ratio_x, (out->dst.y1 - pl->shift_y) * op->ratio_y,
        };
        pl_rect2df_normalize(&op->exact);
        op->covered = (pl_rect2d) {
            floorf(op->exact.x0), floorf(op->exact.y0), ceilf(op->exact.x1), ceilf(op->exact.y1),
        };

        // how this plane reads the finished (full-resolution) image
        uint8_t mask = 0;
        for (int c = 0; c < pl->components; c++) {
            if (pl->component_mapping[c] >= 0)
                mask |= 1 << pl->component_mapping[c];
        }
        op->request = (struct pl_sample_src) {
            .new_w = pl_rect_w(op->covered),
            .new_h = pl_rect_h(op->covered),
            .rect = {
                .x0 = (op->covered.x0 - op->exact.x0) / op->ratio_x,
                .x1 = (op->covered.x1 - op->exact.x0) / op->ratio_x,
                .y0 = (op->covered.y0 - op->exact.y0) / op->ratio_y,
                .y1 = (op->covered.y1 - op->exact.y0) / op->ratio_y,
            },
            .component_mask = mask,
        };

        op->store = op->covered;
        if (flip_x) {
            op->store.x0 = op->covered.x1;
            op->store.x1 = op->covered.x0;
        }
        if (flip_y) {
            op->store.y0 = op->covered.y1;
            op->store.y1 = op->covered.y0;
        }
        if (pl->flipped) {
            const int plane_h = op->ratio_y * ref->params.h;
            op->store.y0 = plane_h - op->store.y0;
            op->store.y1 = plane_h - op->store.y1;
        }
    }
}

// :2282-2295, :2884-2900
enum rp_dither rp_pick_dither(const struct rp_caps *caps, const struct pl_render_params *params,
                              int depth, int plane_h)
{
    if (!depth)
        return RP_DITHER_NONE;
    if (params->error_diffusion && !caps->errdiff_broken &&
        pl_error_diffusion_shmem_req(params->error_diffusion, plane_h) <= caps->max_shmem)
        return RP_DITHER_ERROR_DIFFUSION;
    return params->dither_params ? RP_DITHER_ORDERED : RP_DITHER_NONE;
}

/* ======================================================================================== */
/* summary                                                                                   */

static void say(struct rp_summary *s, const char *fmt, ...)
{
    const size_t used = strlen(s->text);
    if (used + 2 >= sizeof(s->text))
        return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(s->text + used, sizeof(s->text) - used, fmt, ap);
    va_end(ap);
}

static const char *scaler_name(const struct rp_scaler *sc)
{
    static const char *const kinds[] = { "builtin", "nearest", "bicubic", "hermite", "gaussian",
                                         "oversample", "filter" };
    if (sc->kind == RP_SCALER_FILTER)
        return sc->filter->polar ? "polar" : "separable";
    return kinds[sc->kind];
}

static const char *dir_name(enum rp_direction d)
{
    return d == RP_DIR_UP ? "up" : d == RP_DIR_DOWN ? "down" : "none";
}

void rp_summarise(const struct rp_caps *caps, const struct pl_frame *pimage,
                  const struct pl_frame *ptarget, const struct pl_render_params *params,
                  struct rp_summary *s)
{
    s->text[0] = '\0';
    struct pl_frame image = *pimage, target = *ptarget;
    const char *bad = rp_frame_problem(&image, false);
    bad = bad ? bad : rp_frame_problem(&target, true);
    if (bad) {
        say(s, "invalid: %s\n", bad);
        return;
    }

    pl_tex iref = image.planes[rp_reference_plane(&image)].texture,
           tref = target.planes[rp_reference_plane(&target)].texture;
    const struct rp_geometry geo = rp_fit_rects(image.crop, iref->params.w, iref->params.h,
                                                image.rotation, target.crop, tref->params.w,
                                                tref->params.h, target.rotation);
    image.crop = geo.src;
    target.crop = geo.dstf;
    rp_complete_frames(&image, &target);
    say(s, "geometry: src %g,%g-%g,%g dst %d,%d-%d,%d rot %d\n", geo.src.x0, geo.src.y0,
        geo.src.x1, geo.src.y1, geo.dst.x0, geo.dst.y0, geo.dst.x1, geo.dst.y1, geo.rotation);

    struct rp_image_layout lay;
    rp_layout_image(&image, &lay);
    const bool deband = params->deband_params && !caps->deband_broken && caps->fbo[4];
    for (int i = 0; i < image.num_planes; i++) {
        if (!lay.planes[i].role)
            continue;
        struct pl_sample_src req = rp_plane_request(&lay, i);
        const bool ident = rp_plane_request_is_identity(&req);
        req.tex = lay.planes[i].plane.texture;
        const struct rp_scaler sc = rp_pick_scaler(caps, params, RP_USE_PLANE, &req,
                                                   req.tex->params.format);
        say(s, "plane %d: role %d%s%s -> %dx%d %s\n", i, lay.planes[i].role,
            i == lay.ref ? " (reference)" : "", deband ? " deband" : "", req.new_w, req.new_h,
            ident && (deband || i == lay.ref) ? "as is" : scaler_name(&sc));
    }

    const int comps = image.repr.alpha == PL_ALPHA_NONE ? 3 : 4;
    const int out_w = abs(pl_rect_w(geo.dst)), out_h = abs(pl_rect_h(geo.dst));
    struct pl_color_space img_color = image.color;
    if (caps->fbo[comps]) {
        const struct pl_sample_src req = {
            .components = comps, .new_w = out_w, .new_h = out_h,
            .rect = { lay.off_x, lay.off_y, lay.off_x + pl_rect_w(lay.planes[lay.ref].rect),
                      lay.off_y + pl_rect_h(lay.planes[lay.ref].rect) },
        };
        // a debanded reference plane has a fixed size: it cannot be resampled in place
        const bool fixed = deband && (req.new_w != pl_rect_w(lay.grid) ||
                                      req.new_h != pl_rect_h(lay.grid));
        const struct rp_scale_stage st = rp_plan_scale(caps, params, &req, NULL, &img_color,
                                                       comps, fixed);
        const char *why = rp_peak_skip_reason(caps, params, &image.color, &img_color, &target.color);
        if (!why && st.peak_before)
            say(s, "peak: measured before scaling\n");
        if (st.skip) {
            say(s, "scale: none\n");
        } else if (st.defer) {
            say(s, "scale: deferred to the output pass (%s)\n", scaler_name(&st.scaler));
        } else {
            say(s, "scale: %s %s%s%s%s -> %dx%d\n", scaler_name(&st.scaler),
                dir_name(st.scaler.dir), st.linear ? " linear" : "", st.sigmoid ? " sigmoid" : "",
                st.scaler.kind == RP_SCALER_FILTER && !st.scaler.filter->polar &&
                st.scaler.axis[0] && st.scaler.axis[1] ? " two-pass" : "", st.out_w, st.out_h);
            if (st.linear || st.sigmoid)
                img_color.transfer = PL_COLOR_TRC_LINEAR;
        }
        if (!why && !st.peak_before)
            say(s, "peak: measured after scaling\n");
        if (why && params->peak_detect_params)
            say(s, "peak: skipped (%s)\n", why);
    } else {
        say(s, "scale: no intermediate format, output pass samples directly\n");
    }

    int mw, mh;
    if (rp_wants_feature_map(caps, params, &img_color, &target.color, out_w, out_h, &mw, &mh))
        say(s, "contrast recovery: feature map %dx%d\n", mw, mh);
    say(s, "colour: %s -> %s%s\n", pl_color_transfer_name(image.color.transfer),
        pl_color_transfer_name(target.color.transfer),
        img_color.transfer == PL_COLOR_TRC_LINEAR && image.repr.alpha != PL_ALPHA_PREMULTIPLIED
            ? " (prelinearized)" : "");

    struct rp_output_stage out;
    rp_plan_output(params, &target, &geo, comps, image.repr.alpha, &out);
    for (int i = 0; i < out.num_planes; i++) {
        const struct rp_output_plane *op = &out.planes[i];
        static const char *const dn[] = { "none", "ordered", "error diffusion" };
        say(s, "output plane %d: store %d,%d-%d,%d dither %s/%d scale 1/%g%s%s\n", i,
            op->store.x0, op->store.y0, op->store.x1, op->store.y1,
            dn[rp_pick_dither(caps, params, out.dither_depth, pl_rect_h(op->covered))],
            out.dither_depth, out.scale, out.blend ? " blend" : "",
            out.transposed ? " transposed" : "");
    }
}
