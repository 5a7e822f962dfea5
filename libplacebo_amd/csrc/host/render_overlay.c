/*
 * libplacebo-hip -- overlays of a frame (subtitles, on-screen display) drawn over a rendered plane.
 *
 * Counterpart of draw_overlays (/root/reference/src/renderer.c:811-1020). The reference builds a
 * vertex buffer of two triangles per part and lets the rasteriser and the blend unit do the rest;
 * here the parts are handed to the dispatch as rectangles of the plane with an affine map into the
 * overlay texture (dispatch.c: plh_dispatch_overlay, k_overlay.hip), and the recorded shader is the
 * colour half only: decode the overlay's representation, map its colour space to the target's,
 * encode for the target, swizzle into the plane.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "renderer_priv.h"
#include "shaders_priv.h"

// src_to_dst of :833-850: image coordinates (texels of the reference plane) -> target pixels
static bool image_to_target(const struct frame_job *job, pl_transform2x2 *out)
{
    const struct pl_frame *image = &job->image;
    const pl_rect2d dst = job->geo.dst;
    if (!pl_rect_w(image->crop) || !pl_rect_h(image->crop))
        return false;
    const float rx = pl_rect_w(dst) / pl_rect_w(image->crop),
                ry = pl_rect_h(dst) / pl_rect_h(image->crop);
    *out = (pl_transform2x2) {
        .mat.m = {{ rx, 0 }, { 0, ry }},
        .c = { dst.x0 - rx * image->crop.x0, dst.y0 - ry * image->crop.y0 },
    };
    if (job->geo.rotation % PL_ROTATION_180 == PL_ROTATION_90) {
        const float t = out->c[0];
        out->c[0] = out->c[1];
        out->c[1] = t;
        out->mat = (pl_matrix2x2) {{{ 0, ry }, { rx, 0 }}};
    }
    return true;
}

// One part as the kernel takes it: where its corners land on the plane (`tf`), and the texture
// coordinate as an affine function of the pixel centre. `tf` only ever scales, flips and swaps the
// axes, so the quad stays a rectangle of the plane.
static struct plh_overlay_part place_part(const struct pl_overlay_part *part,
                                          const pl_transform2x2 *tf, pl_tex tex)
{
    float p00[2] = { part->dst.x0, part->dst.y0 }, p10[2] = { part->dst.x1, part->dst.y0 },
          p01[2] = { part->dst.x0, part->dst.y1 }, p11[2] = { part->dst.x1, part->dst.y1 };
    pl_transform2x2_apply(tf, p00);
    pl_transform2x2_apply(tf, p10);
    pl_transform2x2_apply(tf, p01);
    pl_transform2x2_apply(tf, p11);

    const float du = (part->src.x1 - part->src.x0) / tex->params.w,
                dv = (part->src.y1 - part->src.y0) / tex->params.h;
    struct plh_overlay_part q = {
        .x0 = PL_MIN(p00[0], p11[0]), .x1 = PL_MAX(p00[0], p11[0]),
        .y0 = PL_MIN(p00[1], p11[1]), .y1 = PL_MAX(p00[1], p11[1]),
        .ox = p00[0], .oy = p00[1],
        .u0 = part->src.x0 / tex->params.w, .v0 = part->src.y0 / tex->params.h,
        .color = { part->color[0], part->color[1], part->color[2], part->color[3] },
    };
    const bool swapped = tf->mat.m[0][0] == 0.0f && tf->mat.m[1][1] == 0.0f;
    const float along_u = swapped ? p10[1] - p00[1] : p10[0] - p00[0];  // the edge src.x runs along
    const float along_v = swapped ? p01[0] - p00[0] : p01[1] - p00[1];
    if (along_u == 0.0f || along_v == 0.0f) {
        q.x1 = q.x0;    // a part without area covers no pixel centre
        return q;
    }
    if (swapped) {
        q.uy = du / along_u;
        q.vx = dv / along_v;
    } else {
        q.ux = du / along_u;
        q.vy = dv / along_v;
    }
    return q;
}

static void append_forced_swizzle(pl_shader sh, int comps, const int mapping[4])
{
    static const int identity[4] = {0, 1, 2, 3};
    if (!mapping)
        mapping = identity;
    uint32_t from = 0;
    for (int c = 0; c < 4; c++) {
        const int m = c < comps ? mapping[c] : -1;
        from |= (uint32_t) (m < 0 ? 0xff : m) << (8 * c);
    }
    struct plh_op *op = sh_op(sh, PLH_OP_SWIZZLE);
    if (!op)
        return;
    op->i0 = from;
    op->i1 = comps;
    op->i2 = 1;     // color.a = orig.a whatever the plane holds: the blend unit needs it
    sh_listf(sh, "swizzle(comps=%d, map=0x%08x, keep alpha)\n", comps, (unsigned) from);
}

void plh_draw_overlays(struct frame_job *job, pl_tex fbo, int comps, const int comp_map[4],
                       const struct pl_overlay *overlays, int num, bool have_image,
                       struct pl_color_space color, struct pl_color_repr repr,
                       const pl_transform2x2 *output_shift)
{
    pl_renderer rr = job->rr;
    if (num <= 0 || (rr->errors & PL_RENDER_ERR_OVERLAY))
        return;

    const struct pl_frame *image = have_image ? &job->image : NULL;
    const struct pl_frame *target = &job->target;
    pl_transform2x2 src_to_dst = pl_transform2x2_identity;
    if (image && !image_to_target(job, &src_to_dst))
        image = NULL;

    pl_rect2df dst_crop = target->crop;
    pl_rect2df_rotate(&dst_crop, -job->geo.rotation);
    pl_rect2df_normalize(&dst_crop);

    for (int n = 0; n < num; n++) {
        struct pl_overlay ol = overlays[n];
        if (!ol.num_parts)
            continue;
        if (!ol.tex || !ol.parts) {
            RR_LOG(rr, PL_LOG_ERR, "Overlay %d has parts but no texture", n);
            rr->errors |= PL_RENDER_ERR_OVERLAY;
            return;
        }
        if (!ol.coords) {
            ol.coords = overlays == target->overlays ? PL_OVERLAY_COORDS_DST_FRAME
                                                     : PL_OVERLAY_COORDS_SRC_FRAME;
        }

        pl_transform2x2 tf = pl_transform2x2_identity;
        switch (ol.coords) {
        case PL_OVERLAY_COORDS_SRC_CROP:
            if (!image)
                continue;
            tf.c[0] = image->crop.x0;
            tf.c[1] = image->crop.y0;
            pl_transform2x2_rmul(&src_to_dst, &tf);
            break;
        case PL_OVERLAY_COORDS_SRC_FRAME:
            if (!image)
                continue;
            pl_transform2x2_rmul(&src_to_dst, &tf);
            break;
        case PL_OVERLAY_COORDS_DST_CROP:
            tf.c[0] = dst_crop.x0;
            tf.c[1] = dst_crop.y0;
            break;
        case PL_OVERLAY_COORDS_DST_FRAME:
            break;
        default:
            RR_LOG(rr, PL_LOG_ERR, "Overlay %d: invalid coordinate space %d", n, (int) ol.coords);
            continue;
        }
        if (output_shift)
            pl_transform2x2_rmul(output_shift, &tf);

        if (ol.num_parts > rr->osd_cap) {
            void *grown = realloc(rr->osd_parts, ol.num_parts * sizeof(*rr->osd_parts));
            if (!grown) {
                rr->errors |= PL_RENDER_ERR_OVERLAY;
                return;
            }
            rr->osd_parts = grown;
            rr->osd_cap = ol.num_parts;
        }
        for (int i = 0; i < ol.num_parts; i++)
            rr->osd_parts[i] = place_part(&ol.parts[i], &tf, ol.tex);

        // the colour half (:950-994)
        pl_shader sh = pl_dispatch_begin(rr->dp);
        if (!sh) {
            rr->errors |= PL_RENDER_ERR_OVERLAY;
            return;
        }
        sh_describef(sh, "overlay");
        sh->output = PL_SHADER_SIG_COLOR;
        pl_shader_decode_color(sh, &ol.repr, NULL);

        // the overlay takes the image's way through the colour management only where it is in
        // the image's colour space; anything else is mapped statelessly
        static const struct pl_color_map_params osd_params = {
            PL_COLOR_MAP_DEFAULTS
            .tone_mapping_function = &pl_tone_map_linear,
            .gamut_mapping         = &pl_gamut_map_saturation,
        };
        struct pl_color_space ol_color = ol.color, target_color = target->color;
        pl_color_space_infer_map(&ol_color, &target_color);
        if (image && pl_color_space_equal(&ol_color, &image->color)) {
            pl_shader_color_map_ex(sh, job->params->color_map_params, pl_color_map_args(
                .src   = ol_color,
                .dst   = color,
                .state = &rr->tone_map_state,
            ));
        } else {
            pl_shader_color_map_ex(sh, &osd_params, pl_color_map_args(ol.color, color));
        }

        const bool premul = repr.alpha == PL_ALPHA_PREMULTIPLIED;
        struct pl_color_repr enc = repr;
        pl_shader_encode_color(sh, &enc);
        const int coverage_at = sh->pass.num_ops;   // (a glyph's coverage goes in here, :987-991)
        append_forced_swizzle(sh, comps, comp_map);

        const bool blending = !(rr->errors & PL_RENDER_ERR_BLENDING);
        const struct pl_blend_params blend = {
            .src_rgb   = premul ? PL_BLEND_ONE : PL_BLEND_SRC_ALPHA,
            .src_alpha = PL_BLEND_ONE,
            .dst_rgb   = PL_BLEND_ONE_MINUS_SRC_ALPHA,
            .dst_alpha = PL_BLEND_ONE_MINUS_SRC_ALPHA,
        };
        const bool ok = plh_dispatch_overlay(rr->dp, &sh, fbo, &(struct plh_overlay_draw) {
            .tex = ol.tex,
            .mode = ol.mode == PL_OVERLAY_MONOCHROME ? PLH_OVERLAY_MONOCHROME : PLH_OVERLAY_NORMAL,
            .linear = (ol.tex->params.format->caps & PL_FMT_CAP_LINEAR) != 0,
            .premultiplied = premul,
            .coverage_at = coverage_at,
            .blend = blending ? &blend : NULL,
            .parts = rr->osd_parts,
            .num_parts = ol.num_parts,
        });
        if (!ok) {
            RR_LOG(rr, PL_LOG_ERR, "Failed rendering overlays!");
            rr->errors |= PL_RENDER_ERR_OVERLAY;
            return;
        }
    }
}

// The plane's own transform from target pixels (:2918-2934, :3405-3420): subsampling ratio, the
// chroma sample position, a flipped plane
pl_transform2x2 plh_plane_shift(const struct pl_plane *plane, pl_tex ref)
{
    const float rx = (float) plane->texture->params.w / ref->params.w,
                ry = (float) plane->texture->params.h / ref->params.h;
    const float rrx = rx >= 1 ? roundf(rx) : 1.0 / roundf(1.0 / rx),
                rry = ry >= 1 ? roundf(ry) : 1.0 / roundf(1.0 / ry);
    pl_transform2x2 tscale = {
        .mat = {{{ rrx, 0.0 }, { 0.0, rry }}},
        .c = { -plane->shift_x, -plane->shift_y },
    };
    if (plane->flipped) {
        tscale.mat.m[1][1] = -tscale.mat.m[1][1];
        tscale.c[1] += plane->texture->params.h;
    }
    return tscale;
}
