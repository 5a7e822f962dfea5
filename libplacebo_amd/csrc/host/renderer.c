/*
 * libplacebo-hip — pl_renderer: executor of the pl_render_image hot path.
 *
 * render_plan.c decides, this file does: it turns the planner's decisions into recorded
 * shaders (pl_shader_* calls, i.e. op lists for the HIP kernels) and dispatches them. A frame
 * moves through four stages, the same ones the reference's src/renderer.c has, with a pass
 * boundary (round trip through an rgba16hf image) exactly where the reference has one, because
 * that rounding is part of the numerics:
 *
 *   read     planes -> one image on the reference plane's grid: [deband], sample, merge,
 *            decode, premultiply                                    (reference :1553-1960)
 *   scale    [peak], linearize / sigmoidize, main scaler            (:1964-2087)
 *   colours  alpha mode, LUTs, tone + gamut mapping                 (:2157-2280)
 *   output   background, encode, dither / error diffusion, 1/scale, swizzle, store per plane
 *                                                                   (:2586-2960)
 *
 * Everything between two boundaries is ONE kernel launch. Where the main scaler is a polar
 * filter and what precedes it is a plain fetch plus colour ops, even that boundary is folded
 * into the polar kernel (the ops run on the source texels while they are staged, with the
 * rgba16hf rounding the intermediate image would have applied): same values, one pass less.
 *
 * Outside the scope of this backend: hooks (refused); ICC profiles, film grain
 * (ignored with an error bit / warning).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/shaders/deinterlacing.h>

#include "renderer_priv.h"

const struct pl_render_params pl_render_fast_params = { PL_RENDER_DEFAULTS };

const struct pl_render_params pl_render_default_params = {
    PL_RENDER_DEFAULTS
    .upscaler           = &pl_filter_lanczos,
    .downscaler         = &pl_filter_hermite,
    .sigmoid_params     = &pl_sigmoid_default_params,
    .dither_params      = &pl_dither_default_params,
    .peak_detect_params = &pl_peak_detect_default_params,
};

const struct pl_render_params pl_render_high_quality_params = {
    PL_RENDER_DEFAULTS
    .upscaler           = &pl_filter_ewa_lanczossharp,
    .downscaler         = &pl_filter_hermite,
    .sigmoid_params     = &pl_sigmoid_default_params,
    .peak_detect_params = &pl_peak_detect_high_quality_params,
    .color_map_params   = &pl_color_map_high_quality_params,
    .dither_params      = &pl_dither_default_params,
    .deband_params      = &pl_deband_default_params,
};

/* ---- object ---------------------------------------------------------------------------- */

pl_renderer pl_renderer_create(pl_log log, pl_gpu gpu)
{
    struct pl_renderer_t *rr = calloc(1, sizeof(*rr));
    if (!rr)
        return NULL;
    rr->gpu = gpu;
    rr->log = log;
    rr->dp = pl_dispatch_create(log, gpu);
    if (!rr->dp) {
        free(rr);
        return NULL;
    }
    return rr;
}

static void slot_release(struct scaler_slot *slot)
{
    pl_shader_obj_destroy(&slot->up);
    pl_shader_obj_destroy(&slot->down);
}

// The pre-v6 preset lists some option parsers still walk (src/renderer.c:226-244): the frame
// mixers, and the scalers = "none", "oversample" and then the common filter presets (the entries of
// pl_filter_presets behind its own "none").
const struct pl_filter_preset pl_frame_mixers[] = {
    { "none",           NULL,                       "No frame mixing" },
    { "linear",         &pl_filter_bilinear,        "Linear frame mixing" },
    { "oversample",     &pl_filter_oversample,      "Oversample (AKA SmoothMotion)" },
    { "mitchell_clamp", &pl_filter_mitchell_clamp,  "Clamped Mitchell spline" },
    { "hermite",        &pl_filter_hermite,         "Cubic spline (Hermite)" },
    {0}
};
const int pl_num_frame_mixers = sizeof(pl_frame_mixers) / sizeof(pl_frame_mixers[0]) - 1;

const struct pl_filter_preset pl_scale_filters[] = {
    { "none",       NULL,                   "Built-in sampling" },
    { "oversample", &pl_filter_oversample,  "Oversample (Aspect-preserving NN)" },
    PLH_COMMON_FILTER_PRESETS
    {0}
};
const int pl_num_scale_filters = sizeof(pl_scale_filters) / sizeof(pl_scale_filters[0]) - 1;

// Deprecated front ends of the gpu's pl_cache (src/renderer.c:184-192)
size_t pl_renderer_save(pl_renderer rr, uint8_t *out)
{
    return pl_cache_save(plh_gpu_cache(rr->gpu), out, out ? SIZE_MAX : 0);
}

void pl_renderer_load(pl_renderer rr, const uint8_t *cache)
{
    pl_cache_load(plh_gpu_cache(rr->gpu), cache, SIZE_MAX);
}

void pl_renderer_flush_cache(pl_renderer rr)
{
    for (int i = 0; i < rr->num_cached; i++)
        pl_tex_destroy(rr->gpu, &rr->cache[i].tex);
    for (int i = 0; i < rr->num_spare; i++)
        pl_tex_destroy(rr->gpu, &rr->spare[i]);
    for (int i = 0; i < rr->num_fbos; i++)
        pl_tex_destroy(rr->gpu, &rr->fbos[i]);
    for (int i = 0; i < RR_MEASURE_FBOS; i++)
        pl_tex_destroy(rr->gpu, &rr->measure_fbo[i]);
    rr->num_cached = rr->num_spare = rr->num_fbos = 0;
    pl_reset_detected_peak(rr->tone_map_state);
}

void pl_renderer_destroy(pl_renderer *ptr)
{
    pl_renderer rr = ptr ? *ptr : NULL;
    if (!rr)
        return;
    pl_gpu_finish(rr->gpu);     // nothing recorded may still reference our objects
    pl_renderer_flush_cache(rr);
    slot_release(&rr->scale_main);
    slot_release(&rr->scale_ref);
    slot_release(&rr->scale_contrast);
    for (int i = 0; i < PL_MAX_PLANES; i++) {
        slot_release(&rr->scale_plane[i]);
        slot_release(&rr->scale_out[i]);
    }
    for (int i = 0; i < RR_LUT_COUNT; i++)
        pl_shader_obj_destroy(&rr->lut_state[i]);
    pl_shader_obj_destroy(&rr->tone_map_state);
    pl_shader_obj_destroy(&rr->dither_state);
    pl_dispatch_destroy(&rr->dp);
    free(rr->osd_parts);
    free(rr);
    *ptr = NULL;
}

struct pl_render_errors pl_renderer_get_errors(pl_renderer rr)
{
    return (struct pl_render_errors) { .errors = rr->errors };
}

void pl_renderer_reset_errors(pl_renderer rr, const struct pl_render_errors *errors)
{
    rr->errors = errors ? rr->errors & ~errors->errors : PL_RENDER_ERR_NONE;
}

bool pl_renderer_get_hdr_metadata(pl_renderer rr, struct pl_hdr_metadata *metadata)
{
    return pl_get_detected_hdr_metadata(rr->tone_map_state, metadata);
}

pl_shader_obj pl_hip_renderer_tone_map_state(pl_renderer rr)
{
    return rr->tone_map_state;
}

static void raise(pl_renderer rr, enum pl_render_error bit, enum pl_log_level lev, const char *msg)
{
    RR_LOG(rr, lev, "%s", msg);
    rr->errors |= bit;
}

/* ---- intermediate images ------------------------------------------------------------------ */

// Preference order for the intermediate format (reference :383-434): 16-bit float first, then
// 16-bit integer, then 8 bit; linearly sampleable before merely sampleable.
static void choose_fbo_formats(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    struct rp_caps *caps = &job->caps;
    if (job->params->disable_fbos || (rr->errors & PL_RENDER_ERR_FBO) || caps->fbo[4])
        return;

    static const struct candidate {
        enum pl_fmt_type type;
        int depth;
        bool linear;
    } order[] = {
        { PL_FMT_FLOAT, 16, true }, { PL_FMT_FLOAT, 16, false },
        { PL_FMT_UNORM, 16, true }, { PL_FMT_SNORM, 16, true },
        { PL_FMT_UNORM, 16, false }, { PL_FMT_SNORM, 16, false },
        { PL_FMT_UNORM, 8, true }, { PL_FMT_UNORM, 8, false },
    };

    for (size_t i = 0; i < PL_ARRAY_SIZE(order); i++) {
        const struct candidate *c = &order[i];
        if (c->depth > 8 && job->params->force_low_bit_depth_fbos)
            continue;
        const enum pl_fmt_caps need = PL_FMT_CAP_RENDERABLE |
            (c->linear ? PL_FMT_CAP_LINEAR : PL_FMT_CAP_SAMPLEABLE);
        pl_fmt four = pl_find_fmt(rr->gpu, c->type, 4, c->depth, 0, need);
        if (!four)
            continue;
        caps->fbo[4] = four;
        // narrower variants with the same capabilities, else the next wider one
        for (int n = 3; n >= 1; n--) {
            pl_fmt f = pl_find_fmt(rr->gpu, c->type, n, c->depth, 0, four->caps);
            caps->fbo[n] = f ? f : caps->fbo[n + 1];
        }
        return;
    }
    raise(rr, PL_RENDER_ERR_FBO, PL_LOG_WARN,
          "Found no renderable FBO format! Most features disabled");
}

// An unused pooled image that fits (w, h, fmt) best, recreated to fit exactly
static pl_tex borrow_fbo(struct frame_job *job, int w, int h, pl_fmt fmt, int comps)
{
    pl_renderer rr = job->rr;
    if (!fmt)
        fmt = job->caps.fbo[comps ? comps : 4];
    if (!fmt)
        return NULL;

    int pick = -1, pick_cost = 0;
    for (int i = 0; i < rr->num_fbos; i++) {
        if (job->fbo_busy[i])
            continue;
        const struct pl_tex_params *have = &rr->fbos[i]->params;
        const int cost = abs(have->w - w) + abs(have->h - h) + (have->format == fmt ? 0 : 1000);
        if (pick < 0 || cost < pick_cost) {
            pick = i;
            pick_cost = cost;
        }
    }
    if (pick < 0) {
        if (rr->num_fbos == RR_MAX_FBOS)
            return NULL;
        pick = rr->num_fbos++;
        rr->fbos[pick] = NULL;
    }

    const struct pl_tex_params want = {
        .w = w, .h = h, .format = fmt,
        .sampleable = true, .renderable = true,
        .storable = !!(fmt->caps & PL_FMT_CAP_STORABLE),
    };
    if (!pl_tex_recreate(rr->gpu, &rr->fbos[pick], &want))
        return NULL;
    job->fbo_busy[pick] = true;
    return rr->fbos[pick];
}

// The intermediate of a pass that goes on the measurement stream (see renderer_priv.h)
static pl_tex borrow_measure_fbo(struct frame_job *job, int w, int h, pl_fmt fmt, int comps)
{
    pl_renderer rr = job->rr;
    if (!fmt)
        fmt = job->caps.fbo[comps ? comps : 4];
    if (!fmt || !(fmt->caps & PL_FMT_CAP_STORABLE) || job->measure_fbo)
        return NULL;
    pl_tex *slot = &rr->measure_fbo[rr->measure_flip % RR_MEASURE_FBOS];
    const struct pl_tex_params want = {
        .w = w, .h = h, .format = fmt,
        .sampleable = true, .renderable = true, .storable = true,
    };
    if (!pl_tex_recreate(rr->gpu, slot, &want))
        return NULL;
    rr->measure_flip++;
    job->measure_fbo = *slot;
    plh_tex_fence_reads(*slot);
    return *slot;
}

// true if `rec` would do nothing but copy `copy_of`: then that texture is the answer
static bool is_pure_copy(const struct work_image *img)
{
    const pl_shader sh = img->rec;
    pl_tex o = img->copy_of;
    if (!sh || !o || img->store_as || sh->kind != PLH_SHADER_PASS || pl_shader_is_failed(sh))
        return false;
    const struct plh_pass *p = &sh->pass;
    const bool plain_fetch = p->s.type == PLH_SAMPLE_NEAREST || p->s.type == PLH_SAMPLE_BILINEAR;
    return plain_fetch && !p->num_ops && !p->num_pre_ops && p->s.scale == 1.0f &&
           !sh->detect_peak && o->params.w == img->w && o->params.h == img->h;
}

// Make the image resident (reference _img_tex :505-547)
pl_tex plh_work_texture(struct frame_job *job, struct work_image *img)
{
    pl_renderer rr = job->rr;
    if (img->tex)
        return img->tex;

    if (is_pure_copy(img)) {
        pl_dispatch_abort(rr->dp, &img->rec);
        img->tex = img->copy_of;
        img->copy_of = NULL;
        return img->tex;
    }
    img->copy_of = NULL;

    pl_tex fbo = NULL;
    if (plh_gpu_async(rr->gpu) && plh_shader_aux_eligible(img->rec)) {
        fbo = borrow_measure_fbo(job, img->w, img->h, img->store_as, img->comps);
        img->rec->on_aux = !!fbo;
    }
    if (!fbo)
        fbo = borrow_fbo(job, img->w, img->h, img->store_as, img->comps);
    img->store_as = NULL;
    if (!fbo) {
        // without intermediates only the simplest pipeline remains
        raise(rr, PL_RENDER_ERR_FBO, PL_LOG_ERR,
              "Failed creating FBO texture! Disabling advanced rendering..");
        memset(job->caps.fbo, 0, sizeof(job->caps.fbo));
        pl_dispatch_abort(rr->dp, &img->rec);
        return img->fail_tex;
    }

    const bool ok = pl_dispatch_finish(rr->dp, pl_dispatch_params(
        .shader = &img->rec,
        .target = fbo,
    ));
    const char *msg = img->fail_msg ? img->fail_msg : "Failed dispatching intermediate pass!";
    const enum pl_render_error bit = img->fail_bit;
    pl_tex fallback = img->fail_tex;
    img->fail_msg = NULL;
    img->fail_bit = PL_RENDER_ERR_NONE;
    img->fail_tex = NULL;

    if (!ok) {
        raise(rr, bit, PL_LOG_ERR, msg);
        img->tex = fallback;
        return fallback;
    }
    img->tex = fbo;
    return fbo;
}

// Make the image recordable (reference img_sh :552-567)
pl_shader plh_work_shader(struct frame_job *job, struct work_image *img)
{
    if (!img->rec) {
        img->rec = pl_dispatch_begin(job->rr->dp);
        pl_shader_sample_direct(img->rec, pl_sample_src( .tex = img->tex ));
        img->copy_of = img->tex;
        img->tex = NULL;
    }
    return img->rec;
}

struct plh_op *plh_append_scale(pl_shader sh, float k, bool with_alpha)
{
    struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
    if (!op)
        return NULL;
    op->f[0] = op->f[1] = op->f[2] = k;
    op->f[3] = with_alpha ? k : 1.0f;
    sh_listf(sh, "scale(%g%s)\n", k, with_alpha ? "" : ", rgb only");
    return op;
}

/* ---- scalers -------------------------------------------------------------------------------- */

// Separable filter along both axes: vertical pass into an intermediate, then horizontal
static bool run_two_pass(struct frame_job *job, pl_shader sh, const struct pl_sample_src *req,
                         const struct pl_sample_filter_params *fp)
{
    pl_renderer rr = job->rr;
    struct pl_sample_src vert = *req, horiz = *req;
    vert.new_w = req->tex->params.w;
    vert.rect.x0 = 0;
    vert.rect.x1 = vert.new_w;
    horiz.rect.y0 = 0;
    horiz.rect.y1 = vert.new_h;

    pl_shader first = pl_dispatch_begin(rr->dp);
    if (!pl_shader_sample_ortho2(first, &vert, fp)) {
        pl_dispatch_abort(rr->dp, &first);
        return false;
    }
    struct work_image mid = {
        .rec = first, .w = vert.new_w, .h = vert.new_h, .comps = req->components,
    };
    horiz.tex = plh_work_texture(job, &mid);
    horiz.scale = 1.0;
    return horiz.tex && pl_shader_sample_ortho2(sh, &horiz, fp);
}

// The low-pass of the contrast-recovery feature map (a separable downscale of a one-component
// plane: two passes through an intermediate in the reference, src/renderer.c:2089-2154) as ONE launch
// that keeps the intermediate in LDS (k_lowpass2). true = `small` is written; false = nothing was
// done (not a separable filter, not a shape the kernel takes): the caller runs the two passes.
static bool try_fused_lowpass(struct frame_job *job, const struct pl_sample_src *req, pl_tex small)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    struct rp_scaler sc = rp_pick_scaler(&job->caps, params, RP_USE_LOWPASS, req,
                                         req->tex ? req->tex->params.format : NULL);
    if (sc.kind != RP_SCALER_FILTER || sc.filter->polar || !(sc.axis[0] && sc.axis[1]) ||
        sc.dir == RP_DIR_NONE)
        return false;
    const struct pl_sample_filter_params fp = {
        .filter      = *sc.filter,
        .antiring    = params->antiringing_strength,
        .no_widening = false,
        .lut         = sc.dir == RP_DIR_UP ? &rr->scale_contrast.up : &rr->scale_contrast.down,
    };
    // (run_two_pass's two requests)
    struct pl_sample_src vert = *req, horiz = *req;
    vert.new_w = req->tex->params.w;
    vert.rect.x0 = 0;
    vert.rect.x1 = vert.new_w;
    horiz.rect.y0 = 0;
    horiz.rect.y1 = vert.new_h;
    // the intermediate exists as a texture (the horizontal pass is recorded against it) and is
    // never written
    horiz.tex = borrow_fbo(job, vert.new_w, vert.new_h, NULL, 1);
    horiz.scale = 1.0;
    if (!horiz.tex)
        return false;
    pl_shader first = pl_dispatch_begin(rr->dp), second = pl_dispatch_begin(rr->dp);
    int done = 0;
    if (pl_shader_sample_ortho2(first, &vert, &fp) && pl_shader_sample_ortho2(second, &horiz, &fp))
        done = plh_dispatch_lowpass2(rr->dp, &first, &second, small);
    if (done <= 0) {
        // declined: the ordinary two dispatches, from the shaders as recorded
        bool ok = done == 0 && first && second && !first->failed && !second->failed;
        if (ok)
            ok = pl_dispatch_finish(rr->dp, pl_dispatch_params( .shader = &first, .target = horiz.tex ));
        if (ok)
            ok = pl_dispatch_finish(rr->dp, pl_dispatch_params( .shader = &second, .target = small ));
        pl_dispatch_abort(rr->dp, &first);
        pl_dispatch_abort(rr->dp, &second);
        return ok;
    }
    return true;
}

// Record the sampling of `req` into `sh` with whatever rp_pick_scaler chooses for it
static void run_scaler(struct frame_job *job, pl_shader sh, struct scaler_slot *slot,
                       enum rp_usage usage, const struct pl_sample_src *req)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    struct rp_scaler sc = { .kind = RP_SCALER_BUILTIN };
    if (slot)
        sc = rp_pick_scaler(&job->caps, params, usage, req, req->tex ? req->tex->params.format : NULL);
    if (sc.dir == RP_DIR_NONE)
        sc.kind = RP_SCALER_BUILTIN;    // 1:1: a plain fetch

    switch (sc.kind) {
    case RP_SCALER_NEAREST:
        pl_shader_sample_nearest(sh, req);
        return;
    case RP_SCALER_BICUBIC:
        pl_shader_sample_bicubic(sh, req);
        return;
    case RP_SCALER_HERMITE:
        pl_shader_sample_hermite(sh, req);
        return;
    case RP_SCALER_GAUSSIAN:
        pl_shader_sample_gaussian(sh, req);
        return;
    case RP_SCALER_OVERSAMPLE:
        pl_shader_sample_oversample(sh, req, sc.filter->kernel->params[0]);
        return;
    case RP_SCALER_BUILTIN:
        pl_shader_sample_direct(sh, req);
        return;
    case RP_SCALER_FILTER:
        break;
    }

    const struct pl_sample_filter_params fp = {
        .filter      = *sc.filter,
        .antiring    = params->antiringing_strength,
        .no_widening = params->skip_anti_aliasing && usage != RP_USE_LOWPASS,
        .lut         = sc.dir == RP_DIR_UP ? &slot->up : &slot->down,
    };
    bool ok;
    if (sc.filter->polar)
        ok = pl_shader_sample_polar(sh, req, &fp);
    else if (sc.axis[0] && sc.axis[1])
        ok = run_two_pass(job, sh, req, &fp);
    else
        ok = pl_shader_sample_ortho2(sh, req, &fp);

    if (!ok) {
        raise(rr, PL_RENDER_ERR_SAMPLING, PL_LOG_ERR, "Failed dispatching scaler.. disabling");
        pl_shader_sample_direct(sh, req);
    }
}

// Fold the pending image (`pre`: a plain whole-plane fetch + colour ops) into a polar main
// scaler instead of writing it to an intermediate first. Only valid where the intermediate
// would be rgba16hf, because the fused tile rounds to f16.
static bool try_fused_polar(struct frame_job *job, pl_shader sh, const struct pl_sample_src *req,
                            pl_shader pre, int w, int h)
{
    const struct pl_render_params *params = job->params;
    const char *off = getenv("PL_HIP_NO_FUSION");
    const int force = off && (off[0] == '0' || off[0] == '1') ? off[0] - '0' : -1;
    if (!pre || force == 1)
        return false;
    pl_fmt fmt = job->caps.fbo[job->img.comps];
    if (!fmt || fmt->type != PL_FMT_FLOAT || fmt->component_depth[0] != 16)
        return false;

    // the scaler choice looks at the source format: describe the intermediate that is skipped
    const struct pl_tex_t stand_in = { .params = { .w = w, .h = h, .format = fmt } };
    struct pl_sample_src probe = *req;
    probe.tex = &stand_in;
    const struct rp_scaler sc = rp_pick_scaler(&job->caps, params, RP_USE_MAIN, &probe, fmt);
    if (sc.kind != RP_SCALER_FILTER || !sc.filter->polar)
        return false;
    // (the rule -- which directions fuse, with which pending ops -- is the planner's:
    // render_plan.h, tests/test_render_plan.py; measured in profiles/r04_44_*)
    bool pending_lite = true;
    for (int i = 0; i < pre->pass.num_ops; i++) {
        switch (pre->pass.ops[i].kind) {
        case PLH_OP_SCALE: case PLH_OP_AFFINE: case PLH_OP_PREMULTIPLY: case PLH_OP_ALPHA_ONE:
        case PLH_OP_QUANT_F16: case PLH_OP_SWIZZLE: case PLH_OP_CLAMP01: case PLH_OP_PLANE_MAP:
            break;
        default:
            pending_lite = false;
        }
    }
    const float ar = sc.filter->antiring ? sc.filter->antiring : params->antiringing_strength;
    if (!rp_fuse_into_polar(sc.dir, pending_lite, ar, force))
        return false;

    const struct pl_sample_filter_params fp = {
        .filter      = *sc.filter,
        .antiring    = params->antiringing_strength,
        .no_widening = params->skip_anti_aliasing,
        .lut         = sc.dir == RP_DIR_UP ? &job->rr->scale_main.up : &job->rr->scale_main.down,
    };
    return plh_shader_sample_polar_fused(sh, pre, &probe, &fp);
}

/* ---- job set-up --------------------------------------------------------------------------------- */

static void forward_pass_info(void *priv, const struct pl_dispatch_info *dinfo)
{
    struct frame_job *job = priv;
    if (!job->params->info_callback)
        return;
    job->info.pass = dinfo;
    job->params->info_callback(job->params->info_priv, &job->info);
    job->info.index++;
}

// Only when somebody listens (the reference: `if (params->info_callback)`, renderer.c pass_init):
// a watched pass is bracketed by two timer events, and an event recorded behind a kernel costs the
// stream about 3 us (tools/ubench/launch_gap.hip) -- installed unconditionally, that was 7.7 us of
// every 27 us bilinear frame and 4-12 us of the others.
void plh_job_watch_passes(struct frame_job *job)
{
    pl_dispatch_reset_frame(job->rr->dp);
    if (job->params->info_callback)
        pl_dispatch_callback(job->rr->dp, job, forward_pass_info);
    else
        pl_dispatch_callback(job->rr->dp, NULL, NULL);
}

bool plh_params_supported(pl_renderer rr, const struct pl_render_params *p)
{
    const char *what = p->num_hooks ? "hooks" : NULL;
    if (!what)
        return true;
    RR_LOG(rr, PL_LOG_ERR, "pl_render_params.%s requests a stage this backend does not have "
           "(outside the pl_render_image hot path, SURVEY.md 8)", what);
    return false;
}

// things a frame may carry that are not rendered here: say so once, keep going
static void note_ignored_members(pl_renderer rr, const struct pl_frame *f)
{
    if ((f->icc || f->profile.data) && !rr->warned_icc) {
        rr->warned_icc = true;
        RR_LOG(rr, PL_LOG_WARN, "ICC profiles are not interpreted by this build (no lcms2): "
               "rendering from the frame's pl_color_space");
    }
    if (f->film_grain.type != PL_FILM_GRAIN_NONE && !rr->warned_grain) {
        rr->warned_grain = true;
        raise(rr, PL_RENDER_ERR_FILM_GRAIN, PL_LOG_WARN,
              "Film grain synthesis is not part of this backend");
    }
}

void plh_job_end(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    pl_dispatch_abort(rr->dp, &job->img.rec);
    pl_dispatch_callback(rr->dp, NULL, NULL);
    if (job->measure_fbo) {
        // every reader of this frame's measured intermediate is queued by now: the measuring
        // pass that reuses the texture two frames on waits for this point
        plh_tex_read_so_far(rr->gpu, job->measure_fbo, 0);
        job->measure_fbo = NULL;
    }
    if (job->prev_acquired && job->prev.release)
        job->prev.release(rr->gpu, &job->prev);
    if (job->next_acquired && job->next.release)
        job->next.release(rr->gpu, &job->next);
    job->prev_acquired = job->next_acquired = false;
    if (job->image_acquired && job->image.release)
        job->image.release(rr->gpu, &job->image);
    if (job->target_acquired && !job->target_borrowed && job->target.release)
        job->target.release(rr->gpu, &job->target);
    job->image_acquired = job->target_acquired = false;
}

// validate_deinterlace_ref (:2989-3001, :3033-3039)
static const char *deinterlace_refs_problem(const struct pl_frame *image)
{
    if (image->field == PL_FIELD_NONE)
        return NULL;
    if (image->first_field == PL_FIELD_NONE)
        return "an interlaced frame must say which field comes first (first_field)";
    const struct pl_frame *refs[2] = { image->prev, image->next };
    for (int r = 0; r < 2; r++) {
        const struct pl_frame *ref = refs[r];
        if (!ref)
            continue;
        if (ref->num_planes != image->num_planes)
            return "prev / next must have the planes of the frame they surround";
        for (int p = 0; p < image->num_planes; p++) {
            pl_tex a = image->planes[p].texture, b = ref->planes[p].texture;
            if (!b || !b->params.sampleable || a->params.w != b->params.w ||
                a->params.h != b->params.h ||
                a->params.format->num_components != b->params.format->num_components)
                return "prev / next must have sampleable planes of the same size and components";
        }
    }
    return NULL;
}

// acquire both frames, validate, fit the rects, complete the descriptions (:3317-3428)
bool plh_job_begin(struct frame_job *job, bool acquire_image)
{
    pl_renderer rr = job->rr;
    if (!job->target_acquired && !job->target_borrowed && job->target.acquire) {
        if (!job->target.acquire(rr->gpu, &job->target))
            return false;
        job->target_acquired = true;
    }
    if (acquire_image && job->image.acquire) {
        if (!job->image.acquire(rr->gpu, &job->image)) {
            plh_job_end(job);
            return false;
        }
        job->image_acquired = true;
    }

    // the frames a temporal deinterlacer reads beside the image (:3329-3350)
    const struct pl_deinterlace_params *deint = job->params->deinterlace_params;
    if (acquire_image && job->image.field != PL_FIELD_NONE && deint &&
        pl_deinterlace_needs_refs(deint->algo))
    {
        if (job->image.prev) {
            job->prev = *job->image.prev;
            job->image.prev = &job->prev;
            if (job->prev.acquire && !job->prev.acquire(rr->gpu, &job->prev)) {
                plh_job_end(job);
                return false;
            }
            job->prev_acquired = true;
        }
        if (job->image.next) {
            job->next = *job->image.next;
            job->image.next = &job->next;
            if (job->next.acquire && !job->next.acquire(rr->gpu, &job->next)) {
                plh_job_end(job);
                return false;
            }
            job->next_acquired = true;
        }
    }

    const char *bad = rp_frame_problem(&job->image, false);
    if (!bad)
        bad = deinterlace_refs_problem(&job->image);
    if (bad) {
        RR_LOG(rr, PL_LOG_ERR, "Image frame: %s", bad);
    } else if ((bad = rp_frame_problem(&job->target, true))) {
        RR_LOG(rr, PL_LOG_ERR, "Target frame: %s", bad);
    }
    if (bad) {
        plh_job_end(job);
        return false;
    }
    note_ignored_members(rr, &job->image);
    note_ignored_members(rr, &job->target);

    job->caps.max_shmem = rr->gpu->glsl.max_shmem_size;
    job->caps.sampling_broken = rr->errors & PL_RENDER_ERR_SAMPLING;
    job->caps.peak_broken = rr->errors & PL_RENDER_ERR_PEAK_DETECT;
    job->caps.deband_broken = rr->errors & PL_RENDER_ERR_DEBANDING;
    job->caps.contrast_broken = rr->errors & PL_RENDER_ERR_CONTRAST_RECOVERY;
    job->caps.errdiff_broken = rr->errors & PL_RENDER_ERR_ERROR_DIFFUSION;
    choose_fbo_formats(job);

    pl_tex iref = job->image.planes[rp_reference_plane(&job->image)].texture,
           tref = job->target.planes[rp_reference_plane(&job->target)].texture;
    job->geo = rp_fit_rects(job->image.crop, iref->params.w, iref->params.h, job->image.rotation,
                            job->target.crop, tref->params.w, tref->params.h,
                            job->target.rotation);
    job->image.crop = job->geo.src;
    job->target.crop = job->geo.dstf;
    rp_complete_frames(&job->image, &job->target);
    return true;
}

void pl_frames_infer(pl_renderer rr, struct pl_frame *image, struct pl_frame *target)
{
    if (rp_frame_problem(image, false) || rp_frame_problem(target, true))
        return;
    pl_tex iref = image->planes[rp_reference_plane(image)].texture,
           tref = target->planes[rp_reference_plane(target)].texture;
    const struct rp_geometry geo = rp_fit_rects(image->crop, iref->params.w, iref->params.h,
                                                image->rotation, target->crop, tref->params.w,
                                                tref->params.h, target->rotation);
    image->crop = geo.src;
    target->crop = geo.dstf;
    rp_complete_frames(image, target);
    (void) rr;
}

void pl_frame_set_chroma_location(struct pl_frame *frame, enum pl_chroma_location loc)
{
    pl_tex ref = frame->planes[rp_reference_plane(frame)].texture;
    for (int i = 0; i < frame->num_planes; i++) {
        struct pl_plane *plane = &frame->planes[i];
        // a plane smaller than the reference is subsampled; without textures, go by content
        bool subsampled;
        if (ref && plane->texture) {
            subsampled = plane->texture->params.w < ref->params.w ||
                         plane->texture->params.h < ref->params.h;
        } else {
            subsampled = rp_plane_role(plane, &frame->repr) == RP_PLANE_CHROMA;
        }
        if (subsampled)
            pl_chroma_location_offset(loc, &plane->shift_x, &plane->shift_y);
    }
}

bool pl_frame_is_cropped(const struct pl_frame *frame)
{
    if (!frame->num_planes)
        return false;
    pl_tex ref = frame->planes[rp_reference_plane(frame)].texture;
    if (!ref)
        return false;
    pl_rect2df crop = frame->crop;
    if (!crop.x0 && !crop.y0 && !crop.x1 && !crop.y1)
        return false;   // unset = the whole plane
    pl_rect2df_normalize(&crop);
    const pl_rect2d r = pl_rect2df_round(&crop);
    return r.x0 > 0 || r.y0 > 0 || r.x1 < ref->params.w || r.y1 < ref->params.h;
}

/* ---- stage 1: read -------------------------------------------------------------------------- */

// Replace a plane's texture by its debanded version (recorded, not yet run). :1318-1350
static void deband_plane(struct frame_job *job, struct work_image *pimg, const float neutral[3])
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    if (!params->deband_params || job->caps.deband_broken || !job->caps.fbo[4])
        return;

    pl_tex source = plh_work_texture(job, pimg);
    struct pl_color_repr repr = pimg->repr;
    const struct pl_sample_src src = {
        .tex = source,
        .components = pimg->comps,
        .scale = pl_color_repr_normalize(&repr),
    };

    struct pl_deband_params dp = *params->deband_params;
    // grain strength is specified for SDR white: keep it there for brighter sources
    dp.grain /= job->image.color.hdr.max_luma / PL_COLOR_SDR_WHITE;
    memcpy(dp.grain_neutral, neutral, sizeof(dp.grain_neutral));

    pimg->tex = NULL;
    pimg->rec = pl_dispatch_begin(rr->dp);
    pl_shader_deband(pimg->rec, &src, &dp);
    pimg->repr = repr;
    pimg->fail_msg = "Failed applying debanding... disabling!";
    pimg->fail_bit = PL_RENDER_ERR_DEBANDING;
    pimg->fail_tex = source;
}

// An interlaced frame: the plane becomes its deinterlaced version (recorded, not yet run),
// stored -- if it has to be -- in the plane's own format (:1591-1612)
static void deinterlace_plane(struct frame_job *job, struct work_image *pimg, int plane)
{
    pl_renderer rr = job->rr;
    const struct pl_frame *image = &job->image;
    const struct pl_render_params *params = job->params;
    if (image->field == PL_FIELD_NONE || !params->deinterlace_params || !job->caps.fbo[4] ||
        (rr->errors & PL_RENDER_ERR_DEINTERLACING))
        return;

    pl_tex source = pimg->tex;
    const struct pl_deinterlace_source src = {
        .cur.top  = source,
        .prev.top = image->prev ? image->prev->planes[plane].texture : NULL,
        .next.top = image->next ? image->next->planes[plane].texture : NULL,
        .field    = image->field,
        .first_field = image->first_field,
        .component_mask = (1 << pimg->comps) - 1,
    };
    pimg->tex = NULL;
    pimg->rec = pl_dispatch_begin(rr->dp);
    pl_shader_deinterlace(pimg->rec, &src, params->deinterlace_params);
    if (source->params.format->caps & PL_FMT_CAP_STORABLE)
        pimg->store_as = source->params.format;
    pimg->fail_msg = "Failed deinterlacing plane.. disabling!";
    pimg->fail_bit = PL_RENDER_ERR_DEINTERLACING;
    pimg->fail_tex = source;
}

// `fetch` must be a bare nearest / bilinear fetch: its texture is read from inside `sh`
bool plh_append_plane_fetch(pl_shader sh, const pl_shader fetch, const struct pl_plane *plane)
{
    const struct plh_sampler_args *s = &fetch->pass.s;
    const bool bilinear = s->type == PLH_SAMPLE_BILINEAR;
    if (fetch->kind != PLH_SHADER_PASS || fetch->pass.num_ops ||
        (s->type != PLH_SAMPLE_NEAREST && !bilinear))
        return false;
    if (s->src.w > 0xffff || s->src.h > 0xffff)
        return false;   // sizes are packed into 16 bits each

    struct plh_op *op = sh_op(sh, PLH_OP_PLANE_FETCH);
    if (!op)
        return false;

    uint32_t dest = 0;      // 4 bits per fetched component: where it goes (0xf = nowhere)
    for (int c = 0; c < 4; c++) {
        const int to = c < plane->components ? plane->component_mapping[c] : -1;
        dest |= (uint32_t) (to < 0 ? 0xf : to) << (4 * c);
    }
    memcpy(op->f, s->pos, sizeof(s->pos));
    op->f[8] = s->scale;
    op->f[9] = s->rect_w;
    op->f[10] = s->rect_h;
    op->ptr = s->src.ptr;
    op->i0 = s->src.w | (s->src.h << 16);
    op->i1 = s->src.pitch;
    op->i2 = s->src.fmt | (plane->components << 8) | (bilinear << 12) | (s->address_mode << 13) |
             ((bilinear && s->rect_on_grid) << 15) | (dest << 16);
    sh_listf(sh, "plane_fetch(tex=%dx%d, %s, scale=%g, comps=%d, map=0x%04x)\n", s->src.w,
             s->src.h, bilinear ? "bilinear" : "nearest", s->scale, plane->components, dest);
    for (int i = 0; i < fetch->num_held; i++)
        sh_hold(sh, fetch->held[i]);
    return true;
}

// color = (neutral luma, neutral chroma x 2, 1), then the sampled components go where the
// plane's mapping says. Not needed for a plain RGBA plane.
static bool append_plane_map(pl_shader sh, const struct pl_plane *plane, bool only_plane,
                             float neutral_luma, float neutral_chroma)
{
    bool identity = true;
    uint32_t dest = 0;
    for (int c = 0; c < 4; c++) {
        const int to = c < plane->components ? plane->component_mapping[c] : -1;
        dest |= (uint32_t) (to < 0 ? 0xff : to) << (8 * c);
        if (c < plane->components && to != c)
            identity = false;
    }
    if (identity && plane->components == 4 && only_plane)
        return true;

    struct plh_op *op = sh_op(sh, PLH_OP_PLANE_MAP);
    if (!op)
        return false;
    op->f[0] = neutral_luma;
    op->f[1] = op->f[2] = neutral_chroma;
    op->f[3] = 1.0f;
    op->i0 = dest;
    op->i1 = plane->components;
    op->i2 = identity;      // the sampled components stay where they are
    sh_listf(sh, "plane_map(comps=%d, map=0x%08x, neutral=%g/%g)\n", plane->components,
             (unsigned) dest, neutral_luma, neutral_chroma);
    return true;
}

bool plh_stage_read(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    const struct pl_frame *image = &job->image;

    struct rp_image_layout lay;
    rp_layout_image(image, &lay);

    // every plane: texture -> [deband] -> sampled onto the reference grid
    struct work_image pimg[PL_MAX_PLANES];
    float gain[PL_MAX_PLANES];
    for (int i = 0; i < image->num_planes; i++) {
        const struct rp_plane_layout *pl = &lay.planes[i];
        if (!pl->role)
            continue;
        pl_tex tex = image->planes[i].texture;
        pimg[i] = (struct work_image) {
            .tex = tex, .w = tex->params.w, .h = tex->params.h,
            .repr = image->repr, .color = image->color,
            .comps = image->planes[i].components,
            .rect = pl->rect,
        };
        deinterlace_plane(job, &pimg[i], i);
        deband_plane(job, &pimg[i], pl->neutral);

        struct pl_sample_src req = rp_plane_request(&lay, i);
        req.scale = pl_color_repr_normalize(&pimg[i].repr);
        bool as_is = pimg[i].rec && pimg[i].w == req.new_w && pimg[i].h == req.new_h &&
                     rp_plane_request_is_identity(&req);
        if (as_is && pimg[i].rec->pass.s.type == PLH_SAMPLE_DEINTERLACE && !pimg[i].rec->pass.num_ops) {
            // A deinterlaced plane that IS the reference grid continues unrounded in the reference
            // (the deinterlacing shader goes on to become the decoding pass). Same values here, but
            // through a float image of the plane's layout: the deinterlacer then runs as the
            // row-dword kernel it has for whole planes (k_deint_rows: 4 pixels of an r8 plane per
            // lane) and everything behind it as the kernels tuned for a texture source -- against
            // one kernel that does both through the interpreter (1080i luma: 87 us -> 4 + 22)
            static const char *const names[] = { NULL, "r32f", "rg32f", NULL, "rgba32f" };
            const int nc = image->planes[i].texture->params.format->num_components;
            pl_fmt exact = nc <= 4 && names[nc] ? pl_find_named_fmt(rr->gpu, names[nc]) : NULL;
            if (exact && (exact->caps & PL_FMT_CAP_STORABLE)) {
                pimg[i].store_as = exact;
                if (plh_work_texture(job, &pimg[i]))
                    as_is = false;
            }
        }
        if (!as_is) {
            req.tex = plh_work_texture(job, &pimg[i]);
            if (!req.tex)
                return false;
            pimg[i].tex = NULL;
            pimg[i].rec = pl_dispatch_begin(rr->dp);
            run_scaler(job, pimg[i].rec, i == lay.ref ? &rr->scale_ref : &rr->scale_plane[i],
                       RP_USE_PLANE, &req);
            pimg[i].fail_bit |= PL_RENDER_ERR_SAMPLING;
            pimg[i].w = req.new_w;
            pimg[i].h = req.new_h;
            pimg[i].rect = (pl_rect2df) { 0, 0, req.new_w, req.new_h };
            req.scale = 1.0;    // applied by the sampler
        }
        gain[i] = req.scale;
    }

    // the reference plane's recording becomes the pass; the others are fetched into it
    struct work_image *ref = &pimg[lay.ref];
    pl_shader sh = plh_work_shader(job, ref);
    ref->rec = NULL;
    if (gain[lay.ref] != 1.0f && !plh_append_scale(sh, gain[lay.ref], true))
        return false;
    if (!append_plane_map(sh, &lay.planes[lay.ref].plane, image->num_planes == 1,
                          lay.neutral_luma, lay.neutral_chroma))
        return false;

    for (int i = 0; i < image->num_planes; i++) {
        if (i == lay.ref || !lay.planes[i].role)
            continue;
        const struct pl_plane *plane = &lay.planes[i].plane;
        pl_shader psh = plh_work_shader(job, &pimg[i]);
        if (gain[i] != 1.0f || !plh_append_plane_fetch(sh, psh, plane)) {
            // more than a fetch (debanded, or scaled by a real filter): run it, fetch the result
            if (gain[i] != 1.0f)
                plh_append_scale(psh, gain[i], true);
            pimg[i].comps = plane->components;
            bool merged = plh_work_texture(job, &pimg[i]) != NULL;
            if (merged) {
                psh = plh_work_shader(job, &pimg[i]);
                merged = plh_append_plane_fetch(sh, psh, plane);
            }
            if (!merged) {
                pl_dispatch_abort(rr->dp, &pimg[i].rec);
                pl_dispatch_abort(rr->dp, &sh);
                return false;
            }
        }
        pl_dispatch_abort(rr->dp, &pimg[i].rec);
    }

    job->img = (struct work_image) {
        .rec   = sh,
        .w     = pl_rect_w(lay.grid),
        .h     = pl_rect_h(lay.grid),
        .repr  = ref->repr,
        .color = image->color,
        .comps = ref->repr.alpha == PL_ALPHA_NONE ? 3 : 4,
        .rect  = {
            lay.off_x, lay.off_y,
            lay.off_x + pl_rect_w(lay.planes[lay.ref].rect),
            lay.off_y + pl_rect_h(lay.planes[lay.ref].rect),
        },
        .fail_msg = ref->fail_msg, .fail_bit = ref->fail_bit, .fail_tex = ref->fail_tex,
    };
    struct work_image *img = &job->img;

    // Frame LUT (:1920-1946). NATIVE and CONVERSION see the raw samples (bit depth fixed up),
    // CONVERSION also does the decoding; NORMALIZED sees decoded RGB.
    const enum pl_lut_type lut = rp_frame_lut_type(image, false);
    const bool raw_lut = lut == PL_LUT_NATIVE || lut == PL_LUT_CONVERSION;
    if (raw_lut) {
        plh_append_scale(sh, pl_color_repr_normalize(&img->repr), true);
        pl_shader_custom_lut(sh, image->lut, &rr->lut_state[RR_LUT_IMAGE]);
    }
    if (lut == PL_LUT_CONVERSION) {
        img->repr.sys = PL_COLOR_SYSTEM_RGB;
        img->repr.levels = PL_COLOR_LEVELS_FULL;
    } else {
        if (img->repr.sys == PL_COLOR_SYSTEM_XYZ) {
            // the XYZ matrix applies to linear light
            pl_shader_linearize(sh, &img->color);
            img->color.transfer = PL_COLOR_TRC_LINEAR;
        }
        if (img->repr.sys == PL_COLOR_SYSTEM_DOLBYVISION && sh->pass.s.type >= PLH_SAMPLE_POLAR) {
            // the Dolby Vision ops live in the generic kernel only: a plane that arrives through a
            // sampler with a kernel of its own (debanded) is stored first, unrounded
            img->store_as = pl_find_named_fmt(rr->gpu, "rgba32f");
            if (!plh_work_texture(job, img))
                return false;
            sh = plh_work_shader(job, img);
        }
        pl_shader_decode_color(sh, &img->repr, params->color_adjustment);
    }
    if (lut == PL_LUT_NORMALIZED)
        pl_shader_custom_lut(sh, image->lut, &rr->lut_state[RR_LUT_IMAGE]);

    // transparent regions must not bleed into opaque ones while filtering
    pl_shader_set_alpha(sh, &img->repr, PL_ALPHA_PREMULTIPLIED);
    return !pl_shader_is_failed(sh);
}

/* ---- HDR peak measurement ------------------------------------------------------------------- */

static bool owns_workgroup_shape(const pl_shader sh)
{
    const int t = sh->pass.s.type;
    if (t == PLH_SAMPLE_POLAR || t == PLH_SAMPLE_ORTHO || t == PLH_SAMPLE_DEBAND ||
        t == PLH_SAMPLE_DEINTERLACE)
        return true;
    // ... and so do the Dolby Vision ops: one variant of the generic kernel, without the
    // measurement's workgroup state
    for (int i = 0; i < sh->pass.num_ops; i++) {
        if (sh->pass.ops[i].kind == PLH_OP_DOVI_RESHAPE || sh->pass.ops[i].kind == PLH_OP_DOVI_LMS)
            return true;
    }
    return false;
}

static void measure_peak(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    struct work_image *img = &job->img;

    const char *why_not = rp_peak_skip_reason(&job->caps, params, &job->image.color, &img->color,
                                              &job->target.color);
    if (!why_not && !job->caps.fbo[4] && !params->peak_detect_params->allow_delayed) {
        raise(rr, PL_RENDER_ERR_PEAK_DETECT, PL_LOG_WARN, "Disabling peak detection because "
              "`pl_peak_detect_params.allow_delayed` is false, but lack of FBOs forces the "
              "result to be delayed.");
        why_not = "no intermediates";
    }
    if (why_not) {
        pl_reset_detected_peak(rr->tone_map_state);
        return;
    }

    if (img->rec && job->caps.fbo[4] && owns_workgroup_shape(img->rec)) {
        // The scaler / deband kernels have their own workgroup shape, the measurement needs the
        // reference's 16x16 tiling: make the image resident and measure it with a pass that
        // only reads (the reference merges the two into one compute shader; same image, same
        // tiling, the values having gone through the intermediate's f16 rounding).
        pl_tex tex = plh_work_texture(job, img);
        bool ok = tex != NULL;
        if (ok) {
            // If the tone mapper is going to ask for a contrast-recovery feature map of this very
            // image, the pass that reads it for the measurement extracts the features as well: ONE
            // read of the intermediate instead of two (66 MB each at 4K). The reference runs them
            // as two passes (renderer.c:2089-2154 after :1964-2087); the values are the same, the
            // measurement sees the colours before the FEATURES op replaces them.
            int mw, mh;
            pl_tex full = NULL;
            const char *off = getenv("PL_HIP_FUSED_FEATURES");
            if (!(off && off[0] == '0') && !job->features_full &&
                rp_wants_feature_map(&job->caps, params, &img->color, &job->target.color,
                                     abs(pl_rect_w(job->geo.dst)), abs(pl_rect_h(job->geo.dst)), &mw, &mh))
                full = borrow_fbo(job, img->w, img->h, NULL, 1);
            pl_shader probe = pl_dispatch_begin(rr->dp);
            ok = pl_shader_sample_direct(probe, pl_sample_src( .tex = tex )) &&
                 pl_shader_detect_peak(probe, img->color, &rr->tone_map_state,
                                       params->peak_detect_params);
            if (ok && full) {
                pl_shader_extract_features(probe, img->color);
                ok = pl_dispatch_finish(rr->dp, pl_dispatch_params( .shader = &probe, .target = full ));
                if (ok) {
                    job->features_full = full;
                    job->features_src = tex;
                    job->features_color = img->color;
                }
            } else if (ok) {
                ok = pl_dispatch_compute(rr->dp, pl_dispatch_compute_params(
                    .shader = &probe, .width = tex->params.w, .height = tex->params.h,
                ));
            } else {
                pl_dispatch_abort(rr->dp, &probe);
            }
        }
        if (!ok) {
            raise(rr, PL_RENDER_ERR_PEAK_DETECT, PL_LOG_WARN,
                  "Failed measuring the HDR peak.. disabling");
            pl_reset_detected_peak(rr->tone_map_state);
        }
        job->peak_pending = false;  // already complete in stream order
        return;
    }

    if (!pl_shader_detect_peak(plh_work_shader(job, img), img->color, &rr->tone_map_state,
                               params->peak_detect_params)) {
        raise(rr, PL_RENDER_ERR_PEAK_DETECT, PL_LOG_WARN,
              "Failed creating HDR peak detection shader.. disabling");
        pl_reset_detected_peak(rr->tone_map_state);
        return;
    }
    // the pass carrying the measurement has to run before this frame's tone mapping
    job->peak_pending = !params->peak_detect_params->allow_delayed;
}

/* ---- stage 2: scale ------------------------------------------------------------------------- */

bool plh_stage_scale(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    struct work_image *img = &job->img;
    if (!job->caps.fbo[img->comps])
        return true;    // no intermediates: the output pass samples the source directly

    struct pl_sample_src req = {
        .components = img->comps,
        .new_w      = abs(pl_rect_w(job->geo.dst)),
        .new_h      = abs(pl_rect_h(job->geo.dst)),
        .rect       = img->rect,
    };
    // a recording with a fixed output size (deband, ...) cannot be resampled in place
    int fw, fh;
    const bool fixed = img->rec && pl_shader_output_size(img->rec, &fw, &fh) &&
                       (fw != req.new_w || fh != req.new_h);
    const struct rp_scale_stage st = rp_plan_scale(&job->caps, params, &req, NULL, &img->color,
                                                   img->comps, fixed);
    const pl_rect2df full = { .x1 = st.out_w, .y1 = st.out_h };

    if (st.peak_before)
        measure_peak(job);

    if (st.defer) {
        img->w = st.out_w;
        img->h = st.out_h;
        img->rect = full;
    } else if (!st.skip) {
        if (st.restore_transfer) {
            // decoding left linear light (XYZ), but scaling is to happen non-linearly
            img->color.transfer = job->image.color.transfer;
            if (img->color.transfer == PL_COLOR_TRC_LINEAR)
                img->color.transfer = PL_COLOR_TRC_GAMMA22;     // arbitrary, as the reference
            pl_shader_delinearize(plh_work_shader(job, img), &img->color);
        }
        if (st.linear || st.sigmoid) {
            pl_shader_linearize(plh_work_shader(job, img), &img->color);
            img->color.transfer = PL_COLOR_TRC_LINEAR;
        }
        if (st.sigmoid)
            pl_shader_sigmoidize(plh_work_shader(job, img), params->sigmoid_params);

        // pass boundary: what is recorded so far either fuses into the polar kernel or lands
        // in an intermediate image the scaler reads
        pl_shader sh = pl_dispatch_begin(rr->dp);
        if (img->rec && try_fused_polar(job, sh, &req, img->rec, img->w, img->h)) {
            pl_dispatch_abort(rr->dp, &img->rec);
        } else {
            req.tex = plh_work_texture(job, img);
            if (!req.tex) {
                pl_dispatch_abort(rr->dp, &sh);
                return false;
            }
            run_scaler(job, sh, &rr->scale_main, RP_USE_MAIN, &req);
        }
        job->peak_pending = false;  // whatever rode on the previous pass has run
        img->tex = NULL;
        img->copy_of = NULL;
        img->rec = sh;
        img->w = st.out_w;
        img->h = st.out_h;
        img->rect = full;
        if (st.sigmoid)
            pl_shader_unsigmoidize(sh, params->sigmoid_params);
    }

    if (!st.peak_before)
        measure_peak(job);
    return true;
}

/* ---- stage 3: colours ----------------------------------------------------------------------- */

// What pl_shader_extract_features depends on: the primaries (RGB -> LMS) and the linearisation --
// which reads the HDR metadata for every curve but the linear one and PQ (black / peak scaling of
// the SDR curves, HLG's system gamma: plh_fill_linearize, colorspace.c:640-719).
static bool features_color_same(const struct pl_color_space *a, const struct pl_color_space *b)
{
    if (a->primaries != b->primaries || a->transfer != b->transfer)
        return false;
    if (a->transfer == PL_COLOR_TRC_LINEAR || a->transfer == PL_COLOR_TRC_PQ)
        return true;
    return pl_hdr_metadata_equal(&a->hdr, &b->hdr);
}

// Low-resolution luminance of the image for the tone mapper's contrast recovery: I of IPT at
// full size, then low-passed (bicubic, mirrored edges) to 1/smoothness of the output size.
static pl_tex make_feature_map(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    struct work_image *img = &job->img;
    int mw, mh;
    if (!rp_wants_feature_map(&job->caps, job->params, &img->color, &job->target.color,
                              abs(pl_rect_w(job->geo.dst)), abs(pl_rect_h(job->geo.dst)),
                              &mw, &mh))
        return NULL;
    if (!plh_work_texture(job, img))
        return NULL;

    // (the full-size plane may exist already: the measuring pass of this image writes it when it
    // can, measure_peak)
    // -- valid only while the image is still that texture: anything recorded on it since (cone
    // distortion, an alpha conversion) made plh_work_texture above produce another one
    // ... and while it is still described by the colour space the features were extracted with:
    // the reference extracts them after hdr_update_peak (renderer.c:2089-2154 behind :1964-2087),
    // and the linearisation of an HLG / BT.1886 image depends on the metadata detected there
    const bool have = job->features_full && job->features_src == img->tex &&
                      features_color_same(&job->features_color, &img->color);
    pl_tex full = have ? job->features_full : borrow_fbo(job, img->w, img->h, NULL, 1);
    pl_tex small = borrow_fbo(job, mw, mh, NULL, 1);
    bool ok = full && small;
    if (ok && !have) {
        pl_shader sh = pl_dispatch_begin(rr->dp);
        pl_shader_sample_direct(sh, pl_sample_src( .tex = img->tex ));
        pl_shader_extract_features(sh, img->color);
        ok = pl_dispatch_finish(rr->dp, pl_dispatch_params( .shader = &sh, .target = full ));
    }
    if (ok) {
        const struct pl_sample_src req = {
            .tex = full, .rect = img->rect, .address_mode = PL_TEX_ADDRESS_MIRROR,
            .components = 1, .new_w = mw, .new_h = mh,
        };
        if (!try_fused_lowpass(job, &req, small)) {
            pl_shader sh = pl_dispatch_begin(rr->dp);
            run_scaler(job, sh, &rr->scale_contrast, RP_USE_LOWPASS, &req);
            ok = pl_dispatch_finish(rr->dp, pl_dispatch_params( .shader = &sh, .target = small ));
        }
    }
    if (!ok) {
        raise(rr, PL_RENDER_ERR_CONTRAST_RECOVERY, PL_LOG_ERR,
              "Failed extracting luma for contrast recovery, disabling");
        return NULL;
    }
    return small;
}

// pl_render_params.lut between the image's and the target's colour space (:2199-2247).
// Returns false if the LUT replaces the regular conversion.
static bool apply_params_lut(struct frame_job *job, pl_shader sh, bool *prelinearized)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    struct work_image *img = &job->img;
    const struct pl_custom_lut *lut = params->lut;
    struct pl_color_space in = lut->color_in, out = lut->color_out;
    const bool normalized = params->lut_type == PL_LUT_NORMALIZED;
    const bool conversion = params->lut_type == PL_LUT_CONVERSION;

    if (normalized && !*prelinearized) {
        pl_shader_linearize(sh, &img->color);   // this placement wants linear input
        img->color.transfer = PL_COLOR_TRC_LINEAR;
        *prelinearized = true;
    }
    // whatever the LUT leaves unspecified is taken from the image (as it is at this point)
    const struct pl_color_space *fill = normalized ? &img->color : &job->image.color;
    pl_color_space_merge(&in, fill);
    if (!conversion)
        pl_color_space_merge(&out, fill);

    pl_shader_color_map_ex(sh, params->color_map_params, pl_color_map_args(
        .src = job->image.color, .dst = in, .prelinearized = *prelinearized));
    if (normalized)
        plh_append_scale(sh, 1.0f / pl_color_transfer_nominal_peak(in.transfer), false);
    pl_shader_custom_lut(sh, lut, &rr->lut_state[RR_LUT_PARAMS]);
    if (normalized)
        plh_append_scale(sh, pl_color_transfer_nominal_peak(out.transfer), false);
    if (conversion)
        return false;
    pl_shader_color_map_ex(sh, params->color_map_params, pl_color_map_args(
        .src = out, .dst = img->color));
    return true;
}

void plh_stage_colors(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    const struct pl_frame *image = &job->image, *target = &job->target;
    struct work_image *img = &job->img;

    // the tricubic 3D-LUT lookup exists in one variant of the generic pass kernel only: keep
    // the colour conversion out of a pending scaler pass
    const struct pl_color_map_params *cm = params->color_map_params;
    if (cm && cm->lut3d_tricubic && img->rec && !plh_work_texture(job, img)) {
        RR_LOG(rr, PL_LOG_ERR, "Failed flushing the image ahead of the tricubic colour map");
        return;
    }
    pl_shader sh = plh_work_shader(job, img);

    bool prelinearized = false;
    if (img->color.transfer == PL_COLOR_TRC_LINEAR) {
        if (img->repr.alpha == PL_ALPHA_PREMULTIPLIED) {
            // scaled in linear light *with premultiplied alpha*; the mapping wants independent
            // alpha: return to the encoded signal first, so that the alpha division happens
            // where it was multiplied in
            img->color.transfer = image->color.transfer;
            pl_shader_delinearize(sh, &img->color);
        } else {
            prelinearized = true;
        }
    } else if (image->color.transfer == PL_COLOR_TRC_LINEAR) {
        pl_shader_linearize(sh, &img->color);
        img->color.transfer = PL_COLOR_TRC_LINEAR;
    }
    pl_shader_set_alpha(sh, &img->repr, PL_ALPHA_INDEPENDENT);

    if (params->cone_params)
        pl_shader_cone_distort(sh, img->color, params->cone_params);

    const bool convert = !params->lut || apply_params_lut(job, sh, &prelinearized);
    if (convert) {
        // a measurement made by this frame has to be on the device before it is read
        if (job->peak_pending && !plh_work_texture(job, img))
            return;
        pl_tex features = make_feature_map(job);
        sh = plh_work_shader(job, img);
        pl_shader_color_map_ex(sh, params->color_map_params, pl_color_map_args(
            .src           = image->color,
            .dst           = target->color,
            .prelinearized = prelinearized,
            .state         = &rr->tone_map_state,
            .feature_map   = features,
        ));
    }

    // a target LUT working on RGB acts here; a NATIVE one after the encoding
    const enum pl_lut_type tl = rp_frame_lut_type(target, true);
    if (tl == PL_LUT_NORMALIZED || tl == PL_LUT_CONVERSION)
        pl_shader_custom_lut(sh, target->lut, &rr->lut_state[RR_LUT_TARGET]);
    img->color = target->color;
}

/* ---- stage 4: output ------------------------------------------------------------------------ */

// An sRGB colour (background, tiles, pl_frame_clear_*) in the colour space `csp` (:2555-2584).
// The display-referred gamma curves (BT.1886, sRGB, gamma 2.2) are taken to be the curve the
// colour was given in -- the round trip through linear light then changes nothing but the
// primaries, and black keeps the target's level; for every other target the colour is read as
// sRGB of infinite contrast, so that no black point is lifted into it.
static void background_in(const struct pl_color_space *csp, const float srgb[3], float out[3])
{
    struct pl_color_space from = pl_color_space_srgb;
    const bool gamma_like = csp->transfer == PL_COLOR_TRC_BT_1886 ||
                            csp->transfer == PL_COLOR_TRC_SRGB ||
                            csp->transfer == PL_COLOR_TRC_GAMMA22;
    if (gamma_like)
        from.transfer = csp->transfer;
    from.hdr.min_luma = gamma_like ? csp->hdr.min_luma : PL_COLOR_HDR_BLACK;

    memcpy(out, srgb, 3 * sizeof(float));
    pl_color_linearize(&from, out);
    const pl_matrix3x3 m = pl_get_color_mapping_matrix(
        pl_raw_primaries_get(from.primaries), pl_raw_primaries_get(csp->primaries),
        PL_INTENT_RELATIVE_COLORIMETRIC);
    pl_matrix3x3_apply(&m, out);
    pl_color_delinearize(csp, out);
}

// the inverse of the frame's decoding: normalised RGB -> the values its planes hold
static pl_transform3x3 frame_encoding(const struct pl_frame *frame)
{
    struct pl_color_repr repr = frame->repr;
    pl_transform3x3 enc = pl_color_repr_decode(&repr, NULL);
    pl_transform3x3_invert(&enc);
    return enc;
}

// :4172-4199
void pl_frame_clear_rgba(pl_gpu gpu, const struct pl_frame *frame, const float rgba[4])
{
    float enc[3];
    background_in(&frame->color, rgba, enc);
    const pl_transform3x3 tr = frame_encoding(frame);
    pl_transform3x3_apply(&tr, enc);

    const float cover = frame->repr.alpha == PL_ALPHA_PREMULTIPLIED ? rgba[3] : 1.0f;
    for (int i = 0; i < frame->num_planes; i++) {
        const struct pl_plane *pl = &frame->planes[i];
        float texel[4] = { 0.0f, 0.0f, 0.0f, rgba[3] };
        for (int c = 0; c < pl->components; c++) {
            const int ch = pl->component_mapping[c];
            if (ch >= 0 && ch < 3)
                texel[c] = cover * enc[ch];
        }
        pl_tex_clear(gpu, pl->texture, texel);
    }
}

// :4116-4170. The tile period of a plane follows its size relative to the reference plane, rounded
// to a whole (or whole-reciprocal) ratio; the period is an integer number of texels.
void pl_frame_clear_tiles(pl_gpu gpu, const struct pl_frame *frame,
                          const float tile_colors[2][3], int tile_size)
{
    if (!frame->num_planes || tile_size <= 0)
        return;
    const pl_transform3x3 tr = frame_encoding(frame);
    float enc[2][3];
    for (int t = 0; t < 2; t++) {
        background_in(&frame->color, tile_colors[t], enc[t]);
        pl_transform3x3_apply(&tr, enc[t]);
    }

    pl_tex ref = frame->planes[rp_reference_plane(frame)].texture;
    for (int i = 0; i < frame->num_planes; i++) {
        const struct pl_plane *pl = &frame->planes[i];
        float tiles[2][4] = { { 0.0f, 0.0f, 0.0f, 1.0f }, { 0.0f, 0.0f, 0.0f, 1.0f } };
        for (int c = 0; c < pl->components; c++) {
            const int ch = pl->component_mapping[c];
            if (ch >= 0 && ch < 3) {
                tiles[0][c] = enc[0][ch];
                tiles[1][c] = enc[1][ch];
            }
        }
        const float rx = (float) pl->texture->params.w / ref->params.w,
                    ry = (float) pl->texture->params.h / ref->params.h;
        const float sx = rx >= 1 ? roundf(rx) : 1.0 / roundf(1.0 / rx),
                    sy = ry >= 1 ? roundf(ry) : 1.0 / roundf(1.0 / ry);
        const int period_x = tile_size * sx, period_y = tile_size * sy;
        if (period_x <= 0 || period_y <= 0)
            continue;   // (a plane subsampled beyond the tile size has no tiles to show)
        if (!pl->texture->params.blit_dst && !pl->texture->params.renderable) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_frame_clear_tiles: plane %d is neither renderable nor blit_dst", i);
            continue;
        }
        plh_tex_clear_tiles(gpu, pl->texture, tiles[0], tiles[1],
                            (float) (1.0 / period_x), (float) (1.0 / period_y));
    }
}

// the border of a cropped target: every plane filled before the image is drawn into its rect
// (:2941-2964: colour -> pl_frame_clear_rgba, tiles -> pl_frame_clear_tiles)
static void clear_planes(struct frame_job *job, enum pl_clear_mode mode)
{
    const struct pl_render_params *params = job->params;
    const struct pl_frame *target = &job->target;
    if (mode == PL_CLEAR_TILES) {
        static const float unset[2][3] = {{0}};
        const bool custom = memcmp(params->tile_colors, unset, sizeof(unset)) != 0;
        pl_frame_clear_tiles(job->rr->gpu, target,
                             custom ? params->tile_colors : pl_render_default_params.tile_colors,
                             PL_DEF(params->tile_size, pl_render_default_params.tile_size));
        return;
    }
    const float rgba[4] = {
        params->background_color[0], params->background_color[1], params->background_color[2],
        1.0 - params->background_transparency,
    };
    pl_frame_clear_rgba(job->rr->gpu, target, rgba);
}

// color = (0, 0, 0, 1) with color[c] = previous[mapping[c]] (reference swizzle_color :791-808)
static void append_swizzle(pl_shader sh, int comps, const int mapping[4], bool force_alpha)
{
    bool identity = comps == 4;
    uint32_t from = 0;
    for (int c = 0; c < 4; c++) {
        const int m = c < comps ? mapping[c] : -1;
        from |= (uint32_t) (m < 0 ? 0xff : m) << (8 * c);
        identity &= c >= comps || m == c;
    }
    if (identity)
        return;
    struct plh_op *op = sh_op(sh, PLH_OP_SWIZZLE);
    if (!op)
        return;
    op->i0 = from;
    op->i1 = comps;
    op->i2 = force_alpha;
    sh_listf(sh, "swizzle(comps=%d, map=0x%08x%s)\n", comps, (unsigned) from,
             force_alpha ? ", keep alpha" : "");
}

// Run `*sh` into an intermediate, diffuse the quantisation error over it (one workgroup, the
// whole image's error ring in LDS), and continue from the result (:2282-2344)
static bool run_error_diffusion(struct frame_job *job, pl_shader *sh, int depth, int comps,
                                int w, int h)
{
    pl_renderer rr = job->rr;
    pl_fmt fmt = job->caps.fbo[comps];
    if (!fmt || !(fmt->caps & PL_FMT_CAP_STORABLE)) {
        raise(rr, PL_RENDER_ERR_ERROR_DIFFUSION, PL_LOG_ERR,
              "Error diffusion requires storable FBOs.. disabling!");
        return false;
    }
    struct pl_error_diffusion_params ed = {
        .input_tex  = borrow_fbo(job, w, h, fmt, comps),
        .output_tex = borrow_fbo(job, w, h, fmt, comps),
        .new_depth  = depth,
        .kernel     = job->params->error_diffusion,
    };
    pl_shader diffuse = ed.input_tex && ed.output_tex ? pl_dispatch_begin(rr->dp) : NULL;
    if (diffuse && !pl_shader_error_diffusion(diffuse, &ed))
        pl_dispatch_abort(rr->dp, &diffuse);
    if (!diffuse) {
        rr->errors |= PL_RENDER_ERR_ERROR_DIFFUSION;
        return false;
    }

    bool ok = pl_dispatch_finish(rr->dp, pl_dispatch_params( .shader = sh, .target = ed.input_tex ));
    if (ok) {
        ok = pl_dispatch_compute(rr->dp, pl_dispatch_compute_params(
            .shader = &diffuse, .dispatch_size = {1, 1, 1},
        ));
    } else {
        pl_dispatch_abort(rr->dp, &diffuse);
    }
    *sh = pl_dispatch_begin(rr->dp);
    pl_shader_sample_direct(*sh, pl_sample_src( .tex = ok ? ed.output_tex : ed.input_tex ));
    return ok;
}

// pl_render_params.distort_params (:2655-2701): the finished image through an affine map, onto a
// target rect grown (within the target) to hold the result
static bool distort_image(struct frame_job *job, struct work_image *img, struct rp_geometry *geo)
{
    pl_renderer rr = job->rr;
    const struct pl_frame *target = &job->target;
    struct pl_distort_params dpars = *job->params->distort_params;
    if (dpars.alpha_mode) {
        pl_shader_set_alpha(plh_work_shader(job, img), &img->repr, dpars.alpha_mode);
        img->repr.alpha = dpars.alpha_mode;
        img->comps = 4;
    }
    pl_tex tex = plh_work_texture(job, img);
    if (!tex)
        return false;

    // the bounding box of the transformed image, in units of the target rect
    const float ar = pl_rect2df_aspect(&target->crop);
    const float sx = fminf(ar, 1.0f), sy = fminf(1.0f / ar, 1.0f);
    const pl_rect2df unit = { .x0 = -sx, .x1 = sx, .y0 = -sy, .y1 = sy };
    const pl_rect2df bb = pl_transform2x2_bounds(&dpars.transform, &unit);
    pl_rect2df tmp = target->crop;
    pl_rect2df_stretch(&tmp, pl_rect_w(bb) / (2 * sx), pl_rect_h(bb) / (2 * sy));
    const float tmp_w = pl_rect_w(tmp), tmp_h = pl_rect_h(tmp);
    pl_tex ref = target->planes[rp_reference_plane(target)].texture;
    int canvas_w = ref->params.w, canvas_h = ref->params.h;
    if (geo->rotation % PL_ROTATION_180 == PL_ROTATION_90) {
        const int t = canvas_w;
        canvas_w = canvas_h;
        canvas_h = t;
    }
    tmp.x0 = PL_CLAMP(tmp.x0, 0.0f, canvas_w);
    tmp.x1 = PL_CLAMP(tmp.x1, 0.0f, canvas_w);
    tmp.y0 = PL_CLAMP(tmp.y0, 0.0f, canvas_h);
    tmp.y1 = PL_CLAMP(tmp.y1, 0.0f, canvas_h);
    if (dpars.constrain) {
        const float rx = pl_rect_w(tmp) / tmp_w, ry = pl_rect_h(tmp) / tmp_h;
        pl_rect2df_stretch(&tmp, fminf(ry / rx, 1.0f), fminf(rx / ry, 1.0f));
    }
    geo->dstf = (pl_rect2df) { roundf(tmp.x0), roundf(tmp.y0), roundf(tmp.x1), roundf(tmp.y1) };
    geo->dst = (pl_rect2d) { geo->dstf.x0, geo->dstf.y0, geo->dstf.x1, geo->dstf.y1 };
    if (!pl_rect_w(geo->dst) || !pl_rect_h(geo->dst)) {
        RR_LOG(rr, PL_LOG_ERR, "Distortion leaves nothing of the image inside the target");
        return false;
    }

    dpars.unscaled = true;
    img->w = abs(pl_rect_w(geo->dst));
    img->h = abs(pl_rect_h(geo->dst));
    img->rect = (pl_rect2df) { 0, 0, img->w, img->h };
    img->tex = NULL;
    img->rec = pl_dispatch_begin(rr->dp);
    pl_shader_distort(img->rec, tex, img->w, img->h, &dpars);
    return true;
}

bool plh_stage_output(struct frame_job *job)
{
    pl_renderer rr = job->rr;
    const struct pl_render_params *params = job->params;
    const struct pl_frame *target = &job->target;
    struct work_image *img = &job->img;
    struct rp_geometry geo = job->geo;
    if (params->distort_params && !distort_image(job, img, &geo))
        return false;
    pl_shader sh = plh_work_shader(job, img);

    struct rp_output_stage out;
    rp_plan_output(params, target, &geo, img->comps, img->repr.alpha, &out);

    if (out.premultiply)
        pl_shader_set_alpha(sh, &img->repr, PL_ALPHA_PREMULTIPLIED);
    if (out.blend && out.background == PL_CLEAR_TILES) {
        struct plh_op *op = sh_op(sh, PLH_OP_BLEND_TILES);
        if (!op)
            return false;
        static const float unset[2][3] = {{0}};
        const float (*tc)[3] = memcmp(params->tile_colors, unset, sizeof(unset))
                             ? params->tile_colors : pl_render_default_params.tile_colors;
        background_in(&target->color, tc[0], op->f);
        background_in(&target->color, tc[1], op->f + 4);
        op->f[8] = 1.0 / PL_DEF(params->tile_size, pl_render_default_params.tile_size);
        sh_listf(sh, "blend_tiles(%g %g %g | %g %g %g, 1/%g)\n", op->f[0], op->f[1], op->f[2],
                 op->f[4], op->f[5], op->f[6], 1.0 / op->f[8]);
    } else if (out.blend) {
        struct plh_op *op = sh_op(sh, PLH_OP_BLEND_BG);
        if (!op)
            return false;
        background_in(&target->color, params->background_color, op->f);
        op->f[3] = 1.0 - params->background_transparency;
        sh_listf(sh, "blend_background(%g %g %g %g)\n", op->f[0], op->f[1], op->f[2], op->f[3]);
    }
    if (out.drop_alpha) {
        img->repr.alpha = PL_ALPHA_NONE;
        img->comps = 3;
    }
    if (out.unpremultiply)
        pl_shader_set_alpha(sh, &img->repr, PL_ALPHA_INDEPENDENT);

    // The integer scale of the encoding is applied last, separately, so that an intermediate
    // (error diffusion) still holds normalised values
    if (out.encode) {
        pl_shader_encode_color(sh, &out.repr);
        if (out.delinearize_xyz) {
            img->color.transfer = PL_COLOR_TRC_ST428;
            pl_shader_delinearize(sh, &img->color);
        }
    }
    if (out.target_lut == PL_LUT_NATIVE)
        pl_shader_custom_lut(sh, target->lut, &rr->lut_state[RR_LUT_TARGET]);

    if (out.transposed) {
        const int t = img->w;
        img->w = img->h;
        img->h = t;
        sh->transpose = true;
    }
    if (out.clear_border)
        clear_planes(job, out.border);

    // a planar target samples the finished image once per plane
    pl_tex finished = NULL;
    if (out.num_planes > 1) {
        img->rec = sh;
        finished = plh_work_texture(job, img);
        sh = NULL;
        if (!finished) {
            RR_LOG(rr, PL_LOG_ERR, "Output requires multiple planes, but FBOs are unavailable.");
            return false;
        }
    } else {
        img->rec = NULL;
    }

    bool ok = true;
    for (int i = 0; i < out.num_planes && ok; i++) {
        const struct pl_plane *plane = &target->planes[i];
        const struct rp_output_plane *op = &out.planes[i];
        const int pw = pl_rect_w(op->covered), ph = pl_rect_h(op->covered);

        if (finished) {
            struct pl_sample_src req = op->request;
            req.tex = finished;
            sh = pl_dispatch_begin(rr->dp);
            run_scaler(job, sh, &rr->scale_out[i], RP_USE_PLANE, &req);
        }

        int dithered = 0;
        switch (rp_pick_dither(&job->caps, params, out.dither_depth, ph)) {
        case RP_DITHER_ERROR_DIFFUSION:
            if (run_error_diffusion(job, &sh, out.dither_depth, plane->components, pw, ph)) {
                dithered = out.dither_depth;
                break;
            }
            if (!params->dither_params)
                break;
            __attribute__((fallthrough));   // ordered dither instead
        case RP_DITHER_ORDERED: {
            struct pl_dither_params dp = *params->dither_params;
            if (!params->disable_dither_gamma_correction)
                dp.transfer = target->color.transfer;
            pl_shader_dither(sh, out.dither_depth, &rr->dither_state, &dp);
            dithered = out.dither_depth;
            break;
        }
        case RP_DITHER_NONE:
            break;
        }
        if (dithered != rr->last_dither_depth) {
            if (dithered)
                RR_LOG(rr, PL_LOG_INFO, "Dithering to %d bit depth", dithered);
            else
                RR_LOG(rr, PL_LOG_INFO, "Dithering disabled");
            rr->last_dither_depth = dithered;
        }

        // (a blended output keeps its alpha as it is: the blend unit's, not a stored value, :2911-2917)
        if (!plh_append_scale(sh, 1.0f / out.scale, !params->blend_params)) {
            pl_dispatch_abort(rr->dp, &sh);
            return false;
        }
        append_swizzle(sh, plane->components, plane->component_mapping, params->blend_params);
        ok = pl_dispatch_finish(rr->dp, pl_dispatch_params(
            .shader = &sh,
            .target = plane->texture,
            .rect   = op->store,
            .blend_params = params->blend_params,
        ));
        if (!ok)
            break;

        // the frames' overlays over the finished plane (:2948-2958); a mixed frame's own
        // overlays went onto its cached intermediate (render_mix.c)
        const pl_transform2x2 shift = plh_plane_shift(plane,
            target->planes[rp_reference_plane(target)].texture);
        if (job->info.stage != PL_RENDER_STAGE_BLEND) {
            plh_draw_overlays(job, plane->texture, plane->components, plane->component_mapping,
                              job->image.overlays, job->image.num_overlays, true,
                              target->color, target->repr, &shift);
        }
        plh_draw_overlays(job, plane->texture, plane->components, plane->component_mapping,
                          target->overlays, target->num_overlays,
                          job->info.stage != PL_RENDER_STAGE_BLEND, target->color, target->repr,
                          &shift);
    }
    *img = (struct work_image) {0};
    return ok;
}

/* ---- pl_render_image ------------------------------------------------------------------------ */

// clear the target and draw its overlays: what remains of a render without an image
static bool render_nothing(pl_renderer rr, const struct pl_frame *ptarget,
                           const struct pl_render_params *params)
{
    struct frame_job job = { .rr = rr, .params = params, .target = *ptarget };
    if (job.target.acquire) {
        if (!job.target.acquire(rr->gpu, &job.target))
            return false;
        job.target_acquired = true;
    }
    const char *bad = rp_frame_problem(&job.target, true);
    if (bad) {
        RR_LOG(rr, PL_LOG_ERR, "Target frame: %s", bad);
    } else {
        rp_complete_frame(&job.target);
        pl_color_space_infer(&job.target.color);
        // (draw_empty_overlays -> clear_target, :2491-2553: the border mode decides; a blurred
        // border has nothing to blur without an image)
        enum pl_clear_mode mode = params->skip_target_clearing ? PL_CLEAR_SKIP : params->border;
        if (mode == PL_CLEAR_BLUR)
            mode = PL_CLEAR_COLOR;
        if (mode != PL_CLEAR_SKIP)
            clear_planes(&job, mode);
        // (:3397-3424) the target's overlays over the cleared planes
        pl_tex ref = job.target.planes[rp_reference_plane(&job.target)].texture;
        job.geo = rp_fit_target(job.target.crop, ref->params.w, ref->params.h,
                                job.target.rotation);
        job.target.crop = job.geo.dstf;
        for (int i = 0; i < job.target.num_planes; i++) {
            const struct pl_plane *plane = &job.target.planes[i];
            const pl_transform2x2 shift = plh_plane_shift(plane, ref);
            plh_draw_overlays(&job, plane->texture, plane->components, plane->component_mapping,
                              job.target.overlays, job.target.num_overlays, false,
                              job.target.color, job.target.repr, &shift);
        }
    }
    plh_job_end(&job);
    return !bad;
}

bool pl_render_image(pl_renderer rr, const struct pl_frame *pimage, const struct pl_frame *ptarget,
                     const struct pl_render_params *params)
{
    params = params ? params : &pl_render_default_params;
    if (!ptarget) {
        RR_LOG(rr, PL_LOG_ERR, "pl_render_image: a target is required");
        return false;
    }
    if (!plh_params_supported(rr, params))
        return false;
    if (!pimage)
        return render_nothing(rr, ptarget, params);

    // pre-v6.254 spelling of pl_peak_detect_params.allow_delayed
    struct pl_render_params local;
    struct pl_peak_detect_params peak;
    if (params->allow_delayed_peak_detect && params->peak_detect_params &&
        !params->peak_detect_params->allow_delayed)
    {
        local = *params;
        peak = *params->peak_detect_params;
        peak.allow_delayed = true;
        local.peak_detect_params = &peak;
        params = &local;
    }

    struct frame_job job = {
        .rr = rr, .params = params, .image = *pimage, .target = *ptarget,
    };
    if (!plh_job_begin(&job, true))
        return false;
    if (!pl_rect_w(job.geo.dst) || !pl_rect_h(job.geo.dst)) {
        plh_job_end(&job);
        return true;    // nothing visible
    }

    plh_job_watch_passes(&job);
    bool ok = plh_stage_read(&job) && plh_stage_scale(&job);
    if (ok) {
        plh_stage_colors(&job);
        ok = (job.img.rec || job.img.tex) && plh_stage_output(&job);
    }
    if (!ok)
        RR_LOG(rr, PL_LOG_ERR, "Failed rendering image!");
    plh_job_end(&job);
    return ok;
}

/* Test hooks (tests/): record `color *= s` the way the output stage does -- there is no public
 * pl_shader_* entry point for it -- and print the plan of a frame without a GPU. */
PL_API void plh_test_op_scale(pl_shader sh, float s);
void plh_test_op_scale(pl_shader sh, float s)
{
    struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
    if (op)
        op->f[0] = op->f[1] = op->f[2] = op->f[3] = s;
}

// Plan a frame from descriptions alone (the textures only need valid `params`) and print the
// decisions. `fbos`: whether intermediate images (rgba16hf) are available.
PL_API int plh_test_fuse_into_polar(int dir, int pending_lite, float antiring, int force);
int plh_test_fuse_into_polar(int dir, int pending_lite, float antiring, int force)
{
    return rp_fuse_into_polar((enum rp_direction) dir, pending_lite != 0, antiring, force);
}

PL_API pl_fmt plh_test_format(const char *name);
PL_API size_t plh_test_plan(const struct pl_frame *image, const struct pl_frame *target,
                            const struct pl_render_params *params, bool fbos,
                            size_t max_shmem, char *out, size_t out_size);
size_t plh_test_plan(const struct pl_frame *image, const struct pl_frame *target,
                     const struct pl_render_params *params, bool fbos, size_t max_shmem,
                     char *out, size_t out_size)
{
    struct rp_caps caps = { .max_shmem = max_shmem };
    if (fbos) {
        caps.fbo[4] = caps.fbo[3] = plh_test_format("rgba16hf");
        caps.fbo[2] = plh_test_format("rg16hf");
        caps.fbo[1] = plh_test_format("r16hf");
    }
    struct rp_summary sum;
    rp_summarise(&caps, image, target, params ? params : &pl_render_default_params, &sum);
    const size_t len = strlen(sum.text);
    if (out && out_size) {
        const size_t n = len < out_size - 1 ? len : out_size - 1;
        memcpy(out, sum.text, n);
        out[n] = '\0';
    }
    return len;
}
