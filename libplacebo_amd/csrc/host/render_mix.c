/*
 * libplacebo-hip — pl_render_image_mix: temporal frame mixing on top of the single-frame
 * stages of renderer.c (behaviour of the reference's src/renderer.c:3477-4080).
 *
 * Every source frame that contributes to the vsync being drawn is rendered once -- up to and
 * including the colour conversion, at the output size, in the target's colour space -- into a
 * texture that is cached by the frame's signature. A vsync then costs one blending pass over
 * the cached frames (in linear light, premultiplied alpha) followed by the regular output
 * stage. Structure here: weights() decides who contributes, the cache_* functions own the
 * texture cache, blend_frames() records the one pass.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "renderer_priv.h"
#include "cache_priv.h"

const struct pl_frame *pl_frame_mix_current(const struct pl_frame_mix *mix)
{
    // the last frame whose presentation time has come (zero-order hold)
    int at = -1;
    while (at + 1 < mix->num_frames && mix->timestamps[at + 1] <= 0.0f)
        at++;
    return at < 0 ? NULL : mix->frames[at];
}

const struct pl_frame *pl_frame_mix_nearest(const struct pl_frame_mix *mix)
{
    if (!mix->num_frames)
        return NULL;
    // sorted timestamps: the distance to zero falls, then rises
    int at = 0;
    while (at + 1 < mix->num_frames &&
           fabsf(mix->timestamps[at + 1]) < fabsf(mix->timestamps[at]))
        at++;
    return mix->frames[at];
}

void pl_frames_infer_mix(pl_renderer rr, const struct pl_frame_mix *mix, struct pl_frame *target,
                         struct pl_frame *out_ref)
{
    const struct pl_frame *nearest = pl_frame_mix_nearest(mix);
    struct pl_frame ref = {0};
    if (nearest) {
        ref = *nearest;
        pl_frames_infer(rr, &ref, target);
    }
    if (out_ref)
        *out_ref = ref;
}

/* ---- what a cached frame depends on ---------------------------------------------------------- */

static void mix_in(uint64_t *acc, const void *data, size_t size)
{
    plh_hash_merge(acc, plh_mem_hash(data, size));
}

#define MIX_VALUE(acc, v) mix_in(acc, &(v), sizeof(v))

static void mix_filter(uint64_t *acc, const struct pl_filter_config *f)
{
    const uint8_t present = !!f;
    MIX_VALUE(acc, present);
    if (!f)
        return;
    // the two function tables are static objects: their addresses identify them
    MIX_VALUE(acc, f->kernel);
    MIX_VALUE(acc, f->window);
    MIX_VALUE(acc, f->radius);
    mix_in(acc, f->params, sizeof(f->params));
    mix_in(acc, f->wparams, sizeof(f->wparams));
    MIX_VALUE(acc, f->clamp);
    MIX_VALUE(acc, f->blur);
    MIX_VALUE(acc, f->taper);
    MIX_VALUE(acc, f->polar);
    MIX_VALUE(acc, f->antiring);
}

// Digest of everything in pl_render_params that changes how a frame rendered for mixing looks.
// Field by field: padding bytes and callbacks never enter, a LUT enters by its signature.
static uint64_t digest_params(const struct pl_render_params *p)
{
    uint64_t acc = 0x9ae16a3b2f90404full;
    mix_filter(&acc, p->upscaler);
    mix_filter(&acc, p->downscaler);
    mix_filter(&acc, p->plane_upscaler);
    mix_filter(&acc, p->plane_downscaler);
    MIX_VALUE(&acc, p->antiringing_strength);

#define MIX_POINTEE(ptr, ...)                                       \
    do {                                                            \
        const uint8_t present = !!(ptr);                            \
        MIX_VALUE(&acc, present);                                   \
        if (ptr) { __VA_ARGS__ }                                    \
    } while (0)

    MIX_POINTEE(p->deband_params,
        MIX_VALUE(&acc, p->deband_params->iterations);
        MIX_VALUE(&acc, p->deband_params->threshold);
        MIX_VALUE(&acc, p->deband_params->radius);
        MIX_VALUE(&acc, p->deband_params->grain);
        mix_in(&acc, p->deband_params->grain_neutral, sizeof(p->deband_params->grain_neutral)););
    MIX_POINTEE(p->sigmoid_params,
        MIX_VALUE(&acc, p->sigmoid_params->center);
        MIX_VALUE(&acc, p->sigmoid_params->slope););
    MIX_POINTEE(p->color_adjustment,
        const struct pl_color_adjustment *a = p->color_adjustment;
        const float v[6] = { a->brightness, a->contrast, a->saturation, a->hue, a->gamma,
                             a->temperature };
        mix_in(&acc, v, sizeof(v)););
    MIX_POINTEE(p->peak_detect_params,
        const struct pl_peak_detect_params *d = p->peak_detect_params;
        const float v[6] = { d->smoothing_period, d->scene_threshold_low, d->scene_threshold_high,
                             d->percentile, d->black_cutoff, d->minimum_peak };
        mix_in(&acc, v, sizeof(v));
        MIX_VALUE(&acc, d->allow_delayed););
    MIX_POINTEE(p->color_map_params,
        const struct pl_color_map_params *c = p->color_map_params;
        MIX_VALUE(&acc, c->gamut_mapping);
        mix_in(&acc, &c->gamut_constants, sizeof(c->gamut_constants));
        mix_in(&acc, c->lut3d_size, sizeof(c->lut3d_size));
        MIX_VALUE(&acc, c->lut3d_tricubic);
        MIX_VALUE(&acc, c->gamut_expansion);
        MIX_VALUE(&acc, c->tone_mapping_function);
        mix_in(&acc, &c->tone_constants, sizeof(c->tone_constants));
        MIX_VALUE(&acc, c->inverse_tone_mapping);
        MIX_VALUE(&acc, c->metadata);
        MIX_VALUE(&acc, c->lut_size);
        MIX_VALUE(&acc, c->contrast_recovery);
        MIX_VALUE(&acc, c->contrast_smoothness);
        MIX_VALUE(&acc, c->force_tone_mapping_lut);
        MIX_VALUE(&acc, c->tone_mapping_param);
        MIX_VALUE(&acc, c->intent);
        MIX_VALUE(&acc, c->gamut_mode););
    MIX_POINTEE(p->cone_params,
        MIX_VALUE(&acc, p->cone_params->cones);
        MIX_VALUE(&acc, p->cone_params->strength););
    MIX_POINTEE(p->lut, MIX_VALUE(&acc, p->lut->signature););
#undef MIX_POINTEE

    MIX_VALUE(&acc, p->lut_type);
    const uint8_t flags[] = {
        p->skip_anti_aliasing, p->disable_linear_scaling, p->disable_builtin_scalers,
        p->correct_subpixel_offsets, p->disable_fbos, p->force_low_bit_depth_fbos,
    };
    mix_in(&acc, flags, sizeof(flags));
    return acc;
}

/* ---- who contributes, and how much ------------------------------------------------------------ */

struct mixer_setup {
    struct pl_filter_config filter;     // .kernel == NULL: visibility-based ("oversample")
    bool single;                        // only the nearest frame
};

// The temporal filter is stretched when source frames / vsyncs are longer than one unit, like a
// spatial kernel is widened when downscaling
static struct mixer_setup prepare_mixer(const struct pl_frame_mix *mix,
                                        const struct pl_render_params *params)
{
    struct mixer_setup ms = { .single = !params->frame_mixer || mix->num_frames == 1 };
    if (!params->frame_mixer)
        return ms;
    ms.filter = *params->frame_mixer;
    if (!ms.filter.blur)
        ms.filter.blur = 1.0f;
    if (params->skip_anti_aliasing)
        return ms;
    for (int i = 1; i < mix->num_frames; i++) {
        // the frame pair that straddles the vsync tells the current frame duration
        if (mix->timestamps[i - 1] < 0.0f && mix->timestamps[i] >= 0.0f) {
            const float span = PL_MAX(mix->timestamps[i] - mix->timestamps[i - 1],
                                      mix->vsync_duration);
            if (span > 1.0f)
                ms.filter.blur *= span;
            break;
        }
    }
    return ms;
}

// false: frame `i` takes no part at all
static bool frame_weight(const struct pl_frame_mix *mix, int i, const struct mixer_setup *ms,
                         const struct pl_frame *nearest, float *weight)
{
    if (ms->single) {
        *weight = 1.0f;
        return mix->frames[i] == nearest;
    }
    const float ts = mix->timestamps[i];
    const struct pl_filter_function *kernel = ms->filter.kernel;
    if (!kernel || kernel == &pl_filter_function_oversample) {
        // share of the vsync interval during which this frame is the one on screen
        const float until = i + 1 < mix->num_frames ? mix->timestamps[i + 1] : INFINITY;
        if (ts > mix->vsync_duration || until < 0.0f)
            return false;
        const float from = PL_MAX(ts, 0.0f), to = PL_MIN(until, mix->vsync_duration);
        *weight = (to - from) / mix->vsync_duration;
        if (kernel && *weight < kernel->params[0])
            *weight = 0.0f;     // below the oversampling threshold
        return true;
    }
    if (fabsf(ts) >= pl_filter_radius_bound(&ms->filter))
        return false;
    *weight = pl_filter_sample(&ms->filter, ts);
    return true;
}

/* ---- the cache ---------------------------------------------------------------------------------- */

static struct mix_entry *cache_find(pl_renderer rr, uint64_t signature)
{
    for (int i = 0; i < rr->num_cached; i++) {
        if (rr->cache[i].signature == signature)
            return &rr->cache[i];
    }
    return NULL;
}

static struct mix_entry *cache_add(pl_renderer rr, uint64_t signature)
{
    if (rr->num_cached == RR_MAX_CACHED_FRAMES)
        return NULL;
    struct mix_entry *e = &rr->cache[rr->num_cached++];
    *e = (struct mix_entry) { .signature = signature };
    return e;
}

// drop every entry the mix under construction did not touch; keep their textures for reuse
static void cache_collect(pl_renderer rr)
{
    int kept = 0;
    for (int i = 0; i < rr->num_cached; i++) {
        struct mix_entry *e = &rr->cache[i];
        if (!e->stale) {
            rr->cache[kept++] = *e;
            continue;
        }
        if (e->tex && rr->num_spare < RR_MAX_CACHED_FRAMES)
            rr->spare[rr->num_spare++] = e->tex;
        else
            pl_tex_destroy(rr->gpu, &e->tex);
    }
    rr->num_cached = kept;
}

static bool same_rect(pl_rect2df a, pl_rect2df b)
{
    return a.x0 == b.x0 && a.y0 == b.y0 && a.x1 == b.x1 && a.y1 == b.y1;
}

// Render `image` through read / scale / colours into the entry's texture
static bool cache_fill(struct frame_job *outer, struct mix_entry *e, const struct pl_frame *image,
                       const struct pl_frame *ptarget, int out_w, int out_h, uint64_t digest)
{
    pl_renderer rr = outer->rr;
    if (!e->tex && rr->num_spare)
        e->tex = rr->spare[--rr->num_spare];
    pl_fmt fmt = outer->caps.fbo[4];
    const struct pl_tex_params want = {
        .w = out_w, .h = out_h, .format = fmt,
        .sampleable = true, .renderable = true, .storable = true,
        .blit_dst = !!(fmt->caps & PL_FMT_CAP_BLITTABLE),
    };
    if (!pl_tex_recreate(rr->gpu, &e->tex, &want)) {
        RR_LOG(rr, PL_LOG_ERR, "Could not create intermediate texture for frame mixing.. disabling!");
        return false;
    }

    struct frame_job job = {
        .rr = rr, .params = outer->params, .image = *image, .target = *ptarget,
        .target_borrowed = true,    // `outer` holds the target
        .info.stage = PL_RENDER_STAGE_FRAME,
    };
    if (!plh_job_begin(&job, true))
        return false;
    job.target = outer->target;

    plh_job_watch_passes(&job);
    bool ok = plh_stage_read(&job) && plh_stage_scale(&job);
    if (ok) {
        plh_stage_colors(&job);
        ok = job.img.rec || job.img.tex;
    }
    if (ok) {
        pl_shader sh = plh_work_shader(&job, &job.img);
        pl_shader_set_alpha(sh, &job.img.repr, PL_ALPHA_PREMULTIPLIED);    // mixable
        ok = job.img.w == out_w && job.img.h == out_h &&
             pl_dispatch_finish(rr->dp, pl_dispatch_params( .shader = &job.img.rec, .target = e->tex ));
    }
    if (ok && image->num_overlays) {
        // the frame's own overlays go onto its intermediate (:3855-3877): target pixels -> texels
        // of the intermediate, which holds the target rect only
        const pl_rect2d dst = job.geo.dst;
        const float sx = out_w / (float) pl_rect_w(dst), sy = out_h / (float) pl_rect_h(dst);
        pl_transform2x2 shift = {
            .mat.m = {{ sx, 0 }, { 0, sy }},
            .c = { -sx * dst.x0, -sy * dst.y0 },
        };
        if (job.geo.rotation % PL_ROTATION_180 == PL_ROTATION_90) {
            shift.mat = (pl_matrix2x2) {{{ shift.mat.m[0][1], shift.mat.m[0][0] },
                                         { shift.mat.m[1][1], shift.mat.m[1][0] }}};
        }
        plh_draw_overlays(&job, e->tex, job.img.comps, NULL, image->overlays, image->num_overlays,
                          true, job.img.color, job.img.repr, &shift);
    }
    if (ok) {
        e->params_digest = digest;
        e->crop = image->crop;
        e->color = job.img.color;
        e->repr = job.img.repr;
        e->comps = job.img.comps;
    }
    plh_job_end(&job);
    return ok;
}

/* ---- the blending pass ------------------------------------------------------------------------ */

struct blend_input {
    struct mix_entry frame;
    float weight;
};

// color = sum_i weight_i * linear(frame_i), in premultiplied alpha; with one frame, that frame
static pl_shader blend_frames(struct frame_job *job, const struct blend_input *in, int n,
                              int out_w, int out_h, float wsum, int *out_comps,
                              enum pl_alpha_mode *out_alpha)
{
    pl_renderer rr = job->rr;
    const struct pl_frame *target = &job->target;
    pl_shader sh = pl_dispatch_begin(rr->dp);
    const bool blend = n > 1;
    struct pl_color_space space = target->color;
    if (blend)
        space.transfer = PL_COLOR_TRC_LINEAR;

    int comps = 0;
    bool ok = true;
    for (int i = 0; i < n && ok; i++) {
        pl_tex tex = in[i].frame.tex;
        const bool resample = (tex->params.w != out_w || tex->params.h != out_h) &&
                              (tex->params.format->caps & PL_FMT_CAP_LINEAR);
        const struct pl_sample_src src = { .tex = tex, .new_w = i ? 0 : out_w, .new_h = i ? 0 : out_h };
        if (i == 0) {
            // the first frame is the pass' own sampler
            ok = resample ? pl_shader_sample_bilinear(sh, &src) : pl_shader_sample_nearest(sh, &src);
        } else {
            // the others are fetched into it
            pl_shader fetch = pl_dispatch_begin(rr->dp);
            ok = resample ? pl_shader_sample_bilinear(fetch, &src)
                          : pl_shader_sample_nearest(fetch, &src);
            static const struct pl_plane rgba = { .components = 4, .component_mapping = {0, 1, 2, 3} };
            ok = ok && plh_append_plane_fetch(sh, fetch, &rgba);
            pl_dispatch_abort(rr->dp, &fetch);
        }
        if (!ok)
            break;

        // normally just the linearization; also reconciles frames cached under another target
        // colour space (preserve_mixing_cache). HDR metadata differences are ignored.
        struct pl_color_repr repr = in[i].frame.repr;
        struct pl_color_space have = in[i].frame.color;
        have.hdr = space.hdr;
        if (!pl_color_space_equal(&have, &space)) {
            pl_shader_set_alpha(sh, &repr, PL_ALPHA_INDEPENDENT);
            pl_shader_color_map_ex(sh, NULL, pl_color_map_args( .src = have, .dst = space ));
        }
        pl_shader_set_alpha(sh, &repr, PL_ALPHA_PREMULTIPLIED);

        if (blend) {
            struct plh_op *op = sh_op(sh, PLH_OP_MIX_ADD);
            ok = op != NULL;
            if (ok) {
                op->f[0] = in[i].weight / wsum;
                sh_listf(sh, "mix_color += %g * color\n", op->f[0]);
            }
        }
        comps = PL_MAX(comps, in[i].frame.comps);
    }
    if (ok && blend) {
        ok = sh_op(sh, PLH_OP_MIX_END) != NULL;
        sh_listf(sh, "color = mix_color\n");
    }
    if (!ok || pl_shader_is_failed(sh)) {
        pl_dispatch_abort(rr->dp, &sh);
        return NULL;
    }
    sh_describef(sh, "frame mixing (%d frame%s)", n, n > 1 ? "s" : "");

    // back from linear light to the target's transfer, in independent alpha (the output stage
    // premultiplies again where it needs to)
    struct pl_color_repr repr = { .alpha = comps >= 4 ? PL_ALPHA_PREMULTIPLIED : PL_ALPHA_NONE };
    if (!pl_color_space_equal(&space, &target->color)) {
        pl_shader_set_alpha(sh, &repr, PL_ALPHA_INDEPENDENT);
        pl_shader_color_map_ex(sh, NULL, pl_color_map_args( .src = space, .dst = target->color ));
    }
    *out_comps = comps;
    *out_alpha = repr.alpha;
    return sh;
}

/* ---- entry point ------------------------------------------------------------------------------ */

bool pl_render_image_mix(pl_renderer rr, const struct pl_frame_mix *mix,
                         const struct pl_frame *ptarget, const struct pl_render_params *params)
{
    params = params ? params : &pl_render_default_params;
    if (!mix || !mix->num_frames)
        return pl_render_image(rr, NULL, ptarget, params);
    if (!plh_params_supported(rr, params))
        return false;
    if (!(mix->vsync_duration > 0.0f)) {
        RR_LOG(rr, PL_LOG_ERR, "pl_render_image_mix: vsync_duration must be positive");
        return false;
    }
    for (int i = 1; i < mix->num_frames; i++) {
        if (!(mix->timestamps[i - 1] <= mix->timestamps[i])) {
            RR_LOG(rr, PL_LOG_ERR, "pl_render_image_mix: timestamps must be sorted");
            return false;
        }
    }

    const struct pl_frame *nearest = pl_frame_mix_nearest(mix);
    struct frame_job job = {
        .rr = rr, .params = params, .image = *nearest, .target = *ptarget,
        .info.stage = PL_RENDER_STAGE_BLEND,
    };
    bool mixed = false, disable = false;

    if (rr->errors & PL_RENDER_ERR_FRAME_MIXING)
        goto single;
    if (!plh_job_begin(&job, false))
        return false;
    const int out_w = abs(pl_rect_w(job.geo.dst)), out_h = abs(pl_rect_h(job.geo.dst));
    if (!job.caps.fbo[4] || !out_w || !out_h)
        goto single;

    const uint64_t digest = digest_params(params);
    struct mixer_setup ms = prepare_mixer(mix, params);
    struct blend_input inputs[RR_MAX_MIX_FRAMES];
    int n = 0;
    float wsum = 0.0f;

    for (int attempt = 0; attempt < 2 && !n; attempt++) {
        // second attempt: nothing had a usable weight -> hold the nearest frame
        if (attempt)
            ms.single = true;
        for (int i = 0; i < rr->num_cached && !attempt; i++)
            rr->cache[i].stale = true;

        for (int i = 0; i < mix->num_frames && n < RR_MAX_MIX_FRAMES; i++) {
            const struct pl_frame *image = mix->frames[i];
            float weight;
            if (pl_rotation_normalize(image->rotation - nearest->rotation) != 0 ||
                !frame_weight(mix, i, &ms, nearest, &weight))
                continue;

            struct mix_entry *e = cache_find(rr, mix->signatures[i]);
            if (e)
                e->stale = false;   // in range: keep it, even if it contributes nothing now
            if (fabsf(weight) <= 1e-3f && image != nearest)
                continue;

            const bool uncached = ms.single && params->skip_caching_single_frame;
            if (!e && uncached)
                goto single;
            if (!e && !(e = cache_add(rr, mix->signatures[i]))) {
                RR_LOG(rr, PL_LOG_WARN, "Frame mixing cache is full, rendering without mixing");
                goto single;
            }

            bool usable = e->tex != NULL;
            const bool strict = uncached || ms.single || !params->preserve_mixing_cache;
            if (usable && strict) {
                usable = e->tex->params.w == out_w && e->tex->params.h == out_h &&
                         same_rect(e->crop, image->crop) && e->params_digest == digest &&
                         pl_color_space_equal(&e->color, &job.target.color);
            }
            if (!usable && uncached)
                goto single;
            if (!usable && !cache_fill(&job, e, image, ptarget, out_w, out_h, digest)) {
                disable = true;
                goto single;
            }

            inputs[n].frame = *e;
            inputs[n].weight = weight;
            wsum += weight;
            n++;
        }
        cache_collect(rr);
        if (ms.single)
            break;
    }
    if (!n)
        goto single;

    plh_job_watch_passes(&job);
    job.info.count = n;
    int comps = 0;
    enum pl_alpha_mode alpha = PL_ALPHA_NONE;
    pl_shader sh = blend_frames(&job, inputs, n, out_w, out_h, wsum, &comps, &alpha);
    if (!sh) {
        RR_LOG(rr, PL_LOG_WARN, "Frame mixing pass could not be recorded (%d frames), rendering "
               "the nearest frame instead", n);
        goto single;
    }
    job.img = (struct work_image) {
        .rec = sh, .w = out_w, .h = out_h, .comps = comps,
        .color = job.target.color,
        .rect = { 0, 0, out_w, out_h },
        .repr = {
            .sys = PL_COLOR_SYSTEM_RGB,
            .levels = PL_COLOR_LEVELS_FULL,
            .alpha = alpha,
        },
    };
    mixed = plh_stage_output(&job);

single:
    if (disable) {
        RR_LOG(rr, PL_LOG_ERR, "Could not render image for frame mixing.. disabling!");
        rr->errors |= PL_RENDER_ERR_FRAME_MIXING;
    }
    plh_job_end(&job);
    return mixed || pl_render_image(rr, nearest, ptarget, params);
}
