/*
 * libplacebo-hip — pl_renderer: the pl_render_image hot path.
 *
 * Host-side pass graph of the reference's src/renderer.c, restated for the op-recording
 * shaders of this backend. Same stages, same order, same decisions:
 *
 *   pass_fix_frames     crop rounding, bit-depth / colour-space inference   renderer.c:3068-3293
 *   pass_read_image     [deband] -> sample plane -> decode -> premultiply  :1553-1960
 *   pass_scale_main     sampler choice, [peak detect], linearize/sigmoidize,
 *                       PASS A into an FBO, main scaler, unsigmoidize       :597-682, :1964-2087
 *   pass_convert_colors [PASS B for same-frame peak], colour mapping        :2157-2280
 *   pass_output_target  background, encode, dither / error diffusion,
 *                       1/scale, swizzle, final pass into the target        :2586-2960
 *
 * A pass boundary (FBO) appears exactly where the reference has one; everything between two
 * boundaries runs fused in one HIP launch (sampler + colour ops).
 *
 * Scope: packed, semi-planar and planar (subsampled) frames in and out, rotation, custom LUTs,
 * contrast recovery, frame mixing (pl_render_image_mix). Hooks, ICC, overlays, blending,
 * deinterlacing and distortion are outside the hot path and rejected with an error.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/renderer.h>

#include "shaders_priv.h"

#define MAX_FBOS 16
enum { LUT_IMAGE, LUT_TARGET, LUT_PARAMS };
#define MAX_MIX_FRAMES 16        // renderer.c:3610
#define MAX_CACHED_FRAMES 32

struct sampler {
    pl_shader_obj upscaler_state;
    pl_shader_obj downscaler_state;
};

struct pl_renderer_t {
    pl_gpu gpu;
    pl_dispatch dp;
    pl_log log;
    enum pl_render_error errors;

    pl_tex fbos[MAX_FBOS];
    int num_fbos;

    struct sampler sampler_main;
    struct sampler sampler_src;
    struct sampler sampler_contrast;             // feature map of the contrast recovery
    struct sampler samplers_aux[PL_MAX_PLANES];  // chroma / alpha plane scalers
    struct sampler samplers_dst[PL_MAX_PLANES];  // planar output
    pl_shader_obj tone_map_state;
    pl_shader_obj dither_state;
    pl_shader_obj lut_state[3];         // LUT_IMAGE, LUT_TARGET, LUT_PARAMS
    int prev_dither;

    // frame mixing cache (pl_render_image_mix, renderer.c:82-110 `struct cached_frame`)
    struct cached_frame {
        uint64_t signature;
        uint64_t params_hash;       // of the params it was rendered with
        struct pl_color_space color;
        struct pl_color_repr repr;
        pl_tex tex;
        int comps;
        pl_rect2df crop;
        bool evict;                 // for garbage collection
    } frames[MAX_CACHED_FRAMES];
    int num_frames;
    pl_tex frame_fbos[MAX_CACHED_FRAMES];   // textures of evicted frames, for reuse
    int num_frame_fbos;
};

enum sampler_type {
    SAMPLER_DIRECT,     // pick based on texture caps
    SAMPLER_NEAREST,
    SAMPLER_BICUBIC,
    SAMPLER_HERMITE,
    SAMPLER_GAUSSIAN,
    SAMPLER_COMPLEX,    // polar / separable filters
    SAMPLER_OVERSAMPLE,
};

enum sampler_dir { SAMPLER_NOOP, SAMPLER_UP, SAMPLER_DOWN };
enum sampler_usage { SAMPLER_MAIN, SAMPLER_PLANE, SAMPLER_LOWPASS };

struct sampler_info {
    const struct pl_filter_config *config;
    enum sampler_usage usage;
    enum sampler_type type;
    enum sampler_dir dir;
    enum sampler_dir dir_sep[2];
};

// An image in flight: either a recorded-but-not-yet-dispatched shader or a texture
struct img {
    pl_shader sh;
    pl_tex tex;
    int w, h;
    pl_rect2df rect;
    struct pl_color_repr repr;
    struct pl_color_space color;
    int comps;
    pl_fmt fmt;             // FBO format override
    const char *err_msg;
    enum pl_render_error err_enum;
    pl_tex err_tex;
    pl_tex sh_origin;       // texture that `sh` started from as a plain 1:1 fetch (img_sh)
};

struct pass_state {
    pl_renderer rr;
    const struct pl_render_params *params;
    struct pl_frame image, target;
    pl_rect2d dst_rect;
    pl_rect2df ref_rect;
    struct img img;
    pl_fmt fbofmt[5];
    bool fbos_used[MAX_FBOS];
    bool need_peak_fbo;
    bool acquired_image, acquired_target;
    struct pl_render_info info;
    pl_rotation rotation;   // logical end-to-end rotation
};

static void info_callback(void *priv, const struct pl_dispatch_info *dinfo)
{
    struct pass_state *pass = priv;
    const struct pl_render_params *params = pass->params;
    if (!params->info_callback)
        return;
    pass->info.pass = dinfo;
    params->info_callback(params->info_priv, &pass->info);
    pass->info.index++;
}

#define RR_ERR(rr, ...)  pl_msg((rr)->log, PL_LOG_ERR, __VA_ARGS__)
#define PL_WARN_RR(rr, ...) pl_msg((rr)->log, PL_LOG_WARN, __VA_ARGS__)
#define RR_WARN(rr, ...) pl_msg((rr)->log, PL_LOG_WARN, __VA_ARGS__)
#define RR_INFO(rr, ...) pl_msg((rr)->log, PL_LOG_INFO, __VA_ARGS__)

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

pl_renderer pl_renderer_create(pl_log log, pl_gpu gpu)
{
    pl_renderer rr = calloc(1, sizeof(*rr));
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

static void sampler_destroy(struct sampler *s)
{
    pl_shader_obj_destroy(&s->upscaler_state);
    pl_shader_obj_destroy(&s->downscaler_state);
}

static void frame_cache_flush(pl_renderer rr)
{
    for (int i = 0; i < rr->num_frames; i++)
        pl_tex_destroy(rr->gpu, &rr->frames[i].tex);
    for (int i = 0; i < rr->num_frame_fbos; i++)
        pl_tex_destroy(rr->gpu, &rr->frame_fbos[i]);
    rr->num_frames = rr->num_frame_fbos = 0;
}

void pl_renderer_flush_cache(pl_renderer rr)
{
    frame_cache_flush(rr);
    for (int i = 0; i < rr->num_fbos; i++)
        pl_tex_destroy(rr->gpu, &rr->fbos[i]);
    rr->num_fbos = 0;
    pl_reset_detected_peak(rr->tone_map_state);
}

void pl_renderer_destroy(pl_renderer *p_rr)
{
    pl_renderer rr = *p_rr;
    if (!rr)
        return;
    pl_gpu_finish(rr->gpu);
    pl_renderer_flush_cache(rr);
    sampler_destroy(&rr->sampler_main);
    sampler_destroy(&rr->sampler_src);
    sampler_destroy(&rr->sampler_contrast);
    for (int i = 0; i < PL_MAX_PLANES; i++) {
        sampler_destroy(&rr->samplers_aux[i]);
        sampler_destroy(&rr->samplers_dst[i]);
    }
    pl_shader_obj_destroy(&rr->tone_map_state);
    pl_shader_obj_destroy(&rr->dither_state);
    for (int i = 0; i < 3; i++)
        pl_shader_obj_destroy(&rr->lut_state[i]);
    pl_dispatch_destroy(&rr->dp);
    free(rr);
    *p_rr = NULL;
}

struct pl_render_errors pl_renderer_get_errors(pl_renderer rr)
{
    return (struct pl_render_errors) { .errors = rr->errors };
}

void pl_renderer_reset_errors(pl_renderer rr, const struct pl_render_errors *errors)
{
    if (!errors) {
        rr->errors = PL_RENDER_ERR_NONE;
        return;
    }
    rr->errors &= ~errors->errors;
}

bool pl_renderer_get_hdr_metadata(pl_renderer rr, struct pl_hdr_metadata *metadata)
{
    return pl_get_detected_hdr_metadata(rr->tone_map_state, metadata);
}

pl_shader_obj pl_hip_renderer_tone_map_state(pl_renderer rr)
{
    return rr->tone_map_state;
}

/* ---- FBOs (find_fbo_format :383-434, get_fbo :448-503) ----------------------------------- */

static void find_fbo_format(struct pass_state *pass)
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    if (params->disable_fbos || (rr->errors & PL_RENDER_ERR_FBO) || pass->fbofmt[4])
        return;

    static const struct { enum pl_fmt_type type; int depth; enum pl_fmt_caps caps; } configs[] = {
        {PL_FMT_FLOAT, 16, PL_FMT_CAP_LINEAR},
        {PL_FMT_FLOAT, 16, PL_FMT_CAP_SAMPLEABLE},
        {PL_FMT_UNORM, 16, PL_FMT_CAP_LINEAR},
        {PL_FMT_SNORM, 16, PL_FMT_CAP_LINEAR},
        {PL_FMT_UNORM, 16, PL_FMT_CAP_SAMPLEABLE},
        {PL_FMT_SNORM, 16, PL_FMT_CAP_SAMPLEABLE},
        {PL_FMT_UNORM, 8, PL_FMT_CAP_LINEAR},
        {PL_FMT_UNORM, 8, PL_FMT_CAP_SAMPLEABLE},
    };

    for (size_t i = 0; i < sizeof(configs) / sizeof(configs[0]); i++) {
        if (params->force_low_bit_depth_fbos && configs[i].depth > 8)
            continue;
        pl_fmt fmt = pl_find_fmt(rr->gpu, configs[i].type, 4, configs[i].depth, 0,
                                 PL_FMT_CAP_RENDERABLE | configs[i].caps);
        if (!fmt)
            continue;
        pass->fbofmt[4] = fmt;
        for (int c = 3; c >= 1; c--) {
            pass->fbofmt[c] = pl_find_fmt(rr->gpu, configs[i].type, c, configs[i].depth, 0,
                                          fmt->caps);
            pass->fbofmt[c] = PL_DEF(pass->fbofmt[c], pass->fbofmt[c + 1]);
        }
        return;
    }

    RR_WARN(rr, "Found no renderable FBO format! Most features disabled");
    rr->errors |= PL_RENDER_ERR_FBO;
}

static pl_tex get_fbo(struct pass_state *pass, int w, int h, pl_fmt fmt, int comps)
{
    pl_renderer rr = pass->rr;
    comps = PL_DEF(comps, 4);
    fmt = PL_DEF(fmt, pass->fbofmt[comps]);
    if (!fmt)
        return NULL;

    const struct pl_tex_params params = {
        .w = w, .h = h, .format = fmt,
        .sampleable = true, .renderable = true,
        .storable = fmt->caps & PL_FMT_CAP_STORABLE,
    };

    // best fit among the unused FBOs: |dw| + |dh| + 1000 * (format mismatch)
    int best_idx = -1, best_diff = 0;
    for (int i = 0; i < rr->num_fbos; i++) {
        if (pass->fbos_used[i])
            continue;
        const int diff = abs(rr->fbos[i]->params.w - w) + abs(rr->fbos[i]->params.h - h) +
                         (rr->fbos[i]->params.format != fmt ? 1000 : 0);
        if (best_idx < 0 || diff < best_diff) {
            best_idx = i;
            best_diff = diff;
        }
    }
    if (best_idx < 0) {
        if (rr->num_fbos == MAX_FBOS)
            return NULL;
        best_idx = rr->num_fbos++;
        rr->fbos[best_idx] = NULL;
    }
    if (!pl_tex_recreate(rr->gpu, &rr->fbos[best_idx], &params))
        return NULL;
    pass->fbos_used[best_idx] = true;
    return rr->fbos[best_idx];
}

// Forcibly convert an img to `tex`, dispatching where necessary (:505-547)
static pl_tex img_tex(struct pass_state *pass, struct img *img)
{
    if (img->tex)
        return img->tex;

    pl_renderer rr = pass->rr;
    // img_sh() followed by img_tex() with nothing recorded in between (the reference's
    // get_feature_map / need_peak_fbo sequences): the texture the shader would copy is the answer
    if (img->sh && img->sh_origin && !img->fmt) {
        const struct plh_pass *p = &img->sh->pass;
        pl_tex o = img->sh_origin;
        if (img->sh->kind == PLH_SHADER_PASS && !p->num_ops && !p->num_pre_ops &&
            p->s.scale == 1.0f && !img->sh->detect_peak &&
            (p->s.type == PLH_SAMPLE_NEAREST || p->s.type == PLH_SAMPLE_BILINEAR) &&
            o->params.w == img->w && o->params.h == img->h && !pl_shader_is_failed(img->sh))
        {
            pl_dispatch_abort(rr->dp, &img->sh);
            img->sh_origin = NULL;
            img->tex = o;
            return o;
        }
    }
    img->sh_origin = NULL;
    pl_tex tex = get_fbo(pass, img->w, img->h, img->fmt, img->comps);
    img->fmt = NULL;
    if (!tex) {
        RR_ERR(rr, "Failed creating FBO texture! Disabling advanced rendering..");
        memset(pass->fbofmt, 0, sizeof(pass->fbofmt));
        pl_dispatch_abort(rr->dp, &img->sh);
        rr->errors |= PL_RENDER_ERR_FBO;
        return img->err_tex;
    }

    const bool ok = pl_dispatch_finish(rr->dp, pl_dispatch_params(
        .shader = &img->sh,
        .target = tex,
    ));

    const char *err_msg = img->err_msg;
    const enum pl_render_error err_enum = img->err_enum;
    pl_tex err_tex = img->err_tex;
    img->err_msg = NULL;
    img->err_enum = PL_RENDER_ERR_NONE;
    img->err_tex = NULL;

    if (!ok) {
        RR_ERR(rr, "%s", PL_DEF(err_msg, "Failed dispatching intermediate pass!"));
        rr->errors |= err_enum;
        img->sh = pl_dispatch_begin(rr->dp);
        img->tex = err_tex;
        return img->tex;
    }

    img->tex = tex;
    return img->tex;
}

// Forcibly convert an img to `sh`, sampling where necessary (:552-567)
static pl_shader img_sh(struct pass_state *pass, struct img *img)
{
    if (img->sh)
        return img->sh;
    img->sh = pl_dispatch_begin(pass->rr->dp);
    pl_shader_sample_direct(img->sh, pl_sample_src( .tex = img->tex ));
    img->sh_origin = img->tex;
    img->tex = NULL;
    return img->sh;
}

/* ---- samplers (sample_src_info :597-682, dispatch_sampler :684-789) ------------------------ */

static struct sampler_info sample_src_info(struct pass_state *pass, const struct pl_sample_src *src,
                                           enum sampler_usage usage)
{
    const struct pl_render_params *params = pass->params;
    struct sampler_info info = { .usage = usage };
    pl_renderer rr = pass->rr;

    const float rx = src->new_w / fabsf(pl_rect_w(src->rect));
    if (rx < 1.0 - 1e-6) {
        info.dir_sep[0] = SAMPLER_DOWN;
    } else if (rx > 1.0 + 1e-6) {
        info.dir_sep[0] = SAMPLER_UP;
    }
    const float ry = src->new_h / fabsf(pl_rect_h(src->rect));
    if (ry < 1.0 - 1e-6) {
        info.dir_sep[1] = SAMPLER_DOWN;
    } else if (ry > 1.0 + 1e-6) {
        info.dir_sep[1] = SAMPLER_UP;
    }

    if (params->correct_subpixel_offsets) {
        if (!info.dir_sep[0] && fabsf(src->rect.x0) > 1e-6f)
            info.dir_sep[0] = SAMPLER_UP;
        if (!info.dir_sep[1] && fabsf(src->rect.y0) > 1e-6f)
            info.dir_sep[1] = SAMPLER_UP;
    }

    // downscaling overrides upscaling when choosing scalers
    info.dir = PL_MAX(info.dir_sep[0], info.dir_sep[1]);
    switch (info.dir) {
    case SAMPLER_DOWN:
        if (usage == SAMPLER_LOWPASS)
            info.config = &pl_filter_bicubic;   // (:630-631)
        else
            info.config = usage == SAMPLER_PLANE && params->plane_downscaler
                            ? params->plane_downscaler : params->downscaler;
        break;
    case SAMPLER_UP:
        info.config = usage == SAMPLER_PLANE && params->plane_upscaler
                        ? params->plane_upscaler : params->upscaler;
        break;
    case SAMPLER_NOOP:
        info.type = SAMPLER_NEAREST;
        return info;
    }

    if ((rr->errors & PL_RENDER_ERR_SAMPLING) || !info.config) {
        info.type = SAMPLER_DIRECT;
    } else if (info.config->kernel == &pl_filter_function_oversample) {
        info.type = SAMPLER_OVERSAMPLE;
    } else {
        info.type = SAMPLER_COMPLEX;

        // faster replacements for the scalers a texture unit provides
        pl_fmt texfmt = src->tex ? src->tex->params.format : pass->fbofmt[4];
        const bool can_linear = texfmt->caps & PL_FMT_CAP_LINEAR;
        const bool can_fast = info.dir == SAMPLER_UP || params->skip_anti_aliasing;
        if (can_fast && !params->disable_builtin_scalers) {
            if (can_linear && pl_filter_config_eq(info.config, &pl_filter_bicubic))
                info.type = SAMPLER_BICUBIC;
            if (can_linear && pl_filter_config_eq(info.config, &pl_filter_hermite))
                info.type = SAMPLER_HERMITE;
            if (can_linear && pl_filter_config_eq(info.config, &pl_filter_gaussian))
                info.type = SAMPLER_GAUSSIAN;
            if (can_linear && pl_filter_config_eq(info.config, &pl_filter_bilinear))
                info.type = SAMPLER_DIRECT;
            if (pl_filter_config_eq(info.config, &pl_filter_nearest))
                info.type = can_linear ? SAMPLER_NEAREST : SAMPLER_DIRECT;
        }
    }

    // no advanced scaling without FBOs
    if (!pass->fbofmt[4] && info.type == SAMPLER_COMPLEX)
        info.type = SAMPLER_DIRECT;
    return info;
}

static void dispatch_sampler(struct pass_state *pass, pl_shader sh, struct sampler *sampler,
                             enum sampler_usage usage, const struct pl_sample_src *src)
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    if (!sampler)
        goto fallback;

    const struct sampler_info info = sample_src_info(pass, src, usage);
    pl_shader_obj *lut = NULL;
    switch (info.dir) {
    case SAMPLER_NOOP:
        goto fallback;
    case SAMPLER_DOWN:
        lut = &sampler->downscaler_state;
        break;
    case SAMPLER_UP:
        lut = &sampler->upscaler_state;
        break;
    }

    switch (info.type) {
    case SAMPLER_DIRECT:
        goto fallback;
    case SAMPLER_NEAREST:
        pl_shader_sample_nearest(sh, src);
        return;
    case SAMPLER_OVERSAMPLE:
        pl_shader_sample_oversample(sh, src, info.config->kernel->params[0]);
        return;
    case SAMPLER_BICUBIC:
        pl_shader_sample_bicubic(sh, src);
        return;
    case SAMPLER_HERMITE:
        pl_shader_sample_hermite(sh, src);
        return;
    case SAMPLER_GAUSSIAN:
        pl_shader_sample_gaussian(sh, src);
        return;
    case SAMPLER_COMPLEX:
        break;
    }

    struct pl_sample_filter_params fparams = {
        .filter      = *info.config,
        .antiring    = params->antiringing_strength,
        .no_widening = params->skip_anti_aliasing && usage != SAMPLER_LOWPASS,
        .lut         = lut,
    };

    bool ok;
    if (info.config->polar) {
        ok = pl_shader_sample_polar(sh, src, &fparams);
    } else if (info.dir_sep[0] && info.dir_sep[1]) {
        // both directions: vertical pass into an FBO, then the horizontal pass (:745-772)
        struct pl_sample_src src1 = *src, src2 = *src;
        src1.new_w = src->tex->params.w;
        src1.rect.x0 = 0;
        src1.rect.x1 = src1.new_w;
        src2.rect.y0 = 0;
        src2.rect.y1 = src1.new_h;

        pl_shader tsh = pl_dispatch_begin(rr->dp);
        ok = pl_shader_sample_ortho2(tsh, &src1, &fparams);
        if (!ok) {
            pl_dispatch_abort(rr->dp, &tsh);
            goto done;
        }
        struct img img = {
            .sh = tsh, .w = src1.new_w, .h = src1.new_h, .comps = src->components,
        };
        src2.tex = img_tex(pass, &img);
        src2.scale = 1.0;
        ok = src2.tex && pl_shader_sample_ortho2(sh, &src2, &fparams);
    } else {
        ok = pl_shader_sample_ortho2(sh, src, &fparams);
    }

done:
    if (!ok) {
        RR_ERR(rr, "Failed dispatching scaler.. disabling");
        rr->errors |= PL_RENDER_ERR_SAMPLING;
        goto fallback;
    }
    return;

fallback:
    pl_shader_sample_direct(sh, src);
}

// PASS A fusion (not in the reference, which always renders `pre` into an FBO before a complex
// scaler, renderer.c:2064): if the main scaler is a polar one and `pre` - everything recorded
// so far - is a plain fetch of the whole plane followed by colour ops, the polar kernel runs
// those ops on the source texels while it stages them, with the FBO's rgba16hf rounding, and
// the intermediate pass (one full-frame write + read) disappears. Same values, one pass less.
static bool try_fused_polar(struct pass_state *pass, pl_shader sh, const struct pl_sample_src *src,
                            pl_shader pre, int fbo_w, int fbo_h)
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    const char *env = getenv("PL_HIP_NO_FUSION");
    if (!pre || (env && env[0] == '1'))
        return false;
    pl_fmt fbofmt = pass->fbofmt[pass->img.comps];
    if (!fbofmt || fbofmt->type != PL_FMT_FLOAT || fbofmt->component_depth[0] != 16)
        return false; // the fused tile rounds to f16: only valid if the FBO would too

    // sample_src_info wants a texture for its format checks: the FBO that would be used
    struct pl_sample_src probe = *src;
    const struct pl_tex_params tp = { .w = fbo_w, .h = fbo_h, .format = fbofmt };
    const struct pl_tex_t fake = { .params = tp };
    probe.tex = &fake;
    const struct sampler_info info = sample_src_info(pass, &probe, SAMPLER_MAIN);
    if (info.type != SAMPLER_COMPLEX || !info.config->polar || info.dir == SAMPLER_NOOP)
        return false;
    if (PL_DEF(info.config->antiring, params->antiringing_strength) > 0)
        return false;

    struct pl_sample_filter_params fparams = {
        .filter      = *info.config,
        .antiring    = params->antiringing_strength,
        .no_widening = params->skip_anti_aliasing,
        .lut         = info.dir == SAMPLER_UP ? &rr->sampler_main.upscaler_state
                                              : &rr->sampler_main.downscaler_state,
    };
    return plh_shader_sample_polar_fused(sh, pre, &probe, &fparams);
}

/* ---- planes (detect_plane_type :287-335, frame_ref :3048-3066) -------------------------------- */

enum plane_type { PLANE_INVALID = 0, PLANE_ALPHA, PLANE_CHROMA, PLANE_LUMA, PLANE_RGB, PLANE_XYZ };

static enum plane_type detect_plane_type(const struct pl_plane *plane,
                                         const struct pl_color_repr *repr)
{
    if (pl_color_system_is_ycbcr_like(repr->sys)) {
        int t = PLANE_INVALID;
        for (int c = 0; c < plane->components; c++) {
            switch (plane->component_mapping[c]) {
            case PL_CHANNEL_Y: t = PL_MAX(t, PLANE_LUMA); continue;
            case PL_CHANNEL_A: t = PL_MAX(t, PLANE_ALPHA); continue;
            case PL_CHANNEL_CB:
            case PL_CHANNEL_CR: t = PL_MAX(t, PLANE_CHROMA); continue;
            default: continue;
            }
        }
        return t;
    }
    if (plane->components == 1 && plane->component_mapping[0] == PL_CHANNEL_A)
        return PLANE_ALPHA;
    return repr->sys == PL_COLOR_SYSTEM_XYZ ? PLANE_XYZ : PLANE_RGB;
}

static int frame_ref(const struct pl_frame *frame)
{
    for (int i = 0; i < frame->num_planes; i++) {
        switch (detect_plane_type(&frame->planes[i], &frame->repr)) {
        case PLANE_RGB: case PLANE_LUMA: case PLANE_XYZ:
            return i;
        default:
            continue;
        }
    }
    return 0;
}

/* ---- frame fix-ups (:3068-3293) -------------------------------------------------------------- */

static void default_rect(pl_rect2df *rc, const pl_rect2df *backup)
{
    if (!rc->x0 && !rc->y0 && !rc->x1 && !rc->y1)
        *rc = *backup;
}

bool pl_frame_is_cropped(const struct pl_frame *frame)
{
    if (!frame->num_planes || !frame->planes[frame_ref(frame)].texture)
        return false;
    pl_tex ref = frame->planes[frame_ref(frame)].texture;
    pl_rect2df crop = frame->crop;
    default_rect(&crop, &(pl_rect2df) { 0, 0, ref->params.w, ref->params.h });
    pl_rect2df_normalize(&crop);
    const int x0 = roundf(crop.x0), y0 = roundf(crop.y0),
              x1 = roundf(crop.x1), y1 = roundf(crop.y1);
    return x0 > 0 || y0 > 0 || x1 < ref->params.w || y1 < ref->params.h;
}

static void fix_refs_and_rects(struct pass_state *pass)
{
    struct pl_frame *target = &pass->target, *image = &pass->image;
    pl_rect2df *dst = &target->crop, *src = &image->crop;
    pl_tex dst_ref = target->planes[frame_ref(target)].texture,
           src_ref = image->planes[frame_ref(image)].texture;
    int dst_w = dst_ref->params.w, dst_h = dst_ref->params.h;

    if ((!dst->x0 && !dst->x1) || (!dst->y0 && !dst->y1)) {
        dst->x1 = dst_w;
        dst->y1 = dst_h;
    }
    if ((!src->x0 && !src->x1) || (!src->y0 && !src->y1)) {
        src->x1 = src_ref->params.w;
        src->y1 = src_ref->params.h;
    }

    // end-to-end rotation (:3113-3117): the image is processed in its own orientation, the
    // target rect is counter-rotated into it, the output stage transposes / flips the stores
    pass->rotation = pl_rotation_normalize(image->rotation - target->rotation);
    pl_rect2df_rotate(dst, -pass->rotation);
    if (pass->rotation % PL_ROTATION_180 == PL_ROTATION_90) {
        const int t = dst_w;
        dst_w = dst_h;
        dst_h = t;
    }

    // is the end-to-end rendering flipped?
    const bool flipped_x = (src->x0 > src->x1) != (dst->x0 > dst->x1),
               flipped_y = (src->y0 > src->y1) != (dst->y0 > dst->y1);
    pl_rect2df_normalize(src);
    pl_rect2df_normalize(dst);

    // round the output rect and clip it to the framebuffer
    const float rx0 = roundf(PL_CLAMP(dst->x0, 0.0, dst_w)),
                ry0 = roundf(PL_CLAMP(dst->y0, 0.0, dst_h)),
                rx1 = roundf(PL_CLAMP(dst->x1, 0.0, dst_w)),
                ry1 = roundf(PL_CLAMP(dst->y1, 0.0, dst_h));

    // adjust the src rect for the rounded crop
    const float scale_x = pl_rect_w(*src) / pl_rect_w(*dst),
                scale_y = pl_rect_h(*src) / pl_rect_h(*dst),
                base_x = src->x0, base_y = src->y0;
    src->x0 = base_x + (rx0 - dst->x0) * scale_x;
    src->x1 = base_x + (rx1 - dst->x0) * scale_x;
    src->y0 = base_y + (ry0 - dst->y0) * scale_y;
    src->y1 = base_y + (ry1 - dst->y0) * scale_y;

    // flips always go to the dst rect (keeps compute samplers usable)
    *dst = (pl_rect2df) {
        .x0 = flipped_x ? rx1 : rx0,
        .y0 = flipped_y ? ry1 : ry0,
        .x1 = flipped_x ? rx0 : rx1,
        .y1 = flipped_y ? ry0 : ry1,
    };
    pass->ref_rect = *src;
    pass->dst_rect = (pl_rect2d) { dst->x0, dst->y0, dst->x1, dst->y1 };
}

static void fix_frame(struct pl_frame *frame)
{
    pl_tex tex = frame->planes[frame_ref(frame)].texture;
    if (frame->repr.sys == PL_COLOR_SYSTEM_XYZ) {
        // XYZ is implicitly converted to linear DCI-P3 in pl_color_repr_decode
        frame->color.primaries = PL_COLOR_PRIM_DCI_P3;
        frame->color.transfer = PL_COLOR_TRC_ST428;
    }
    if (tex && !frame->color.primaries)
        frame->color.primaries = pl_color_primaries_guess(tex->params.w, tex->params.h);

    bool has_alpha = false;
    for (int p = 0; p < frame->num_planes; p++) {
        for (int c = 0; c < frame->planes[p].components; c++)
            has_alpha |= frame->planes[p].component_mapping[c] == PL_CHANNEL_A;
    }
    if (!has_alpha)
        frame->repr.alpha = PL_ALPHA_NONE;

    // UNORM textures tell us the sampled bit depth
    struct pl_bit_encoding *bits = &frame->repr.bits;
    if (!bits->sample_depth && tex && tex->params.format->type == PL_FMT_UNORM) {
        bits->sample_depth = tex->params.format->component_depth[0];
        bits->color_depth = PL_DEF(bits->color_depth, bits->sample_depth);
        bits->color_depth = PL_MIN(bits->color_depth, bits->sample_depth);
        bits->bit_shift += bits->sample_depth - bits->color_depth;
    }
}

static void pass_fix_frames(struct pass_state *pass)
{
    struct pl_frame *image = &pass->image, *target = &pass->target;
    fix_refs_and_rects(pass);
    fix_frame(image);
    pl_color_space_infer_map(&image->color, &target->color);
    fix_frame(target); // only after infer_map
    if (image->repr.alpha == PL_ALPHA_UNKNOWN)
        image->repr.alpha = PL_ALPHA_INDEPENDENT;
    if (target->repr.alpha == PL_ALPHA_UNKNOWN)
        target->repr.alpha = PL_ALPHA_PREMULTIPLIED;
}

static bool validate_frame(pl_renderer rr, const struct pl_frame *f, const char *what, bool dst)
{
    if (f->num_planes < 1 || f->num_planes > PL_MAX_PLANES) {
        RR_ERR(rr, "%s frame has an invalid number of planes: %d", what, f->num_planes);
        return false;
    }
    for (int i = 0; i < f->num_planes; i++) {
        const struct pl_plane *pi = &f->planes[i];
        if (!pi->texture || pi->components < 1 || pi->components > 4) {
            RR_ERR(rr, "%s plane %d: missing texture or invalid number of components", what, i);
            return false;
        }
        if (!dst && !pi->texture->params.sampleable) {
            RR_ERR(rr, "Image textures must be sampleable");
            return false;
        }
        if (dst && !pi->texture->params.storable) {
            RR_ERR(rr, "Target textures must be storable (every pass is a compute pass)");
            return false;
        }
    }
    const struct pl_plane *pl = &f->planes[frame_ref(f)];
    if (pl->shift_x || pl->shift_y) {
        RR_ERR(rr, "%s reference plane must have no shift", what);
        return false;
    }
    return true;
}

void pl_frame_set_chroma_location(struct pl_frame *frame, enum pl_chroma_location chroma_loc)
{
    pl_tex ref = frame->planes[frame_ref(frame)].texture;
    for (int i = 0; i < frame->num_planes; i++) {
        struct pl_plane *plane = &frame->planes[i];
        pl_tex tex = plane->texture;
        const bool apply = ref && tex
            ? tex->params.w < ref->params.w || tex->params.h < ref->params.h
            : detect_plane_type(plane, &frame->repr) == PLANE_CHROMA;
        if (apply)
            pl_chroma_location_offset(chroma_loc, &plane->shift_x, &plane->shift_y);
    }
}

void pl_frames_infer(pl_renderer rr, struct pl_frame *image, struct pl_frame *target)
{
    struct pass_state pass = { .rr = rr, .image = *image, .target = *target };
    if (!validate_frame(rr, image, "Image", false) || !validate_frame(rr, target, "Target", true))
        return;
    pass_fix_frames(&pass);
    *image = pass.image;
    *target = pass.target;
}

/* ---- peak detection (hdr_update_peak :1183-1250) --------------------------------------------- */

static void hdr_update_peak(struct pass_state *pass)
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    if (!params->peak_detect_params || !pl_color_space_is_hdr(&pass->image.color))
        goto cleanup;
    if (rr->errors & PL_RENDER_ERR_PEAK_DETECT)
        goto cleanup;
    if (pass->fbofmt[4] && !(pass->fbofmt[4]->caps & PL_FMT_CAP_STORABLE))
        goto cleanup;

    float max_peak = pl_color_transfer_nominal_peak(pass->image.color.transfer) *
                     PL_COLOR_SDR_WHITE;
    if (pass->image.color.transfer == PL_COLOR_TRC_HLG)
        max_peak = pass->img.color.hdr.max_luma;
    if (max_peak <= pass->target.color.hdr.max_luma + 1e-6)
        goto cleanup; // no adaptation needed
    if (pass->img.color.hdr.avg_pq_y)
        goto cleanup; // dynamic metadata already present

    enum pl_hdr_metadata_type metadata = PL_HDR_METADATA_ANY;
    if (params->color_map_params)
        metadata = params->color_map_params->metadata;
    if (metadata && metadata != PL_HDR_METADATA_CIE_Y)
        goto cleanup; // measurement would be unused

    const struct pl_color_map_params *cpars = params->color_map_params;
    const bool uses_ootf = cpars && cpars->tone_mapping_function == &pl_tone_map_st2094_40;
    if (uses_ootf && pass->img.color.hdr.ootf.num_anchors)
        goto cleanup; // HDR10+ OOTF is being used
    if (params->lut && params->lut_type == PL_LUT_CONVERSION)
        goto cleanup; // LUT handles tone mapping

    if (!pass->fbofmt[4] && !params->peak_detect_params->allow_delayed) {
        RR_WARN(rr, "Disabling peak detection because `pl_peak_detect_params.allow_delayed` "
                "is false, but lack of FBOs forces the result to be delayed.");
        rr->errors |= PL_RENDER_ERR_PEAK_DETECT;
        goto cleanup;
    }

    // The polar / separable / deband kernels own their workgroup shape, so a measurement cannot
    // ride on them (the reference merges it into the scaler's compute shader): materialise the
    // image and measure it with a target-less pass that only reads it. Same 16x16 tiling of the
    // same image; the values it sees went through the FBO's f16 rounding.
    struct img *img = &pass->img;
    if (img->sh && pass->fbofmt[4] &&
        (img->sh->pass.s.type == PLH_SAMPLE_POLAR || img->sh->pass.s.type == PLH_SAMPLE_ORTHO ||
         img->sh->pass.s.type == PLH_SAMPLE_DEBAND))
    {
        pl_tex tex = img_tex(pass, img);
        if (!tex)
            goto cleanup;
        pl_shader msh = pl_dispatch_begin(rr->dp);
        bool mok = pl_shader_sample_direct(msh, pl_sample_src( .tex = tex )) &&
                   pl_shader_detect_peak(msh, img->color, &rr->tone_map_state,
                                         params->peak_detect_params);
        if (mok) {
            mok = pl_dispatch_compute(rr->dp, pl_dispatch_compute_params(
                .shader = &msh, .width = tex->params.w, .height = tex->params.h,
            ));
        } else {
            pl_dispatch_abort(rr->dp, &msh);
        }
        if (!mok) {
            RR_WARN(rr, "Failed measuring the HDR peak.. disabling");
            rr->errors |= PL_RENDER_ERR_PEAK_DETECT;
            goto cleanup;
        }
        pass->need_peak_fbo = false; // already complete (stream order)
        return;
    }

    const bool ok = pl_shader_detect_peak(img_sh(pass, &pass->img), pass->img.color,
                                          &rr->tone_map_state, params->peak_detect_params);
    if (!ok) {
        RR_WARN(rr, "Failed creating HDR peak detection shader.. disabling");
        rr->errors |= PL_RENDER_ERR_PEAK_DETECT;
        goto cleanup;
    }
    pass->need_peak_fbo = !params->peak_detect_params->allow_delayed;
    return;

cleanup:
    pl_reset_detected_peak(rr->tone_map_state);
}

/* ---- pass_read_image (:1553-1960) ------------------------------------------------------------ */

static bool plane_deband(struct pass_state *pass, struct img *img, const float neutral[3])
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    if ((rr->errors & PL_RENDER_ERR_DEBANDING) || !params->deband_params || !pass->fbofmt[4])
        return false;

    struct pl_color_repr repr = img->repr;
    struct pl_sample_src src = {
        .tex = img_tex(pass, img),
        .components = img->comps,
        .scale = pl_color_repr_normalize(&repr),
    };

    // keep the grain intensity independent of the source's nominal peak (:1337-1342)
    struct pl_deband_params dparams = *params->deband_params;
    dparams.grain /= pass->image.color.hdr.max_luma / PL_COLOR_SDR_WHITE;
    memcpy(dparams.grain_neutral, neutral, sizeof(dparams.grain_neutral));

    img->tex = NULL;
    img->sh = pl_dispatch_begin(rr->dp);
    pl_shader_deband(img->sh, &src, &dparams);
    img->err_msg = "Failed applying debanding... disabling!";
    img->err_enum = PL_RENDER_ERR_DEBANDING;
    img->err_tex = src.tex;
    img->repr = repr;
    return true;
}

struct plane_state {
    enum plane_type type;
    struct pl_plane plane;
    struct img img;
    float plane_w, plane_h; // logical plane dimensions
};

// color = scale * texel of another shader's plain fetch, merged into `sh` (the reference's
// sh_subpass). Returns false if `psh` is more than a plain fetch.
static bool merge_plane_fetch(pl_shader sh, const pl_shader psh, const struct pl_plane *plane)
{
    const struct plh_sampler_args *ps = &psh->pass.s;
    if (psh->pass.num_ops || psh->kind != PLH_SHADER_PASS ||
        (ps->type != PLH_SAMPLE_NEAREST && ps->type != PLH_SAMPLE_BILINEAR))
        return false;
    if (ps->src.w > 0xffff || ps->src.h > 0xffff)
        return false;
    struct plh_op *op = sh_op(sh, PLH_OP_PLANE_FETCH);
    if (!op)
        return false;
    memcpy(op->f, ps->pos, sizeof(ps->pos));
    op->f[8] = ps->scale;
    op->f[9] = ps->rect_w;
    op->f[10] = ps->rect_h;
    op->ptr = ps->src.ptr;
    op->i0 = ps->src.w | (ps->src.h << 16);
    op->i1 = ps->src.pitch;
    uint32_t map = 0;
    for (int c = 0; c < 4; c++) {
        const int m = c < plane->components ? plane->component_mapping[c] : -1;
        map |= (uint32_t) (m < 0 ? 0xf : m) << (4 * c);
    }
    op->i2 = ps->src.fmt | (plane->components << 8) |
             ((ps->type == PLH_SAMPLE_BILINEAR) << 12) | (ps->address_mode << 13) |
             ((ps->type == PLH_SAMPLE_BILINEAR && ps->rect_on_grid) << 15) | (map << 16);
    sh_listf(sh, "plane_fetch(tex=%dx%d, %s, scale=%g, comps=%d, map=0x%04x)\n", ps->src.w,
             ps->src.h, ps->type == PLH_SAMPLE_BILINEAR ? "bilinear" : "nearest", ps->scale,
             plane->components, map);
    for (int i = 0; i < psh->num_held; i++)
        sh_hold(sh, psh->held[i]);
    return true;
}

// guess_frame_lut_type (:1447-1468)
static enum pl_lut_type guess_frame_lut_type(const struct pl_frame *frame, bool reversed)
{
    if (!frame->lut)
        return PL_LUT_UNKNOWN;
    if (frame->lut_type)
        return frame->lut_type;
    enum pl_color_system sys_in = frame->lut->repr_in.sys, sys_out = frame->lut->repr_out.sys;
    if (reversed) {
        const enum pl_color_system t = sys_in;
        sys_in = sys_out;
        sys_out = t;
    }
    if (sys_in == PL_COLOR_SYSTEM_RGB && sys_out == sys_in)
        return PL_LUT_NORMALIZED;
    if (sys_in == frame->repr.sys && sys_out == PL_COLOR_SYSTEM_RGB)
        return PL_LUT_CONVERSION;
    return PL_LUT_NATIVE; // unknown: the default
}

static bool pass_read_image(struct pass_state *pass)
{
    const struct pl_render_params *params = pass->params;
    struct pl_frame *image = &pass->image;
    pl_renderer rr = pass->rr;
    const int src_ref = frame_ref(image);

    struct plane_state planes[PL_MAX_PLANES];
    struct plane_state *ref = &planes[src_ref];
    for (int i = 0; i < image->num_planes; i++) {
        planes[i] = (struct plane_state) {
            .type = detect_plane_type(&image->planes[i], &image->repr),
            .plane = image->planes[i],
            .img = {
                .w = image->planes[i].texture->params.w,
                .h = image->planes[i].texture->params.h,
                .tex = image->planes[i].texture,
                .repr = image->repr,
                .color = image->color,
                .comps = image->planes[i].components,
            },
        };
        // an overridden alpha mode drops the alpha channel / plane
        if (image->repr.alpha == PL_ALPHA_NONE) {
            if (planes[i].type == PLANE_ALPHA) {
                planes[i].type = PLANE_INVALID;
                continue;
            }
            for (int j = 0; j < planes[i].plane.components; j++) {
                if (planes[i].plane.component_mapping[j] == PL_CHANNEL_A)
                    planes[i].plane.component_mapping[j] = PL_CHANNEL_NONE;
            }
        }
    }
    pl_tex ref_tex = ref->plane.texture;

    const int bits = image->repr.bits.sample_depth;
    const float out_scale = bits ? (1llu << bits) / ((1llu << bits) - 1.0f) : 1.0f;
    float neutral_luma = 0.0, neutral_chroma = 0.5f * out_scale;
    if (pl_color_levels_guess(&image->repr) == PL_COLOR_LEVELS_LIMITED)
        neutral_luma = 16 / 256.0f * out_scale;
    if (!pl_color_system_is_ycbcr_like(image->repr.sys))
        neutral_chroma = neutral_luma;

    // sampling rect of every plane (:1724-1790)
    for (int i = 0; i < image->num_planes; i++) {
        struct plane_state *st = &planes[i];
        if (!st->type)
            continue;
        const float rx = (float) st->plane.texture->params.w / ref_tex->params.w,
                    ry = (float) st->plane.texture->params.h / ref_tex->params.h;
        // integer subsampling ratios only (fractionally subsampled planes are rounded up)
        const float rrx = rx >= 1 ? roundf(rx) : 1.0 / roundf(1.0 / rx),
                    rry = ry >= 1 ? roundf(ry) : 1.0 / roundf(1.0 / ry);
        const float sx = st->plane.shift_x, sy = st->plane.shift_y;
        st->img.rect = (pl_rect2df) {
            .x0 = (image->crop.x0 - sx) * rrx,
            .y0 = (image->crop.y0 - sy) * rry,
            .x1 = (image->crop.x1 - sx) * rrx,
            .y1 = (image->crop.y1 - sy) * rry,
        };
        st->plane_w = ref_tex->params.w * rrx;
        st->plane_h = ref_tex->params.h * rry;

        float neutral[3] = {0.0};
        for (int c = 0, idx = 0; c < st->plane.components; c++) {
            switch (st->plane.component_mapping[c]) {
            case PL_CHANNEL_Y: neutral[idx++] = neutral_luma; break;
            case PL_CHANNEL_U: // fall through
            case PL_CHANNEL_V: neutral[idx++] = neutral_chroma; break;
            }
        }
        plane_deband(pass, &st->img, neutral);
    }

    // Drop subpixel offsets from the ref rect and re-add them as part of `pass->img.rect`,
    // always rounding towards 0; drop anamorphic subpixel mismatches (:1810-1828)
    const pl_rect2df ref_rc = ref->img.rect;
    pl_rect2d ref_rounded;
    ref_rounded.x0 = truncf(ref_rc.x0);
    ref_rounded.y0 = truncf(ref_rc.y0);
    ref_rounded.x1 = ref_rounded.x0 + roundf(pl_rect_w(ref_rc));
    ref_rounded.y1 = ref_rounded.y0 + roundf(pl_rect_h(ref_rc));
    const float off_x = ref_rc.x0 - ref_rounded.x0, off_y = ref_rc.y0 - ref_rounded.y0,
                stretch_x = pl_rect_w(ref_rounded) / pl_rect_w(ref_rc),
                stretch_y = pl_rect_h(ref_rounded) / pl_rect_h(ref_rc);

    // every plane becomes a shader producing it on the (rounded) reference grid (:1830-1872)
    float plane_scale[PL_MAX_PLANES];
    for (int i = 0; i < image->num_planes; i++) {
        struct plane_state *st = &planes[i];
        const struct pl_plane *plane = &st->plane;
        if (!st->type)
            continue;

        const float scale_x = pl_rect_w(st->img.rect) / pl_rect_w(ref_rc),
                    scale_y = pl_rect_h(st->img.rect) / pl_rect_h(ref_rc),
                    base_x = st->img.rect.x0 - scale_x * off_x,
                    base_y = st->img.rect.y0 - scale_y * off_y;
        struct pl_sample_src src = {
            .components = plane->components,
            .address_mode = plane->address_mode,
            .scale      = pl_color_repr_normalize(&st->img.repr),
            .new_w      = pl_rect_w(ref_rounded),
            .new_h      = pl_rect_h(ref_rounded),
            .rect = {
                base_x, base_y,
                base_x + stretch_x * pl_rect_w(st->img.rect),
                base_y + stretch_y * pl_rect_h(st->img.rect),
            },
        };
        if (plane->flipped) {
            src.rect.y0 = st->plane_h - src.rect.y0;
            src.rect.y1 = st->plane_h - src.rect.y1;
        }

        const bool unscaled = src.rect.x0 == 0 && src.rect.y0 == 0 &&
                              src.rect.x1 == src.new_w && src.rect.y1 == src.new_h;
        if (st->img.sh && st->img.w == src.new_w && st->img.h == src.new_h && unscaled) {
            // image rects are already equal, no indirect scaling needed
        } else {
            src.tex = img_tex(pass, &st->img);
            if (!src.tex)
                return false;
            st->img.tex = NULL;
            st->img.sh = pl_dispatch_begin(rr->dp);
            dispatch_sampler(pass, st->img.sh, i == src_ref ? &rr->sampler_src : &rr->samplers_aux[i],
                             SAMPLER_PLANE, &src);
            st->img.err_enum |= PL_RENDER_ERR_SAMPLING;
            st->img.rect.x0 = st->img.rect.y0 = 0.0f;
            st->img.w = st->img.rect.x1 = src.new_w;
            st->img.h = st->img.rect.y1 = src.new_h;
            src.scale = 1.0;
        }
        plane_scale[i] = src.scale;
    }

    // "pass_read_image": color = (neutral_luma, neutral_chroma x2, 1); per plane
    // tmp = scale * plane(); color[mapping[c]] = tmp[c]                            (:1790-1890)
    // The reference plane's shader is the pass itself; the other planes are fetched into it.
    pl_shader sh = img_sh(pass, &ref->img);
    ref->img.sh = NULL;
    if (plane_scale[src_ref] != 1.0f) {
        struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
        if (!op)
            return false;
        op->f[0] = op->f[1] = op->f[2] = op->f[3] = plane_scale[src_ref];
        sh_listf(sh, "scale(%g)\n", plane_scale[src_ref]);
    }
    const struct pl_plane *rplane = &ref->plane;
    bool trivial = rplane->components == 4 && image->num_planes == 1;
    for (int c = 0; c < rplane->components; c++)
        trivial &= rplane->component_mapping[c] == c;
    if (!trivial) {
        struct plh_op *op = sh_op(sh, PLH_OP_PLANE_MAP);
        if (!op)
            return false;
        op->f[0] = neutral_luma;
        op->f[1] = op->f[2] = neutral_chroma;
        op->f[3] = 1.0f;
        op->i1 = rplane->components;
        op->i0 = 0;
        op->i2 = 1; // identity prefix?
        for (int c = 0; c < 4; c++) {
            const int m = c < rplane->components ? rplane->component_mapping[c] : -1;
            op->i0 |= (m < 0 ? 0xff : m) << (8 * c);
            if (c < rplane->components && m != c)
                op->i2 = 0;
        }
        sh_listf(sh, "plane_map(comps=%d, map=0x%08x, neutral=%g/%g)\n", rplane->components,
                 (unsigned) op->i0, neutral_luma, neutral_chroma);
    }

    for (int i = 0; i < image->num_planes; i++) {
        struct plane_state *st = &planes[i];
        if (!st->type || i == src_ref)
            continue;
        pl_shader psh = img_sh(pass, &st->img);
        if (plane_scale[i] != 1.0f || !merge_plane_fetch(sh, psh, &st->plane)) {
            // not a plain fetch (debanded / scaled by a complex filter): render it, fetch 1:1
            st->img.sh = psh;
            if (plane_scale[i] != 1.0f) {
                struct plh_op *op = sh_op(psh, PLH_OP_SCALE);
                if (op)
                    op->f[0] = op->f[1] = op->f[2] = op->f[3] = plane_scale[i];
            }
            st->img.comps = st->plane.components;
            if (!img_tex(pass, &st->img)) {
                pl_dispatch_abort(rr->dp, &sh);
                return false;
            }
            psh = img_sh(pass, &st->img);
            if (!merge_plane_fetch(sh, psh, &st->plane)) {
                pl_dispatch_abort(rr->dp, &psh);
                pl_dispatch_abort(rr->dp, &sh);
                return false;
            }
        }
        pl_dispatch_abort(rr->dp, &psh);
        st->img.sh = NULL;
    }

    pass->img = (struct img) {
        .sh     = sh,
        .w      = pl_rect_w(ref_rounded),
        .h      = pl_rect_h(ref_rounded),
        .repr   = ref->img.repr,
        .color  = image->color,
        .comps  = ref->img.repr.alpha == PL_ALPHA_NONE ? 3 : 4,
        .rect   = { off_x, off_y, off_x + pl_rect_w(ref_rc), off_y + pl_rect_h(ref_rc) },
        .err_msg = ref->img.err_msg, .err_enum = ref->img.err_enum, .err_tex = ref->img.err_tex,
    };
    pass->ref_rect = pass->img.rect;

    // frame LUT (:1920-1946): NATIVE / CONVERSION act on the raw (bit-depth-fixed) samples, a
    // CONVERSION LUT replaces the decoding, NORMALIZED acts on the decoded RGB
    const enum pl_lut_type lut_type = guess_frame_lut_type(image, false);
    bool needs_conversion = true;
    if (lut_type == PL_LUT_NATIVE || lut_type == PL_LUT_CONVERSION) {
        const float scale = pl_color_repr_normalize(&pass->img.repr);
        struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
        if (op) {
            op->f[0] = op->f[1] = op->f[2] = op->f[3] = scale;
            sh_listf(sh, "scale(%g)\n", scale);
        }
        pl_shader_custom_lut(sh, image->lut, &rr->lut_state[LUT_IMAGE]);
        if (lut_type == PL_LUT_CONVERSION) {
            pass->img.repr.sys = PL_COLOR_SYSTEM_RGB;
            pass->img.repr.levels = PL_COLOR_LEVELS_FULL;
            needs_conversion = false;
        }
    }
    if (needs_conversion) {
        if (pass->img.repr.sys == PL_COLOR_SYSTEM_XYZ) {
            pl_shader_linearize(sh, &pass->img.color);
            pass->img.color.transfer = PL_COLOR_TRC_LINEAR;
        }
        pl_shader_decode_color(sh, &pass->img.repr, params->color_adjustment);
    }
    if (lut_type == PL_LUT_NORMALIZED)
        pl_shader_custom_lut(sh, image->lut, &rr->lut_state[LUT_IMAGE]);

    // pre-multiply alpha before the rest of the pipeline, to avoid bleeding colours from
    // transparent regions into opaque ones
    pl_shader_set_alpha(sh, &pass->img.repr, PL_ALPHA_PREMULTIPLIED);
    return !pl_shader_is_failed(sh);
}

/* ---- pass_scale_main (:1964-2087) ------------------------------------------------------------- */

static bool pass_scale_main(struct pass_state *pass)
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    pl_fmt fbofmt = pass->fbofmt[pass->img.comps];
    if (!fbofmt)
        return true; // no FBOs: skip the main scaler

    const pl_rect2df new_rect = {
        .x1 = abs(pl_rect_w(pass->dst_rect)),
        .y1 = abs(pl_rect_h(pass->dst_rect)),
    };

    struct img *img = &pass->img;
    struct pl_sample_src src = {
        .components = img->comps,
        .new_w      = pl_rect_w(new_rect),
        .new_h      = pl_rect_h(new_rect),
        .rect       = img->rect,
    };

    const struct pl_frame *image = &pass->image;
    bool need_fbo = false;

    // force FBO indirection if this shader is non-resizable
    int out_w, out_h;
    if (img->sh && pl_shader_output_size(img->sh, &out_w, &out_h))
        need_fbo |= out_w != src.new_w || out_h != src.new_h;

    const struct sampler_info info = sample_src_info(pass, &src, SAMPLER_MAIN);
    bool use_sigmoid = info.dir == SAMPLER_UP && params->sigmoid_params;
    bool use_linear  = info.dir == SAMPLER_DOWN;

    // opportunistically measure the peak here if that saves a pass
    if (info.dir == SAMPLER_UP)
        hdr_update_peak(pass);

    if (info.dir == SAMPLER_NOOP && !need_fbo)
        goto done; // no-op

    if (info.type == SAMPLER_DIRECT && !need_fbo) {
        // "free" sampling: the final pass samples the source at the output size
        img->w = src.new_w;
        img->h = src.new_h;
        img->rect = new_rect;
        goto done;
    }

    // hard-disable sigmoidization and linearization when required
    if (params->disable_linear_scaling || fbofmt->component_depth[0] < 16)
        use_sigmoid = use_linear = false;

    // sigmoidization clips to [0,1]: not for HDR; linear HDR needs a float FBO
    if (pl_color_space_is_hdr(&img->color)) {
        use_sigmoid = false;
        if (fbofmt->type != PL_FMT_FLOAT)
            use_linear = false;
    }

    if (!(use_linear || use_sigmoid) && img->color.transfer == PL_COLOR_TRC_LINEAR) {
        img->color.transfer = image->color.transfer;
        if (image->color.transfer == PL_COLOR_TRC_LINEAR)
            img->color.transfer = PL_COLOR_TRC_GAMMA22; // arbitrary fallback
        pl_shader_delinearize(img_sh(pass, img), &img->color);
    }

    if (use_linear || use_sigmoid) {
        pl_shader_linearize(img_sh(pass, img), &img->color);
        img->color.transfer = PL_COLOR_TRC_LINEAR;
    }
    if (use_sigmoid)
        pl_shader_sigmoidize(img_sh(pass, img), params->sigmoid_params);

    // ---- PASS A: everything recorded so far lands in an FBO (or is fused, see above) ----
    pl_shader sh = pl_dispatch_begin(rr->dp);
    if (img->sh && try_fused_polar(pass, sh, &src, img->sh, img->w, img->h)) {
        pl_dispatch_abort(rr->dp, &img->sh);
    } else {
        src.tex = img_tex(pass, img);
        if (!src.tex) {
            pl_dispatch_abort(rr->dp, &sh);
            return false;
        }
        dispatch_sampler(pass, sh, &rr->sampler_main, SAMPLER_MAIN, &src);
    }
    pass->need_peak_fbo = false;

    img->tex  = NULL;
    img->sh   = sh;
    img->w    = src.new_w;
    img->h    = src.new_h;
    img->rect = new_rect;

    if (use_sigmoid)
        pl_shader_unsigmoidize(img_sh(pass, img), params->sigmoid_params);

done:
    if (info.dir != SAMPLER_UP)
        hdr_update_peak(pass);
    return true;
}

/* ---- pass_convert_colors (:2157-2280) ---------------------------------------------------------- */

// Low-resolution luminance map for the tone mapper's contrast recovery (get_feature_map
// :2089-2154): I of IPT at full resolution, low-passed (bicubic, mirrored edges) down to
// 1/contrast_smoothness of the output size.
static pl_tex get_feature_map(struct pass_state *pass)
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    const struct pl_color_map_params *cparams =
        PL_DEF(params->color_map_params, &pl_color_map_default_params);
    if (!cparams->contrast_recovery || cparams->contrast_smoothness <= 1)
        return NULL;
    if (!pass->fbofmt[4] || !pass->fbofmt[1])
        return NULL;
    if (!pl_color_space_is_hdr(&pass->img.color))
        return NULL;
    if (rr->errors & (PL_RENDER_ERR_SAMPLING | PL_RENDER_ERR_CONTRAST_RECOVERY))
        return NULL;
    if (pass->img.color.hdr.max_luma <= pass->target.color.hdr.max_luma + 1e-6)
        return NULL; // no adaptation needed
    if (params->lut && params->lut_type == PL_LUT_CONVERSION)
        return NULL; // LUT handles tone mapping

    struct img *img = &pass->img;
    if (!img_tex(pass, img))
        return NULL;

    const float ratio = cparams->contrast_smoothness;
    const int cr_w = ceilf(abs(pl_rect_w(pass->dst_rect)) / ratio);
    const int cr_h = ceilf(abs(pl_rect_h(pass->dst_rect)) / ratio);
    pl_tex inter_tex = get_fbo(pass, img->w, img->h, NULL, 1);
    pl_tex out_tex = get_fbo(pass, cr_w, cr_h, NULL, 1);
    if (!inter_tex || !out_tex)
        goto error;

    pl_shader sh = pl_dispatch_begin(rr->dp);
    pl_shader_sample_direct(sh, pl_sample_src( .tex = img->tex ));
    pl_shader_extract_features(sh, img->color);
    if (!pl_dispatch_finish(rr->dp, pl_dispatch_params(.shader = &sh, .target = inter_tex)))
        goto error;

    const struct pl_sample_src src = {
        .tex          = inter_tex,
        .rect         = img->rect,
        .address_mode = PL_TEX_ADDRESS_MIRROR,
        .components   = 1,
        .new_w        = cr_w,
        .new_h        = cr_h,
    };
    sh = pl_dispatch_begin(rr->dp);
    dispatch_sampler(pass, sh, &rr->sampler_contrast, SAMPLER_LOWPASS, &src);
    if (!pl_dispatch_finish(rr->dp, pl_dispatch_params(.shader = &sh, .target = out_tex)))
        goto error;
    return out_tex;

error:
    RR_ERR(rr, "Failed extracting luma for contrast recovery, disabling");
    rr->errors |= PL_RENDER_ERR_CONTRAST_RECOVERY;
    return NULL;
}

static void pass_convert_colors(struct pass_state *pass)
{
    const struct pl_render_params *params = pass->params;
    const struct pl_frame *image = &pass->image, *target = &pass->target;
    pl_renderer rr = pass->rr;
    struct img *img = &pass->img;

    // lut3d_tricubic exists in one variant of the generic pass kernel only (k_pass.hip): give the
    // colour conversion a pass of its own instead of fusing it into the pending sampler
    if (params->color_map_params && params->color_map_params->lut3d_tricubic && img->sh) {
        if (!img_tex(pass, img)) {
            RR_ERR(rr, "Failed flushing the image ahead of the tricubic colour map");
            return;
        }
    }
    pl_shader sh = img_sh(pass, img);

    bool prelinearized = false;
    if (img->color.transfer == PL_COLOR_TRC_LINEAR) {
        if (img->repr.alpha == PL_ALPHA_PREMULTIPLIED) {
            // prelinearization happened with premultiplied alpha, colour mapping wants
            // independent alpha: go back to the non-linear representation *before* the alpha
            // mode conversion, to avoid distortion
            img->color.transfer = image->color.transfer;
            pl_shader_delinearize(sh, &img->color);
        } else {
            prelinearized = true;
        }
    } else if (img->color.transfer != image->color.transfer) {
        if (image->color.transfer == PL_COLOR_TRC_LINEAR) {
            pl_shader_linearize(sh, &img->color);
            img->color.transfer = PL_COLOR_TRC_LINEAR;
        }
    }

    // all processing in independent alpha, to avoid nonlinear distortions
    pl_shader_set_alpha(sh, &img->repr, PL_ALPHA_INDEPENDENT);

    // colour blindness simulation (:2194-2196)
    if (params->cone_params)
        pl_shader_cone_distort(sh, img->color, params->cone_params);

    // ---- PASS B: a same-frame peak measurement must finish before it is consumed ----
    // main LUT (:2199-2247): between the image's and the target's colour space
    bool need_conversion = true;
    if (params->lut) {
        struct pl_color_space lut_in = params->lut->color_in;
        struct pl_color_space lut_out = params->lut->color_out;
        switch (params->lut_type) {
        case PL_LUT_UNKNOWN:
        case PL_LUT_NATIVE:
            pl_color_space_merge(&lut_in, &image->color);
            pl_color_space_merge(&lut_out, &image->color);
            break;
        case PL_LUT_CONVERSION:
            pl_color_space_merge(&lut_in, &image->color);
            need_conversion = false; // the LUT is the conversion
            break;
        case PL_LUT_NORMALIZED:
            if (!prelinearized) {
                // PL_LUT_NORMALIZED wants linear input data
                pl_shader_linearize(sh, &img->color);
                img->color.transfer = PL_COLOR_TRC_LINEAR;
                prelinearized = true;
            }
            pl_color_space_merge(&lut_in, &img->color);
            pl_color_space_merge(&lut_out, &img->color);
            break;
        }

        pl_shader_color_map_ex(sh, params->color_map_params, pl_color_map_args(
            .src = image->color, .dst = lut_in, .prelinearized = prelinearized));
        if (params->lut_type == PL_LUT_NORMALIZED) {
            struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
            if (op) {
                const float k = 1.0f / pl_color_transfer_nominal_peak(lut_in.transfer);
                op->f[0] = op->f[1] = op->f[2] = k;
                op->f[3] = 1.0f;
            }
        }
        pl_shader_custom_lut(sh, params->lut, &rr->lut_state[LUT_PARAMS]);
        if (params->lut_type == PL_LUT_NORMALIZED) {
            struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
            if (op) {
                const float k = pl_color_transfer_nominal_peak(lut_out.transfer);
                op->f[0] = op->f[1] = op->f[2] = k;
                op->f[3] = 1.0f;
            }
        }
        if (params->lut_type != PL_LUT_CONVERSION) {
            pl_shader_color_map_ex(sh, params->color_map_params, pl_color_map_args(
                .src = lut_out, .dst = img->color));
        }
    }

    if (need_conversion) {
        if (pass->need_peak_fbo && !img_tex(pass, img))
            return;

        // HDR feature map for the contrast recovery, if required (dispatches the image so far)
        pl_tex feature_map = get_feature_map(pass);
        sh = img_sh(pass, img);

        pl_shader_color_map_ex(sh, params->color_map_params, pl_color_map_args(
            .src           = image->color,
            .dst           = target->color,
            .prelinearized = prelinearized,
            .state         = &rr->tone_map_state,
            .feature_map   = feature_map,
        ));
    }

    // target LUT (:2272-2274): NORMALIZED / CONVERSION (RGB -> native) act while encoding
    const enum pl_lut_type tlut = guess_frame_lut_type(target, true);
    if (tlut == PL_LUT_NORMALIZED || tlut == PL_LUT_CONVERSION)
        pl_shader_custom_lut(sh, target->lut, &rr->lut_state[LUT_TARGET]);
    img->color = target->color;
}

/* ---- pass_output_target (:2586-2960) ------------------------------------------------------------ */

// sRGB background colour -> target colour space (translate_srgb_color :2557-2584)
static void translate_srgb_color(float out[3], const float in[3], const struct pl_color_space *csp)
{
    memcpy(out, in, 3 * sizeof(float));
    if (csp->primaries == PL_COLOR_PRIM_BT_709 && csp->transfer == PL_COLOR_TRC_SRGB)
        return;
    struct pl_color_space srgb = pl_color_space_srgb;
    pl_color_linearize(&srgb, out);
    if (csp->primaries != PL_COLOR_PRIM_BT_709) {
        const pl_matrix3x3 m = pl_get_color_mapping_matrix(
            pl_raw_primaries_get(PL_COLOR_PRIM_BT_709), pl_raw_primaries_get(csp->primaries),
            PL_INTENT_RELATIVE_COLORIMETRIC);
        pl_matrix3x3_apply(&m, out);
    }
    pl_color_delinearize(csp, out);
}

static void record_swizzle(pl_shader sh, int comps, const int mapping[4])
{
    // swizzle_color (:791-808): color = (0,0,0,1); color[c] = orig[mapping[c]]
    bool trivial = comps == 4;
    for (int c = 0; c < comps; c++)
        trivial &= mapping[c] == c;
    if (trivial)
        return;
    struct plh_op *op = sh_op(sh, PLH_OP_SWIZZLE);
    if (!op)
        return;
    op->i1 = comps;
    op->i0 = 0;
    for (int c = 0; c < 4; c++) {
        const int m = c < comps ? mapping[c] : -1;
        op->i0 |= (m < 0 ? 0xff : m) << (8 * c);
    }
    sh_listf(sh, "swizzle(comps=%d, map=0x%08x)\n", comps, (unsigned) op->i0);
}

// Returns true if error diffusion was performed (pass_error_diffusion :2282-2344)
static bool pass_error_diffusion(struct pass_state *pass, pl_shader *sh, int new_depth,
                                 int comps, int out_w, int out_h)
{
    const struct pl_render_params *params = pass->params;
    pl_renderer rr = pass->rr;
    if (!params->error_diffusion || (rr->errors & PL_RENDER_ERR_ERROR_DIFFUSION))
        return false;

    const size_t shmem_req = pl_error_diffusion_shmem_req(params->error_diffusion, out_h);
    if (shmem_req > rr->gpu->glsl.max_shmem_size)
        return false;

    pl_fmt fmt = pass->fbofmt[comps];
    if (!fmt || !(fmt->caps & PL_FMT_CAP_STORABLE)) {
        RR_ERR(rr, "Error diffusion requires storable FBOs.. disabling!");
        goto error;
    }

    struct pl_error_diffusion_params edpars = {
        .new_depth = new_depth,
        .kernel = params->error_diffusion,
    };
    edpars.input_tex = get_fbo(pass, out_w, out_h, fmt, comps);
    edpars.output_tex = get_fbo(pass, out_w, out_h, fmt, comps);
    if (!edpars.input_tex || !edpars.output_tex)
        goto error;

    pl_shader dsh = pl_dispatch_begin(rr->dp);
    if (!pl_shader_error_diffusion(dsh, &edpars)) {
        pl_dispatch_abort(rr->dp, &dsh);
        goto error;
    }

    bool ok = pl_dispatch_finish(rr->dp, pl_dispatch_params(
        .shader = sh,
        .target = edpars.input_tex,
    ));
    if (ok) {
        ok = pl_dispatch_compute(rr->dp, pl_dispatch_compute_params(
            .shader = &dsh,
            .dispatch_size = {1, 1, 1},
        ));
    } else {
        pl_dispatch_abort(rr->dp, &dsh);
    }

    *sh = pl_dispatch_begin(rr->dp);
    pl_shader_sample_direct(*sh, pl_sample_src(
        .tex = ok ? edpars.output_tex : edpars.input_tex,
    ));
    return ok;

error:
    rr->errors |= PL_RENDER_ERR_ERROR_DIFFUSION;
    return false;
}

// clear_target (:2410-2555), PL_CLEAR_COLOR flavour (tiles / blur degrade to it), every plane
static void clear_target(struct pass_state *pass, float scale)
{
    const struct pl_render_params *params = pass->params;
    const struct pl_frame *target = &pass->target;
    pl_renderer rr = pass->rr;
    float bg[3];
    translate_srgb_color(bg, params->background_color, &target->color);
    float enc[3] = { bg[0], bg[1], bg[2] };
    struct pl_color_repr crepr = target->repr;
    pl_transform3x3 tr = pl_color_repr_decode(&crepr, NULL);
    pl_transform3x3_invert(&tr);
    pl_transform3x3_apply(&tr, enc);
    const float alpha = 1.0 - params->background_transparency;
    for (int pi = 0; pi < target->num_planes; pi++) {
        const struct pl_plane *cp = &target->planes[pi];
        float clear[4];
        for (int c = 0; c < 4; c++) {
            const int m = c < cp->components ? cp->component_mapping[c] : -1;
            clear[c] = m == PL_CHANNEL_A ? alpha : m >= 0 && m < 3 ? enc[m] / scale : 0.0f;
        }
        pl_tex_clear(rr->gpu, cp->texture, clear);
    }
}

static bool pass_output_target(struct pass_state *pass)
{
    const struct pl_render_params *params = pass->params;
    const struct pl_frame *target = &pass->target;
    const struct pl_plane *plane = &target->planes[frame_ref(target)];
    pl_renderer rr = pass->rr;
    struct img *img = &pass->img;
    pl_shader sh = img_sh(pass, img);
    pl_rect2d dst_rect = pass->dst_rect;
    const bool need_clear = pl_frame_is_cropped(target);

    enum pl_clear_mode background = params->background;
    if (background == PL_CLEAR_TILES || background == PL_CLEAR_BLUR)
        background = PL_CLEAR_COLOR; // (unsupported modes degrade to a plain colour)

    // avoid an unnecessary round trip through premultiplied alpha
    const bool has_alpha = target->repr.alpha != PL_ALPHA_NONE;
    if (params->background_transparency >= 1.0 && has_alpha)
        background = PL_CLEAR_SKIP;

    const bool need_blend = background != PL_CLEAR_SKIP || !has_alpha;
    if (img->comps == 4 && need_blend) {
        pl_shader_set_alpha(sh, &img->repr, PL_ALPHA_PREMULTIPLIED);
        if (background == PL_CLEAR_COLOR) {
            float bg[3];
            translate_srgb_color(bg, params->background_color, &target->color);
            struct plh_op *op = sh_op(sh, PLH_OP_BLEND_BG);
            if (!op)
                return false;
            op->f[0] = bg[0]; op->f[1] = bg[1]; op->f[2] = bg[2];
            op->f[3] = 1.0 - params->background_transparency;
            sh_listf(sh, "blend_background(%g %g %g %g)\n", bg[0], bg[1], bg[2], op->f[3]);
            if (!params->background_transparency || !has_alpha) {
                img->repr.alpha = PL_ALPHA_NONE;
                img->comps = 3;
            }
        }
    }

    // the colour scale is applied separately, after encoding, so that an intermediate FBO
    // (error diffusion) has the right precision
    struct pl_color_repr repr = target->repr;
    const float scale = pl_color_repr_normalize(&repr);

    // don't double-apply an alpha mode that is already in effect
    if (img->repr.alpha == repr.alpha || img->comps < 4) {
        repr.alpha = PL_ALPHA_NONE;
    } else {
        pl_shader_set_alpha(sh, &img->repr, PL_ALPHA_INDEPENDENT);
    }

    // (a CONVERSION LUT on the target already produced native samples, pass_convert_colors)
    const enum pl_lut_type tlut = guess_frame_lut_type(target, true);
    if (tlut != PL_LUT_CONVERSION) {
        pl_shader_encode_color(sh, &repr);
        if (repr.sys == PL_COLOR_SYSTEM_XYZ) {
            img->color.transfer = PL_COLOR_TRC_ST428;
            pl_shader_delinearize(sh, &img->color);
        }
    }
    if (tlut == PL_LUT_NATIVE)
        pl_shader_custom_lut(sh, target->lut, &rr->lut_state[LUT_TARGET]);

    // rotation by an odd number of quarter turns (:2787-2793): back to the target's
    // orientation, the stores are transposed
    if (pass->rotation % PL_ROTATION_180 == PL_ROTATION_90) {
        int t;
        t = dst_rect.x0; dst_rect.x0 = dst_rect.y0; dst_rect.y0 = t;
        t = dst_rect.x1; dst_rect.x1 = dst_rect.y1; dst_rect.y1 = t;
        t = img->w; img->w = img->h; img->h = t;
        sh->transpose = true;
    }

    const bool flipped_x = dst_rect.x1 < dst_rect.x0, flipped_y = dst_rect.y1 < dst_rect.y0;

    if (need_clear && params->border != PL_CLEAR_SKIP)
        clear_target(pass, scale);

    pl_tex ref_tex = target->planes[frame_ref(target)].texture;
    pl_tex img_fbo = NULL;
    if (target->num_planes > 1) {
        // planar output: every plane samples the finished image from an intermediate FBO
        img->sh = sh;
        img_fbo = img_tex(pass, img);
        sh = NULL;
        if (!img_fbo) {
            RR_ERR(rr, "Output requires multiple planes, but FBOs are unavailable.");
            return false;
        }
    } else {
        img->sh = NULL;
    }

    bool ok = true;
    for (int pi = 0; pi < target->num_planes && ok; pi++) {
        plane = &target->planes[pi];
        const float prx = (float) plane->texture->params.w / ref_tex->params.w,
                    pry = (float) plane->texture->params.h / ref_tex->params.h;
        // integer subsampling ratios only (fractional sizes are rounded up: over-render)
        const float rrx = prx >= 1 ? roundf(prx) : 1.0 / roundf(1.0 / prx),
                    rry = pry >= 1 ? roundf(pry) : 1.0 / roundf(1.0 / pry);
        const float psx = plane->shift_x, psy = plane->shift_y;

        pl_rect2df plane_rectf = {
            .x0 = (dst_rect.x0 - psx) * rrx,
            .y0 = (dst_rect.y0 - psy) * rry,
            .x1 = (dst_rect.x1 - psx) * rrx,
            .y1 = (dst_rect.y1 - psy) * rry,
        };
        pl_rect2df_normalize(&plane_rectf);
        const int rx0 = floorf(plane_rectf.x0), ry0 = floorf(plane_rectf.y0),
                  rx1 =  ceilf(plane_rectf.x1), ry1 =  ceilf(plane_rectf.y1);

        if (target->num_planes > 1) {
            uint8_t mask = 0;
            for (int c = 0; c < plane->components; c++) {
                if (plane->component_mapping[c] >= 0)
                    mask |= 1 << plane->component_mapping[c];
            }
            struct pl_sample_src src = {
                .tex        = img_fbo,
                .new_w      = rx1 - rx0,
                .new_h      = ry1 - ry0,
                .rect = {
                    .x0 = (rx0 - plane_rectf.x0) / rrx,
                    .x1 = (rx1 - plane_rectf.x0) / rrx,
                    .y0 = (ry0 - plane_rectf.y0) / rry,
                    .y1 = (ry1 - plane_rectf.y0) / rry,
                },
                .component_mask = mask,
            };
            sh = pl_dispatch_begin(rr->dp);
            dispatch_sampler(pass, sh, &rr->samplers_dst[pi], SAMPLER_PLANE, &src);
        }

        // > 16-bit outputs are not dithered by default (:2884-2900)
        const int depth = target->repr.bits.color_depth;
        int applied_dither = 0;
        if (depth && (depth < 16 || params->force_dither)) {
            if (pass_error_diffusion(pass, &sh, depth, plane->components, rx1 - rx0, ry1 - ry0)) {
                applied_dither = depth;
            } else if (params->dither_params) {
                struct pl_dither_params dparams = *params->dither_params;
                if (!params->disable_dither_gamma_correction)
                    dparams.transfer = target->color.transfer;
                pl_shader_dither(sh, depth, &rr->dither_state, &dparams);
                applied_dither = depth;
            }
        }
        if (applied_dither != rr->prev_dither) {
            if (applied_dither) {
                RR_INFO(rr, "Dithering to %d bit depth", applied_dither);
            } else {
                RR_INFO(rr, "Dithering disabled");
            }
            rr->prev_dither = applied_dither;
        }

        // color *= 1 / scale                                                          (:2911)
        struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
        if (!op) {
            pl_dispatch_abort(rr->dp, &sh);
            return false;
        }
        op->f[0] = op->f[1] = op->f[2] = op->f[3] = 1.0f / scale;
        sh_listf(sh, "scale(1/%g)\n", scale);

        record_swizzle(sh, plane->components, plane->component_mapping);

        pl_rect2d plane_rect = {
            .x0 = flipped_x ? rx1 : rx0,
            .x1 = flipped_x ? rx0 : rx1,
            .y0 = flipped_y ? ry1 : ry0,
            .y1 = flipped_y ? ry0 : ry1,
        };
        if (plane->flipped) {
            const int plane_h = rry * ref_tex->params.h;
            plane_rect.y0 = plane_h - plane_rect.y0;
            plane_rect.y1 = plane_h - plane_rect.y1;
        }

        ok = pl_dispatch_finish(rr->dp, pl_dispatch_params(
            .shader = &sh,
            .target = plane->texture,
            .rect = plane_rect,
        ));
    }
    *img = (struct img) {0};
    return ok;
}

/* ---- entry point (:3433-3480) --------------------------------------------------------------------- */

static void pass_uninit(struct pass_state *pass)
{
    pl_renderer rr = pass->rr;
    pl_dispatch_abort(rr->dp, &pass->img.sh);
    pl_dispatch_callback(rr->dp, NULL, NULL);
    if (pass->acquired_image && pass->image.release)
        pass->image.release(rr->gpu, &pass->image);
    if (pass->acquired_target && pass->target.release)
        pass->target.release(rr->gpu, &pass->target);
}

static bool unsupported(pl_renderer rr, const struct pl_render_params *p)
{
    if (p->blend_params || p->deinterlace_params || p->distort_params ||
        p->num_hooks)
    {
        RR_ERR(rr, "pl_render_params requests a stage outside this backend's hot path "
               "(blend / deinterlace / distort / hooks)");
        return true;
    }
    return false;
}

// acquire + validate + infer (pass_init :3391-3428)
static bool pass_init(struct pass_state *pass, bool acquire_image)
{
    pl_renderer rr = pass->rr;
    if (!pass->acquired_target && pass->target.acquire) {
        if (!pass->target.acquire(rr->gpu, &pass->target))
            return false;
        pass->acquired_target = true;
    }
    if (acquire_image && pass->image.acquire) {
        if (!pass->image.acquire(rr->gpu, &pass->image)) {
            pass_uninit(pass);
            return false;
        }
        pass->acquired_image = true;
    }
    if (!validate_frame(rr, &pass->image, "Image", false) ||
        !validate_frame(rr, &pass->target, "Target", true))
    {
        pass_uninit(pass);
        return false;
    }
    find_fbo_format(pass);
    pass_fix_frames(pass);
    return true;
}

bool pl_render_image(pl_renderer rr, const struct pl_frame *pimage, const struct pl_frame *ptarget,
                     const struct pl_render_params *params)
{
    params = PL_DEF(params, &pl_render_default_params);
    if (!ptarget) {
        RR_ERR(rr, "pl_render_image: a target is required");
        return false;
    }
    if (unsupported(rr, params))
        return false;
    if (!pimage) {
        // no image (:3463-3476): the target is cleared (there are no overlays to draw here)
        struct pass_state pass = { .rr = rr, .params = params, .target = *ptarget };
        if (pass.target.acquire) {
            if (!pass.target.acquire(rr->gpu, &pass.target))
                return false;
            pass.acquired_target = true;
        }
        bool ok = validate_frame(rr, &pass.target, "Target", true);
        if (ok) {
            fix_frame(&pass.target);
            pl_color_space_infer(&pass.target.color);
            struct pl_color_repr repr = pass.target.repr;
            clear_target(&pass, pl_color_repr_normalize(&repr));
        }
        pass_uninit(&pass);
        return ok;
    }

    struct pass_state pass = {
        .rr = rr,
        .params = params,
        .image = *pimage,
        .target = *ptarget,
    };
    if (!pass_init(&pass, true))
        return false;

    // no-op (empty crop)
    if (!pl_rect_w(pass.dst_rect) || !pl_rect_h(pass.dst_rect)) {
        pass_uninit(&pass);
        return true;
    }

    pl_dispatch_reset_frame(rr->dp);
    pl_dispatch_callback(rr->dp, &pass, info_callback);
    if (!pass_read_image(&pass))
        goto error;
    if (!pass_scale_main(&pass))
        goto error;
    pass_convert_colors(&pass);
    if (!pass.img.sh && !pass.img.tex)
        goto error;
    if (!pass_output_target(&pass))
        goto error;

    pass_uninit(&pass);
    return true;

error:
    RR_ERR(rr, "Failed rendering image!");
    pass_uninit(&pass);
    return false;
}

/* ---- frame mixing (pl_render_image_mix, renderer.c:3477-3508, 3612-4028) ---------------------- */

const struct pl_frame *pl_frame_mix_current(const struct pl_frame_mix *mix)
{
    const struct pl_frame *cur = NULL;
    for (int i = 0; i < mix->num_frames && mix->timestamps[i] <= 0.0f; i++)
        cur = mix->frames[i];
    return cur;
}

const struct pl_frame *pl_frame_mix_nearest(const struct pl_frame_mix *mix)
{
    if (!mix->num_frames)
        return NULL;
    // timestamps are sorted: |ts| falls, then rises
    int best = 0;
    for (int i = 1; i < mix->num_frames; i++) {
        if (fabsf(mix->timestamps[i]) < fabsf(mix->timestamps[best]))
            best = i;
        else
            break;
    }
    return mix->frames[best];
}

// Everything in pl_render_params that changes how a cached frame looks (render_params_info
// :3510-3560 hashes the same set): the struct itself minus callbacks, plus what it points to.
static uint64_t fnv1a(uint64_t h, const void *data, size_t size)
{
    const uint8_t *p = data;
    for (size_t i = 0; i < size; i++)
        h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

static uint64_t params_hash(const struct pl_render_params *params)
{
    struct pl_render_params p = *params;
    p.info_callback = NULL;
    p.info_priv = NULL;
    uint64_t h = 0xcbf29ce484222325ull;
#define HASH_PTR(field)                                         \
    do {                                                        \
        if (p.field)                                            \
            h = fnv1a(h, p.field, sizeof(*p.field));            \
        p.field = NULL;                                         \
    } while (0)
    HASH_PTR(upscaler); HASH_PTR(downscaler); HASH_PTR(plane_upscaler); HASH_PTR(plane_downscaler);
    HASH_PTR(frame_mixer); HASH_PTR(deband_params); HASH_PTR(sigmoid_params);
    HASH_PTR(color_adjustment); HASH_PTR(peak_detect_params); HASH_PTR(color_map_params);
    HASH_PTR(dither_params); HASH_PTR(error_diffusion); HASH_PTR(cone_params);
#undef HASH_PTR
    return fnv1a(h, &p, sizeof(p));
}

static bool rect2df_eq(pl_rect2df a, pl_rect2df b)
{
    return a.x0 == b.x0 && a.y0 == b.y0 && a.x1 == b.x1 && a.y1 == b.y1;
}

// `color = texel of frame` for every frame but the first (which the pass' sampler reads)
static bool mix_fetch(pl_renderer rr, pl_shader sh, pl_tex tex, bool linear)
{
    pl_shader psh = pl_dispatch_begin(rr->dp);
    const struct pl_sample_src src = { .tex = tex };
    bool ok = linear ? pl_shader_sample_bilinear(psh, &src) : pl_shader_sample_nearest(psh, &src);
    const struct pl_plane whole = { .components = 4, .component_mapping = { 0, 1, 2, 3 } };
    ok = ok && merge_plane_fetch(sh, psh, &whole);
    pl_dispatch_abort(rr->dp, &psh);
    return ok;
}

bool pl_render_image_mix(pl_renderer rr, const struct pl_frame_mix *images,
                         const struct pl_frame *ptarget, const struct pl_render_params *params)
{
    params = PL_DEF(params, &pl_render_default_params);
    if (!images || !images->num_frames)
        return pl_render_image(rr, NULL, ptarget, params);
    if (unsupported(rr, params))
        return false;
    if (!(images->vsync_duration > 0.0f)) {
        RR_ERR(rr, "pl_render_image_mix: vsync_duration must be positive");
        return false;
    }
    for (int i = 0; i + 1 < images->num_frames; i++) {
        if (!(images->timestamps[i] <= images->timestamps[i + 1])) {
            RR_ERR(rr, "pl_render_image_mix: timestamps must be sorted");
            return false;
        }
    }

    const uint64_t phash = params_hash(params);
    const struct pl_frame *refimg = pl_frame_mix_nearest(images);
    struct pass_state pass = {
        .rr = rr,
        .params = params,
        .image = *refimg,
        .target = *ptarget,
        .info.stage = PL_RENDER_STAGE_BLEND,
    };

    if (rr->errors & PL_RENDER_ERR_FRAME_MIXING)
        goto fallback;
    if (!pass_init(&pass, false))
        return false;
    if (!pass.fbofmt[4])
        goto fallback;

    const struct pl_frame *target = &pass.target;
    const int out_w = abs(pl_rect_w(pass.dst_rect)), out_h = abs(pl_rect_h(pass.dst_rect));
    if (!out_w || !out_h)
        goto fallback;

    int fidx = 0;
    struct cached_frame frames[MAX_MIX_FRAMES];
    float weights[MAX_MIX_FRAMES];
    float wsum = 0.0f;

    // garbage collection: everything not touched below is evicted
    for (int i = 0; i < rr->num_frames; i++)
        rr->frames[i].evict = true;

    // blur the mixer by the vsync ratio (source / display)
    struct pl_filter_config mixer = {0};
    if (params->frame_mixer) {
        mixer = *params->frame_mixer;
        mixer.blur = PL_DEF(mixer.blur, 1.0f);
        for (int i = 1; i < images->num_frames; i++) {
            if (images->timestamps[i] >= 0.0f && images->timestamps[i - 1] < 0.0f) {
                const float frame_dur = images->timestamps[i] - images->timestamps[i - 1];
                const float sample_dur = PL_MAX(frame_dur, images->vsync_duration);
                if (sample_dur > 1.0f && !params->skip_anti_aliasing)
                    mixer.blur *= sample_dur;
                break;
            }
        }
    }

    bool single_frame = !params->frame_mixer || images->num_frames == 1;
retry:
    for (int i = 0; i < images->num_frames; i++) {
        const uint64_t sig = images->signatures[i];
        float rts = images->timestamps[i];
        const struct pl_frame *img = images->frames[i];
        if (img->rotation != refimg->rotation)
            continue; // (only PL_ROTATION_0 passes validation anyway)

        float weight;
        if (single_frame) {
            // only the reference image is rendered
            if (img != refimg)
                continue;
            weight = 1.0f;
        } else if (!mixer.kernel || mixer.kernel == &pl_filter_function_oversample) {
            // weight = fraction of the vsync interval during which the frame is visible
            float end = i + 1 < images->num_frames ? images->timestamps[i + 1] : INFINITY;
            if (rts > images->vsync_duration || end < 0.0f)
                continue;
            rts = PL_MAX(rts, 0.0f);
            end = PL_MIN(end, images->vsync_duration);
            weight = (end - rts) / images->vsync_duration;
            if (mixer.kernel && weight < mixer.kernel->params[0])
                weight = 0.0f; // culled by the oversampling threshold
        } else {
            if (fabsf(rts) >= pl_filter_radius_bound(&mixer))
                continue;
            weight = pl_filter_sample(&mixer, rts);
        }

        struct cached_frame *f = NULL;
        for (int j = 0; j < rr->num_frames; j++) {
            if (rr->frames[j].signature == sig) {
                f = &rr->frames[j];
                f->evict = false;
                break;
            }
        }

        // negligible contributions are skipped -- after the lookup, so that these frames are
        // not evicted yet; never the reference image (at least one frame must remain)
        if (fabsf(weight) <= 1e-3f && img != refimg)
            continue;

        // (the reference also bypasses the cache for "trivial" params; that only saves a copy)
        const bool skip_cache = single_frame && params->skip_caching_single_frame;
        if (!f && skip_cache)
            goto fallback;

        if (!f) {
            if (rr->num_frames == MAX_CACHED_FRAMES) {
                PL_WARN_RR(rr, "Frame mixing cache is full, rendering without mixing");
                goto fallback;
            }
            f = &rr->frames[rr->num_frames++];
            *f = (struct cached_frame) { .signature = sig };
        }

        bool can_reuse = f->tex;
        const bool strict_reuse = skip_cache || single_frame || !params->preserve_mixing_cache;
        if (can_reuse && strict_reuse) {
            can_reuse = f->tex->params.w == out_w && f->tex->params.h == out_h &&
                        rect2df_eq(f->crop, img->crop) && f->params_hash == phash &&
                        pl_color_space_equal(&f->color, &target->color);
        }
        if (!can_reuse && skip_cache)
            goto fallback;

        if (!can_reuse) {
            // (re-)render this frame, up to where pass_output_target would take over
            if (!f->tex && rr->num_frame_fbos)
                f->tex = rr->frame_fbos[--rr->num_frame_fbos];
            pl_fmt fmt = pass.fbofmt[4];
            if (!pl_tex_recreate(rr->gpu, &f->tex, pl_tex_params(
                    .w = out_w, .h = out_h, .format = fmt,
                    .sampleable = true, .renderable = true, .storable = true,
                    .blit_dst = !!(fmt->caps & PL_FMT_CAP_BLITTABLE))))
            {
                RR_ERR(rr, "Could not create intermediate texture for frame mixing.. disabling!");
                rr->errors |= PL_RENDER_ERR_FRAME_MIXING;
                goto fallback;
            }

            struct pass_state inter = {
                .rr = rr,
                .params = params,
                .image = *img,
                .target = *ptarget,
                .info.stage = PL_RENDER_STAGE_FRAME,
                .acquired_target = pass.acquired_target, // (already acquired by `pass`)
            };
            if (!pass_init(&inter, true))
                goto fail;
            inter.acquired_target = false; // released by `pass`
            inter.target = pass.target;

            pl_dispatch_reset_frame(rr->dp);
            pl_dispatch_callback(rr->dp, &inter, info_callback);
            bool ok = pass_read_image(&inter) && pass_scale_main(&inter);
            if (ok) {
                pass_convert_colors(&inter);
                ok = inter.img.sh || inter.img.tex;
            }
            if (ok) {
                pl_shader sh = img_sh(&inter, &inter.img);
                pl_shader_set_alpha(sh, &inter.img.repr, PL_ALPHA_PREMULTIPLIED); // for mixing
                ok = inter.img.w == out_w && inter.img.h == out_h &&
                     pl_dispatch_finish(rr->dp, pl_dispatch_params(
                         .shader = &inter.img.sh, .target = f->tex));
            }
            if (ok) {
                f->params_hash = phash;
                f->crop = img->crop;
                f->color = inter.img.color;
                f->repr = inter.img.repr;
                f->comps = inter.img.comps;
            }
            pass_uninit(&inter);
            if (!ok)
                goto fail;
        }

        if (fidx == MAX_MIX_FRAMES)
            break;
        frames[fidx] = *f;
        weights[fidx] = weight;
        wsum += weight;
        fidx++;
    }

    // evict what this mix did not touch
    for (int i = 0; i < rr->num_frames; ) {
        if (!rr->frames[i].evict) {
            i++;
            continue;
        }
        if (rr->frames[i].tex) {
            if (rr->num_frame_fbos < MAX_CACHED_FRAMES)
                rr->frame_fbos[rr->num_frame_fbos++] = rr->frames[i].tex;
            else
                pl_tex_destroy(rr->gpu, &rr->frames[i].tex);
        }
        rr->frames[i] = rr->frames[--rr->num_frames];
    }

    // nothing left: zero-order hold
    if (!fidx) {
        if (single_frame)
            goto fallback;
        single_frame = true;
        goto retry;
    }

    // ---- sample and mix --------------------------------------------------------------------
    pl_dispatch_reset_frame(rr->dp);
    pl_dispatch_callback(rr->dp, &pass, info_callback);
    pass.info.count = fidx;

    pl_shader sh = pl_dispatch_begin(rr->dp);
    // with a single frame there is nothing to mix: no linearize / delinearize round trip
    const bool mixing = fidx > 1;
    struct pl_color_space mix_csp = target->color;
    if (mixing)
        mix_csp.transfer = PL_COLOR_TRC_LINEAR;

    int comps = 0;
    bool ok = true;
    for (int i = 0; i < fidx && ok; i++) {
        const struct pl_tex_params *tp = &frames[i].tex->params;
        const bool linear = (tp->w != out_w || tp->h != out_h) &&
                            (tp->format->caps & PL_FMT_CAP_LINEAR);
        if (i == 0) {
            const struct pl_sample_src src = { .tex = frames[i].tex, .new_w = out_w, .new_h = out_h };
            ok = linear ? pl_shader_sample_bilinear(sh, &src) : pl_shader_sample_nearest(sh, &src);
        } else {
            ok = mix_fetch(rr, sh, frames[i].tex, linear);
        }
        if (!ok)
            break;

        // usually just the linearization; handles mixed-colorspace frames when
        // preserve_mixing_cache spans target changes (differences in HDR metadata are ignored)
        struct pl_color_repr frame_repr = frames[i].repr;
        struct pl_color_space frame_csp = frames[i].color;
        frame_csp.hdr = mix_csp.hdr;
        if (!pl_color_space_equal(&frame_csp, &mix_csp)) {
            pl_shader_set_alpha(sh, &frame_repr, PL_ALPHA_INDEPENDENT);
            pl_shader_color_map_ex(sh, NULL, pl_color_map_args(.src = frame_csp, .dst = mix_csp));
        }
        pl_shader_set_alpha(sh, &frame_repr, PL_ALPHA_PREMULTIPLIED);

        if (mixing) {
            struct plh_op *op = sh_op(sh, PLH_OP_MIX_ADD);
            if (!op) {
                ok = false;
                break;
            }
            op->f[0] = weights[i] / wsum;
            sh_listf(sh, "mix_color += %g * color\n", op->f[0]);
        }
        comps = PL_MAX(comps, frames[i].comps);
    }
    if (ok && mixing) {
        ok = !!sh_op(sh, PLH_OP_MIX_END);
        sh_listf(sh, "color = mix_color\n");
    }
    if (!ok || pl_shader_is_failed(sh)) {
        // (more frames than one pass can hold ops for)
        PL_WARN_RR(rr, "Frame mixing pass could not be recorded (%d frames), rendering the "
                   "nearest frame instead", fidx);
        pl_dispatch_abort(rr->dp, &sh);
        goto fallback;
    }
    sh_describef(sh, "frame mixing (%d frame%s)", fidx, fidx > 1 ? "s" : "");

    pass.img = (struct img) {
        .sh = sh,
        .w = out_w,
        .h = out_h,
        .comps = comps,
        .color = target->color,
        .rect = { 0, 0, out_w, out_h },
        .repr = {
            .sys = PL_COLOR_SYSTEM_RGB,
            .levels = PL_COLOR_LEVELS_FULL,
            .alpha = comps >= 4 ? PL_ALPHA_PREMULTIPLIED : PL_ALPHA_NONE,
        },
    };

    // re-encode to the target transfer (in practice: delinearize)
    if (!pl_color_space_equal(&mix_csp, &pass.img.color)) {
        pl_shader_set_alpha(sh, &pass.img.repr, PL_ALPHA_INDEPENDENT);
        pl_shader_color_map_ex(sh, NULL, pl_color_map_args(.src = mix_csp, .dst = pass.img.color));
    }

    if (!pass_output_target(&pass))
        goto fallback;

    pass_uninit(&pass);
    return true;

fail:
    RR_ERR(rr, "Could not render image for frame mixing.. disabling!");
    rr->errors |= PL_RENDER_ERR_FRAME_MIXING;
    // fall through

fallback:
    pass_uninit(&pass);
    return pl_render_image(rr, refimg, ptarget, params);
}

void pl_frames_infer_mix(pl_renderer rr, const struct pl_frame_mix *mix, struct pl_frame *target,
                         struct pl_frame *out_ref)
{
    const struct pl_frame *refimg = pl_frame_mix_nearest(mix);
    if (!refimg) {
        if (out_ref)
            *out_ref = (struct pl_frame) {0};
        return;
    }
    struct pl_frame ref = *refimg;
    pl_frames_infer(rr, &ref, target);
    if (out_ref)
        *out_ref = ref;
}

/* Test hook: record `color *= s` the way pass_output_target does (there is no public
 * pl_shader_* entry point for it); used by tests/test_gpu_renderer.py to rebuild the
 * renderer's passes by hand. */
PL_API void plh_test_op_scale(pl_shader sh, float s);
void plh_test_op_scale(pl_shader sh, float s)
{
    struct plh_op *op = sh_op(sh, PLH_OP_SCALE);
    if (op)
        op->f[0] = op->f[1] = op->f[2] = op->f[3] = s;
}
