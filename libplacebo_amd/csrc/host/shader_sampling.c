/*
 * libplacebo-hip — sampling stages: host half of src/shaders/sampling.c.
 *
 * Each pl_shader_sample_* fills the sampler part of the recorded pass. The
 * generation-time decisions the reference makes while emitting GLSL are made
 * here with the same arithmetic:
 *   setup_src                 sampling.c:45-181   ratios, scale, component mask
 *   polar filter + widening   sampling.c:608-631
 *   polar tap pruning/order   sampling.c:503-523 (flags), :776-783 (compute
 *                             order), :798-893 (gather order)
 *   LDS tile size             sampling.c:661-699
 *   ortho filter / LUT        sampling.c:914-942, 1004-1063
 *   deband constants          sampling.c:183-275
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <libplacebo/shaders/sampling.h>

#include "shaders_priv.h"

const struct pl_deband_params pl_deband_default_params = { PL_DEBAND_DEFAULTS };

enum filter_req { REQ_NEAREST, REQ_LINEAR, REQ_BEST, REQ_FASTEST };

struct src_info {
    float ratio_x, ratio_y;
    float scale;
    uint8_t comp_mask;
    bool linear; // bound with LINEAR filtering
};

// Common source setup; mirrors setup_src (sampling.c:45-181)
static bool setup_src(pl_shader sh, const struct pl_sample_src *src, struct src_info *out,
                      bool resizeable, enum filter_req req)
{
    if (!src->tex) {
        SH_FAIL(sh, "pl_sample_src without `tex`: external samplers are not "
                "supported by the HIP backend");
        return false;
    }

    pl_fmt fmt = src->tex->params.format;
    const bool can_linear = fmt->caps & PL_FMT_CAP_LINEAR;
    if (req == REQ_LINEAR && !can_linear) {
        SH_FAIL(sh, "Trying to use a shader that requires linear sampling with a "
                "texture whose format (%s) does not support PL_FMT_CAP_LINEAR", fmt->name);
        return false;
    }
    out->linear = req == REQ_LINEAR || (req == REQ_BEST && can_linear);

    float src_w = pl_rect_w(src->rect), src_h = pl_rect_h(src->rect);
    src_w = PL_DEF(src_w, (float) src->tex->params.w);
    src_h = PL_DEF(src_h, (float) src->tex->params.h);

    const int out_w = PL_DEF(src->new_w, (int) roundf(fabsf(src_w)));
    const int out_h = PL_DEF(src->new_h, (int) roundf(fabsf(src_h)));
    if (!out_w || !out_h) {
        SH_FAIL(sh, "Degenerate output size %dx%d", out_w, out_h);
        return false;
    }

    out->ratio_x = out_w / fabs(src_w);
    out->ratio_y = out_h / fabs(src_h);
    out->scale = PL_DEF(src->scale, 1.0);

    const uint8_t tex_mask = (1 << fmt->num_components) - 1;
    uint8_t src_mask = src->component_mask;
    if (!src_mask)
        src_mask = (1 << PL_DEF(src->components, 4)) - 1;
    out->comp_mask = tex_mask & src_mask;

    if (sh->pass.s.type != PLH_SAMPLE_NONE || sh->output != PL_SHADER_SIG_NONE) {
        SH_FAIL(sh, "Illegal sequence of shader operations: a sampling stage must "
                "be the first stage of a shader");
        return false;
    }
    if (!sh_require(sh, PL_SHADER_SIG_NONE, resizeable ? 0 : out_w, resizeable ? 0 : out_h))
        return false;

    const pl_rect2df rect = {
        .x0 = src->rect.x0,
        .y0 = src->rect.y0,
        .x1 = src->rect.x0 + src_w,
        .y1 = src->rect.y0 + src_h,
    };
    if (!sh_bind(sh, src->tex, src->address_mode, &rect))
        return false;

    sh->pass.s.scale = out->scale;
    sh->pass.s.comp_mask = out->comp_mask;
    sh->pass.s.linear = out->linear;
    return true;
}

static bool sample_simple(pl_shader sh, const struct pl_sample_src *src, enum filter_req req,
                          const char *desc)
{
    struct src_info info;
    if (!setup_src(sh, src, &info, true, req))
        return false;
    // 1:1 sampling on the texel grid (what img_sh()/PASS A does): a texture unit
    // returns the texel itself there (its fixed-point lerp weights snap to 0),
    // whereas an exact fp32 lerp would blend in ~1e-5 of a neighbour from the
    // rounding noise in `pos`. Lower such identity fetches to nearest.
    const bool identity = fabsf(info.ratio_x - 1.0f) < 1e-6f && fabsf(info.ratio_y - 1.0f) < 1e-6f &&
                          src->rect.x0 == truncf(src->rect.x0) &&
                          src->rect.y0 == truncf(src->rect.y0);
    sh->pass.s.type = info.linear && !identity ? PLH_SAMPLE_BILINEAR : PLH_SAMPLE_NEAREST;
    info.linear &= !identity;
    if (desc)
        sh_describef(sh, "%s", desc);
    sh_listf(sh, "sample_%s(tex=%dx%d %s, scale=%g)\n", info.linear ? "bilinear" : "nearest",
             src->tex->params.w, src->tex->params.h, src->tex->params.format->name, info.scale);
    return true;
}

bool pl_shader_sample_direct(pl_shader sh, const struct pl_sample_src *src)
{
    return sample_simple(sh, src, REQ_BEST, NULL);
}

bool pl_shader_sample_nearest(pl_shader sh, const struct pl_sample_src *src)
{
    return sample_simple(sh, src, REQ_NEAREST, "nearest");
}

bool pl_shader_sample_bilinear(pl_shader sh, const struct pl_sample_src *src)
{
    return sample_simple(sh, src, REQ_LINEAR, "bilinear");
}

static bool sample_fast(pl_shader sh, const struct pl_sample_src *src, int type, const char *name)
{
    struct src_info info;
    if (!setup_src(sh, src, &info, true, REQ_LINEAR))
        return false;
    if (info.ratio_x < 1 || info.ratio_y < 1) {
        pl_msg(sh->log, PL_LOG_TRACE, "Using fast %s sampling when downscaling. This "
               "will most likely result in nasty aliasing!", name);
    }
    sh->pass.s.type = type;
    sh->pass.s.ratio[0] = info.ratio_x;
    sh->pass.s.ratio[1] = info.ratio_y;
    sh_describef(sh, "%s", name);
    sh_listf(sh, "sample_%s(scale=%g)\n", name, info.scale);
    return true;
}

bool pl_shader_sample_bicubic(pl_shader sh, const struct pl_sample_src *src)
{
    return sample_fast(sh, src, PLH_SAMPLE_BICUBIC, "bicubic");
}

bool pl_shader_sample_hermite(pl_shader sh, const struct pl_sample_src *src)
{
    return sample_fast(sh, src, PLH_SAMPLE_HERMITE, "hermite");
}

bool pl_shader_sample_gaussian(pl_shader sh, const struct pl_sample_src *src)
{
    return sample_fast(sh, src, PLH_SAMPLE_GAUSSIAN, "gaussian");
}

bool pl_shader_sample_oversample(pl_shader sh, const struct pl_sample_src *src, float threshold)
{
    if (!sample_fast(sh, src, PLH_SAMPLE_OVERSAMPLE, "oversample"))
        return false;
    sh->pass.s.threshold = PL_CLAMP(threshold, 0.0f, 0.5f);
    return true;
}

/* ------------------------------------------------------------------------ */
/* complex (LUT based) scalers                                               */

#define SCALER_LUT_SIZE     256
#define SCALER_LUT_CUTOFF   1e-3f

struct sh_sampler_obj {
    pl_filter filter;
    pl_buf lut;         // polar: 256 {L[i], L[i+1]} pairs; ortho: rows
    pl_buf taps;        // polar: packed tap list
    int num_taps;
    bool taps_gather;   // tap order the list was generated for
    pl_shader_obj pass2; // second ortho pass
};

static void sh_sampler_uninit(pl_gpu gpu, void *ptr)
{
    struct sh_sampler_obj *obj = ptr;
    pl_buf_destroy(gpu, &obj->lut);
    pl_buf_destroy(gpu, &obj->taps);
    pl_shader_obj_destroy(&obj->pass2);
    pl_filter_free(&obj->filter);
    memset(obj, 0, sizeof(*obj));
}

static void describe_filter(pl_shader sh, const struct pl_filter_config *cfg,
                            const char *stage, float rx, float ry)
{
    const char *dir = rx > 1 && ry > 1 ? "up" : rx < 1 && ry < 1 ? "down"
                    : rx == 1 && ry == 1 ? "noop" : "ana";
    if (cfg->name) {
        sh_describef(sh, "%s %sscaling (%s)", stage, dir, cfg->name);
    } else if (cfg->window) {
        sh_describef(sh, "%s %sscaling (%s+%s)", stage, dir,
                     PL_DEF(cfg->kernel->name, "unknown"), PL_DEF(cfg->window->name, "unknown"));
    } else {
        sh_describef(sh, "%s %sscaling (%s)", stage, dir, PL_DEF(cfg->kernel->name, "unknown"));
    }
}

// Flags of one polar tap, or -1 if it is pruned at generation time
// (polar_sample, sampling.c:508-522)
static int polar_tap_flags(pl_filter filter, int x, int y, bool use_ar)
{
    const int yy = y > 0 ? y - 1 : y;
    const int xx = x > 0 ? x - 1 : x;
    const float dmin = sqrt(xx * xx + yy * yy);
    if (dmin >= filter->radius)
        return -1;
    int fl = 0;
    if (dmin >= filter->radius - M_SQRT2)
        fl |= PLH_TAP_SKIPPABLE;
    if (use_ar && dmin < filter->radius_zero)
        fl |= PLH_TAP_AR;
    return fl;
}

static int add_tap(uint32_t *taps, int n, pl_filter filter, int x, int y, bool use_ar)
{
    const int fl = polar_tap_flags(filter, x, y, use_ar);
    if (fl >= 0)
        taps[n++] = PLH_TAP_PACK(x, y, fl);
    return n;
}

// Evaluation order of the compute-shader formulation (sampling.c:776-783)
static int polar_taps_compute(uint32_t *taps, pl_filter filter, int bound, bool use_ar)
{
    int n = 0;
    for (int y = 1 - bound; y <= bound; y++) {
        for (int x = 1 - bound; x <= bound; x++)
            n = add_tap(taps, n, filter, x, y, use_ar);
    }
    return n;
}

// Evaluation order of the textureGather formulation (sampling.c:798-893),
// which the reference uses for radius >= 6 or when compute is unavailable
static int polar_taps_gather(uint32_t *taps, pl_filter filter, int bound, bool use_ar,
                             const struct pl_glsl_version *glsl)
{
    int n = 0;
    uint64_t gathered_cur = 0x0, gathered_next = 0x0;
    const float radius2 = PL_SQUARE(filter->radius);
    const int base = bound - 1;

    for (int y = 1 - bound; y <= bound; y++) {
        for (int x = 1 - bound; x <= bound; x++) {
            const uint64_t bit = 1llu << (base + x);
            if (gathered_cur & bit)
                continue; // fetched by the previous row's gather

            const int xx = x * x, xx1 = (x + 1) * (x + 1);
            const int yy = y * y, yy1 = (y + 1) * (y + 1);
            bool use_gather = PL_MAX(xx, xx1) + PL_MAX(yy, yy1) < radius2;
            use_gather &= PL_MAX(x, y) <= glsl->max_gather_offset;
            use_gather &= PL_MIN(x, y) >= glsl->min_gather_offset;
            if (!use_gather) {
                n = add_tap(taps, n, filter, x, y, use_ar);
                continue;
            }

            // 2x2 quad, counter-clockwise from the bottom left
            static const int xo[4] = {0, 1, 1, 0};
            static const int yo[4] = {1, 1, 0, 0};
            for (int p = 0; p < 4; p++) {
                if (x + xo[p] > bound || y + yo[p] > bound)
                    continue;
                if (!yo[p] && (gathered_cur & (bit << xo[p])))
                    continue;
                n = add_tap(taps, n, filter, x + xo[p], y + yo[p], use_ar);
            }

            gathered_next |= bit | (bit << 1);
            x++;
        }
        gathered_cur = gathered_next;
        gathered_next = 0;
    }
    return n;
}

// Output tile of the polar kernel (csrc/hip/k_polar.hip): 32 columns, 8 lanes
// rows x `rows` rows per lane
#define POLAR_BW 32
#define POLAR_BH 8

bool pl_shader_sample_polar(pl_shader sh, const struct pl_sample_src *src,
                            const struct pl_sample_filter_params *params)
{
    if (!params->filter.polar) {
        SH_FAIL(sh, "Trying to use polar sampling with a non-polar filter?");
        return false;
    }

    struct src_info info;
    if (!setup_src(sh, src, &info, false, REQ_FASTEST))
        return false;

    pl_gpu gpu = SH_GPU(sh);
    struct sh_sampler_obj *obj = SH_OBJ(sh, params->lut, PL_SHADER_OBJ_SAMPLER,
                                        struct sh_sampler_obj, sh_sampler_uninit);
    if (!obj) {
        SH_FAIL(sh, "pl_shader_sample_polar requires `params->lut` state");
        return false;
    }

    float inv_scale = 1.0 / PL_MIN(info.ratio_x, info.ratio_y);
    inv_scale = PL_MAX(inv_scale, 1.0);
    if (params->no_widening)
        inv_scale = 1.0;

    struct pl_filter_config cfg = params->filter;
    cfg.antiring = PL_DEF(cfg.antiring, params->antiring);
    cfg.blur = PL_DEF(cfg.blur, 1.0f) * inv_scale;
    const bool update = !obj->filter || !pl_filter_config_eq(&obj->filter->params.config, &cfg);
    if (update) {
        pl_filter_free(&obj->filter);
        obj->filter = pl_filter_generate(sh->log, pl_filter_params(
            .config         = cfg,
            .lut_entries    = SCALER_LUT_SIZE,
            .cutoff         = SCALER_LUT_CUTOFF,
        ));
        if (!obj->filter) {
            SH_FAIL(sh, "Failed initializing polar filter!");
            return false;
        }
    }

    describe_filter(sh, &cfg, "polar", info.ratio_x, info.ratio_y);
    pl_filter filter = obj->filter;
    const bool use_ar = cfg.antiring > 0;
    const int bound = ceil(filter->radius);
    if (2 * bound - 1 >= 64 || bound > 127) {
        SH_FAIL(sh, "Polar radius %f exceeds implementation capacity!", filter->radius);
        return false;
    }

    // The reference switches from the LDS formulation to the gather one at
    // radius 6 (sampling.c:671-674); both run on the same LDS kernel here, but
    // the tap *order* (hence fp32 summation order) follows the reference's pick
    const struct pl_glsl_version glsl = sh_glsl(sh);
    const bool gather_order = params->no_compute || !(filter->radius < 6.0);

    if (update || !obj->lut || !obj->taps || obj->taps_gather != gather_order) {
        // weight LUT as {L[i], L[min(i+1, 255)]} pairs: one ds_read_b64 per tap
        float pairs[2 * SCALER_LUT_SIZE];
        for (int i = 0; i < SCALER_LUT_SIZE; i++) {
            pairs[2 * i + 0] = filter->weights[i];
            pairs[2 * i + 1] = filter->weights[PL_MIN(i + 1, SCALER_LUT_SIZE - 1)];
        }

        const int max_taps = 4 * bound * bound;
        uint32_t *taps = malloc(max_taps * sizeof(uint32_t));
        if (!taps)
            return false;
        obj->num_taps = gather_order ? polar_taps_gather(taps, filter, bound, use_ar, &glsl)
                                     : polar_taps_compute(taps, filter, bound, use_ar);
        obj->taps_gather = gather_order;

        pl_buf_destroy(gpu, &obj->lut);
        pl_buf_destroy(gpu, &obj->taps);
        obj->lut = pl_buf_create(gpu, pl_buf_params(
            .size = sizeof(pairs), .storable = true, .initial_data = pairs));
        obj->taps = pl_buf_create(gpu, pl_buf_params(
            .size = PL_MAX(obj->num_taps, 1) * sizeof(uint32_t), .storable = true,
            .initial_data = taps));
        free(taps);
        if (!obj->lut || !obj->taps) {
            SH_FAIL(sh, "Failed initializing polar LUT!");
            return false;
        }
    }

    // LDS tile: footprint of a 32 x (8*rows) output tile + filter support
    // (+2: one texel of rounding slack per side, see k_polar.hip)
    const int padding = 2 * bound - 1;
    const float margin = 1e-5;
    const bool fp32_tile = src->tex->params.format->component_depth[0] > 16;
    const size_t texel = fp32_tile ? 16 : 8;
    const size_t max_lds = 160 * 1024 / 2; // keep two workgroups per CU resident
    int rows = 4, tile_w, tile_h;
    for (;;) {
        tile_w = (int) ceilf(POLAR_BW / info.ratio_x - margin) + padding + 1 + 2;
        tile_h = (int) ceilf(POLAR_BH * rows / info.ratio_y - margin) + padding + 1 + 2;
        if (2048 + (size_t) tile_w * tile_h * texel <= max_lds || rows == 1)
            break;
        rows >>= 1;
    }
    const size_t shmem = 2048 + (size_t) tile_w * tile_h * texel;
    if (shmem > 160 * 1024) {
        SH_FAIL(sh, "Polar filter footprint (%dx%d texels) does not fit in LDS", tile_w, tile_h);
        return false;
    }
    sh_try_compute(sh, POLAR_BW, POLAR_BH * rows, false, 0);
    sh->shmem = shmem;

    struct plh_sampler_args *s = &sh->pass.s;
    s->type = PLH_SAMPLE_POLAR;
    s->lut = pl_hip_buf_ptr(obj->lut);
    s->taps = pl_hip_buf_ptr(obj->taps);
    s->num_taps = obj->num_taps;
    s->bound = bound;
    s->radius = filter->radius;
    s->rcp_radius = 1.0f / filter->radius;
    s->radius_zero = filter->radius_zero;
    s->antiring = cfg.antiring;
    s->tile_w = tile_w;
    s->tile_h = tile_h;
    s->tile_rows = rows;
    s->tile_fp32 = fp32_tile;
    sh_hold(sh, *params->lut);

    sh_listf(sh, "sample_polar(filter=%s, radius=%f, radius_zero=%f, taps=%d (%s order), "
             "tile=%dx%d %s, rows=%d, antiring=%g, scale=%g, mask=0x%x)\n",
             PL_DEF(cfg.name, "custom"), filter->radius, filter->radius_zero, obj->num_taps,
             gather_order ? "gather" : "compute", tile_w, tile_h, fp32_tile ? "f32" : "f16",
             rows, cfg.antiring, info.scale, info.comp_mask);
    return true;
}

bool pl_shader_sample_ortho2(pl_shader sh, const struct pl_sample_src *src,
                             const struct pl_sample_filter_params *params)
{
    (void) src; (void) params;
    SH_FAIL(sh, "pl_shader_sample_ortho2: not implemented yet");
    return false;
}

void pl_shader_deband(pl_shader sh, const struct pl_sample_src *src,
                      const struct pl_deband_params *params)
{
    (void) src; (void) params;
    SH_FAIL(sh, "pl_shader_deband: not implemented yet");
}
