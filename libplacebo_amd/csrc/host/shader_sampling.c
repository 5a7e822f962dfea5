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
#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
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
    // 1:1 sampling on the texel grid (what img_sh()/PASS A does): a texture unit returns the
    // texel itself there (its fixed-point lerp weights snap to 0), whereas an exact fp32 lerp
    // would blend in ~1e-5 of a neighbour from the rounding noise in `pos`. Such identity
    // fetches are lowered to nearest - by the dispatch, because these shaders are resizable
    // and only the final output size tells whether the fetch is 1:1 (dispatch.c here).
    sh->pass.s.type = info.linear ? PLH_SAMPLE_BILINEAR : PLH_SAMPLE_NEAREST;
    float rw = pl_rect_w(src->rect), rh = pl_rect_h(src->rect);
    sh->pass.s.rect_w = fabsf(PL_DEF(rw, (float) src->tex->params.w));
    sh->pass.s.rect_h = fabsf(PL_DEF(rh, (float) src->tex->params.h));
    sh->pass.s.rect_on_grid = src->rect.x0 == truncf(src->rect.x0) &&
                              src->rect.y0 == truncf(src->rect.y0);
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

// Geometry a set of polar phase-class tables was built for (plh_polar_pp_setup)
struct polar_pp_key {
    float pos[4][2];
    int src_w, src_h, width, height;
    int bound, num_taps, fp32_tile;
    float scale, radius;
    uint64_t filter_gen;
};

struct sh_sampler_obj {
    pl_filter filter;
    pl_buf lut;         // polar: 256 {L[i], L[i+1]} pairs; ortho: rows
    pl_buf taps;        // polar: packed tap list
    int num_taps;
    bool taps_gather;   // tap order the list was generated for
    pl_shader_obj pass2; // second ortho pass

    // polar phase classes (k_polar_pp): one device blob holding struct plh_polar_pp
    // and every table it points to
    uint64_t filter_gen;            // bumped whenever lut/taps are regenerated
    struct polar_pp_key pp_key;
    int pp_state;                   // 0 = not built, 1 = usable, -1 = not applicable
    pl_buf pp_blob;
    struct plh_polar_pp pp_host;    // host copy (device pointers)
    int pp_tile_w, pp_tile_h, pp_rows, pp_lds_weights;

    // matrix-pipe variant of the same geometry (k_polar_mx): B fragments + tile origin
    pl_buf mx_blob;
    struct plh_polar_mx mx_host;    // .enabled = 0: geometry not eligible
    bool mx_announced;
};

static void sh_sampler_uninit(pl_gpu gpu, void *ptr)
{
    struct sh_sampler_obj *obj = ptr;
    pl_buf_destroy(gpu, &obj->lut);
    pl_buf_destroy(gpu, &obj->taps);
    pl_buf_destroy(gpu, &obj->pp_blob);
    pl_buf_destroy(gpu, &obj->mx_blob);
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
#ifndef POLAR_BW
#define POLAR_BW 32
#define POLAR_BH 8
#endif

static bool sample_polar(pl_shader sh, const struct pl_sample_src *src,
                         const struct pl_sample_filter_params *params, bool force_f16_tile);

bool pl_shader_sample_polar(pl_shader sh, const struct pl_sample_src *src,
                            const struct pl_sample_filter_params *params)
{
    return sample_polar(sh, src, params, false);
}

static bool sample_polar(pl_shader sh, const struct pl_sample_src *src,
                         const struct pl_sample_filter_params *params, bool force_f16_tile)
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
        obj->filter_gen++;

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
    // the LDS tile holds f16 texels only where that is lossless: an rgba16hf source, or the fused
    // PASS A whose result the reference rounds to an rgba16hf FBO anyway. unorm8/16 and fp32
    // sources are staged as fp32 (k/255 and k/65535 are not f16 numbers)
    const pl_fmt sfmt = src->tex->params.format;
    const bool f16_src = sfmt->type == PL_FMT_FLOAT && sfmt->component_depth[0] == 16;
    const bool fp32_tile = !force_f16_tile && !f16_src;
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
    s->pp = NULL;
    sh->polar_obj = use_ar ? NULL : obj;    // anti-ringing needs per-pixel d, see k_polar
    sh_hold(sh, *params->lut);

    sh_listf(sh, "sample_polar(filter=%s, radius=%f, radius_zero=%f, taps=%d (%s order), "
             "tile=%dx%d %s, rows=%d, antiring=%g, scale=%g, mask=0x%x)\n",
             PL_DEF(cfg.name, "custom"), filter->radius, filter->radius_zero, obj->num_taps,
             gather_order ? "gather" : "compute", tile_w, tile_h, fp32_tile ? "f32" : "f16",
             rows, cfg.antiring, info.scale, info.comp_mask);
    return true;
}


/* ---- polar phase classes (device side: k_polar.hip, struct plh_polar_pp) ---------------- */

static int cmp_u32(const void *a, const void *b)
{
    const uint32_t x = *(const uint32_t *) a, y = *(const uint32_t *) b;
    return x < y ? -1 : x > y;
}

// distinct bit patterns of fc[0..n) -> sorted class values; ids[i] = class of element i
static int classify_axis(const float *fc, int n, float *cls, uint16_t *ids, int max_cls)
{
    uint32_t *tmp = malloc(n * sizeof(uint32_t));
    if (!tmp)
        return -1;
    memcpy(tmp, fc, n * sizeof(uint32_t));
    qsort(tmp, n, sizeof(uint32_t), cmp_u32);
    int nc = 0;
    for (int i = 0; i < n; i++) {
        if (i && tmp[i] == tmp[i - 1])
            continue;
        if (nc == max_cls) {
            free(tmp);
            return -1;
        }
        memcpy(&cls[nc++], &tmp[i], 4);
    }
    free(tmp);
    for (int i = 0; i < n; i++) {
        uint32_t key;
        memcpy(&key, &fc[i], 4);
        int lo = 0, hi = nc - 1;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            uint32_t v;
            memcpy(&v, &cls[mid], 4);
            if (v < key)
                lo = mid + 1;
            else
                hi = mid;
        }
        ids[i] = lo;
    }
    return nc;
}

// Can `n` consecutive outputs [n*c - pad, n*c - pad + n) always share a base texel?
static bool cells_share_base(const int32_t *base, int len, int n, int pad)
{
    for (int c0 = -pad; c0 < len; c0 += n) {
        int b = 0;
        bool have = false;
        for (int i = 0; i < n; i++) {
            const int x = c0 + i;
            if (x < 0 || x >= len)
                continue;
            if (have && base[x] != b)
                return false;
            b = base[x];
            have = true;
        }
    }
    return true;
}

struct axis_tiles {
    int ntiles;
    uint8_t *loc;       // [len]
    uint16_t *list;     // [ntiles][PLH_PP_LMAX]
    uint8_t *cnt;       // [ntiles]
    int32_t *org;       // [ntiles]
    int extent;         // LDS tile extent needed along this axis (texels)
    int max_cnt;
};

// Split an axis of `len` outputs into tiles of `tile_cells` cells of `n` outputs
static bool build_axis_tiles(struct axis_tiles *t, const uint16_t *ids, const int32_t *base,
                             int len, int n, int pad, int tile_cells, int bound)
{
    const int cells = (len + pad + n - 1) / n;
    t->ntiles = (cells + tile_cells - 1) / tile_cells;
    t->loc = calloc(len, 1);
    t->list = calloc((size_t) t->ntiles * PLH_PP_LMAX, sizeof(uint16_t));
    t->cnt = calloc(t->ntiles, 1);
    t->org = calloc(t->ntiles, sizeof(int32_t));
    t->extent = 0;
    t->max_cnt = 0;
    if (!t->loc || !t->list || !t->cnt || !t->org)
        return false;
    for (int ti = 0; ti < t->ntiles; ti++) {
        const int x0 = PL_MAX(ti * tile_cells * n - pad, 0);
        const int x1 = PL_MIN((ti + 1) * tile_cells * n - pad, len);
        uint16_t *list = t->list + (size_t) ti * PLH_PP_LMAX;
        int cnt = 0, bmin = INT32_MAX, bmax = INT32_MIN;
        for (int x = x0; x < x1; x++) {
            int l = 0;
            while (l < cnt && list[l] != ids[x])
                l++;
            if (l == cnt) {
                if (cnt == PLH_PP_LMAX)
                    return false;
                list[cnt++] = ids[x];
            }
            t->loc[x] = l;
            bmin = PL_MIN(bmin, base[x]);
            bmax = PL_MAX(bmax, base[x]);
        }
        if (x1 <= x0) {
            bmin = bmax = 0;
            cnt = 1;
        }
        t->cnt[ti] = cnt;
        t->max_cnt = PL_MAX(t->max_cnt, cnt);
        // taps span [base - (bound-1), base + bound]; one texel of slack per side for the
        // rare pixel whose own base is off by one (per-pixel path inside k_polar_pp)
        t->org[ti] = bmin - (bound - 1) - 1;
        t->extent = PL_MAX(t->extent, bmax - bmin + 2 * bound + 2);
    }
    return true;
}

static void free_axis_tiles(struct axis_tiles *t)
{
    free(t->loc);
    free(t->list);
    free(t->cnt);
    free(t->org);
    memset(t, 0, sizeof(*t));
}

static inline size_t align16(size_t x)
{
    return (x + 15) & ~(size_t) 15;
}

int plh_launch_polar_classify(plh_stream stream, const struct plh_pass *pass, void *out);
int plh_launch_polar_weights(plh_stream stream, const struct plh_pass *pass, const float *clsx,
                             int ncx, const float *clsy, int ncy, float *weights);


/* ---- polar on the matrix pipe (device side: k_polar_mx.hiph, struct plh_polar_mx) ------- */

// IEEE binary32 -> binary16, round to nearest even (subnormals and overflow included)
static uint16_t f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u)
        return sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u);
    if (x >= 0x477ff000u)   // rounds to >= 65520: infinity
        return sign | 0x7c00u;
    if (x < 0x38800000u) {  // below the smallest normal half: a multiple of 2^-24
        const float scaled = fabsf(f) * 16777216.0f;       // exact
        return sign | (uint16_t) lrintf(scaled);            // (round-half-even rounding mode)
    }
    const uint32_t mant = x & 0x007fffffu, exp = (x >> 23) - 112;
    uint32_t h = (exp << 10) | (mant >> 13);
    const uint32_t rest = mant & 0x1fffu;
    if (rest > 0x1000u || (rest == 0x1000u && (h & 1)))
        h++;                // (a carry into the exponent is the correct result)
    return sign | (uint16_t) h;
}

static float f16_to_f32(uint16_t h)
{
    const int exp = (h >> 10) & 0x1f, mant = h & 0x3ff;
    float v;
    if (exp == 0)
        v = ldexpf((float) mant, -24);
    else if (exp == 31)
        v = mant ? NAN : INFINITY;
    else
        v = ldexpf((float) (mant | 0x400), exp - 25);
    return (h & 0x8000) ? -v : v;
}

// The geometry k_polar_mx covers: an axis whose outputs alternate between two phases and step
// one source texel per two outputs (an exact 2x upscale, any sub-texel offset). Returns the
// class of each parity and c1 = base(1) - base(0); false if the axis does not have that shape.
static bool mx_axis(const float *fc, const int32_t *base, const uint16_t *ids, int len,
                    int canon[2], int *c1)
{
    if (len < 2)
        return false;
    *c1 = base[1] - base[0];
    if (*c1 != 0 && *c1 != 1)
        return false;
    for (int q = 0; q < 2; q++) {
        canon[q] = ids[q];
        // (a phase next to 0 or 1 could flip its base texel with the rounding of one pixel)
        if (fc[q] < 0.02f || fc[q] > 0.98f)
            return false;
    }
    for (int i = 0; i < len; i++) {
        const int q = i & 1;
        if (base[i] != base[0] + (i >> 1) + (q ? *c1 : 0))
            return false;
        // the phase of a column differs from its parity's by the fp32 rounding of the attribute
        // interpolation, which grows with the coordinate: a few ulps of the source position
        if (fabsf(fc[i] - fc[q]) > 1e-5f + 1.5f * FLT_EPSILON * (float) len)
            return false;
    }
    return true;
}

static bool polar_mxd_build(pl_gpu gpu, pl_log log, struct sh_sampler_obj *obj,
                            const struct plh_pass *pass, const float *wall, const uint32_t *taps,
                            int ntaps, int ncx, int ncy, const float *clsx, const float *clsy,
                            const float *colfc, const int32_t *colbase,
                            const float *rowfc, const int32_t *rowbase);

// B fragments (plh_device.h): frag f = 4 * (py * npairs + j) + kind, lane l, element e hold
//   T(py, wy)[k][n] with n = l & 15, k = 8 * ((l >> 4) & 1) + e, wy = first[py] + 2 j + (l >> 5),
//   = w'(phase py, phase n & 1, tap (k - dbx[n] - 3, wy - 3)): kind 0 / 1 its hi / lo f16 halves,
//   kind 2 / 3 its derivative in fcoord_x / fcoord_y times 2^-PLH_MX_DSHIFT.
// first[py] = the first tap row of row phase py that carries a weight for either column phase;
// npairs = 3 when both row phases have at most six such rows (every centred 2x upscale with a
// radius <= 3.25: rows -3 and 4 of the reference's 8 x 8 tap square lie 3.25 / 3.75 texels from
// the sample), else 4.
// The derivatives are the slopes of a least-squares line through the normalised weights of the
// phase classes of that parity -- the weights the per-pixel kernels actually use for them.
static bool polar_mx_build(pl_gpu gpu, pl_log log, struct sh_sampler_obj *obj,
                           const struct plh_pass *pass, const float *wall, const uint32_t *taps,
                           int ntaps, int ncx, int ncy, const float *clsx, const float *clsy,
                           const float *colfc, const int32_t *colbase, const uint16_t *idx,
                           const float *rowfc, const int32_t *rowbase, const uint16_t *idy)
{
    const struct plh_sampler_args *s = &pass->s;
    const int W = pass->width, H = pass->height;
    obj->mx_host = (struct plh_polar_mx) {0};
    // LDS of the widest variant (RGBA tile, 4 wave-tile columns: 36 KiB of B fragments + 4 planes
    // of 41 rows x 96 B; the RGB tile of 8 columns needs 55.5 KiB) against the limit the backend
    // was created with (pl_hip_params.max_shmem_size)
    if (gpu->glsl.max_shmem_size < 64 * 1024) {
        pl_msg(log, PL_LOG_DEBUG, "matrix-pipe polar: needs 64 KiB of shared memory, the limit is %zu",
               (size_t) gpu->glsl.max_shmem_size);
        return false;
    }
    if (s->bound > 4 || s->tile_fp32 || s->address_mode != PLH_ADDRESS_CLAMP || pass->transpose ||
        s->src.w < 2 || s->antiring > 0) {
        pl_msg(log, PL_LOG_DEBUG, "matrix-pipe polar: not this pass (bound %d, fp32 tile %d, address "
               "mode %d, transpose %d, antiring %g)", s->bound, s->tile_fp32, s->address_mode,
               pass->transpose, s->antiring);
        return false;
    }
    int cx[2], cy[2], c1x, c1y;
    if (!mx_axis(colfc, colbase, idx, W, cx, &c1x) || !mx_axis(rowfc, rowbase, idy, H, cy, &c1y)) {
        pl_msg(log, PL_LOG_DEBUG, "matrix-pipe polar: not an exact 2x geometry");
        return false;
    }

    // tap (x, y) -> index in the list, x, y in [-3, 4]
    int tap_at[8][8];
    for (int y = 0; y < 8; y++) {
        for (int x = 0; x < 8; x++)
            tap_at[y][x] = -1;
    }
    for (int t = 0; t < ntaps; t++) {
        const int x = (int8_t) (taps[t] & 0xff), y = (int8_t) ((taps[t] >> 8) & 0xff);
        if (x < -3 || x > 4 || y < -3 || y > 4)
            return false;
        tap_at[y + 3][x + 3] = t;
    }

    // normalised weight (w * scale / wsum) of tap t for the class pair (kx, ky)
#define WN(kx, ky, t) ((double) wall[((size_t) (ky) * ncx + (kx)) * (ntaps + 1) + (t)] * \
                       (double) wall[((size_t) (ky) * ncx + (kx)) * (ntaps + 1) + ntaps])
    // d w' / d fcoord along one axis at the class pair (cx[px], cy[py])
    double *slope[2];       // [axis][(py * 2 + px) * ntaps + t]
    slope[0] = calloc((size_t) 4 * PL_MAX(ntaps, 1), sizeof(double));
    slope[1] = calloc((size_t) 4 * PL_MAX(ntaps, 1), sizeof(double));
    const size_t nfx = ((size_t) W + PLH_MX_PAD - 1) / PLH_MX_PAD * PLH_MX_PAD;
    const size_t nfy = ((size_t) H + PLH_MX_PAD - 1) / PLH_MX_PAD * PLH_MX_PAD;
    const size_t frag_bytes = (size_t) PLH_MX_NFRAG * 64 * 8 * sizeof(uint16_t);
    const size_t o_dfx = frag_bytes, o_dfy = o_dfx + nfx * 4;
    const size_t o_sink = o_dfy + nfy * 4;      // (512 bytes nobody reads: plh_polar_mx.sink)
    const size_t bytes = o_sink + 512;
    uint8_t *blob = calloc(1, bytes);
    if (!slope[0] || !slope[1] || !blob) {
        free(slope[0]); free(slope[1]); free(blob);
        return false;
    }
    for (int py = 0; py < 2; py++) {
        for (int px = 0; px < 2; px++) {
            double *sx = slope[0] + (size_t) (py * 2 + px) * ntaps;
            double *sy = slope[1] + (size_t) (py * 2 + px) * ntaps;
            double den = 0.0;
            for (int c = 0; c < ncx; c++) {
                const double d = (double) clsx[c] - (double) clsx[cx[px]];
                if (fabs(d) > 0.01)
                    continue;   // the other parity
                den += d * d;
                for (int t = 0; t < ntaps; t++)
                    sx[t] += d * (WN(c, cy[py], t) - WN(cx[px], cy[py], t));
            }
            for (int t = 0; t < ntaps; t++)
                sx[t] = den > 0.0 ? sx[t] / den : 0.0;
            den = 0.0;
            for (int c = 0; c < ncy; c++) {
                const double d = (double) clsy[c] - (double) clsy[cy[py]];
                if (fabs(d) > 0.01)
                    continue;
                den += d * d;
                for (int t = 0; t < ntaps; t++)
                    sy[t] += d * (WN(cx[px], c, t) - WN(cx[px], cy[py], t));
            }
            for (int t = 0; t < ntaps; t++)
                sy[t] = den > 0.0 ? sy[t] / den : 0.0;
        }
    }

    // live tap rows of each row phase: a row counts when any column phase has a weight, a slope
    // included (the slopes are fitted through neighbouring classes, whose tap sets are the same:
    // mx_axis holds every class of a parity within 1e-5 of it)
    int first[2], npairs = 3;
    for (int py = 0; py < 2; py++) {
        int lo = 8, hi = -1;
        for (int wy = 0; wy < 8; wy++) {
            bool live = false;
            for (int wx = 0; wx < 8 && !live; wx++) {
                const int t = tap_at[wy][wx];
                if (t < 0)
                    continue;
                for (int px = 0; px < 2 && !live; px++) {
                    live = WN(cx[px], cy[py], t) != 0.0 ||
                           slope[0][(size_t) (py * 2 + px) * ntaps + t] != 0.0 ||
                           slope[1][(size_t) (py * 2 + px) * ntaps + t] != 0.0;
                }
            }
            if (live) {
                lo = PL_MIN(lo, wy);
                hi = wy;
            }
        }
        if (hi < lo)
            lo = hi = 3;
        first[py] = lo;
        if (hi - lo + 1 > 6)
            npairs = 4;
    }
    for (int py = 0; py < 2; py++)
        first[py] = PL_MIN(first[py], 8 - 2 * npairs);  // (the pairs stay inside the 8 tap rows)

    uint16_t *frag = (uint16_t *) blob;
    const double dscale = ldexp(1.0, -PLH_MX_DSHIFT);
    double worst = 0.0;
    for (int py = 0; py < 2; py++) {
        for (int j = 0; j < npairs; j++) {
            for (int l = 0; l < 64; l++) {
                const int n = l & 15, px = n & 1;
                const int dbx = (n >> 1) + (px ? c1x : 0);
                const int wy = first[py] + 2 * j + (l >> 5);
                for (int e = 0; e < 8; e++) {
                    const int k = 8 * ((l >> 4) & 1) + e, wx = k - dbx;
                    double v = 0.0, vx = 0.0, vy = 0.0;
                    if (wx >= 0 && wx < 8 && wy >= 0 && wy < 8 && tap_at[wy][wx] >= 0) {
                        const int t = tap_at[wy][wx];
                        v = WN(cx[px], cy[py], t);
                        vx = slope[0][(size_t) (py * 2 + px) * ntaps + t];
                        vy = slope[1][(size_t) (py * 2 + px) * ntaps + t];
                    }
                    const uint16_t hi = f32_to_f16((float) v);
                    const uint16_t lo = f32_to_f16((float) (v - (double) f16_to_f32(hi)));
                    const double err = fabs(v - (double) f16_to_f32(hi) - (double) f16_to_f32(lo));
                    worst = PL_MAX(worst, err);
                    const size_t f = 4 * (size_t) (py * npairs + j);
                    frag[((f + 0) * 64 + l) * 8 + e] = hi;
                    frag[((f + 1) * 64 + l) * 8 + e] = lo;
                    frag[((f + 2) * 64 + l) * 8 + e] = f32_to_f16((float) (vx * dscale));
                    frag[((f + 3) * 64 + l) * 8 + e] = f32_to_f16((float) (vy * dscale));
                }
            }
        }
    }
#undef WN
    free(slope[0]);
    free(slope[1]);

    // how far a pixel's own phase lies from the one its parity is expanded about
    float *dfx = (float *) (blob + o_dfx), *dfy = (float *) (blob + o_dfy);
    float dev = 0.0f;
    const float up = ldexpf(1.0f, PLH_MX_DSHIFT);
    for (int i = 0; i < W; i++) {
        const float d = colfc[i] - colfc[i & 1];
        dev = fmaxf(dev, fabsf(d));
        dfx[i] = d * up;
    }
    for (int i = 0; i < H; i++) {
        const float d = rowfc[i] - rowfc[i & 1];
        dev = fmaxf(dev, fabsf(d));
        dfy[i] = d * up;
    }

    pl_buf_destroy(gpu, &obj->mx_blob);
    obj->mx_blob = pl_buf_create(gpu, pl_buf_params(.size = bytes, .storable = true,
                                                    .initial_data = blob));
    free(blob);
    if (!obj->mx_blob)
        return false;

    const char *base = pl_hip_buf_ptr(obj->mx_blob);
    obj->mx_host = (struct plh_polar_mx) {
        .enabled = 1,
        .org_x = colbase[0] - 3, .org_y = rowbase[0] - 3,
        .npairs = npairs,
        // tile row of tap row first[py] for the output row pair 0: rows 2 m + py sample from base
        // rowbase[0] + m + (py ? c1y : 0)
        .row_first = { first[0], first[1] + c1y },
        .sink = (void *) (base + o_sink),
        .bfrag = base,
        .dfx = (const float *) (base + o_dfx), .dfy = (const float *) (base + o_dfy),
    };
    obj->mx_announced = false;
    pl_msg(log, PL_LOG_DEBUG, "matrix-pipe tables for the polar pass: 2 x 2 phases (fcoord %.6f %.6f / %.6f %.6f, "
           "per-pixel phases within %.2e: first-order terms), %d row pairs per phase from tile rows %d / %d, "
           "weight split error <= %.2e",
           colfc[0], colfc[1], rowfc[0], rowfc[1], dev, npairs, first[0], first[1] + c1y, worst);
    return true;
}

// The geometry k_polar_mxr covers: an axis of an exact upscale by R : G (R outputs per G source
// texels; G = 1: the integer ratios). Output i belongs to base index (i + shift) / R and phase
// (i + shift) % R; the base texel of an output is origin + G * index + off[phase] with the same small
// offset for every output of a phase (G = 1: none); the phase of an output is its phase class' up
// to the fp32 rounding of the attribute interpolation. A phase at fcoord = 0 (odd integer ratios
// have one) is where that rounding decides between (base b, fcoord +eps) and (base b - 1, fcoord
// 1 - eps): the same sample position -- the tap that enters at one end and the one that leaves at
// the other lie beyond the filter's radius -- so such an output is taken as (b, fcoord - 1), a small
// negative deviation from the phase (`canon`: the outputs' canonical fcoord, which the caller turns
// into the deviations). Returns the shift, the origin, the offsets and a representative, unwrapped
// output of every phase.
static bool mxr_axis(const float *fc, const int32_t *base, int len, int R, int G, int *shift,
                     int *origin, int off[PLH_MXR_MAX_RATIO], int rep[PLH_MXR_MAX_RATIO], float *canon)
{
    if (len < 3 * R)
        return false;
    for (int i = 0; i < len; i++)
        canon[i] = fc[i] > 0.98f ? fc[i] - 1.0f : fc[i];
#define CANON_BASE(i) (base[i] + (fc[i] > 0.98f ? 1 : 0))
    // the shift: the one under which the offsets are consistent and smallest
    int best = -1, best_max = 0, best_org = 0;
    for (int sh = 0; sh < R; sh++) {
        int org = INT_MAX;
        for (int i = 0; i < 2 * R; i++)
            org = PL_MIN(org, CANON_BASE(i) - G * ((i + sh) / R));
        int o[PLH_MXR_MAX_RATIO], omax = 0;
        bool ok = true;
        for (int q = 0; q < R; q++)
            o[q] = -1;
        for (int i = 0; i < len && ok; i++) {
            const int q = (i + sh) % R;
            const int d = CANON_BASE(i) - G * ((i + sh) / R) - org;
            if (o[q] < 0)
                o[q] = d;
            ok = d == o[q] && d >= 0 && d <= (G == 1 ? 0 : 2);
            omax = PL_MAX(omax, d);
        }
        if (ok && (best < 0 || omax < best_max)) {
            best = sh;
            best_max = omax;
            best_org = org;
        }
    }
    if (best < 0)
        return false;
    *shift = best;
    *origin = best_org;
    for (int q = 0; q < R; q++)
        rep[q] = off[q] = -1;
    for (int i = 0; i < len; i++) {
        const int q = (i + best) % R;
        if (off[q] < 0)
            off[q] = CANON_BASE(i) - G * ((i + best) / R) - best_org;
        if (rep[q] < 0 && !(fc[i] > 0.98f))
            rep[q] = i;
    }
#undef CANON_BASE
    for (int q = 0; q < R; q++) {
        if (rep[q] < 0 || off[q] < 0)
            return false;
    }
    for (int i = 0; i < len; i++) {
        const int q = (i + best) % R;
        if (fabsf(canon[i] - fc[rep[q]]) > 1e-5f + 1.5f * FLT_EPSILON * (float) len)
            return false;
    }
    return true;
}

// Test hook (tests/test_mxr_axis.py, CPU): mxr_axis on an axis described by its per-output base
// texels and fcoords. out = { shift, origin, off[0..3], rep[0..3] }; canon: len floats.
PL_API int plh_test_mxr_axis(const float *fc, const int32_t *base, int len, int R, int G,
                             int *out, float *canon);
int plh_test_mxr_axis(const float *fc, const int32_t *base, int len, int R, int G, int *out, float *canon)
{
    int shift = 0, origin = 0, off[PLH_MXR_MAX_RATIO] = {0}, rep[PLH_MXR_MAX_RATIO] = {0};
    if (R < 2 || R > PLH_MXR_MAX_RATIO || !mxr_axis(fc, base, len, R, G, &shift, &origin, off, rep, canon))
        return 0;
    out[0] = shift;
    out[1] = origin;
    for (int q = 0; q < PLH_MXR_MAX_RATIO; q++) {
        out[2 + q] = q < R ? off[q] : -1;
        out[2 + PLH_MXR_MAX_RATIO + q] = q < R ? rep[q] : -1;
    }
    return 1;
}

// B fragments of k_polar_mxr (plh_device.h): frag f = 32 py + 4 * (NH * j + h) + kind, lane l,
// element e hold T(py, j, h)[k][n], n = l & 15 the output column within half h of the wave's 8 / G
// bases -- base bi = 4 h + n / R, phase px = n % R, n < 4 R -- and K index (row 2 j + (l >> 5) of the
// base's footprint rows, column k = 8 * ((l >> 4) & 1) + e of the wave's 16-column window):
//   = w'(py, px, tap (k - G bi - offx[px] - 3, 2 j + (l >> 5) - offy[py] - 3)), kinds as in
// polar_mx_build. NH halves and NJ row pairs: 2 and 4 for the integer ratios, 1 and 5 for 3 : 2.
static bool polar_mxr_build(pl_gpu gpu, pl_log log, struct sh_sampler_obj *obj,
                            const struct plh_pass *pass, const float *wall, const uint32_t *taps,
                            int ntaps, int ncx, int ncy, const float *clsx, const float *clsy,
                            const float *colfc, const int32_t *colbase, const uint16_t *idx,
                            const float *rowfc, const int32_t *rowbase, const uint16_t *idy)
{
    const struct plh_sampler_args *s = &pass->s;
    const int W = pass->width, H = pass->height;
    const char *env = getenv("PL_HIP_POLAR_MXR");
    if (env && env[0] == '0')
        return false;
    if (gpu->glsl.max_shmem_size < 64 * 1024 || s->bound > 4 || s->tile_fp32 ||
        s->address_mode != PLH_ADDRESS_CLAMP || pass->transpose || s->src.w < 2 || s->antiring > 0)
        return false;
    int R = 0, G = 0, sx = 0, sy = 0, repx[PLH_MXR_MAX_RATIO], repy[PLH_MXR_MAX_RATIO];
    int bx0 = 0, by0 = 0, offx[PLH_MXR_MAX_RATIO], offy[PLH_MXR_MAX_RATIO];
    float *canx = malloc(((size_t) W + H) * sizeof(float)), *cany = canx ? canx + W : NULL;
    if (!canx)
        return false;
    static const int ratios[][2] = { {3, 1}, {4, 1}, {3, 2} };
    for (int r = 0; r < 3 && !R; r++) {
        if (mxr_axis(colfc, colbase, W, ratios[r][0], ratios[r][1], &sx, &bx0, offx, repx, canx) &&
            mxr_axis(rowfc, rowbase, H, ratios[r][0], ratios[r][1], &sy, &by0, offy, repy, cany)) {
            R = ratios[r][0];
            G = ratios[r][1];
        }
    }
    if (!R) {
        pl_msg(log, PL_LOG_DEBUG, "matrix-pipe polar: not an exact 3x / 4x / 3 : 2 geometry either");
        free(canx);
        return false;
    }
    const int NJ = G == 1 ? 4 : 5, NH = G == 1 ? 2 : 1;
    int tap_at[8][8];
    for (int y = 0; y < 8; y++) {
        for (int x = 0; x < 8; x++)
            tap_at[y][x] = -1;
    }
    for (int t = 0; t < ntaps; t++) {
        const int x = (int8_t) (taps[t] & 0xff), y = (int8_t) ((taps[t] >> 8) & 0xff);
        if (x < -3 || x > 4 || y < -3 || y > 4) {
            free(canx);
            return false;
        }
        tap_at[y + 3][x + 3] = t;
    }
#define WN(kx, ky, t) ((double) wall[((size_t) (ky) * ncx + (kx)) * (ntaps + 1) + (t)] * \
                       (double) wall[((size_t) (ky) * ncx + (kx)) * (ntaps + 1) + ntaps])
    int cx[PLH_MXR_MAX_RATIO], cy[PLH_MXR_MAX_RATIO];
    for (int q = 0; q < R; q++) {
        cx[q] = idx[repx[q]];
        cy[q] = idy[repy[q]];
    }
    // d w' / d fcoord at every phase pair: least-squares slopes over the classes of that phase
    const size_t nt = (size_t) PL_MAX(ntaps, 1);
    double *slx = calloc((size_t) R * R * nt, sizeof(double)), *sly = calloc((size_t) R * R * nt, sizeof(double));
    const size_t nfx = ((size_t) W + PLH_MX_PAD - 1) / PLH_MX_PAD * PLH_MX_PAD;
    const size_t nfy = ((size_t) H + PLH_MX_PAD - 1) / PLH_MX_PAD * PLH_MX_PAD;
    const size_t frag_bytes = (size_t) R * PLH_MXR_FRAGS_PER_PHASE * 64 * 8 * sizeof(uint16_t);
    const size_t o_dfx = frag_bytes, o_dfy = o_dfx + nfx * 4, bytes = o_dfy + nfy * 4;
    uint8_t *blob = calloc(1, bytes);
    if (!slx || !sly || !blob) {
        free(slx); free(sly); free(blob); free(canx);
        return false;
    }
    // tap t sits at (tapx[t], tapy[t]) of the 8 x 8 footprint
    int tapx[64], tapy[64];
    for (int y = 0; y < 8; y++) {
        for (int x = 0; x < 8; x++) {
            if (tap_at[y][x] >= 0 && tap_at[y][x] < 64) {
                tapx[tap_at[y][x]] = x;
                tapy[tap_at[y][x]] = y;
            }
        }
    }
    if (ntaps > 64) {
        free(slx); free(sly); free(blob); free(canx);
        return false;
    }
    // The classes a slope is fitted through: those within 0.01 of the phase -- and, for the phase at
    // fcoord = 0 of an odd ratio, the WRAPPED ones on the other side of it (fcoord 1 - eps on the
    // base one texel lower = -eps on this base: their weight for footprint position (x, y) is their
    // own weight one position further along the axis). Without them that phase may have a single
    // class, fcoord = 0 exactly, no slope, and its wrapped outputs -- up to 1e-4 away -- no
    // first-order term (4 codes on white noise at 720p -> 4K).
    for (int py = 0; py < R; py++) {
        for (int px = 0; px < R; px++) {
            double *vx = slx + (size_t) (py * R + px) * nt, *vy = sly + (size_t) (py * R + px) * nt;
            double den = 0.0;
            for (int c = 0; c < ncx; c++) {
                const bool wrapped = clsx[c] > 0.98f;
                const double d = (double) clsx[c] - (wrapped ? 1.0 : 0.0) - (double) clsx[cx[px]];
                if (fabs(d) > 0.01)
                    continue;   // another phase
                den += d * d;
                for (int t = 0; t < ntaps; t++) {
                    const int ts = !wrapped ? t : tapx[t] + 1 < 8 ? tap_at[tapy[t]][tapx[t] + 1] : -1;
                    vx[t] += d * ((ts >= 0 ? WN(c, cy[py], ts) : 0.0) - WN(cx[px], cy[py], t));
                }
            }
            for (int t = 0; t < ntaps; t++)
                vx[t] = den > 0.0 ? vx[t] / den : 0.0;
            den = 0.0;
            for (int c = 0; c < ncy; c++) {
                const bool wrapped = clsy[c] > 0.98f;
                const double d = (double) clsy[c] - (wrapped ? 1.0 : 0.0) - (double) clsy[cy[py]];
                if (fabs(d) > 0.01)
                    continue;
                den += d * d;
                for (int t = 0; t < ntaps; t++) {
                    const int ts = !wrapped ? t : tapy[t] + 1 < 8 ? tap_at[tapy[t] + 1][tapx[t]] : -1;
                    vy[t] += d * ((ts >= 0 ? WN(cx[px], c, ts) : 0.0) - WN(cx[px], cy[py], t));
                }
            }
            for (int t = 0; t < ntaps; t++)
                vy[t] = den > 0.0 ? vy[t] / den : 0.0;
        }
    }
    uint16_t *frag = (uint16_t *) blob;
    const double dscale = ldexp(1.0, -PLH_MX_DSHIFT);
    double worst = 0.0;
    for (int py = 0; py < R; py++) {
        for (int j = 0; j < NJ; j++) {
            for (int h = 0; h < NH; h++) {
                const size_t f = (size_t) py * PLH_MXR_FRAGS_PER_PHASE + 4 * (size_t) (NH * j + h);
                for (int l = 0; l < 64; l++) {
                    const int n = l & 15, px = n % R, bi = 4 * h + n / R;
                    const int wy = 2 * j + (l >> 5) - offy[py];
                    for (int e = 0; e < 8; e++) {
                        const int k = 8 * ((l >> 4) & 1) + e, wx = k - G * bi - offx[px];
                        double v = 0.0, vx = 0.0, vy = 0.0;
                        if (n < 4 * R && wx >= 0 && wx < 8 && wy >= 0 && wy < 8 && tap_at[wy][wx] >= 0) {
                            const int t = tap_at[wy][wx];
                            v = WN(cx[px], cy[py], t);
                            vx = slx[(size_t) (py * R + px) * nt + t];
                            vy = sly[(size_t) (py * R + px) * nt + t];
                        }
                        const uint16_t hi = f32_to_f16((float) v);
                        const uint16_t lo = f32_to_f16((float) (v - (double) f16_to_f32(hi)));
                        worst = PL_MAX(worst, fabs(v - (double) f16_to_f32(hi) - (double) f16_to_f32(lo)));
                        frag[((f + 0) * 64 + l) * 8 + e] = hi;
                        frag[((f + 1) * 64 + l) * 8 + e] = lo;
                        frag[((f + 2) * 64 + l) * 8 + e] = f32_to_f16((float) (vx * dscale));
                        frag[((f + 3) * 64 + l) * 8 + e] = f32_to_f16((float) (vy * dscale));
                    }
                }
            }
        }
    }
#undef WN
    free(slx);
    free(sly);
    float *dfx = (float *) (blob + o_dfx), *dfy = (float *) (blob + o_dfy);
    float dev = 0.0f;
    const float up = ldexpf(1.0f, PLH_MX_DSHIFT);
    for (int i = 0; i < W; i++) {
        const float d = canx[i] - colfc[repx[(i + sx) % R]];
        dev = fmaxf(dev, fabsf(d));
        dfx[i] = d * up;
    }
    for (int i = 0; i < H; i++) {
        const float d = cany[i] - rowfc[repy[(i + sy) % R]];
        dev = fmaxf(dev, fabsf(d));
        dfy[i] = d * up;
    }
    free(canx);
    pl_buf_destroy(gpu, &obj->mx_blob);
    obj->mx_blob = pl_buf_create(gpu, pl_buf_params(.size = bytes, .storable = true, .initial_data = blob));
    free(blob);
    if (!obj->mx_blob)
        return false;
    const char *base = pl_hip_buf_ptr(obj->mx_blob);
    obj->mx_host = (struct plh_polar_mx) {
        .enabled = 3, .ratio = R, .group = G, .sx = sx, .sy = sy,
        .org_x = bx0 - 3, .org_y = by0 - 3,     // (bx0, by0: the texel of base index 0, offset 0)
        .bfrag = base,
        .dfx = (const float *) (base + o_dfx), .dfy = (const float *) (base + o_dfy),
    };
    obj->mx_announced = false;
    pl_msg(log, PL_LOG_DEBUG, "matrix-pipe tables for the polar pass: %d x %d phases (%d : %d upscale, shifts %d / %d, "
           "per-pixel phases within %.2e: first-order terms), weight split error <= %.2e", R, R, R, G, sx, sy, dev, worst);
    return true;
}

static bool polar_pp_build(pl_gpu gpu, pl_log log, struct sh_sampler_obj *obj,
                           const struct plh_pass *pass)
{
    const struct plh_sampler_args *s = &pass->s;
    const int W = pass->width, H = pass->height, ntaps = s->num_taps;
    const plh_stream stream = plh_gpu_stream(gpu);
    bool ok = false;
    pl_buf tmp = NULL, wbuf = NULL;
    float *host = NULL, *clsx = NULL, *clsy = NULL, *wall = NULL;
    uint16_t *idx = NULL, *idy = NULL;
    uint8_t *blob = NULL;
    struct axis_tiles tx = {0}, ty = {0};
    enum { MAX_CLS = 96 };

    // ---- 1. fcoord / base of every column and row, evaluated by the device ------------------
    const size_t cls_bytes = (size_t) 2 * (W + H) * 4;
    tmp = pl_buf_create(gpu, pl_buf_params(.size = cls_bytes, .storable = true,
                                           .host_readable = true));
    host = malloc(cls_bytes);
    clsx = malloc(MAX_CLS * sizeof(float));
    clsy = malloc(MAX_CLS * sizeof(float));
    idx = malloc(W * sizeof(uint16_t));
    idy = malloc(H * sizeof(uint16_t));
    if (!tmp || !host || !clsx || !clsy || !idx || !idy)
        goto done;
    if (plh_launch_polar_classify(stream, pass, pl_hip_buf_ptr(tmp)) ||
        !plh_buf_read(gpu, tmp, 0, host, cls_bytes))
        goto done;
    const float *colfc = host, *rowfc = host + 2 * W;
    const int32_t *colbase = (const int32_t *) (host + W);
    const int32_t *rowbase = (const int32_t *) (host + 2 * W + H);

    // ---- 2. classes ---------------------------------------------------------------------------
    const int ncx = classify_axis(colfc, W, clsx, idx, MAX_CLS);
    const int ncy = classify_axis(rowfc, H, clsy, idy, MAX_CLS);
    if (ncx < 0 || ncy < 0) {
        // arbitrary (non-rational) ratio: every column has its own phase
        pl_msg(log, PL_LOG_DEBUG, "polar phase classes: more than %d distinct phases per axis "
               "(%dx%d outputs)", MAX_CLS, W, H);
        goto done;
    }

    // ---- 3. outputs per lane: 2x2 when pairs of outputs share their base texel --------------
    int n = 1, padx = 0, pady = 0;
    for (int px = 0; px < 2 && n == 1; px++) {
        if (!cells_share_base(colbase, W, 2, px))
            continue;
        for (int py = 0; py < 2; py++) {
            if (cells_share_base(rowbase, H, 2, py)) {
                n = 2; padx = px; pady = py;
                break;
            }
        }
    }

    // ---- 4. tiles: 32 x 8*rows cells, LDS = lut + weights + source tile ----------------------
    const size_t texel = s->tile_fp32 ? 16 : 8;
    const size_t max_lds = 64 * 1024;   // >= 2 workgroups per CU
    int rows = n == 2 ? 3 : 4, tp = 0, ntc = 0;    // (measured: 3 beats 4 by ~2 % for 2x2 cells)
    const char *env_rows = getenv("PL_HIP_PP_ROWS");     // profiling aid
    if (env_rows && atoi(env_rows) > 0)
        rows = PL_MIN(atoi(env_rows), 8);
    rows = PL_MIN(rows, 64 / (POLAR_BH * n));   // the kernel stages <= 64 output rows of info
    // a small output (the chroma planes of 1080p video: 1920 x 1080 = 690 workgroups at 3 rows) does
    // not fill 256 CUs twice with such tiles, and the kernel lives on latency hiding: fewer rows
    // per workgroup until there are two rounds of them (NV12 1080p -> 4K, the chroma pass:
    // 37.5 -> 26.2 us, profiles/r04_49_pp_rows_small.txt)
    if (!(env_rows && atoi(env_rows) > 0)) {
        while (rows > 1 && (size_t) ((W + POLAR_BW * n - 1) / (POLAR_BW * n)) *
                           (size_t) ((H + POLAR_BH * rows * n - 1) / (POLAR_BH * rows * n)) < 1024)
            rows--;
    }
    size_t lds_w = 0;
    for (;; rows >>= 1) {
        free_axis_tiles(&tx);
        free_axis_tiles(&ty);
        if (getenv("PL_HIP_PP_TRACE"))
            pl_msg(log, PL_LOG_DEBUG, "polar phase classes: %dx%d classes, n=%d rows=%d", ncx, ncy, n, rows);
        if (!build_axis_tiles(&tx, idx, colbase, W, n, padx, POLAR_BW, s->bound) ||
            !build_axis_tiles(&ty, idy, rowbase, H, n, pady, POLAR_BH * rows, s->bound))
            goto done;
        // worst-case weights slice; the compacted tap count is only known later
        lds_w = align16((size_t) tx.max_cnt * ty.max_cnt * (ntaps + 4) * 4) + align16(ntaps * 4);
        if (2048 + 1024 + 64 + lds_w + (size_t) tx.extent * ty.extent * texel <= max_lds)
            break;
        if (rows == 1)
            goto done;
    }

    // ---- 5. weights of every class pair, by the device; compaction of dead taps -------------
    const size_t wall_bytes = (size_t) ncx * ncy * (ntaps + 1) * 4;
    wbuf = pl_buf_create(gpu, pl_buf_params(.size = wall_bytes + (ncx + ncy) * 4, .storable = true,
                                            .host_readable = true, .host_writable = true));
    wall = malloc(wall_bytes);
    if (!wbuf || !wall)
        goto done;
    plh_buf_write(gpu, wbuf, wall_bytes, clsx, ncx * 4);
    plh_buf_write(gpu, wbuf, wall_bytes + ncx * 4, clsy, ncy * 4);
    const float *dcls = (const float *) ((const char *) pl_hip_buf_ptr(wbuf) + wall_bytes);
    if (plh_launch_polar_weights(stream, pass, dcls, ncx, dcls + ncx, ncy, pl_hip_buf_ptr(wbuf)) ||
        !plh_buf_read(gpu, wbuf, 0, wall, wall_bytes))
        goto done;

    uint32_t *taps_all = malloc(PL_MAX(ntaps, 1) * sizeof(uint32_t));
    int *keep = malloc(PL_MAX(ntaps, 1) * sizeof(int));
    if (!taps_all || !keep || !plh_buf_read(gpu, obj->taps, 0, taps_all, ntaps * sizeof(uint32_t))) {
        free(taps_all);
        free(keep);
        goto done;
    }
    for (int t = 0; t < ntaps; t++) {
        bool used = false;
        for (int pr = 0; pr < ncx * ncy && !used; pr++)
            used = wall[(size_t) pr * (ntaps + 1) + t] != 0.0f;
        if (used)
            keep[ntc++] = t;
    }
    tp = (ntc + 1 + 3) & ~3;    // weights + norm, padded to 16 bytes
    // + the compacted tap offsets, staged at the tail of this area (k_polar_pp)
    lds_w = align16((size_t) tx.max_cnt * ty.max_cnt * tp * 4) + align16(ntc * 4);

    // ---- 6. one device blob: struct + tables ---------------------------------------------------
    size_t off = align16(sizeof(struct plh_polar_pp));
#define PLACE(name, bytes) const size_t o_##name = off; off = align16(off + (bytes))
    PLACE(colfc, (size_t) W * 4);   PLACE(rowfc, (size_t) H * 4);
    PLACE(colbase, (size_t) W * 4); PLACE(rowbase, (size_t) H * 4);
    PLACE(colloc, W);               PLACE(rowloc, H);
    PLACE(collist, (size_t) tx.ntiles * PLH_PP_LMAX * 2);
    PLACE(rowlist, (size_t) ty.ntiles * PLH_PP_LMAX * 2);
    PLACE(coln, (size_t) tx.ntiles * 4); PLACE(rown, (size_t) ty.ntiles * 4);
    PLACE(colorg, (size_t) tx.ntiles * 4); PLACE(roworg, (size_t) ty.ntiles * 4);
    PLACE(weights, (size_t) ncx * ncy * tp * 4);
    PLACE(tapoff, (size_t) PL_MAX(ntc, 1) * 4);
    PLACE(tilemap, (size_t) tx.ntiles * ty.ntiles * 4);
#undef PLACE
    blob = calloc(1, off);
    if (!blob) {
        free(taps_all);
        free(keep);
        goto done;
    }
    memcpy(blob + o_colfc, colfc, (size_t) W * 4);
    memcpy(blob + o_rowfc, rowfc, (size_t) H * 4);
    memcpy(blob + o_colbase, colbase, (size_t) W * 4);
    memcpy(blob + o_rowbase, rowbase, (size_t) H * 4);
    memcpy(blob + o_colloc, tx.loc, W);
    memcpy(blob + o_rowloc, ty.loc, H);
    memcpy(blob + o_collist, tx.list, (size_t) tx.ntiles * PLH_PP_LMAX * 2);
    memcpy(blob + o_rowlist, ty.list, (size_t) ty.ntiles * PLH_PP_LMAX * 2);
    for (int i = 0; i < tx.ntiles; i++)
        ((int32_t *) (blob + o_coln))[i] = tx.cnt[i];
    for (int i = 0; i < ty.ntiles; i++)
        ((int32_t *) (blob + o_rown))[i] = ty.cnt[i];
    memcpy(blob + o_colorg, tx.org, (size_t) tx.ntiles * 4);
    memcpy(blob + o_roworg, ty.org, (size_t) ty.ntiles * 4);
    float *wc = (float *) (blob + o_weights);
    for (int pr = 0; pr < ncx * ncy; pr++) {
        const float *src = wall + (size_t) pr * (ntaps + 1);
        float *dst = wc + (size_t) pr * tp;
        for (int k = 0; k < ntc; k++)
            dst[k] = src[keep[k]];
        dst[ntc] = src[ntaps];      // scale / wsum
    }
    int32_t *tapoff = (int32_t *) (blob + o_tapoff);
    for (int k = 0; k < ntc; k++) {
        const uint32_t tap = taps_all[keep[k]];
        const int x = (int8_t) (tap & 0xff), y = (int8_t) ((tap >> 8) & 0xff);
        tapoff[k] = (y * tx.extent + x) * (int) texel;
    }
    // the same geometry on the matrix pipe, where it has the shape for it
    if (!polar_mx_build(gpu, log, obj, pass, wall, taps_all, ntaps, ncx, ncy, clsx, clsy, colfc, colbase,
                        idx, rowfc, rowbase, idy) &&
        !polar_mxr_build(gpu, log, obj, pass, wall, taps_all, ntaps, ncx, ncy, clsx, clsy, colfc, colbase,
                         idx, rowfc, rowbase, idy))
        polar_mxd_build(gpu, log, obj, pass, wall, taps_all, ntaps, ncx, ncy, clsx, clsy, colfc, colbase,
                        rowfc, rowbase);
    free(taps_all);
    free(keep);

    // XCD-aware launch order: workgroups go to the 8 XCDs round-robin, each XCD has its own L2;
    // XCD x gets the x-th contiguous eighth of the tiles so that neighbours share halo texels
    const uint32_t gx = tx.ntiles, total = (uint32_t) tx.ntiles * ty.ntiles;
    const bool remap = n == 2 && tx.ntiles <= 0xffff && ty.ntiles <= 0xffff;
    if (remap) {
        uint32_t *tm = (uint32_t *) (blob + o_tilemap);
        const uint32_t q = total / 8, r = total % 8;
        for (uint32_t lin = 0; lin < total; lin++) {
            const uint32_t xcd = lin % 8, k = lin / 8;
            const uint32_t tile = xcd * q + PL_MIN(xcd, r) + k;
            tm[lin] = (tile % gx) | ((tile / gx) << 16);
        }
    }

    pl_buf_destroy(gpu, &obj->pp_blob);
    obj->pp_blob = pl_buf_create(gpu, pl_buf_params(.size = off, .storable = true,
                                                    .host_writable = true));
    if (!obj->pp_blob)
        goto done;
    const char *d = pl_hip_buf_ptr(obj->pp_blob);
    struct plh_polar_pp *pp = &obj->pp_host;
    *pp = (struct plh_polar_pp) {
        .n = n, .padx = padx, .pady = pady,
        .cells_w = (W + padx + n - 1) / n, .cells_h = (H + pady + n - 1) / n,
        .ncx = ncx, .ncy = ncy, .ntaps = ntc, .tp = tp,
        .colfc = (const float *) (d + o_colfc), .rowfc = (const float *) (d + o_rowfc),
        .colbase = (const int32_t *) (d + o_colbase), .rowbase = (const int32_t *) (d + o_rowbase),
        .colloc = (const uint8_t *) (d + o_colloc), .rowloc = (const uint8_t *) (d + o_rowloc),
        .collist = (const uint16_t *) (d + o_collist), .rowlist = (const uint16_t *) (d + o_rowlist),
        .coln = (const int32_t *) (d + o_coln), .rown = (const int32_t *) (d + o_rown),
        .colorg = (const int32_t *) (d + o_colorg), .roworg = (const int32_t *) (d + o_roworg),
        .weights = (const float *) (d + o_weights), .tapoff = (const int32_t *) (d + o_tapoff),
        .tilemap = remap ? (const uint32_t *) (d + o_tilemap) : NULL,
    };
    memcpy(blob, pp, sizeof(*pp));
    plh_buf_write(gpu, obj->pp_blob, 0, blob, off);

    obj->pp_tile_w = tx.extent;
    obj->pp_tile_h = ty.extent;
    obj->pp_rows = rows;
    obj->pp_lds_weights = lds_w;
    pl_msg(log, PL_LOG_DEBUG, "polar phase classes: %dx%d classes, %d/%d live taps, %dx%d px per "
           "lane, tile %dx%d, %d rows, %zu B of weights in LDS", ncx, ncy, ntc, ntaps, n, n,
           tx.extent, ty.extent, rows, lds_w);
    ok = true;

done:
    pl_buf_destroy(gpu, &tmp);
    pl_buf_destroy(gpu, &wbuf);
    free_axis_tiles(&tx);
    free_axis_tiles(&ty);
    free(host); free(clsx); free(clsy); free(idx); free(idy); free(wall); free(blob);
    return ok;
}

void plh_polar_pp_setup(pl_gpu gpu, pl_log log, void *polar_obj, struct plh_pass *pass)
{
    struct sh_sampler_obj *obj = polar_obj;
    struct plh_sampler_args *s = &pass->s;
    s->pp = NULL;
    memset(&s->mx, 0, sizeof(s->mx));
    const char *env = getenv("PL_HIP_POLAR_PER_PIXEL");
    if (env && env[0] == '1')
        return;
    const uint32_t cm = s->comp_mask & 0xf;
    if (cm != 0x7 && cm != 0xf && cm != 0x1 && cm != 0x3)
        return; // k_polar_pp is instantiated for RGB / RGBA and for 1- / 2-component planes

    struct polar_pp_key key = {
        .src_w = s->src.w, .src_h = s->src.h, .width = pass->width, .height = pass->height,
        .bound = s->bound, .num_taps = s->num_taps, .fp32_tile = s->tile_fp32,
        .scale = s->scale, .radius = s->radius, .filter_gen = obj->filter_gen,
    };
    memcpy(key.pos, s->pos, sizeof(key.pos));
    if (!obj->pp_state || memcmp(&key, &obj->pp_key, sizeof(key))) {
        obj->pp_key = key;
        obj->pp_state = polar_pp_build(gpu, log, obj, pass) ? 1 : -1;
        if (obj->pp_state < 0)
            pl_msg(log, PL_LOG_DEBUG, "polar phase classes not applicable to this geometry; "
                   "using per-pixel weights");
    }
    if (obj->pp_state != 1)
        return;

    s->pp = pl_hip_buf_ptr(obj->pp_blob);
    s->ppv = obj->pp_host;
    s->pp_n = obj->pp_host.n;
    s->pp_cells_w = obj->pp_host.cells_w;
    s->pp_cells_h = obj->pp_host.cells_h;
    s->pp_lds_weights = obj->pp_lds_weights;
    const char *dbg = getenv("PL_HIP_PP_DEBUG");
    s->pp_debug = dbg ? atoi(dbg) : 0;
    s->tile_w = obj->pp_tile_w;
    s->tile_h = obj->pp_tile_h;
    s->tile_rows = obj->pp_rows;

    // k_polar_mx: the contraction on the f16 matrix pipe, within +-1 code of 16 bits of the
    // sequential-fma kernels. PL_HIP_POLAR_MFMA=0 keeps the bit-exact reference variant.
    // That bound holds behind EVERY epilogue, the ones that amplify near black included -- a pass
    // that scales in linear / sigmoidized light continues with UNSIGMOIDIZE (slope up to 17 at
    // the dark end) and DELINEARIZE ((1 / 2.4) x^-0.58) -- because the contraction's error scales
    // with the taps' products, which are small where the output is dark: measured <= 1 code at
    // 1080p -> 4K on white noise and on a dark field with isolated full-scale texels
    // (tests/test_gpu_default_kernels.py::test_matrix_pipe_behind_sigmoid_measured).
    const char *mfma = getenv("PL_HIP_POLAR_MFMA");
    memset(&s->mx, 0, sizeof(s->mx));
    if (obj->mx_host.enabled && !(mfma && mfma[0] == '0') && (cm == 0x7 || cm == 0xf) &&
        !pass->transpose && s->address_mode == PLH_ADDRESS_CLAMP) {
        s->mx = obj->mx_host;
        if (!obj->mx_announced)
            pl_msg(log, PL_LOG_DEBUG, "polar on the matrix pipe (%s)",
                   s->mx.enabled == 2 ? "k_polar_mxd, where the pass has its shape" :
                   s->mx.enabled == 3 ? "k_polar_mxr, where the pass has its shape" : "k_polar_mx");
        obj->mx_announced = true;
    }
}


/* ---- the 2 : 1 downscale on the matrix pipe ------------------------------------------------------ */

// every output i has its base texel at base[0] + 2 i and a phase within `tol` of 1/2
static bool mxd_axis(const float *fc, const int32_t *base, int len, float *dev)
{
    if (len < 2)
        return false;
    for (int i = 0; i < len; i++) {
        if (base[i] != base[0] + 2 * i)
            return false;
        const float d = fabsf(fc[i] - 0.5f);
        // (first-order expansion about 1/2: its neglected term is (d / a texel)^2 of a weight)
        if (d > 4e-3f)
            return false;
        *dev = fmaxf(*dev, d);
    }
    return true;
}

// B fragments of k_polar_mxd (plh_device.h): frag f = 4 * (2 j + kb) + kind, lane l, element e hold
//   T_j[i], i = 32 kb + k - 2 n, n = l & 15, k = 8 * (l >> 4) + e  (0 outside the 14 taps),
// T_j[i] = the normalised weight w' of tap (i - 6, j - 6) at fcoord (1/2, 1/2) -- kind 0 / 1 its f16
// hi / lo halves -- and kind 2 / 3 its derivative in fcoord_x / fcoord_y (least-squares slope over
// the phase classes that occur) times 2^-PLH_MX_DSHIFT. Rows j and 13 - j are averaged (they agree
// to rounding: the distance of a tap to the sample point is the same).
static bool polar_mxd_build(pl_gpu gpu, pl_log log, struct sh_sampler_obj *obj,
                            const struct plh_pass *pass, const float *wall, const uint32_t *taps,
                            int ntaps, int ncx, int ncy, const float *clsx, const float *clsy,
                            const float *colfc, const int32_t *colbase,
                            const float *rowfc, const int32_t *rowbase)
{
    const struct plh_sampler_args *s = &pass->s;
    const int W = pass->width, H = pass->height;
    enum { NT = PLH_MXD_TAPS };
    if (s->tile_fp32 || s->address_mode != PLH_ADDRESS_CLAMP || pass->transpose || s->src.w < 2 ||
        s->antiring > 0)
        return false;
    // (56 KiB of B fragments + a 140 x 76 tile of three f16 planes: one workgroup per CU)
    if (gpu->glsl.max_shmem_size < 124 * 1024) {
        pl_msg(log, PL_LOG_DEBUG, "matrix-pipe downscale: needs 124 KiB of shared memory, the limit is %zu",
               (size_t) gpu->glsl.max_shmem_size);
        return false;
    }
    float dev = 0.0f;
    if (!mxd_axis(colfc, colbase, W, &dev) || !mxd_axis(rowfc, rowbase, H, &dev))
        return false;
    // the class pair at exactly (1/2, 1/2): the expansion point
    int c0x = -1, c0y = -1;
    for (int c = 0; c < ncx; c++)
        c0x = clsx[c] == 0.5f ? c : c0x;
    for (int c = 0; c < ncy; c++)
        c0y = clsy[c] == 0.5f ? c : c0y;
    if (c0x < 0 || c0y < 0) {
        pl_msg(log, PL_LOG_DEBUG, "matrix-pipe downscale: no output at phase 1/2 exactly");
        return false;
    }
    int tap_at[NT][NT];
    for (int y = 0; y < NT; y++) {
        for (int x = 0; x < NT; x++)
            tap_at[y][x] = -1;
    }
    for (int t = 0; t < ntaps; t++) {
        const int x = (int8_t) (taps[t] & 0xff), y = (int8_t) ((taps[t] >> 8) & 0xff);
        if (x < -6 || x > 7 || y < -6 || y > 7)
            return false;
        tap_at[y + 6][x + 6] = t;
    }
#define WN(kx, ky, t) ((double) wall[((size_t) (ky) * ncx + (kx)) * (ntaps + 1) + (t)] * \
                       (double) wall[((size_t) (ky) * ncx + (kx)) * (ntaps + 1) + ntaps])
    double *sx = calloc(PL_MAX(ntaps, 1), sizeof(double)), *sy = calloc(PL_MAX(ntaps, 1), sizeof(double));
    const size_t nfx = ((size_t) W + PLH_MX_PAD - 1) / PLH_MX_PAD * PLH_MX_PAD;
    const size_t nfy = ((size_t) H + PLH_MX_PAD - 1) / PLH_MX_PAD * PLH_MX_PAD;
    const size_t frag_bytes = (size_t) PLH_MXD_NFRAG * 64 * 8 * sizeof(uint16_t);
    const size_t o_dfx = frag_bytes, o_dfy = o_dfx + nfx * 4, bytes = o_dfy + nfy * 4;
    uint8_t *blob = calloc(1, bytes);
    if (!sx || !sy || !blob) {
        free(sx); free(sy); free(blob);
        return false;
    }
    double den = 0.0;
    for (int c = 0; c < ncx; c++) {
        const double d = (double) clsx[c] - 0.5;
        den += d * d;
        for (int t = 0; t < ntaps; t++)
            sx[t] += d * (WN(c, c0y, t) - WN(c0x, c0y, t));
    }
    for (int t = 0; t < ntaps; t++)
        sx[t] = den > 0.0 ? sx[t] / den : 0.0;
    den = 0.0;
    for (int c = 0; c < ncy; c++) {
        const double d = (double) clsy[c] - 0.5;
        den += d * d;
        for (int t = 0; t < ntaps; t++)
            sy[t] += d * (WN(c0x, c, t) - WN(c0x, c0y, t));
    }
    for (int t = 0; t < ntaps; t++)
        sy[t] = den > 0.0 ? sy[t] / den : 0.0;

    uint16_t *frag = (uint16_t *) blob;
    const double dscale = ldexp(1.0, -PLH_MX_DSHIFT);
    double worst = 0.0, asym = 0.0;
    // the first source row (and, mirrored, the last) that carries a weight at all: at fcoord = 1/2
    // rows -6 and 7 of the reference's 14 x 14 tap square lie 6.5 texels from the sample, beyond
    // twice any radius <= 3.25 (ewa_lanczos: 6.4766) -- the kernel starts its contraction there
    int first_row = NT / 2 - 1;
    for (int j = 0; j < NT / 2; j++) {
        for (int kb = 0; kb < 2; kb++) {
            for (int l = 0; l < 64; l++) {
                const int n = l & 15;
                for (int e = 0; e < 8; e++) {
                    const int i = 32 * kb + 8 * (l >> 4) + e - 2 * n;
                    double v = 0.0, vx = 0.0, vy = 0.0;
                    if (i >= 0 && i < NT) {
                        const int ta = tap_at[j][i], tb = tap_at[NT - 1 - j][i];
                        if ((ta < 0) != (tb < 0)) {
                            free(sx); free(sy); free(blob);
                            return false;   // (a tap list that is not symmetric: not this filter)
                        }
                        if (ta >= 0) {
                            const double wa = WN(c0x, c0y, ta), wb = WN(c0x, c0y, tb);
                            asym = PL_MAX(asym, fabs(wa - wb));
                            v = 0.5 * (wa + wb);
                            vx = 0.5 * (sx[ta] + sx[tb]);
                            vy = 0.5 * (sy[ta] - sy[tb]);
                        }
                    }
                    const uint16_t hi = f32_to_f16((float) v);
                    const uint16_t lo = f32_to_f16((float) (v - (double) f16_to_f32(hi)));
                    worst = PL_MAX(worst, fabs(v - (double) f16_to_f32(hi) - (double) f16_to_f32(lo)));
                    if (v != 0.0 || vx != 0.0 || vy != 0.0)
                        first_row = PL_MIN(first_row, j);
                    const size_t f = 4 * (size_t) (2 * j + kb);
                    frag[((f + 0) * 64 + l) * 8 + e] = hi;
                    frag[((f + 1) * 64 + l) * 8 + e] = lo;
                    frag[((f + 2) * 64 + l) * 8 + e] = f32_to_f16((float) (vx * dscale));
                    frag[((f + 3) * 64 + l) * 8 + e] = f32_to_f16((float) (vy * dscale));
                }
            }
        }
    }
#undef WN
    free(sx);
    free(sy);
    if (asym > 1e-7) {
        pl_msg(log, PL_LOG_DEBUG, "matrix-pipe downscale: weights not symmetric about the sample "
               "point (%.2e)", asym);
        free(blob);
        return false;
    }
    float *dfx = (float *) (blob + o_dfx), *dfy = (float *) (blob + o_dfy);
    const float up = ldexpf(1.0f, PLH_MX_DSHIFT);
    for (int i = 0; i < W; i++)
        dfx[i] = (colfc[i] - 0.5f) * up;
    for (int i = 0; i < H; i++)
        dfy[i] = (rowfc[i] - 0.5f) * up;

    pl_buf_destroy(gpu, &obj->mx_blob);
    obj->mx_blob = pl_buf_create(gpu, pl_buf_params(.size = bytes, .storable = true,
                                                    .initial_data = blob));
    free(blob);
    if (!obj->mx_blob)
        return false;
    const char *base = pl_hip_buf_ptr(obj->mx_blob);
    obj->mx_host = (struct plh_polar_mx) {
        .enabled = 2,
        .org_x = colbase[0] - 6, .org_y = rowbase[0] - 6,
        .row_first = { first_row, 0 },
        .bfrag = base,
        .dfx = (const float *) (base + o_dfx), .dfy = (const float *) (base + o_dfy),
    };
    obj->mx_announced = false;
    pl_msg(log, PL_LOG_DEBUG, "matrix-pipe tables for the polar pass: 2 : 1 downscale, one phase (1/2, 1/2), "
           "per-pixel phases within %.2e: first-order terms; row symmetry %.1e, weight split error <= %.2e",
           dev, asym, worst);
    return true;
}


/* ---- PASS A fusion ------------------------------------------------------------------------------ */

bool plh_shader_sample_polar_fused(pl_shader sh, const pl_shader pre,
                                   const struct pl_sample_src *src,
                                   const struct pl_sample_filter_params *params)
{
    const struct plh_pass *pp = &pre->pass;
    pl_tex tex = pre->src_tex;
    if (!tex || pre->failed || pre->kind != PLH_SHADER_PASS || pre->detect_peak)
        return false;
    // `pre` must be an identity fetch of the whole texture: then FBO texel (i, j) would hold
    // f16(ops(texel (i, j))), including what clamped reads beyond the edges see
    if (pp->s.type != PLH_SAMPLE_NEAREST && !(pp->s.type == PLH_SAMPLE_BILINEAR && pp->s.rect_on_grid))
        return false;
    const pl_rect2df *rc = &pre->src_rect;
    if (rc->x0 != 0 || rc->y0 != 0 || rc->x1 != tex->params.w || rc->y1 != tex->params.h)
        return false;
    int ow, oh;
    if (pl_shader_output_size(pre, &ow, &oh) && (ow != tex->params.w || oh != tex->params.h))
        return false;
    if (src->tex->params.w != tex->params.w || src->tex->params.h != tex->params.h)
        return false; // `src->tex` is the FBO the caller would have rendered `pre` into
    const bool scaled = pp->s.scale != 1.0f;
    if (pp->num_pre_ops || pp->num_ops + scaled > PLH_MAX_OPS - 6)
        return false;
    for (int i = 0; i < pp->num_ops; i++) {
        if (pp->ops[i].kind == PLH_OP_DITHER || pp->ops[i].kind == PLH_OP_PEAK_DETECT ||
            pp->ops[i].kind == PLH_OP_PLANE_FETCH || pp->ops[i].kind == PLH_OP_DOVI_RESHAPE ||
            pp->ops[i].kind == PLH_OP_DOVI_LMS)
            return false; // position dependent / needs its own kernel
    }

    struct pl_sample_src fsrc = *src;
    fsrc.tex = tex;
    fsrc.address_mode = pp->s.address_mode;
    if (!sample_polar(sh, &fsrc, params, true))
        return false;

    // sample -> * scale -> ops, per source texel; the f16 tile rounds like the FBO store would
    struct plh_pass *p = &sh->pass;
    int n = 0;
    if (scaled) {
        struct plh_op *op = &p->ops[n++];
        memset(op, 0, sizeof(*op));
        op->kind = PLH_OP_SCALE;
        op->f[0] = op->f[1] = op->f[2] = op->f[3] = pp->s.scale;
    }
    memcpy(&p->ops[n], pp->ops, pp->num_ops * sizeof(struct plh_op));
    n += pp->num_ops;
    p->num_pre_ops = p->num_ops = n;
    for (int i = 0; i < pre->num_held; i++)
        sh_hold(sh, pre->held[i]);
    sh_listf(sh, "fused_pre_ops(%d ops of '%s' run per source texel, f16 tile)\n", n,
             sh_description(pre));
    return true;
}

/* ---- separable (orthogonal) filters: pl_shader_sample_ortho2, sampling.c:950-1104 -------- */

// The axis a separable pass filters along: the one whose size changes. -1 if both do.
enum ortho_axis { ORTHO_VERT = 0, ORTHO_HORIZ = 1 };
static int ortho_axis_of(const struct src_info *info, float *ratio)
{
    const bool keeps_x = fabs(info->ratio_x - 1.0f) < 1e-6f;
    const bool keeps_y = fabs(info->ratio_y - 1.0f) < 1e-6f;
    if (keeps_x) {
        *ratio = info->ratio_y;
        return ORTHO_VERT;
    }
    if (keeps_y) {
        *ratio = info->ratio_x;
        return ORTHO_HORIZ;
    }
    return -1;
}

// The filter a pass runs: the caller's, with the renderer-wide anti-ringing as its default and
// its kernel stretched by the downscaling factor (a downscale by k sums over k times the
// support) unless widening is switched off.
static struct pl_filter_config effective_filter(const struct pl_sample_filter_params *params, float ratio)
{
    struct pl_filter_config cfg = params->filter;
    float stretch = 1.0 / ratio;
    if (stretch < 1.0f || params->no_widening)
        stretch = 1.0;
    if (!cfg.antiring)
        cfg.antiring = params->antiring;
    cfg.blur = (cfg.blur ? cfg.blur : 1.0f) * stretch;
    return cfg;
}

// Rows of the weight table as the kernel reads them. Filters without negative lobes use the
// "linear trick" (sampling.c:914-942): taps are fetched in pairs through the bilinear unit, so
// a row holds (w0 + w1, w1 / (w0 + w1)) per pair, the padding repeating the last group.
static float *ortho_rows(pl_filter filt, bool paired)
{
    const int taps = filt->row_size, stride = filt->row_stride;
    const size_t entries = (size_t) SCALER_LUT_SIZE * stride;
    float *rows = malloc(entries * sizeof(float));
    if (!rows)
        return NULL;
    memcpy(rows, filt->weights, entries * sizeof(float));
    if (!paired)
        return rows;
    for (int phase = 0; phase < SCALER_LUT_SIZE; phase++) {
        float *row = rows + (size_t) phase * stride;
        for (int t = 0; t < taps; t += 2) {
            const float sum = row[t] + row[t + 1];
            row[t + 1] = row[t + 1] / sum;
            row[t] = sum;
        }
        for (int t = (taps + 1) & ~1; t < stride; t++)
            row[t] = t >= 4 ? row[t - 4] : 0.0f;
    }
    return rows;
}

bool pl_shader_sample_ortho2(pl_shader sh, const struct pl_sample_src *src,
                             const struct pl_sample_filter_params *params)
{
    if (params->filter.polar) {
        SH_FAIL(sh, "Trying to use separated sampling with a polar filter?");
        return false;
    }
    struct src_info info;
    if (!setup_src(sh, src, &info, false, REQ_LINEAR))
        return false;
    float ratio;
    const int pass = ortho_axis_of(&info, &ratio);
    if (pass < 0) {
        SH_FAIL(sh, "Trying to use pl_shader_sample_ortho with a pl_sample_src that requires "
                "scaling in multiple directions (rx=%f, ry=%f), this is not possible!",
                info.ratio_x, info.ratio_y);
        return false;
    }

    // state: one sampler object per axis, the horizontal one hanging off the vertical one
    // (anamorphic content filters the two axes differently, sampling.c:985-995)
    pl_gpu gpu = SH_GPU(sh);
    struct sh_sampler_obj *obj = SH_OBJ(sh, params->lut, PL_SHADER_OBJ_SAMPLER,
                                        struct sh_sampler_obj, sh_sampler_uninit);
    if (obj && pass == ORTHO_HORIZ)
        obj = SH_OBJ(sh, &obj->pass2, PL_SHADER_OBJ_SAMPLER, struct sh_sampler_obj, sh_sampler_uninit);
    if (!obj)
        return false;

    const struct pl_filter_config cfg = effective_filter(params, ratio);
    const bool update = !obj->filter || !pl_filter_config_eq(&obj->filter->params.config, &cfg);
    if (update) {
        pl_filter_free(&obj->filter);
        obj->filter = pl_filter_generate(sh->log, pl_filter_params(
            .config = cfg, .lut_entries = SCALER_LUT_SIZE, .row_stride_align = 4,
            .max_row_size = gpu->limits.max_tex_2d_dim / 4,
        ));
        if (!obj->filter) {
            SH_FAIL(sh, "Failed initializing separated filter!");
            return false;
        }
    }
    pl_filter filt = obj->filter;
    const int N = filt->row_size, stride = filt->row_stride;
    // no negative lobe = the first zero crossing is the radius: pairs of taps per fetch, and
    // nothing for anti-ringing to clamp
    const bool use_linear = filt->radius == filt->radius_zero;
    const bool use_ar = cfg.antiring > 0 && ratio > 1.0 && !use_linear;

    if (update || !obj->lut) {
        float *rows = ortho_rows(filt, use_linear);
        if (!rows)
            return false;
        pl_buf_destroy(gpu, &obj->lut);
        obj->lut = pl_buf_create(gpu, pl_buf_params(
            .size = (size_t) SCALER_LUT_SIZE * stride * sizeof(float), .storable = true,
            .initial_data = rows));
        free(rows);
        if (!obj->lut) {
            SH_FAIL(sh, "Failed initializing separated LUT!");
            return false;
        }
    }

    describe_filter(sh, &cfg, pass ? "ortho (horiz)" : "ortho (vert)", ratio, ratio);

    // A texture unit returns the texel itself at texel centres (its fixed-point weights snap
    // to zero). Along the filtered axis every tap is fetched at a centre by construction;
    // across it the fetch is at a centre when the pass is 1:1 on the texel grid there.
    const float r0 = pass ? src->rect.y0 : src->rect.x0;
    const bool aligned = r0 == truncf(r0);

    struct plh_sampler_args *s = &sh->pass.s;
    s->type = PLH_SAMPLE_ORTHO;
    s->weights = pl_hip_buf_ptr(obj->lut);
    s->row_size = N;
    s->row_stride = stride;
    s->dir = pass ? 0 : 1;      // 0 = horizontal, 1 = vertical
    s->use_linear = use_linear;
    s->use_ar = use_ar;
    s->antiring = cfg.antiring;
    s->linear = !aligned;       // bilinear across the filtered axis
    sh_hold(sh, *params->lut);

    sh_listf(sh, "sample_ortho(filter=%s, dir=%s, taps=%d, stride=%d, linear_trick=%d, "
             "antiring=%g, scale=%g, mask=0x%x, across=%s)\n", PL_DEF(cfg.name, "custom"),
             pass ? "horiz" : "vert", N, stride, use_linear, use_ar ? cfg.antiring : 0.0f,
             info.scale, info.comp_mask, aligned ? "nearest" : "linear");
    return true;
}

/* ---- debanding: pl_shader_deband, sampling.c:183-275 ---------------------------------------- */

void pl_shader_deband(pl_shader sh, const struct pl_sample_src *src,
                      const struct pl_deband_params *params)
{
    struct src_info info;
    if (!setup_src(sh, src, &info, false, REQ_NEAREST))
        return;

    params = PL_DEF(params, &pl_deband_default_params);
    sh_describef(sh, "debanding");

    struct plh_sampler_args *s = &sh->pass.s;
    s->type = PLH_SAMPLE_DEBAND;
    s->linear = false;
    s->comp_mask = info.comp_mask & ~0x8u; // ignore alpha channel
    s->iterations = s->comp_mask ? PL_MAX(params->iterations, 0) : 0;
    s->db_radius = params->radius;
    s->db_threshold = params->threshold / (1000 * info.scale);
    s->db_grain = s->comp_mask && params->grain > 0 ? params->grain / (1000.0 * info.scale) : 0.0f;
    for (int c = 0, k = 0; c < 3; c++) {
        // grain_neutral is indexed by *enabled* component (sampling.c:258-261)
        if (s->comp_mask & (1u << c))
            s->db_neutral[c] = params->grain_neutral[k++] / info.scale;
    }
    s->prng_seed = sh->params.index;
    // (k_deband_lds stages a 98 x 66 texel window: not where the user has lowered the limit)
    s->db_lds = !SH_GPU(sh) || SH_GPU(sh)->glsl.max_shmem_size >= 52 * 1024;

    sh_listf(sh, "deband(iterations=%d, threshold=%g, radius=%g, grain=%g, scale=%g, "
             "mask=0x%x, seed=%u)\n", s->iterations, params->threshold, params->radius,
             params->grain, info.scale, s->comp_mask, s->prng_seed);
}

/* ---- pl_shader_distort (reference src/shaders/sampling.c:1106-1217) ---------------------------- */

const struct pl_distort_params pl_distort_default_params = { PL_DISTORT_DEFAULTS };

void pl_shader_distort(pl_shader sh, pl_tex src_tex, int out_w, int out_h,
                       const struct pl_distort_params *params)
{
    if (!params || !src_tex) {
        SH_FAIL(sh, "pl_shader_distort: parameters and a texture are required");
        return;
    }
    if (sh->pass.s.type != PLH_SAMPLE_NONE || sh->output != PL_SHADER_SIG_NONE) {
        SH_FAIL(sh, "Illegal sequence of shader operations: a sampling stage must "
                "be the first stage of a shader");
        return;
    }
    if (!sh_require(sh, PL_SHADER_SIG_NONE, out_w, out_h))
        return;

    // the image in aspect-normalised coordinates: its longer side spans [-1, 1], y up
    const int src_w = src_tex->params.w, src_h = src_tex->params.h;
    float rx = 1.0f, ry = 1.0f;
    if (src_w > src_h) {
        ry = (float) src_h / src_w;
    } else {
        rx = (float) src_w / src_h;
    }
    const pl_transform2x2 tex2norm = {
        .mat.m = {{ 2 * rx, 0 }, { 0, -2 * ry }},
        .c = { -rx, ry },
    };
    // ... and from there to the canvas [-1, 1]^2
    const float sx = params->unscaled ? (float) src_w / out_w : 1.0f;
    const float sy = params->unscaled ? (float) src_h / out_h : 1.0f;
    const pl_transform2x2 norm2canvas = {
        .mat.m = {{ sx / rx, 0 }, { 0, sy / ry }},
    };

    pl_transform2x2 transform = params->transform;
    pl_transform2x2_mul(&transform, &tex2norm);
    pl_transform2x2_rmul(&norm2canvas, &transform);
    if (params->constrain) {
        const pl_rect2df unit = { .x1 = 1, .y1 = 1 };
        const pl_rect2df bb = pl_transform2x2_bounds(&transform, &unit);
        const float k = fmaxf(fmaxf(pl_rect_w(bb), pl_rect_h(bb)), 2.0f);
        pl_transform2x2_scale(&transform, 2.0f / k);
    }

    // the kernel walks the canvas (a vertex attribute in the reference, :1156-1161: y runs from +1
    // at the top row to -1) and needs the way back: canvas -> texture coordinates
    if (!sh_bind(sh, src_tex, params->address_mode, NULL))
        return;
    pl_transform2x2_invert(&transform);
    sh_describef(sh, "distortion");

    struct plh_pass *pass = &sh->pass;
    struct plh_sampler_args *s = &pass->s;
    s->type = PLH_SAMPLE_DISTORT;
    s->pos[0][0] = -1.0f; s->pos[0][1] =  1.0f;
    s->pos[1][0] =  1.0f; s->pos[1][1] =  1.0f;
    s->pos[2][0] = -1.0f; s->pos[2][1] = -1.0f;
    s->pos[3][0] =  1.0f; s->pos[3][1] = -1.0f;
    s->scale = 1.0f;
    s->comp_mask = 0xf;
    s->linear = true;
    pass->distort = (struct plh_distort_args) {
        .m = { transform.mat.m[0][0], transform.mat.m[0][1],
               transform.mat.m[1][0], transform.mat.m[1][1] },
        .c = { transform.c[0], transform.c[1] },
        .bicubic = params->bicubic,
        .alpha_mode = params->alpha_mode,
    };
    sh_listf(sh, "distort(tf=[%g %g; %g %g] + (%g, %g)%s%s)\n", pass->distort.m[0],
             pass->distort.m[1], pass->distort.m[2], pass->distort.m[3], pass->distort.c[0],
             pass->distort.c[1], params->bicubic ? ", bicubic" : "",
             params->alpha_mode ? ", transparent outside" : "");
}
