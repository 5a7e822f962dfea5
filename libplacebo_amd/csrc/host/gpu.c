/*
 * libplacebo-hip — the `pl_gpu` object of the HIP backend.
 *
 * Role of src/gpu.c (validation front-end, pl_find_fmt :94-128) plus a backend
 * file such as src/dummy.c / src/opengl/gpu.c (object creation, format table,
 * limits) in the reference. Textures are pitched linear device arrays; all
 * work is ordered on one HIP stream, so uploads/passes/downloads issued in API
 * order execute in that order without further fences.
 */
#include <stdlib.h>
#include <string.h>

#include <libplacebo/hip.h>

#include "gpu_priv.h"
#include "shaders_priv.h"
#include "cache_priv.h"

const struct pl_hip_params pl_hip_default_params = {0};

/* ------------------------------------------------------------------------ */
/* format table                                                              */

#define CAPS_ALL (PL_FMT_CAP_SAMPLEABLE | PL_FMT_CAP_STORABLE | PL_FMT_CAP_LINEAR |    \
                  PL_FMT_CAP_RENDERABLE | PL_FMT_CAP_BLENDABLE | PL_FMT_CAP_BLITTABLE | \
                  PL_FMT_CAP_HOST_READABLE | PL_FMT_CAP_READWRITE)

#define FMT(nm, ty, n, bits, plhfmt, vtx, gtype, gfmt) {                                   \
    .pub = {                                                                            \
        .name = nm, .type = ty, .num_components = n,                                    \
        .caps = CAPS_ALL | ((vtx) ? PL_FMT_CAP_VERTEX : 0),                             \
        .component_depth = { bits, (n) > 1 ? bits : 0, (n) > 2 ? bits : 0, (n) > 3 ? bits : 0 }, \
        .host_bits       = { bits, (n) > 1 ? bits : 0, (n) > 2 ? bits : 0, (n) > 3 ? bits : 0 }, \
        .sample_order = {0, 1, 2, 3},                                                   \
        .internal_size = (n) * (bits) / 8, .texel_size = (n) * (bits) / 8,              \
        .texel_align = (bits) / 8, .gatherable = true,                                  \
        .glsl_type = gtype, .glsl_format = gfmt,                                        \
    }, .plh = plhfmt }

static const struct fmt_priv fmt_table[] = {
    FMT("r8",       PL_FMT_UNORM, 1,  8, PLH_FMT_R8,      false, "float", "r8"),
    FMT("rg8",      PL_FMT_UNORM, 2,  8, PLH_FMT_RG8,     false, "vec2",  "rg8"),
    FMT("rgba8",    PL_FMT_UNORM, 4,  8, PLH_FMT_RGBA8,   false, "vec4",  "rgba8"),
    FMT("r16",      PL_FMT_UNORM, 1, 16, PLH_FMT_R16,     false, "float", "r16"),
    FMT("rg16",     PL_FMT_UNORM, 2, 16, PLH_FMT_RG16,    false, "vec2",  "rg16"),
    FMT("rgba16",   PL_FMT_UNORM, 4, 16, PLH_FMT_RGBA16,  false, "vec4",  "rgba16"),
    FMT("r16hf",    PL_FMT_FLOAT, 1, 16, PLH_FMT_R16F,    false, "float", "r16f"),
    FMT("rg16hf",   PL_FMT_FLOAT, 2, 16, PLH_FMT_RG16F,   false, "vec2",  "rg16f"),
    FMT("rgba16hf", PL_FMT_FLOAT, 4, 16, PLH_FMT_RGBA16F, false, "vec4",  "rgba16f"),
    FMT("r32f",     PL_FMT_FLOAT, 1, 32, PLH_FMT_R32F,    true,  "float", "r32f"),
    FMT("rg32f",    PL_FMT_FLOAT, 2, 32, PLH_FMT_RG32F,   true,  "vec2",  "rg32f"),
    FMT("rgba32f",  PL_FMT_FLOAT, 4, 32, PLH_FMT_RGBA32F, true,  "vec4",  "rgba32f"),
};

#define NUM_FMTS ((int) PL_ARRAY_SIZE(fmt_table))

// Same ordering rule as the reference's pl_gpu_finalize (gpu/utils.c:26-81);
// all our formats share caps, so this reduces to "lower depth first, then name"
static int cmp_fmt(const void *pa, const void *pb)
{
    pl_fmt a = *(pl_fmt *) pa, b = *(pl_fmt *) pb;
    for (int i = 0; i < 4; i++) {
        if (a->component_depth[i] != b->component_depth[i])
            return a->component_depth[i] < b->component_depth[i] ? -1 : 1;
        if (a->host_bits[i] != b->host_bits[i])
            return a->host_bits[i] < b->host_bits[i] ? -1 : 1;
    }
    return strcmp(a->name, b->name);
}

bool pl_fmt_is_ordered(pl_fmt fmt)
{
    for (int i = 0; i < fmt->num_components; i++) {
        if (fmt->sample_order[i] != i)
            return false;
    }
    return true;
}

bool pl_fmt_is_float(pl_fmt fmt)
{
    return fmt->type == PL_FMT_UNORM || fmt->type == PL_FMT_SNORM ||
           fmt->type == PL_FMT_FLOAT;
}

pl_fmt pl_find_fmt(pl_gpu gpu, enum pl_fmt_type type, int num_components,
                   int min_depth, int host_bits, enum pl_fmt_caps caps)
{
    for (int n = 0; n < gpu->num_formats; n++) {
        pl_fmt fmt = gpu->formats[n];
        if (fmt->type != type || fmt->num_components != num_components)
            continue;
        if ((fmt->caps & caps) != caps)
            continue;
        if (host_bits && (fmt->opaque || !pl_fmt_is_ordered(fmt) ||
                          fmt->texel_size * 8 != (size_t) host_bits * num_components))
            continue;

        bool ok = true;
        for (int i = 0; i < fmt->num_components; i++) {
            ok &= fmt->component_depth[i] >= min_depth;
            ok &= !host_bits || fmt->host_bits[i] == host_bits;
        }
        if (ok)
            return fmt;
    }
    return NULL;
}

pl_fmt pl_find_vertex_fmt(pl_gpu gpu, enum pl_fmt_type type, int comps)
{
    for (int n = 0; n < gpu->num_formats; n++) {
        pl_fmt fmt = gpu->formats[n];
        if (fmt->type == type && fmt->num_components == comps &&
            (fmt->caps & PL_FMT_CAP_VERTEX) && fmt->host_bits[0] == 32)
            return fmt;
    }
    return NULL;
}

bool pl_fmt_has_modifier(pl_fmt fmt, uint64_t modifier)
{
    for (int i = 0; fmt && i < fmt->num_modifiers; i++) {
        if (fmt->modifiers[i] == modifier)
            return true;
    }
    return false;
}

pl_fmt pl_find_fourcc(pl_gpu gpu, uint32_t fourcc)
{
    for (int n = 0; fourcc && n < gpu->num_formats; n++) {
        if (gpu->formats[n]->fourcc == fourcc)
            return gpu->formats[n];
    }
    return NULL;    // no DRM interop on this backend: nothing carries a fourcc
}

size_t pl_var_type_size(enum pl_var_type type)
{
    return type == PL_VAR_SINT || type == PL_VAR_UINT || type == PL_VAR_FLOAT ? 4 : 0;
}

int pl_desc_namespace(pl_gpu gpu, enum pl_desc_type type)
{
    (void) gpu;
    return (int) type;  // bindings are numbered per descriptor type
}

const struct pl_blend_params pl_alpha_overlay = {
    .src_rgb = PL_BLEND_SRC_ALPHA,
    .dst_rgb = PL_BLEND_ONE_MINUS_SRC_ALPHA,
    .src_alpha = PL_BLEND_ONE,
    .dst_alpha = PL_BLEND_ONE_MINUS_SRC_ALPHA,
};

void pl_gpu_set_cache(pl_gpu gpu, pl_cache cache)
{
    GPU_PRIV(gpu)->cache = cache;
}

pl_cache plh_gpu_cache(pl_gpu gpu)
{
    return gpu ? GPU_PRIV(gpu)->cache : NULL;
}

pl_fmt pl_find_named_fmt(pl_gpu gpu, const char *name)
{
    for (int n = 0; name && n < gpu->num_formats; n++) {
        if (!strcmp(gpu->formats[n]->name, name))
            return gpu->formats[n];
    }
    return NULL;
}

/* Test hook (tests/test_render_plan.py): a format description without a device, so that the
 * renderer's planner can be exercised on CPU-only hosts. */
PL_API pl_fmt plh_test_format(const char *name);
pl_fmt plh_test_format(const char *name)
{
    for (int i = 0; i < NUM_FMTS; i++) {
        if (!strcmp(fmt_table[i].pub.name, name))
            return &fmt_table[i].pub;
    }
    return NULL;
}

/* ------------------------------------------------------------------------ */
/* backend object                                                            */

int pl_hip_device_count(void)
{
    return plh_dev_count();
}

pl_hip pl_hip_create(pl_log log, const struct pl_hip_params *params)
{
    params = PL_DEF(params, &pl_hip_default_params);
    if (plh_dev_count() <= params->device) {
        pl_msg(log, PL_LOG_FATAL, "pl_hip_create: no HIP device %d (found %d). "
               "There is no CPU fallback for this backend.",
               params->device, plh_dev_count());
        return NULL;
    }

    struct gpu_priv *p = calloc(1, sizeof(*p));
    if (!p)
        return NULL;

    int err = plh_dev_open(params->device, &p->info);
    if (err) {
        pl_msg(log, PL_LOG_FATAL, "pl_hip_create: opening device %d failed: %s",
               params->device, plh_strerror(err));
        free(p);
        return NULL;
    }

    p->device = params->device;
    if (params->stream) {
        p->stream = params->stream;
    } else {
        err = plh_stream_create(p->device, &p->stream);
        if (err) {
            pl_msg(log, PL_LOG_FATAL, "pl_hip_create: stream creation failed: %s",
                   plh_strerror(err));
            free(p);
            return NULL;
        }
        p->own_stream = true;
    }

    struct pl_gpu_t *gpu = &p->gpu;
    gpu->log = log;
    gpu->glsl = (struct pl_glsl_version) {
        .version = 450,
        .vulkan = true,
        .compute = true,
        .max_shmem_size = PL_DEF(params->max_shmem_size, 160 * 1024), // CDNA4 LDS per CU
        .max_group_threads = 1024,
        .max_group_size = { 1024, 1024, 1024 },
        .subgroup_size = 64,
        .min_gather_offset = -32,
        .max_gather_offset = 31,
    };
    gpu->limits = (struct pl_gpu_limits) {
        .thread_safe = false,
        .callbacks = false,
        .max_buf_size = p->info.total_mem,
        .max_ubo_size = 65536,
        .max_ssbo_size = p->info.total_mem,
        .max_tex_1d_dim = 1 << 16,
        .max_tex_2d_dim = 1 << 16,
        .max_tex_3d_dim = 0,
        .buf_transfer = true,
        .align_tex_xfer_pitch = 256,
        .align_tex_xfer_offset = 256,
        .max_variable_comps = 0,
        .max_constants = 0,
        .array_size_constants = true,
        .max_pushc_size = 4096, // kernel arguments
        .max_dispatch = { 1u << 31, 65535, 65535 },
        .fragment_queues = 0,   // every pass is a compute pass (dispatch.c:1236)
        .compute_queues = 1,
    };
    memcpy(gpu->uuid, p->info.uuid, 16);
    gpu->pci = (struct pl_gpu_pci_address) {
        .domain = p->info.pci_domain, .bus = p->info.pci_bus, .device = p->info.pci_device,
    };

    for (int i = 0; i < NUM_FMTS; i++) {
        p->fmt_store[i] = fmt_table[i];
        p->fmt_store[i].pub.num_planes = 1;
        p->fmt_store[i].pub.planes[0].format = &p->fmt_store[i].pub;
        p->fmt_store[i].pub.signature = plh_mem_hash(fmt_table[i].pub.name,
                                                     strlen(fmt_table[i].pub.name));
        p->fmts[i] = &p->fmt_store[i].pub;
    }
    qsort(p->fmts, NUM_FMTS, sizeof(p->fmts[0]), cmp_fmt);
    gpu->formats = p->fmts;
    gpu->num_formats = NUM_FMTS;

    p->hip = (struct pl_hip_t) {
        .gpu = gpu,
        .device = p->device,
        .stream = p->stream,
        .arch = p->info.arch,
        .compute_units = p->info.compute_units,
    };

    pl_msg(log, PL_LOG_INFO, "pl_hip: device %d '%s' (%s), %d CUs, %zu MiB",
           p->device, p->info.name, p->info.arch, p->info.compute_units,
           p->info.total_mem >> 20);
    return &p->hip;
}

void pl_hip_destroy(pl_hip *hip)
{
    if (!hip || !*hip)
        return;
    struct gpu_priv *p = GPU_PRIV((*hip)->gpu);
    plh_stream_sync(p->stream);
    for (int i = 0; i < PLH_STAGE_SLOTS; i++) {
        plh_event_destroy(p->stage[i].done);
        plh_host_free(p->stage[i].host);
    }
    if (p->own_stream)
        plh_stream_destroy(p->stream);
    free(p);
    *hip = NULL;
}

pl_hip pl_hip_get(pl_gpu gpu)
{
    return gpu ? &GPU_PRIV(gpu)->hip : NULL;
}

void pl_gpu_flush(pl_gpu gpu)
{
    (void) gpu; // HIP submits eagerly
}

void pl_gpu_finish(pl_gpu gpu)
{
    struct gpu_priv *p = GPU_PRIV(gpu);
    const int err = plh_stream_sync(p->stream);
    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_gpu_finish: %s", plh_strerror(err));
        p->failed = true;
    }
}

bool pl_gpu_is_failed(pl_gpu gpu)
{
    return GPU_PRIV(gpu)->failed;
}

/* ------------------------------------------------------------------------ */
/* textures                                                                  */

static bool check_tex_params(pl_gpu gpu, const struct pl_tex_params *params)
{
    if (!params->format || params->w <= 0 || params->h < 0 || params->d != 0) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_create: invalid parameters "
               "(need format, w > 0, d == 0; got %dx%dx%d)", params->w, params->h, params->d);
        return false;
    }
    if ((uint32_t) params->w > gpu->limits.max_tex_2d_dim ||
        (uint32_t) params->h > gpu->limits.max_tex_2d_dim) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_create: %dx%d exceeds max_tex_2d_dim",
               params->w, params->h);
        return false;
    }
    return true;
}

pl_tex pl_tex_create(pl_gpu gpu, const struct pl_tex_params *params)
{
    if (!check_tex_params(gpu, params))
        return NULL;

    struct gpu_priv *g = GPU_PRIV(gpu);
    struct tex_priv *t = calloc(1, sizeof(*t));
    if (!t)
        return NULL;
    t->tex.params = *params;
    t->tex.params.initial_data = NULL;
    t->tex.sampler_type = PL_SAMPLER_NORMAL;
    t->gpu = gpu;
    t->plh_fmt = FMT_PRIV(params->format)->plh;
    const int rows = PL_MAX(params->h, 1);
    const size_t row_bytes = (size_t) params->w * params->format->texel_size;
    t->pitch = PL_ALIGN2(row_bytes, (size_t) 256);
    t->ptr = plh_malloc(g->device, t->pitch * rows);
    t->owned = true;
    if (!t->ptr) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_create: out of device memory (%zu bytes)",
               t->pitch * rows);
        free(t);
        return NULL;
    }

    if (params->initial_data) {
        const int err = plh_copy2d_h2d(g->stream, t->ptr, t->pitch, params->initial_data,
                                       row_bytes, row_bytes, rows);
        // initial_data may be freed by the caller right away
        if (err || plh_stream_sync(g->stream)) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_create: initial upload failed");
            plh_free(t->ptr);
            free(t);
            return NULL;
        }
    }
    return &t->tex;
}

pl_tex pl_hip_wrap(pl_gpu gpu, const struct pl_hip_wrap_params *params)
{
    if (!params || !params->ptr || !params->format || params->width <= 0 || params->height <= 0) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_hip_wrap: invalid parameters");
        return NULL;
    }
    struct tex_priv *t = calloc(1, sizeof(*t));
    if (!t)
        return NULL;
    t->tex.params = (struct pl_tex_params) {
        .w = params->width, .h = params->height, .format = params->format,
        .sampleable = true, .renderable = true, .storable = true,
        .blit_src = true, .blit_dst = true, .host_writable = true, .host_readable = true,
    };
    t->gpu = gpu;
    t->plh_fmt = FMT_PRIV(params->format)->plh;
    t->ptr = params->ptr;
    t->pitch = PL_DEF(params->row_pitch, (size_t) params->width * params->format->texel_size);
    t->owned = false;
    return &t->tex;
}

void pl_tex_destroy(pl_gpu gpu, pl_tex *tex)
{
    if (!tex || !*tex)
        return;
    struct tex_priv *t = TEX_PRIV(*tex);
    if (t->owned) {
        // the allocation may still be referenced by queued work
        plh_stream_sync(GPU_PRIV(gpu)->stream);
        plh_free(t->ptr);
    }
    free(t);
    *tex = NULL;
}

static bool tex_params_compat(const struct pl_tex_params *a, const struct pl_tex_params *b)
{
    return a->w == b->w && a->h == b->h && a->d == b->d && a->format == b->format &&
           a->sampleable == b->sampleable && a->renderable == b->renderable &&
           a->storable == b->storable && a->blit_src == b->blit_src &&
           a->blit_dst == b->blit_dst && a->host_writable == b->host_writable &&
           a->host_readable == b->host_readable;
}

bool pl_tex_recreate(pl_gpu gpu, pl_tex *tex, const struct pl_tex_params *params)
{
    if (params->initial_data) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_recreate may not be used with `initial_data`!");
        return false;
    }
    if (*tex && tex_params_compat(&(*tex)->params, params)) {
        pl_tex_invalidate(gpu, *tex);
        return true;
    }
    pl_tex_destroy(gpu, tex);
    *tex = pl_tex_create(gpu, params);
    return !!*tex;
}

void pl_tex_invalidate(pl_gpu gpu, pl_tex tex)
{
    (void) gpu; (void) tex; // contents become undefined: nothing to do
}

void plh_tex_view(pl_tex tex, struct plh_view *out)
{
    const struct tex_priv *t = TEX_PRIV(tex);
    *out = (struct plh_view) {
        .ptr = t->ptr, .w = tex->params.w, .h = PL_MAX(tex->params.h, 1),
        .pitch = (int32_t) t->pitch, .fmt = t->plh_fmt,
    };
}

void *pl_hip_tex_ptr(pl_tex tex, size_t *out_row_pitch)
{
    const struct tex_priv *t = TEX_PRIV(tex);
    if (out_row_pitch)
        *out_row_pitch = t->pitch;
    return t->ptr;
}

void pl_tex_clear_ex(pl_gpu gpu, pl_tex dst, const union pl_clear_color color)
{
    struct plh_view v;
    plh_tex_view(dst, &v);
    const int err = plh_launch_clear(GPU_PRIV(gpu)->stream, &v, color.f);
    if (err)
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_clear: %s", plh_strerror(err));
}

void pl_tex_clear(pl_gpu gpu, pl_tex dst, const float color[4])
{
    union pl_clear_color c;
    memcpy(c.f, color, sizeof(c.f));
    pl_tex_clear_ex(gpu, dst, c);
}

// A blit is a pass with a bare nearest / bilinear sampler and no colour stages: the same
// kernels the renderer uses (the reference emulates blits with a compute shader the same way
// on backends without a native one, src/gpu/utils.c:852).
void pl_tex_blit(pl_gpu gpu, const struct pl_tex_blit_params *params)
{
    pl_tex src = params->src, dst = params->dst;
    if (!src || !dst || !src->params.blit_src || !dst->params.blit_dst) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_blit: needs a blit_src source and a blit_dst target");
        return;
    }
    if (pl_fmt_is_float(src->params.format) != pl_fmt_is_float(dst->params.format)) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_blit: incompatible formats");
        return;
    }
    if (params->sample_mode == PL_TEX_SAMPLE_LINEAR &&
        !(src->params.format->caps & PL_FMT_CAP_LINEAR)) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_blit: source format is not linearly sampleable");
        return;
    }

    pl_rect3d sr = params->src_rc, dr = params->dst_rc;
    if (!sr.x0 && !sr.x1) sr.x1 = src->params.w;
    if (!sr.y0 && !sr.y1) sr.y1 = PL_MAX(src->params.h, 1);
    if (!dr.x0 && !dr.x1) dr.x1 = dst->params.w;
    if (!dr.y0 && !dr.y1) dr.y1 = PL_MAX(dst->params.h, 1);
    const int w = abs(dr.x1 - dr.x0), h = abs(dr.y1 - dr.y0);
    if (!w || !h || PL_MIN(dr.x0, dr.x1) < 0 || PL_MIN(dr.y0, dr.y1) < 0 ||
        PL_MAX(dr.x0, dr.x1) > dst->params.w || PL_MAX(dr.y0, dr.y1) > PL_MAX(dst->params.h, 1)) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_blit: target rect outside the texture");
        return;
    }

    struct plh_pass *pass = calloc(1, sizeof(*pass));
    if (!pass)
        return;
    struct plh_sampler_args *s = &pass->s;
    s->type = params->sample_mode == PL_TEX_SAMPLE_LINEAR ? PLH_SAMPLE_BILINEAR : PLH_SAMPLE_NEAREST;
    plh_tex_view(src, &s->src);
    const float sx = 1.0f / s->src.w, sy = 1.0f / s->src.h;
    const float x0 = sx * sr.x0, x1 = sx * sr.x1, y0 = sy * sr.y0, y1 = sy * sr.y1;
    s->pos[0][0] = x0; s->pos[0][1] = y0;
    s->pos[1][0] = x1; s->pos[1][1] = y0;
    s->pos[2][0] = x0; s->pos[2][1] = y1;
    s->pos[3][0] = x1; s->pos[3][1] = y1;
    s->pt[0] = sx;
    s->pt[1] = sy;
    s->scale = 1.0f;
    s->comp_mask = 0xf;
    s->linear = s->type == PLH_SAMPLE_BILINEAR;
    s->rect_w = abs(sr.x1 - sr.x0);
    s->rect_h = abs(sr.y1 - sr.y0);
    s->rect_on_grid = 1;

    plh_tex_view(dst, &pass->dst);
    pass->width = w;
    pass->height = h;
    pass->out_scale[0] = 1.0 / w;
    pass->out_scale[1] = 1.0 / h;
    pass->base_x = dr.x0 - (dr.x0 > dr.x1);
    pass->base_y = dr.y0 - (dr.y0 > dr.y1);
    pass->dir_x = dr.x0 > dr.x1 ? -1 : 1;
    pass->dir_y = dr.y0 > dr.y1 ? -1 : 1;
    // a 1:1 copy returns the texels themselves (what a texture unit does on the grid)
    if (s->type == PLH_SAMPLE_BILINEAR && s->rect_w == w && s->rect_h == h)
        s->type = PLH_SAMPLE_NEAREST;

    const int err = plh_launch_pass(GPU_PRIV(gpu)->stream, pass);
    free(pass);
    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_blit: %s", plh_strerror(err));
        GPU_PRIV(gpu)->failed = true;
    }
}

static bool tex_transfer(pl_gpu gpu, const struct pl_tex_transfer_params *params, bool upload)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    pl_tex tex = params->tex;
    if (!tex || (!params->ptr && !params->buf)) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: missing tex or ptr/buf",
               upload ? "upload" : "download");
        return false;
    }
    if (upload ? !tex->params.host_writable : !tex->params.host_readable) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: texture lacks host_%s",
               upload ? "upload" : "download", upload ? "writable" : "readable");
        return false;
    }

    const struct tex_priv *t = TEX_PRIV(tex);
    pl_rect3d rc = params->rc;
    if (!rc.x0 && !rc.x1) rc.x1 = tex->params.w;
    if (!rc.y0 && !rc.y1) rc.y1 = PL_MAX(tex->params.h, 1);
    if (rc.x0 < 0 || rc.y0 < 0 || rc.x1 > tex->params.w || rc.y1 > PL_MAX(tex->params.h, 1) ||
        rc.x1 <= rc.x0 || rc.y1 <= rc.y0) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: invalid rect", upload ? "upload" : "download");
        return false;
    }

    const char *what = upload ? "upload" : "download";
    const size_t tsz = tex->params.format->texel_size;
    const size_t row_bytes = (size_t) (rc.x1 - rc.x0) * tsz;
    const size_t rows = rc.y1 - rc.y0;
    const size_t host_pitch = PL_DEF(params->row_pitch, row_bytes);
    uint8_t *dev = (uint8_t *) t->ptr + (size_t) rc.y0 * t->pitch + (size_t) rc.x0 * tsz;

    // what the reference's front-end rejects before a backend sees it (src/gpu.c:440-497)
    if (tex->params.d || rc.z0 || (rc.z1 && rc.z1 != 1)) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: 3D textures are not supported", what);
        return false;
    }
    if (host_pitch < row_bytes || host_pitch % PL_DEF(tex->params.format->texel_align, 1)) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: row_pitch %zu is below the row size %zu or not "
               "a multiple of the texel alignment", what, host_pitch, row_bytes);
        return false;
    }
    if (!params->buf == !params->ptr) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: exactly one of `buf` and `ptr` must be set", what);
        return false;
    }
    if (params->buf) {
        pl_buf buf = params->buf;
        const size_t span = (rows - 1) * host_pitch + row_bytes;   // pl_tex_transfer_size
        if (params->buf_offset + span < params->buf_offset ||
            params->buf_offset + span > buf->params.size) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: %zu bytes at offset %zu exceed the buffer "
                   "(%zu bytes)", what, span, params->buf_offset, buf->params.size);
            return false;
        }
        if (!gpu->limits.buf_transfer) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: buffer transfers unsupported", what);
            return false;
        }
    }

    if (params->timer)
        plh_timer_begin(gpu, params->timer);

    int err;
    if (params->buf) {
        uint8_t *bptr = (uint8_t *) BUF_PRIV(params->buf)->ptr + params->buf_offset;
        err = upload ? plh_copy2d_d2d(g->stream, dev, t->pitch, bptr, host_pitch, row_bytes, rows)
                     : plh_copy2d_d2d(g->stream, bptr, host_pitch, dev, t->pitch, row_bytes, rows);
    } else {
        err = upload ? plh_copy2d_h2d(g->stream, dev, t->pitch, params->ptr, host_pitch, row_bytes, rows)
                     : plh_copy2d_d2h(g->stream, params->ptr, host_pitch, dev, t->pitch, row_bytes, rows);
        // pageable host memory: the reference's contract is that `ptr` may be
        // reused / is filled when the call returns (gpu.h, no callback given)
        if (!err && !params->callback)
            err = plh_stream_sync(g->stream);
    }

    if (params->timer)
        plh_timer_end(gpu, params->timer);

    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_%s: %s", upload ? "upload" : "download",
               plh_strerror(err));
        g->failed = true;
        return false;
    }
    if (params->callback) {
        plh_stream_sync(g->stream);
        params->callback(params->priv);
    }
    return true;
}

bool pl_tex_upload(pl_gpu gpu, const struct pl_tex_transfer_params *params)
{
    return tex_transfer(gpu, params, true);
}

bool pl_tex_download(pl_gpu gpu, const struct pl_tex_transfer_params *params)
{
    return tex_transfer(gpu, params, false);
}

// One in-order stream: an object is in use at most for as long as the stream has unfinished
// work. With a timeout the call waits (any non-zero timeout: a frame is milliseconds).
static bool stream_busy(pl_gpu gpu, uint64_t timeout)
{
    if (timeout) {
        pl_gpu_finish(gpu);
        return false;
    }
    return plh_stream_idle(GPU_PRIV(gpu)->stream) == 0;
}

bool pl_tex_poll(pl_gpu gpu, pl_tex tex, uint64_t timeout)
{
    (void) tex;
    return stream_busy(gpu, timeout);
}

/* ------------------------------------------------------------------------ */
/* buffers                                                                   */

pl_buf pl_buf_create(pl_gpu gpu, const struct pl_buf_params *params)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    if (!params->size || params->size > gpu->limits.max_buf_size) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_create: size %zu outside (0, %zu]", params->size,
               gpu->limits.max_buf_size);
        return NULL;
    }
    if (params->import_handle || params->export_handle) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_create: this backend neither imports nor exports "
               "buffer handles (wrap device memory with pl_hip_wrap instead)");
        return NULL;
    }
    struct buf_priv *b = calloc(1, sizeof(*b));
    if (!b)
        return NULL;
    b->buf.params = *params;
    b->buf.params.initial_data = NULL;
    b->ptr = plh_malloc(g->device, params->size);
    if (!b->ptr) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_create: out of device memory");
        free(b);
        return NULL;
    }
    if (params->initial_data) {
        plh_copy2d_h2d(g->stream, b->ptr, params->size, params->initial_data, params->size,
                       params->size, 1);
        plh_stream_sync(g->stream);
    }
    return &b->buf;
}

void pl_buf_destroy(pl_gpu gpu, pl_buf *buf)
{
    if (!buf || !*buf)
        return;
    plh_stream_sync(GPU_PRIV(gpu)->stream);
    plh_free(BUF_PRIV(*buf)->ptr);
    free(BUF_PRIV(*buf));
    *buf = NULL;
}

bool pl_buf_recreate(pl_gpu gpu, pl_buf *buf, const struct pl_buf_params *params)
{
    if (*buf && (*buf)->params.size == params->size && !params->initial_data)
        return true;
    pl_buf_destroy(gpu, buf);
    *buf = pl_buf_create(gpu, params);
    return !!*buf;
}

void *pl_hip_buf_ptr(pl_buf buf)
{
    return BUF_PRIV(buf)->ptr;
}

// Backend half: no validation, used by the library's own tables (which are created without
// host access flags, as device-only storage).
void plh_buf_write(pl_gpu gpu, pl_buf buf, size_t buf_offset, const void *data, size_t size)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    uint8_t *dst = (uint8_t *) BUF_PRIV(buf)->ptr + buf_offset;
    if (size <= PLH_STAGE_BYTES) {
        // through a pinned slot: `data` is the caller's again as soon as it is copied there,
        // the device copy is ordered on the stream like everything else
        const int i = g->stage_next;
        if (!g->stage[i].host) {
            g->stage[i].host = plh_host_alloc(PLH_STAGE_BYTES);
            if (g->stage[i].host && plh_event_create(&g->stage[i].done)) {
                plh_host_free(g->stage[i].host);
                g->stage[i].host = NULL;
            }
        }
        if (g->stage[i].host) {
            if (g->stage[i].in_flight)
                plh_event_sync(g->stage[i].done);   // eight uploads ago: long finished
            memcpy(g->stage[i].host, data, size);
            if (!plh_copy2d_h2d(g->stream, dst, size, g->stage[i].host, size, size, 1) &&
                !plh_event_record(g->stage[i].done, g->stream)) {
                g->stage[i].in_flight = true;
                g->stage_next = (i + 1) % PLH_STAGE_SLOTS;
                return;
            }
        }
    }
    plh_copy2d_h2d(g->stream, dst, size, data, size, size, 1);
    plh_stream_sync(g->stream);
}

bool plh_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    int err = plh_copy2d_d2h(g->stream, dest, size, (uint8_t *) BUF_PRIV(buf)->ptr + buf_offset,
                             size, size, 1);
    err = err ? err : plh_stream_sync(g->stream);
    return !err;
}

// Front-end half: the API contract (reference src/gpu.c:662-716). A violation is reported
// and the call does nothing -- never an out-of-bounds device access.
static bool buf_range_ok(pl_gpu gpu, const char *fn, pl_buf buf, size_t offset, size_t size)
{
    if (offset + size < offset || offset + size > buf->params.size) {
        pl_msg(gpu->log, PL_LOG_ERR, "%s: %zu bytes at offset %zu exceed the buffer (%zu bytes)%s%s",
               fn, size, offset, buf->params.size, buf->params.debug_tag ? " for buffer: " : "",
               buf->params.debug_tag ? buf->params.debug_tag : "");
        return false;
    }
    return true;
}

void pl_buf_write(pl_gpu gpu, pl_buf buf, size_t buf_offset, const void *data, size_t size)
{
    if (!buf->params.host_writable) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_write: buffer was not created host_writable");
        return;
    }
    if (buf_offset % 4) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_write: offset %zu is not a multiple of 4", buf_offset);
        return;
    }
    if (buf_range_ok(gpu, "pl_buf_write", buf, buf_offset, size))
        plh_buf_write(gpu, buf, buf_offset, data, size);
}

bool pl_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size)
{
    if (!buf->params.host_readable) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_read: buffer was not created host_readable");
        return false;
    }
    return buf_range_ok(gpu, "pl_buf_read", buf, buf_offset, size) &&
           plh_buf_read(gpu, buf, buf_offset, dest, size);
}

void pl_buf_copy(pl_gpu gpu, pl_buf dst, size_t dst_offset, pl_buf src, size_t src_offset, size_t size)
{
    if (src == dst) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_copy: source and destination are the same buffer");
        return;
    }
    if (!buf_range_ok(gpu, "pl_buf_copy (src)", src, src_offset, size) ||
        !buf_range_ok(gpu, "pl_buf_copy (dst)", dst, dst_offset, size))
        return;
    plh_copy2d_d2d(GPU_PRIV(gpu)->stream, (uint8_t *) BUF_PRIV(dst)->ptr + dst_offset, size,
                   (uint8_t *) BUF_PRIV(src)->ptr + src_offset, size, size, 1);
}

/* ------------------------------------------------------------------------ */
/* pl_pass: a recorded op list behind the reference's pass interface          */
//
// The reference's pl_pass is a compiled GLSL program (src/gpu.c:1025-1290). This backend compiles
// nothing at run time; what it can turn into a pass is what pl_shader recorded. pl_shader_finalize
// therefore ends pl_shader_res.glsl with a line "#pl_hip_pass <ticket>" that resolves -- for as
// long as that shader is alive, i.e. for as long as the pl_shader_res is valid at all -- to the
// recorded sampler + colour ops. pl_pass_create copies them (and takes references on the
// shader's state objects: LUTs, filter tables, peak buffers), so the pass outlives the shader and
// can be run any number of times; textures the shader sampled are bound by address and must
// outlive the pass, like any descriptor in the reference. Text without such a line (GLSL) is
// refused with a message naming the alternative.

struct pass_priv {
    struct pl_pass_t pub;
    struct plh_pass pass;
    bool transpose, detect_peak;
    void *polar_obj;
    pl_shader_obj peak_state;
    int out_w, out_h;
    pl_shader_obj held[16];
    int num_held;
    pl_buf noise;
    char *text;
    struct pl_desc *descs;
};

pl_pass pl_pass_create(pl_gpu gpu, const struct pl_pass_params *params)
{
    if (!params->glsl_shader) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: `glsl_shader` is NULL");
        return NULL;
    }
    pl_shader sh = plh_shader_from_glsl(params->glsl_shader);
    if (!sh) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: this backend runs precompiled HIP kernels and "
               "has no GLSL compiler. A pass can be created from the text pl_shader_finalize() "
               "returns for a shader recorded on this pl_gpu (while that shader is alive); "
               "otherwise record with pl_shader_* and run with pl_dispatch_finish / _compute.");
        return NULL;
    }
    if (SH_GPU(sh) != gpu || sh->failed || sh->kind != PLH_SHADER_PASS ||
        sh->input != PL_SHADER_SIG_NONE || sh->output != PL_SHADER_SIG_COLOR) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: the shader behind this text belongs to "
               "another pl_gpu, failed, or does not produce a colour from no input");
        return NULL;
    }
    if (params->type != PL_PASS_RASTER && params->type != PL_PASS_COMPUTE) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: invalid pass type");
        return NULL;
    }
    if (params->num_variables || params->num_constants || params->push_constants_size ||
        params->blend_params) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: variables, constants, push constants and "
               "blending do not exist on this backend (the recorded ops carry their values)");
        return NULL;
    }
    if (params->type == PL_PASS_RASTER && (!params->target_format ||
        !(params->target_format->caps & PL_FMT_CAP_STORABLE))) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: raster passes need a storable `target_format` "
               "(every pass is a compute launch here)");
        return NULL;
    }
    if (params->type == PL_PASS_COMPUTE &&
        (params->num_descriptors > 1 ||
         (params->num_descriptors == 1 && params->descriptors[0].type != PL_DESC_STORAGE_IMG))) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: a compute pass takes at most one descriptor, "
               "the storage image it writes");
        return NULL;
    }

    struct pass_priv *p = calloc(1, sizeof(*p));
    if (!p)
        return NULL;
    p->pub.params = *params;
    p->text = strdup(params->glsl_shader);
    p->pub.params.glsl_shader = p->text;
    p->pub.params.vertex_shader = NULL;
    p->pub.params.vertex_attribs = NULL;
    p->pub.params.num_vertex_attribs = 0;
    p->pub.params.variables = NULL;
    p->pub.params.constants = NULL;
    p->pub.params.constant_data = NULL;
    p->pub.params.descriptors = NULL;
    if (params->num_descriptors) {
        p->descs = calloc(params->num_descriptors, sizeof(*p->descs));
        if (p->descs) {
            memcpy(p->descs, params->descriptors, params->num_descriptors * sizeof(*p->descs));
            for (int i = 0; i < params->num_descriptors; i++)
                p->descs[i].name = NULL;
        }
        p->pub.params.descriptors = p->descs;
    }
    if (!p->text || (params->num_descriptors && !p->descs)) {
        free(p->text);
        free(p->descs);
        free(p);
        return NULL;
    }
    p->pass = sh->pass;
    p->transpose = sh->transpose;
    p->detect_peak = sh->detect_peak;
    p->peak_state = sh->peak_state;
    p->polar_obj = sh->polar_obj;
    p->out_w = sh->output_w;
    p->out_h = sh->output_h;
    for (int i = 0; i < sh->num_held; i++) {
        p->held[p->num_held++] = sh->held[i];
        sh->held[i]->refcount++;
    }
    return &p->pub;
}

void pl_pass_destroy(pl_gpu gpu, pl_pass *pass)
{
    if (!pass || !*pass)
        return;
    struct pass_priv *p = (struct pass_priv *) *pass;
    pl_gpu_finish(gpu);     // launches of this pass may still read its objects
    for (int i = 0; i < p->num_held; i++)
        pl_shader_obj_destroy(&p->held[i]);
    pl_buf_destroy(gpu, &p->noise);
    free(p->text);
    free(p->descs);
    free(p);
    *pass = NULL;
}

void pl_pass_run(pl_gpu gpu, const struct pl_pass_run_params *params)
{
    struct gpu_priv *g = GPU_PRIV(gpu);
    struct pass_priv *p = (struct pass_priv *) params->pass;
    if (!p) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: no pass");
        goto error;
    }
    if (params->num_var_updates || params->push_constants || params->vertex_buf ||
        params->index_data || params->index_buf) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: variables, push constants and vertex / index "
               "buffers do not exist on this backend");
        goto error;
    }

    pl_tex target = NULL;
    pl_rect2d rc = {0};
    if (p->pub.params.type == PL_PASS_RASTER) {
        target = params->target;
        if (!target || target->params.format != p->pub.params.target_format) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: `target` missing or not of the pass' "
                   "target_format");
            goto error;
        }
        rc = pl_rect_w(params->scissors) && pl_rect_h(params->scissors) ? params->scissors
                                                                       : params->viewport;
    } else if (p->pub.params.num_descriptors) {
        if (!params->desc_bindings || !params->desc_bindings[0].object) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: the storage image is not bound");
            goto error;
        }
        target = (pl_tex) params->desc_bindings[0].object;
    }

    struct plh_pass local = p->pass;    // the stored op list stays as it was recorded
    const struct plh_pass_exec x = {
        .pass = &local, .transpose = p->transpose, .polar_obj = p->polar_obj,
        .detect_peak = p->detect_peak, .peak_state = p->peak_state,
    };
    int err;
    if (target) {
        if (pl_tex_params_dimension(target->params) != 2 || !target->params.storable) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: the target must be a storable 2D texture");
            goto error;
        }
        if (!pl_rect_w(rc)) { rc.x0 = 0; rc.x1 = target->params.w; }
        if (!pl_rect_h(rc)) { rc.y0 = 0; rc.y1 = target->params.h; }
        const int tw = abs(pl_rect_w(rc)), th = abs(pl_rect_h(rc));
        const int need_w = p->transpose ? p->out_h : p->out_w, need_h = p->transpose ? p->out_w : p->out_h;
        if (need_w && need_h && (need_w != tw || need_h != th)) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: the pass was recorded for a %dx%d output, "
                   "the target rect is %dx%d", need_w, need_h, tw, th);
            goto error;
        }
        err = plh_pass_execute(gpu, gpu->log, &x, target, rc, params->timer, &p->noise);
    } else {
        // a pass without an image output (a measurement): it covers its recorded output size
        if (!p->out_w || !p->out_h) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: a compute pass without a storage image "
                   "needs a shader with a defined output size");
            goto error;
        }
        memset(&local.dst, 0, sizeof(local.dst));
        local.width = p->out_w;
        local.height = p->out_h;
        local.out_scale[0] = 1.0 / p->out_w;
        local.out_scale[1] = 1.0 / p->out_h;
        local.base_x = local.base_y = 0;
        local.dir_x = local.dir_y = 1;
        local.transpose = 0;
        local.frag_x0 = local.frag_y0 = 0;
        if (params->timer)
            plh_timer_begin(gpu, params->timer);
        err = plh_launch_pass(g->stream, &local);
        if (params->timer)
            plh_timer_end(gpu, params->timer);
        if (!err && p->detect_peak)
            plh_peak_pass_launched(gpu, p->peak_state);
    }
    if (err) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: %s", plh_strerror(err));
        g->failed = true;
    }
error:  // (API misuse: reported above, nothing was launched)
    return;
}

bool pl_buf_export(pl_gpu gpu, pl_buf buf)
{
    (void) buf;
    pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_export: buffers of this backend have no exportable handle");
    return false;
}

bool pl_buf_poll(pl_gpu gpu, pl_buf buf, uint64_t timeout)
{
    (void) buf;
    return stream_busy(gpu, timeout);
}

/* ------------------------------------------------------------------------ */
/* timers: ring of hipEvent pairs                                            */

pl_timer pl_timer_create(pl_gpu gpu)
{
    (void) gpu;
    struct pl_timer_t *t = calloc(1, sizeof(*t));
    if (!t)
        return NULL;
    for (int i = 0; i < PLH_TIMER_RING; i++) {
        if (plh_event_create(&t->start[i]) || plh_event_create(&t->stop[i])) {
            pl_timer_destroy(gpu, &t);
            return NULL;
        }
    }
    return t;
}

void pl_timer_destroy(pl_gpu gpu, pl_timer *timer)
{
    (void) gpu;
    if (!timer || !*timer)
        return;
    for (int i = 0; i < PLH_TIMER_RING; i++) {
        plh_event_destroy((*timer)->start[i]);
        plh_event_destroy((*timer)->stop[i]);
    }
    free(*timer);
    *timer = NULL;
}

void plh_timer_begin(pl_gpu gpu, pl_timer t)
{
    if (t->head - t->tail >= PLH_TIMER_RING)
        t->tail++; // drop the oldest sample
    plh_event_record(t->start[t->head % PLH_TIMER_RING], GPU_PRIV(gpu)->stream);
}

void plh_timer_end(pl_gpu gpu, pl_timer t)
{
    plh_event_record(t->stop[t->head % PLH_TIMER_RING], GPU_PRIV(gpu)->stream);
    t->head++;
}

uint64_t pl_timer_query(pl_gpu gpu, pl_timer t)
{
    (void) gpu;
    if (!t || t->tail == t->head)
        return 0;
    const int i = t->tail % PLH_TIMER_RING;
    if (plh_event_query(t->stop[i]) != 1)
        return 0;
    uint64_t ns = 0;
    plh_event_elapsed_ns(t->start[i], t->stop[i], &ns);
    t->tail++;
    return PL_MAX(ns, 1);
}
