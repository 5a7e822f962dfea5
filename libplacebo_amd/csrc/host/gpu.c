/*
 * libplacebo-hip -- the validating front-end of pl_gpu (the role of src/gpu.c in the reference):
 * format queries, and for every pl_tex_* / pl_buf_* / pl_pass_* / pl_timer_* entry point the checks
 * of the API contract (src/gpu.c:440-497, 543-716, 1025-1190), completion of the parameters, and
 * the call into the backend table (gpu_priv.h: struct plh_gpu_fns; implemented by gpu_hip.c).
 * A violation is reported and the call does nothing -- never an out-of-bounds device access.
 */
#include <stdlib.h>
#include <string.h>

#include <libplacebo/hip.h>

#include "gpu_priv.h"
#include "shaders_priv.h"
#include "cache_priv.h"

/* ------------------------------------------------------------------------ */
/* formats                                                                   */

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

// does `fmt` satisfy a pl_find_fmt query?
static bool fmt_answers(pl_fmt fmt, enum pl_fmt_type type, int comps, int min_depth, int host_bits,
                        enum pl_fmt_caps caps)
{
    if (fmt->type != type || fmt->num_components != comps || (fmt->caps & caps) != caps)
        return false;
    for (int i = 0; i < comps; i++) {
        if (fmt->component_depth[i] < min_depth)
            return false;
    }
    if (!host_bits)
        return true;
    // a host representation was asked for: plain, in order, exactly host_bits per component
    if (fmt->opaque || !pl_fmt_is_ordered(fmt) || fmt->texel_size * 8 != (size_t) host_bits * comps)
        return false;
    for (int i = 0; i < comps; i++) {
        if (fmt->host_bits[i] != host_bits)
            return false;
    }
    return true;
}

pl_fmt pl_find_fmt(pl_gpu gpu, enum pl_fmt_type type, int num_components,
                   int min_depth, int host_bits, enum pl_fmt_caps caps)
{
    // (formats are sorted best first)
    for (int n = 0; n < gpu->num_formats; n++) {
        if (fmt_answers(gpu->formats[n], type, num_components, min_depth, host_bits, caps))
            return gpu->formats[n];
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

/* ---- shader variables: types and buffer layouts (src/gpu.c:745-948) --------------------------- */
// One table for the constructors, the named list and the name lookup: GLSL type name, base type,
// columns, rows. (mat3x2 has a name but, as in the reference, no constructor and no list entry.)
#define VAR_TYPES(X)                                                                            \
    X(float, FLOAT, 1, 1) X(vec2, FLOAT, 1, 2) X(vec3, FLOAT, 1, 3) X(vec4, FLOAT, 1, 4)        \
    X(mat2, FLOAT, 2, 2) X(mat2x3, FLOAT, 2, 3) X(mat2x4, FLOAT, 2, 4) X(mat3, FLOAT, 3, 3)      \
    X(mat3x4, FLOAT, 3, 4) X(mat4x2, FLOAT, 4, 2) X(mat4x3, FLOAT, 4, 3) X(mat4, FLOAT, 4, 4)    \
    X(int, SINT, 1, 1) X(ivec2, SINT, 1, 2) X(ivec3, SINT, 1, 3) X(ivec4, SINT, 1, 4)            \
    X(uint, UINT, 1, 1) X(uvec2, UINT, 1, 2) X(uvec3, UINT, 1, 3) X(uvec4, UINT, 1, 4)

#define X(glsl, T, M, V)                                                                        \
    struct pl_var pl_var_##glsl(const char *name)                                                \
    {                                                                                           \
        return (struct pl_var) { .name = name, .type = PL_VAR_##T, .dim_v = V, .dim_m = M, .dim_a = 1 }; \
    }
VAR_TYPES(X)
#undef X

const struct pl_named_var pl_var_glsl_types[] = {
#define X(glsl, T, M, V) { #glsl, { .type = PL_VAR_##T, .dim_v = V, .dim_m = M, .dim_a = 1 } },
    VAR_TYPES(X)
#undef X
    {0},
};

const char *pl_var_glsl_type_name(struct pl_var var)
{
    for (const struct pl_named_var *n = pl_var_glsl_types; n->glsl_name; n++) {
        if (n->var.type == var.type && n->var.dim_m == var.dim_m && n->var.dim_v == var.dim_v)
            return n->glsl_name;
    }
    if (var.type == PL_VAR_FLOAT && var.dim_m == 3 && var.dim_v == 2)
        return "mat3x2";
    return NULL;
}

struct pl_var pl_var_from_fmt(pl_fmt fmt, const char *name)
{
    const enum pl_var_type type = fmt->type == PL_FMT_UINT ? PL_VAR_UINT :
                                  fmt->type == PL_FMT_SINT ? PL_VAR_SINT : PL_VAR_FLOAT;
    return (struct pl_var) { .name = name, .type = type, .dim_v = fmt->num_components,
                             .dim_m = 1, .dim_a = 1 };
}

enum var_packing { PACK_HOST, PACK_STD140, PACK_STD430 };

static struct pl_var_layout var_layout(enum var_packing rule, size_t offset, const struct pl_var *var)
{
    const size_t scalar = pl_var_type_size(var->type);
    const size_t columns = (size_t) var->dim_m * var->dim_a;    // a matrix is an array of columns
    size_t stride = scalar * var->dim_v, align = 1;
    if (rule != PACK_HOST) {
        // a vector aligns to its size, a three-component one like a four-component one; the
        // columns of an array / matrix are spaced by that alignment, which std140 rounds up to
        // a vec4's
        align = scalar * (var->dim_v == 3 ? 4 : var->dim_v);
        if (columns > 1) {
            if (rule == PACK_STD140)
                align = (align + 15) / 16 * 16;
            stride = align;
        }
    }
    return (struct pl_var_layout) {
        .offset = (offset + align - 1) / align * align,
        .stride = stride,
        .size   = stride * columns,
    };
}

struct pl_var_layout pl_var_host_layout(size_t offset, const struct pl_var *var)
{
    return var_layout(PACK_HOST, offset, var);
}

struct pl_var_layout pl_std140_layout(size_t offset, const struct pl_var *var)
{
    return var_layout(PACK_STD140, offset, var);
}

struct pl_var_layout pl_std430_layout(size_t offset, const struct pl_var *var)
{
    return var_layout(PACK_STD430, offset, var);
}

void memcpy_layout(void *dst, struct pl_var_layout dst_layout,
                   const void *src, struct pl_var_layout src_layout)
{
    uint8_t *d = (uint8_t *) dst + dst_layout.offset;
    const uint8_t *s = (const uint8_t *) src + src_layout.offset;
    if (src_layout.stride == dst_layout.stride) {
        memcpy(d, s, src_layout.size);      // same spacing: one block
        return;
    }
    // column by column, as many bytes as the narrower spacing holds
    const size_t column = PL_MIN(src_layout.stride, dst_layout.stride);
    for (size_t done = 0; done < src_layout.size; done += src_layout.stride) {
        memcpy(d, s + done, column);
        d += dst_layout.stride;
    }
}

const char *pl_desc_access_glsl_name(enum pl_desc_access mode)
{
    return mode == PL_DESC_ACCESS_READONLY ? "readonly" :
           mode == PL_DESC_ACCESS_WRITEONLY ? "writeonly" : "";
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

/* ------------------------------------------------------------------------ */
/* device                                                                    */

void pl_gpu_flush(pl_gpu gpu)
{
    GPU_FNS(gpu)->gpu_flush(gpu);
}

void pl_gpu_finish(pl_gpu gpu)
{
    GPU_FNS(gpu)->gpu_finish(gpu);
}

bool pl_gpu_is_failed(pl_gpu gpu)
{
    return GPU_FNS(gpu)->gpu_is_failed(gpu);
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
    return check_tex_params(gpu, params) ? GPU_FNS(gpu)->tex_create(gpu, params) : NULL;
}

void pl_tex_destroy(pl_gpu gpu, pl_tex *tex)
{
    if (!tex || !*tex)
        return;
    GPU_FNS(gpu)->tex_destroy(gpu, *tex);
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
    GPU_FNS(gpu)->tex_invalidate(gpu, tex);
}

void pl_tex_clear_ex(pl_gpu gpu, pl_tex dst, const union pl_clear_color color)
{
    if (!dst->params.blit_dst) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_tex_clear: the texture was not created blit_dst");
        return;
    }
    GPU_FNS(gpu)->tex_clear_ex(gpu, dst, color);
}

void pl_tex_clear(pl_gpu gpu, pl_tex dst, const float color[4])
{
    union pl_clear_color c;
    memcpy(c.f, color, sizeof(c.f));
    pl_tex_clear_ex(gpu, dst, c);
}

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

    struct pl_tex_blit_params fixed = *params;
    fixed.src_rc = sr;
    fixed.dst_rc = dr;
    GPU_FNS(gpu)->tex_blit(gpu, &fixed);
}

// the checks of fix_tex_transfer (src/gpu.c:440-497); `out` = the parameters with rc and
// row_pitch filled in
static bool tex_transfer_ok(pl_gpu gpu, const struct pl_tex_transfer_params *params, bool upload,
                            struct pl_tex_transfer_params *out)
{
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

    *out = *params;
    out->rc = rc;
    out->rc.z0 = 0;
    out->rc.z1 = 1;
    out->row_pitch = host_pitch;
    return true;
}

bool pl_tex_upload(pl_gpu gpu, const struct pl_tex_transfer_params *params)
{
    struct pl_tex_transfer_params fixed;
    return tex_transfer_ok(gpu, params, true, &fixed) && GPU_FNS(gpu)->tex_upload(gpu, &fixed);
}

bool pl_tex_download(pl_gpu gpu, const struct pl_tex_transfer_params *params)
{
    struct pl_tex_transfer_params fixed;
    return tex_transfer_ok(gpu, params, false, &fixed) && GPU_FNS(gpu)->tex_download(gpu, &fixed);
}

bool pl_tex_poll(pl_gpu gpu, pl_tex tex, uint64_t timeout)
{
    return GPU_FNS(gpu)->tex_poll(gpu, tex, timeout);
}

/* ------------------------------------------------------------------------ */
/* buffers                                                                   */

pl_buf pl_buf_create(pl_gpu gpu, const struct pl_buf_params *params)
{
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
    // what each use of a buffer may ask for (src/gpu.c:595-604); limits.max_mapped_size is 0 on
    // this backend, so a host_mapped buffer is refused here instead of coming back without `data`
    const struct { bool wanted; size_t limit; const char *what; } uses[] = {
        { params->uniform, gpu->limits.max_ubo_size, "uniform" },
        { params->storable, gpu->limits.max_ssbo_size, "storable" },
        { params->drawable, gpu->limits.max_vbo_size, "drawable" },
        { params->host_mapped, gpu->limits.max_mapped_size, "host_mapped" },
        { params->host_mapped && params->memory_type == PL_BUF_MEM_DEVICE,
          gpu->limits.max_mapped_vram, "host_mapped in device memory" },
    };
    for (size_t i = 0; i < PL_ARRAY_SIZE(uses); i++) {
        if (uses[i].wanted && params->size > uses[i].limit) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_create: %zu bytes exceed the limit for %s buffers "
                   "(%zu)%s%s", params->size, uses[i].what, uses[i].limit,
                   params->debug_tag ? " for buffer: " : "", params->debug_tag ? params->debug_tag : "");
            return NULL;
        }
    }
    if (params->format) {
        pl_fmt fmt = params->format;
        if (params->size > gpu->limits.max_buffer_texels * fmt->texel_size ||
            (params->uniform && !(fmt->caps & PL_FMT_CAP_TEXEL_UNIFORM)) ||
            (params->storable && !(fmt->caps & PL_FMT_CAP_TEXEL_STORAGE))) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_create: format '%s' cannot back this texel buffer",
                   fmt->name);
            return NULL;
        }
    }
    pl_buf buf = GPU_FNS(gpu)->buf_create(gpu, params);
    if (buf && params->host_mapped && !buf->data) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_create: backend returned an unmapped host_mapped buffer");
        pl_buf_destroy(gpu, &buf);
    }
    return buf;
}

void pl_buf_destroy(pl_gpu gpu, pl_buf *buf)
{
    if (!buf || !*buf)
        return;
    GPU_FNS(gpu)->buf_destroy(gpu, *buf);
    *buf = NULL;
}

bool pl_buf_recreate(pl_gpu gpu, pl_buf *buf, const struct pl_buf_params *params)
{
    // (src/gpu.c:644-660: a recreated buffer has no defined contents, so asking for some is an error)
    if (params->initial_data) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_recreate may not be used with `initial_data`!");
        return false;
    }
    // reusable iff the existing buffer can do everything the new one is asked to -- a LARGER
    // buffer too, as in the reference (pl_buf_params_superset, src/gpu.c:630-641): callers read
    // the size they asked for from their own parameters, not from buf->params
    if (*buf) {
        const struct pl_buf_params *have = &(*buf)->params;
        const bool covers = have->size >= params->size &&
            have->memory_type == params->memory_type && have->format == params->format &&
            have->host_writable >= params->host_writable && have->host_readable >= params->host_readable &&
            have->host_mapped >= params->host_mapped && have->uniform >= params->uniform &&
            have->storable >= params->storable && have->drawable >= params->drawable;
        if (covers)
            return true;
    }
    pl_buf_destroy(gpu, buf);
    *buf = pl_buf_create(gpu, params);
    return !!*buf;
}

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
        GPU_FNS(gpu)->buf_write(gpu, buf, buf_offset, data, size);
}

bool pl_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size)
{
    if (!buf->params.host_readable) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_read: buffer was not created host_readable");
        return false;
    }
    return buf_range_ok(gpu, "pl_buf_read", buf, buf_offset, size) &&
           GPU_FNS(gpu)->buf_read(gpu, buf, buf_offset, dest, size);
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
    GPU_FNS(gpu)->buf_copy(gpu, dst, dst_offset, src, src_offset, size);
}

// (reference: src/gpu/utils.c:1065-1076, the same requirements)
bool pl_buf_copy_swap(pl_gpu gpu, const struct pl_buf_copy_swap_params *params)
{
    pl_buf src = params->src, dst = params->dst;
    if (!src || !dst || !src->params.storable || !dst->params.storable) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_copy_swap: both buffers must be storable");
        return false;
    }
    if (params->src_offset % 4 || params->dst_offset % 4 || params->size % 4) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_copy_swap: offsets and size must be multiples of 4");
        return false;
    }
    if (params->wordsize != 2 && params->wordsize != 4) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_copy_swap: word size %d is neither 2 nor 4", params->wordsize);
        return false;
    }
    if (src == dst && params->src_offset != params->dst_offset) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_buf_copy_swap: an in-place swap needs equal offsets");
        return false;
    }
    if (!buf_range_ok(gpu, "pl_buf_copy_swap (src)", src, params->src_offset, params->size) ||
        !buf_range_ok(gpu, "pl_buf_copy_swap (dst)", dst, params->dst_offset, params->size))
        return false;
    return GPU_FNS(gpu)->buf_copy_swap(gpu, dst, params->dst_offset, src, params->src_offset,
                                       params->size, params->wordsize);
}

bool pl_buf_export(pl_gpu gpu, pl_buf buf)
{
    return GPU_FNS(gpu)->buf_export(gpu, buf);
}

bool pl_buf_poll(pl_gpu gpu, pl_buf buf, uint64_t timeout)
{
    return GPU_FNS(gpu)->buf_poll(gpu, buf, timeout);
}

/* ------------------------------------------------------------------------ */
/* pl_pass: a recorded op list behind the reference's pass interface          */
//
// The reference's pl_pass is a compiled GLSL program (src/gpu.c:1025-1290). This backend compiles
// nothing at run time; what it can turn into a pass is what pl_shader recorded. pl_shader_finalize
// therefore ends pl_shader_res.glsl with a line "#pl_hip_pass <ticket>" that resolves -- for as
// long as that shader is alive, i.e. for as long as the pl_shader_res is valid at all -- to the
// recorded sampler + colour ops. The backend copies them (and takes references on the shader's
// state objects: LUTs, filter tables, peak buffers), so the pass outlives the shader and can be
// run any number of times; textures the shader sampled are bound by address and must outlive the
// pass, like any descriptor in the reference. Text without such a line (GLSL) is refused with a
// message naming the alternative.

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
    for (int i = 0; i < sh->pass.num_ops; i++) {
        if (sh->pass.ops[i].kind == PLH_OP_DOVI_RESHAPE) {
            // its curves sit in a per-pass scratch slot (gpu_priv.h) that a long-lived pass would outlive
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_create: a shader with Dolby Vision reshaping "
                   "cannot be kept as a pass (its curves change per frame): dispatch it");
            return NULL;
        }
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

    return GPU_FNS(gpu)->pass_create(gpu, params, sh);
}

void pl_pass_destroy(pl_gpu gpu, pl_pass *pass)
{
    if (!pass || !*pass)
        return;
    GPU_FNS(gpu)->pass_destroy(gpu, *pass);
    *pass = NULL;
}

void pl_pass_run(pl_gpu gpu, const struct pl_pass_run_params *params)
{
    pl_pass pass = params->pass;
    if (!pass) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: no pass");
        return;
    }
    if (params->num_var_updates || params->push_constants || params->vertex_buf ||
        params->index_data || params->index_buf) {
        pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: variables, push constants and vertex / index "
               "buffers do not exist on this backend");
        return;
    }

    // the image the pass writes: a raster pass' target, a compute pass' storage image
    pl_tex target = NULL;
    pl_rect2d rc = {0};
    if (pass->params.type == PL_PASS_RASTER) {
        target = params->target;
        if (!target || target->params.format != pass->params.target_format) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: `target` missing or not of the pass' "
                   "target_format");
            return;
        }
        rc = pl_rect_w(params->scissors) && pl_rect_h(params->scissors) ? params->scissors
                                                                       : params->viewport;
    } else if (pass->params.num_descriptors) {
        if (!params->desc_bindings || !params->desc_bindings[0].object) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: the storage image is not bound");
            return;
        }
        target = (pl_tex) params->desc_bindings[0].object;
    }
    if (target) {
        if (pl_tex_params_dimension(target->params) != 2 || !target->params.storable) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: the target must be a storable 2D texture");
            return;
        }
        if (!pl_rect_w(rc)) { rc.x0 = 0; rc.x1 = target->params.w; }
        if (!pl_rect_h(rc)) { rc.y0 = 0; rc.y1 = target->params.h; }
        if (PL_MIN(rc.x0, rc.x1) < 0 || PL_MIN(rc.y0, rc.y1) < 0 ||
            PL_MAX(rc.x0, rc.x1) > target->params.w || PL_MAX(rc.y0, rc.y1) > target->params.h) {
            pl_msg(gpu->log, PL_LOG_ERR, "pl_pass_run: the viewport / scissors lie outside the target");
            return;
        }
    }
    GPU_FNS(gpu)->pass_run(gpu, params, target, rc);
}

/* ------------------------------------------------------------------------ */
/* timers                                                                    */

pl_timer pl_timer_create(pl_gpu gpu)
{
    return GPU_FNS(gpu)->timer_create(gpu);
}

void pl_timer_destroy(pl_gpu gpu, pl_timer *timer)
{
    if (!timer || !*timer)
        return;
    GPU_FNS(gpu)->timer_destroy(gpu, *timer);
    *timer = NULL;
}

uint64_t pl_timer_query(pl_gpu gpu, pl_timer timer)
{
    return timer ? GPU_FNS(gpu)->timer_query(gpu, timer) : 0;
}
