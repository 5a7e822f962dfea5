/*
 * libplacebo-hip: the pl_gpu abstraction (textures, buffers, formats, timers).
 *
 * API-compatible subset of the reference's src/include/libplacebo/gpu.h for
 * everything the render hot path touches: pl_gpu_t (:205-226), pl_fmt_t
 * (:306-373), pl_tex_params / pl_tex_t (:672-754), pl_tex_transfer_params
 * (:845-905), pl_buf (:449-560), pl_timer (:400-420). There is no GLSL pass
 * object (pl_pass): on this backend a finished pl_shader is lowered straight
 * to precompiled HIP kernels by pl_dispatch (see dispatch.h).
 */
#ifndef LIBPLACEBO_GPU_H_
#define LIBPLACEBO_GPU_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include <libplacebo/common.h>
#include <libplacebo/log.h>

PL_API_BEGIN

typedef const char *pl_debug_tag;
#define PL_STRINGIFY_(x) #x
#define PL_TOSTRING(x) PL_STRINGIFY_(x)
#define PL_DEBUG_TAG (__FILE__ ":" PL_TOSTRING(__LINE__))

// Capabilities of the kernel "language". Field names follow the reference's
// pl_glsl_version because generic code sizes its work from them
// (sampling.c:668-699, colorspace.c:1176, renderer.c:2290).
struct pl_glsl_version {
    int version;                // 450 (semantics of the reference's GLSL 450 path)
    bool gles;
    bool vulkan;
    bool compute;               // always true on HIP
    size_t max_shmem_size;      // LDS per workgroup (bytes)
    uint32_t max_group_threads; // 1024
    uint32_t max_group_size[3];
    uint32_t subgroup_size;     // 64 (wavefront)
    int16_t min_gather_offset;
    int16_t max_gather_offset;
};

struct pl_gpu_limits {
    bool thread_safe;
    bool callbacks;
    size_t max_buf_size;
    size_t max_ubo_size;
    size_t max_ssbo_size;
    size_t max_vbo_size;
    size_t max_mapped_size;
    uint64_t max_buffer_texels;
    uint32_t max_tex_1d_dim;
    uint32_t max_tex_2d_dim;
    uint32_t max_tex_3d_dim;
    bool blittable_1d_3d;
    bool buf_transfer;
    size_t align_tex_xfer_pitch;
    size_t align_tex_xfer_offset;
    size_t max_variable_comps;
    size_t max_constants;
    bool array_size_constants;
    size_t max_pushc_size;
    uint32_t max_dispatch[3];
    uint32_t fragment_queues;   // 0: every pass is a compute pass
    uint32_t compute_queues;
};

struct pl_gpu_pci_address {
    uint32_t domain, bus, device, function;
};

typedef const struct pl_fmt_t *pl_fmt;

typedef const struct pl_gpu_t {
    pl_log log;
    struct pl_glsl_version glsl;
    struct pl_gpu_limits limits;
    uint8_t uuid[16];
    pl_fmt *formats;            // sorted best-first, like pl_gpu_finalize does
    int num_formats;
    struct pl_gpu_pci_address pci;
} *pl_gpu;

enum pl_fmt_type {
    PL_FMT_UNKNOWN = 0,
    PL_FMT_UNORM,
    PL_FMT_SNORM,
    PL_FMT_UINT,
    PL_FMT_SINT,
    PL_FMT_FLOAT,
    PL_FMT_TYPE_COUNT,
};

enum pl_fmt_caps {
    PL_FMT_CAP_SAMPLEABLE    = 1 << 0,
    PL_FMT_CAP_STORABLE      = 1 << 1,
    PL_FMT_CAP_LINEAR        = 1 << 2,
    PL_FMT_CAP_RENDERABLE    = 1 << 3,
    PL_FMT_CAP_BLENDABLE     = 1 << 4,
    PL_FMT_CAP_BLITTABLE     = 1 << 5,
    PL_FMT_CAP_VERTEX        = 1 << 6,
    PL_FMT_CAP_TEXEL_UNIFORM = 1 << 7,
    PL_FMT_CAP_TEXEL_STORAGE = 1 << 8,
    PL_FMT_CAP_HOST_READABLE = 1 << 9,
    PL_FMT_CAP_READWRITE     = 1 << 10,
};

struct pl_fmt_t {
    const char *name;           // e.g. "rgba16hf"
    uint64_t signature;
    enum pl_fmt_type type;
    enum pl_fmt_caps caps;
    int num_components;
    int component_depth[4];
    size_t internal_size;
    bool opaque;
    bool emulated;
    size_t texel_size;
    size_t texel_align;
    int host_bits[4];
    int sample_order[4];
    bool gatherable;
    const char *glsl_type;
    const char *glsl_format;
};

PL_API bool pl_fmt_is_ordered(pl_fmt fmt);
PL_API bool pl_fmt_is_float(pl_fmt fmt);
PL_API pl_fmt pl_find_fmt(pl_gpu gpu, enum pl_fmt_type type, int num_components,
                          int min_depth, int host_bits, enum pl_fmt_caps caps);
PL_API pl_fmt pl_find_vertex_fmt(pl_gpu gpu, enum pl_fmt_type type, int num_components);
PL_API pl_fmt pl_find_named_fmt(pl_gpu gpu, const char *name);

// GPU timers (hipEvent pairs). pl_timer_query returns elapsed nanoseconds of
// the oldest finished measurement, or 0 if none is available yet.
typedef struct pl_timer_t *pl_timer;
PL_API pl_timer pl_timer_create(pl_gpu gpu);
PL_API void pl_timer_destroy(pl_gpu gpu, pl_timer *);
PL_API uint64_t pl_timer_query(pl_gpu gpu, pl_timer);

// Buffers: plain device allocations with host read/write
struct pl_buf_params {
    size_t size;
    bool host_writable;
    bool host_readable;
    bool host_mapped;
    bool uniform;
    bool storable;
    const void *initial_data;
    void *user_data;
    pl_debug_tag debug_tag;
};

#define pl_buf_params(...) (&(struct pl_buf_params) { .debug_tag = PL_DEBUG_TAG, __VA_ARGS__ })

typedef const struct pl_buf_t {
    struct pl_buf_params params;
    uint8_t *data; // host_mapped only
} *pl_buf;

PL_API pl_buf pl_buf_create(pl_gpu gpu, const struct pl_buf_params *params);
PL_API void pl_buf_destroy(pl_gpu gpu, pl_buf *buf);
PL_API bool pl_buf_recreate(pl_gpu gpu, pl_buf *buf, const struct pl_buf_params *params);
PL_API void pl_buf_write(pl_gpu gpu, pl_buf buf, size_t buf_offset, const void *data, size_t size);
PL_API bool pl_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size);
PL_API void pl_buf_copy(pl_gpu gpu, pl_buf dst, size_t dst_offset,
                        pl_buf src, size_t src_offset, size_t size);
PL_API bool pl_buf_poll(pl_gpu gpu, pl_buf buf, uint64_t timeout);

enum pl_tex_sample_mode {
    PL_TEX_SAMPLE_NEAREST,
    PL_TEX_SAMPLE_LINEAR,
    PL_TEX_SAMPLE_MODE_COUNT,
};

enum pl_tex_address_mode {
    PL_TEX_ADDRESS_CLAMP,
    PL_TEX_ADDRESS_REPEAT,
    PL_TEX_ADDRESS_MIRROR,
    PL_TEX_ADDRESS_MODE_COUNT,
};

enum pl_sampler_type {
    PL_SAMPLER_NORMAL,
    PL_SAMPLER_RECT,
    PL_SAMPLER_EXTERNAL,
    PL_SAMPLER_TYPE_COUNT,
};

struct pl_tex_params {
    int w, h, d;            // d must be 0 (2D only; 1D = h == 0)
    pl_fmt format;
    bool sampleable;
    bool renderable;
    bool storable;
    bool blit_src;
    bool blit_dst;
    bool host_writable;
    bool host_readable;
    const void *initial_data; // tightly packed
    void *user_data;
    pl_debug_tag debug_tag;
};

#define pl_tex_params(...) (&(struct pl_tex_params) { .debug_tag = PL_DEBUG_TAG, __VA_ARGS__ })

static inline int pl_tex_params_dimension(const struct pl_tex_params params)
{
    return params.d ? 3 : params.h ? 2 : 1;
}

typedef const struct pl_tex_t *pl_tex;
struct pl_tex_t {
    struct pl_tex_params params;
    enum pl_sampler_type sampler_type;
};

PL_API pl_tex pl_tex_create(pl_gpu gpu, const struct pl_tex_params *params);
PL_API void pl_tex_destroy(pl_gpu gpu, pl_tex *tex);
PL_API bool pl_tex_recreate(pl_gpu gpu, pl_tex *tex, const struct pl_tex_params *params);
PL_API void pl_tex_invalidate(pl_gpu gpu, pl_tex tex);

union pl_clear_color {
    float f[4];
    int32_t i[4];
    uint32_t u[4];
};

PL_API void pl_tex_clear_ex(pl_gpu gpu, pl_tex dst, const union pl_clear_color color);
PL_API void pl_tex_clear(pl_gpu gpu, pl_tex dst, const float color[4]);

typedef struct pl_rect3d {
    int x0, y0, z0;
    int x1, y1, z1;
} pl_rect3d;

struct pl_tex_transfer_params {
    pl_tex tex;
    pl_rect3d rc;           // region (0 = whole texture)
    size_t row_pitch;       // bytes between rows in host memory (0 = packed)
    size_t depth_pitch;
    pl_timer timer;
    void (*callback)(void *priv);
    void *priv;
    pl_buf buf;             // device-side transfer source/target (optional)
    size_t buf_offset;
    void *ptr;              // host pointer
};

#define pl_tex_transfer_params(...) (&(struct pl_tex_transfer_params) { __VA_ARGS__ })

PL_API bool pl_tex_upload(pl_gpu gpu, const struct pl_tex_transfer_params *params);
PL_API bool pl_tex_download(pl_gpu gpu, const struct pl_tex_transfer_params *params);
PL_API bool pl_tex_poll(pl_gpu gpu, pl_tex tex, uint64_t timeout);

enum pl_blend_mode {
    PL_BLEND_ZERO,
    PL_BLEND_ONE,
    PL_BLEND_SRC_ALPHA,
    PL_BLEND_ONE_MINUS_SRC_ALPHA,
    PL_BLEND_MODE_COUNT,
};

struct pl_blend_params {
    enum pl_blend_mode src_rgb;
    enum pl_blend_mode dst_rgb;
    enum pl_blend_mode src_alpha;
    enum pl_blend_mode dst_alpha;
};

#define pl_blend_params(...) (&(struct pl_blend_params) { __VA_ARGS__ })

// Flush queued work to the device / wait for all of it.
PL_API void pl_gpu_flush(pl_gpu gpu);
PL_API void pl_gpu_finish(pl_gpu gpu);
PL_API bool pl_gpu_is_failed(pl_gpu gpu);

PL_API_END

#endif // LIBPLACEBO_GPU_H_
