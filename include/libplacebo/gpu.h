/*
 * libplacebo-hip: the pl_gpu abstraction (textures, buffers, formats, timers).
 *
 * API-compatible subset of the reference's src/include/libplacebo/gpu.h for
 * everything the render hot path touches: pl_gpu_t (:205-226), pl_fmt_t
 * (:306-373), pl_tex_params / pl_tex_t (:672-754), pl_tex_transfer_params
 * (:845-905), pl_buf (:449-560), pl_timer (:400-420), pl_pass (:1187-1356).
 *
 * Every aggregate declared here has the reference's member order and layout
 * (tests/test_abi_layout.py compiles a probe against both header sets), so an
 * application built against libplacebo's headers can hand its structs across
 * this library's C ABI unchanged. Members that describe features this backend
 * does not have (external memory handles, DRM modifiers, planar formats,
 * vertex input) are declared, must be left zero, and are refused at run time
 * otherwise. A pl_pass on this backend is a precompiled HIP kernel selection:
 * `pl_pass_params.glsl_shader` carries the serialised op list a pl_shader
 * recorded, never GLSL (SURVEY.md 8b "The GLSL problem").
 */
#ifndef LIBPLACEBO_GPU_H_
#define LIBPLACEBO_GPU_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#include <libplacebo/common.h>
#include <libplacebo/cache.h>
#include <libplacebo/log.h>

PL_API_BEGIN

typedef const char *pl_debug_tag;
#define PL_DEBUG_TAG (__FILE__ ":" PL_TOSTRING(__LINE__))

// What a pass binds (reference gpu.h:34-60). This backend knows sampled textures, storage
// images and storage buffers; the other kinds exist so that the numbering matches.
enum pl_desc_type {
    PL_DESC_INVALID = 0,
    PL_DESC_SAMPLED_TEX,
    PL_DESC_STORAGE_IMG,
    PL_DESC_BUF_UNIFORM,
    PL_DESC_BUF_STORAGE,
    PL_DESC_BUF_TEXEL_UNIFORM,
    PL_DESC_BUF_TEXEL_STORAGE,
    PL_DESC_TYPE_COUNT
};

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

#define pl_glsl_desc pl_glsl_version

struct pl_gpu_limits {
    bool thread_safe;
    bool callbacks;
    size_t max_buf_size;
    size_t max_ubo_size;
    size_t max_ssbo_size;
    size_t max_vbo_size;
    size_t max_mapped_size;
    uint64_t max_buffer_texels;
    bool host_cached;           // host-mapped buffers are cached (pinned memory here: no)
    bool host_ptr_slow;
    size_t max_mapped_vram;
    size_t align_host_ptr;      // 0: host pointers cannot be imported
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
    size_t align_vertex_stride;
    uint32_t max_dispatch[3];
    uint32_t fragment_queues;   // 0: every pass is a compute pass
    uint32_t compute_queues;
};

#define max_xfer_size max_buf_size
#define align_tex_xfer_stride align_tex_xfer_pitch

// External memory handles (reference gpu.h:150-203). Nothing can be imported or exported on
// this backend (zero-copy ingest of device memory goes through pl_hip_wrap, hip.h): all caps
// are 0 and a non-zero handle type in a params struct is refused.
typedef uint64_t pl_handle_caps;
enum pl_handle_type {
    PL_HANDLE_FD        = (1 << 0),
    PL_HANDLE_WIN32     = (1 << 1),
    PL_HANDLE_WIN32_KMT = (1 << 2),
    PL_HANDLE_DMA_BUF   = (1 << 3),
    PL_HANDLE_HOST_PTR  = (1 << 4),
    PL_HANDLE_MTL_TEX   = (1 << 5),
    PL_HANDLE_IOSURFACE = (1 << 6),
};

struct pl_gpu_handle_caps {
    pl_handle_caps tex;
    pl_handle_caps buf;
    pl_handle_caps sync;
};

union pl_handle {
    int fd;
    void *handle;
    void *ptr;
};

struct pl_shared_mem {
    union pl_handle handle;
    size_t size;
    size_t offset;
    uint64_t drm_format_mod;
    size_t stride_w;
    size_t stride_h;
    unsigned plane;
};

struct pl_gpu_pci_address {
    uint32_t domain;
    uint32_t bus;
    uint32_t device;
    uint32_t function;
};

typedef const struct pl_fmt_t *pl_fmt;

typedef const struct pl_gpu_t {
    pl_log log;
    struct pl_glsl_version glsl;
    struct pl_gpu_limits limits;
    struct pl_gpu_handle_caps export_caps;  // all 0
    struct pl_gpu_handle_caps import_caps;  // all 0
    uint8_t uuid[16];
    pl_fmt *formats;            // sorted best-first, like pl_gpu_finalize does
    int num_formats;
    struct pl_gpu_pci_address pci;
} *pl_gpu;

// Attach a cache (cache.h) to the GPU: generated LUTs (blue noise, gamut mapping 3D-LUTs) are
// looked up in / added to it by every renderer and shader object created on this GPU.
// There are no compiled programs to cache. NULL detaches.
PL_API void pl_gpu_set_cache(pl_gpu gpu, pl_cache cache);

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

struct pl_fmt_plane {
    pl_fmt format;
    uint8_t shift_x, shift_y;
};

struct pl_fmt_t {
    const char *name;           // e.g. "rgba16hf"
    uint64_t signature;
    enum pl_fmt_type type;
    enum pl_fmt_caps caps;
    int num_components;
    int component_depth[4];
    size_t internal_size;
    struct pl_fmt_plane planes[4]; // no planar formats here: planes[0] = the format itself
    int num_planes;                // always 1
    bool opaque;
    bool emulated;
    size_t texel_size;
    size_t texel_align;
    int host_bits[4];
    int sample_order[4];
    bool gatherable;
    const char *glsl_type;
    const char *glsl_format;
    uint32_t fourcc;            // 0 (no DRM interop)
    const uint64_t *modifiers;  // NULL
    int num_modifiers;
};

PL_API bool pl_fmt_is_ordered(pl_fmt fmt);
PL_API bool pl_fmt_is_float(pl_fmt fmt);
PL_API bool pl_fmt_has_modifier(pl_fmt fmt, uint64_t modifier);
PL_API pl_fmt pl_find_fmt(pl_gpu gpu, enum pl_fmt_type type, int num_components,
                          int min_depth, int host_bits, enum pl_fmt_caps caps);
PL_API pl_fmt pl_find_vertex_fmt(pl_gpu gpu, enum pl_fmt_type type, int num_components);
PL_API pl_fmt pl_find_named_fmt(pl_gpu gpu, const char *name);
PL_API pl_fmt pl_find_fourcc(pl_gpu gpu, uint32_t fourcc);

// GPU timers (hipEvent pairs). pl_timer_query returns elapsed nanoseconds of
// the oldest finished measurement, or 0 if none is available yet.
typedef struct pl_timer_t *pl_timer;
PL_API pl_timer pl_timer_create(pl_gpu gpu);
PL_API void pl_timer_destroy(pl_gpu gpu, pl_timer *);
PL_API uint64_t pl_timer_query(pl_gpu gpu, pl_timer);

enum pl_buf_mem_type {
    PL_BUF_MEM_AUTO = 0,
    PL_BUF_MEM_HOST,    // pinned host memory
    PL_BUF_MEM_DEVICE,  // HBM
    PL_BUF_MEM_TYPE_COUNT,
};

// Buffers: plain device allocations with host read/write
struct pl_buf_params {
    size_t size;
    bool host_writable;
    bool host_readable;
    bool host_mapped;
    bool uniform;
    bool storable;
    bool drawable;                      // refused (no vertex input)
    enum pl_buf_mem_type memory_type;
    pl_fmt format;                      // texel buffers: refused
    enum pl_handle_type export_handle;  // refused
    enum pl_handle_type import_handle;  // refused
    struct pl_shared_mem shared_mem;
    const void *initial_data;
    void *user_data;
    pl_debug_tag debug_tag;
};

#define pl_buf_params(...) (&(struct pl_buf_params) { .debug_tag = PL_DEBUG_TAG, __VA_ARGS__ })

typedef const struct pl_buf_t {
    struct pl_buf_params params;
    uint8_t *data; // host_mapped only
    struct pl_shared_mem shared_mem;    // unused
} *pl_buf;

PL_API pl_buf pl_buf_create(pl_gpu gpu, const struct pl_buf_params *params);
PL_API void pl_buf_destroy(pl_gpu gpu, pl_buf *buf);
PL_API bool pl_buf_recreate(pl_gpu gpu, pl_buf *buf, const struct pl_buf_params *params);
PL_API void pl_buf_write(pl_gpu gpu, pl_buf buf, size_t buf_offset, const void *data, size_t size);
PL_API bool pl_buf_read(pl_gpu gpu, pl_buf buf, size_t buf_offset, void *dest, size_t size);
PL_API void pl_buf_copy(pl_gpu gpu, pl_buf dst, size_t dst_offset,
                        pl_buf src, size_t src_offset, size_t size);
PL_API bool pl_buf_export(pl_gpu gpu, pl_buf buf);   // always false (no exportable handles)
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
    enum pl_handle_type export_handle;  // refused
    enum pl_handle_type import_handle;  // refused
    struct pl_shared_mem shared_mem;
    const void *initial_data; // tightly packed
    void *user_data;
    pl_debug_tag debug_tag;
};

#define pl_tex_params(...) (&(struct pl_tex_params) { .debug_tag = PL_DEBUG_TAG, __VA_ARGS__ })

static inline int pl_tex_params_dimension(const struct pl_tex_params params)
{
    return params.d ? 3 : params.h ? 2 : 1;
}

enum pl_sampler_type {
    PL_SAMPLER_NORMAL,
    PL_SAMPLER_RECT,
    PL_SAMPLER_EXTERNAL,
    PL_SAMPLER_TYPE_COUNT,
};

typedef const struct pl_tex_t *pl_tex;
struct pl_tex_t {
    struct pl_tex_params params;
    pl_tex planes[4];           // planar formats only: always NULL here
    pl_tex parent;
    struct pl_shared_mem shared_mem;
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

struct pl_tex_blit_params {
    pl_tex src;
    pl_tex dst;
    pl_rect3d src_rc;   // 0 = whole texture; flipped rects flip
    pl_rect3d dst_rc;
    enum pl_tex_sample_mode sample_mode;
};

#define pl_tex_blit_params(...) (&(struct pl_tex_blit_params) { __VA_ARGS__ })

// Copy (and scale) a region of one texture into another; formats must have the same type.
PL_API void pl_tex_blit(pl_gpu gpu, const struct pl_tex_blit_params *params);

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
    bool no_import;         // (host pointers are never imported here)
};

#define pl_tex_transfer_params(...) (&(struct pl_tex_transfer_params) { __VA_ARGS__ })

PL_API bool pl_tex_upload(pl_gpu gpu, const struct pl_tex_transfer_params *params);
PL_API bool pl_tex_download(pl_gpu gpu, const struct pl_tex_transfer_params *params);
PL_API bool pl_tex_poll(pl_gpu gpu, pl_tex tex, uint64_t timeout);

/* ---- passes (reference gpu.h:907-1356) ----------------------------------------------------
 * A pl_pass is one precompiled-kernel selection plus its argument block. It is created from
 * the serialised op list a pl_shader recorded (pl_shader_finalize().glsl), bound to textures /
 * buffers at run time through descriptor bindings, and launched over a target rect. There is
 * no rasteriser: PL_PASS_RASTER passes are executed as full-rect compute launches over
 * `viewport` (the reference's own dispatch does the same upgrade when a GPU reports
 * compute_queues > fragment_queues), vertex inputs are refused.
 */

enum pl_var_type {
    PL_VAR_INVALID = 0,
    PL_VAR_SINT,
    PL_VAR_UINT,
    PL_VAR_FLOAT,
    PL_VAR_TYPE_COUNT
};

PL_API size_t pl_var_type_size(enum pl_var_type type);

// A (possibly vector / matrix / array) input variable. The op lists of this backend carry
// their constants inline, so passes normally have none.
struct pl_var {
    const char *name;
    enum pl_var_type type;
    int dim_v;      // vector dimension
    int dim_m;      // matrix dimension (columns)
    int dim_a;      // array dimension
};

// Constructors for the GLSL types a variable can have (gpu.h:975-997), the same list tagged by
// type name and {0}-terminated, and the lookup the other way round (NULL: no such GLSL type; the
// array dimension is not part of a type name).
PL_API struct pl_var pl_var_float(const char *name);
PL_API struct pl_var pl_var_vec2(const char *name);
PL_API struct pl_var pl_var_vec3(const char *name);
PL_API struct pl_var pl_var_vec4(const char *name);
PL_API struct pl_var pl_var_mat2(const char *name);
PL_API struct pl_var pl_var_mat2x3(const char *name);
PL_API struct pl_var pl_var_mat2x4(const char *name);
PL_API struct pl_var pl_var_mat3(const char *name);
PL_API struct pl_var pl_var_mat3x4(const char *name);
PL_API struct pl_var pl_var_mat4x2(const char *name);
PL_API struct pl_var pl_var_mat4x3(const char *name);
PL_API struct pl_var pl_var_mat4(const char *name);
PL_API struct pl_var pl_var_int(const char *name);
PL_API struct pl_var pl_var_ivec2(const char *name);
PL_API struct pl_var pl_var_ivec3(const char *name);
PL_API struct pl_var pl_var_ivec4(const char *name);
PL_API struct pl_var pl_var_uint(const char *name);
PL_API struct pl_var pl_var_uvec2(const char *name);
PL_API struct pl_var pl_var_uvec3(const char *name);
PL_API struct pl_var pl_var_uvec4(const char *name);

struct pl_named_var {
    const char *glsl_name;
    struct pl_var var;
};
PL_API extern const struct pl_named_var pl_var_glsl_types[];
PL_API const char *pl_var_glsl_type_name(struct pl_var var);
// the variable a texel of `fmt` reads as in a shader (normalised formats read as floats)
PL_API struct pl_var pl_var_from_fmt(pl_fmt fmt, const char *name);

struct pl_var_layout {
    size_t offset;
    size_t stride;
    size_t size;
};

// Where a variable goes in a buffer (gpu.h:1019-1050): tightly packed on the host, or by the
// std140 / std430 rules of the GLSL specification (matrices are arrays of columns; vec3 aligns
// like vec4; std140 rounds the stride of arrays and matrices up to a vec4). `offset` is the first
// free byte; the layout's own offset is that rounded up to the variable's alignment.
// memcpy_layout copies a variable column by column between two layouts of it.
PL_API struct pl_var_layout pl_var_host_layout(size_t offset, const struct pl_var *var);
PL_API struct pl_var_layout pl_std140_layout(size_t offset, const struct pl_var *var);
PL_API struct pl_var_layout pl_std430_layout(size_t offset, const struct pl_var *var);
PL_API void memcpy_layout(void *dst, struct pl_var_layout dst_layout,
                          const void *src, struct pl_var_layout src_layout);

struct pl_constant {
    enum pl_var_type type;
    uint32_t id;
    size_t offset;
};

struct pl_vertex_attrib {
    const char *name;
    pl_fmt fmt;
    size_t offset;
    int location;
};

// Binding numbers are per descriptor type on this backend: namespace = the type itself
PL_API int pl_desc_namespace(pl_gpu gpu, enum pl_desc_type type);

enum pl_desc_access {
    PL_DESC_ACCESS_READWRITE,
    PL_DESC_ACCESS_READONLY,
    PL_DESC_ACCESS_WRITEONLY,
    PL_DESC_ACCESS_COUNT,
};

// "", "readonly", "writeonly": the GLSL qualifier of an access mode (gpu.h:1102)
PL_API const char *pl_desc_access_glsl_name(enum pl_desc_access mode);

struct pl_desc {
    const char *name;
    enum pl_desc_type type;
    int binding;
    enum pl_desc_access access;
};

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

PL_API extern const struct pl_blend_params pl_alpha_overlay;

enum pl_prim_type {
    PL_PRIM_TRIANGLE_LIST,
    PL_PRIM_TRIANGLE_STRIP,
    PL_PRIM_TYPE_COUNT,
};

enum pl_index_format {
    PL_INDEX_UINT16 = 0,
    PL_INDEX_UINT32,
    PL_INDEX_FORMAT_COUNT,
};

enum pl_pass_type {
    PL_PASS_INVALID = 0,
    PL_PASS_RASTER,     // output goes to `target` (executed as a compute launch here)
    PL_PASS_COMPUTE,    // output goes to a PL_DESC_STORAGE_IMG descriptor
    PL_PASS_TYPE_COUNT,
};

struct pl_pass_params {
    enum pl_pass_type type;
    struct pl_var *variables;
    int num_variables;
    struct pl_desc *descriptors;
    int num_descriptors;
    struct pl_constant *constants;
    int num_constants;
    void *constant_data;
    size_t push_constants_size;
    const char *glsl_shader;
    enum pl_prim_type vertex_type;
    struct pl_vertex_attrib *vertex_attribs;
    int num_vertex_attribs;
    size_t vertex_stride;
    const char *vertex_shader;
    pl_fmt target_format;               // raster: format of the targets this pass will write
    const struct pl_blend_params *blend_params;
    bool load_target;
    const uint8_t *cached_program;      // deprecated since v6.322, ignored
    size_t cached_program_len;
};

#define pl_pass_params(...) (&(struct pl_pass_params) { __VA_ARGS__ })

// Thread-safety: Unsafe
typedef const struct pl_pass_t {
    struct pl_pass_params params;       // deep copy
} *pl_pass;

// A pl_pass of the reference is a compiled GLSL program (src/gpu.c:1025-1290). This backend has
// no GLSL front-end: what it turns into a pass is a recorded pl_shader. pl_shader_finalize()
// ends pl_shader_res.glsl with a line "#pl_hip_pass <ticket>" naming the recorded sampler +
// colour ops; pl_pass_create() with that text as `glsl_shader` (while the shader is alive)
// copies them into a pass that can be run any number of times, on any target of
// `target_format` (PL_PASS_RASTER: `target` + `viewport` / `scissors`) or into the storage image
// bound as its only descriptor (PL_PASS_COMPUTE). Variables, constants, push constants, vertex
// data and blending do not exist here (the ops carry their values) and must be unset. Any
// other text -- GLSL -- is refused with a message. DEVIATION: see INTEGRATION.md section 2.
PL_API pl_pass pl_pass_create(pl_gpu gpu, const struct pl_pass_params *params);
PL_API void pl_pass_destroy(pl_gpu gpu, pl_pass *pass);

struct pl_desc_binding {
    const void *object;                 // pl_tex or pl_buf, by descriptor type
    enum pl_tex_address_mode address_mode;
    enum pl_tex_sample_mode sample_mode;
};

struct pl_var_update {
    int index;
    const void *data;
};

struct pl_pass_run_params {
    pl_pass pass;
    void *constant_data;
    struct pl_var_update *var_updates;
    int num_var_updates;
    struct pl_desc_binding *desc_bindings; // one per pass descriptor
    void *push_constants;
    pl_timer timer;
    // raster passes: target + viewport (may be flipped); scissors must equal the viewport or
    // be empty
    pl_tex target;
    pl_rect2d viewport;
    pl_rect2d scissors;
    int vertex_count;                   // vertex / index inputs: must be unset
    const void *vertex_data;
    pl_buf vertex_buf;
    size_t buf_offset;
    const void *index_data;
    enum pl_index_format index_fmt;
    pl_buf index_buf;
    size_t index_offset;
    // compute passes: workgroup counts (the launch geometry is derived from the storage
    // image and the kernel's own tiling; these are validated against the device limits)
    int compute_groups[3];
};

#define pl_pass_run_params(...) (&(struct pl_pass_run_params) { __VA_ARGS__ })

PL_API void pl_pass_run(pl_gpu gpu, const struct pl_pass_run_params *params);

// Flush queued work to the device / wait for all of it.
PL_API void pl_gpu_flush(pl_gpu gpu);
PL_API void pl_gpu_finish(pl_gpu gpu);
PL_API bool pl_gpu_is_failed(pl_gpu gpu);

PL_API_END

#endif // LIBPLACEBO_GPU_H_
