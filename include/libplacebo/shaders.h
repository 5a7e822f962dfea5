/*
 * libplacebo-hip: the pl_shader builder.
 *
 * Same role and entry points as the reference's
 * src/include/libplacebo/shaders.h: a pl_shader accumulates a sampling stage
 * plus a chain of colour stages through pl_shader_* calls and is then run by
 * pl_dispatch_finish. The difference is what gets accumulated: typed ops that
 * select and parameterise precompiled HIP kernels, not GLSL text (SURVEY.md
 * §8b "The GLSL problem"). pl_shader_finalize().glsl is the serialised op
 * list; the human-readable stage list is in pl_shader_res.info.
 */
#ifndef LIBPLACEBO_SHADERS_H_
#define LIBPLACEBO_SHADERS_H_

#include <libplacebo/gpu.h>

PL_API_BEGIN

typedef struct pl_shader_t *pl_shader;

struct pl_shader_params {
    uint8_t id;         // unused on this backend (no identifier namespaces)
    pl_gpu gpu;         // required for anything that samples or uses LUTs
    uint8_t index;      // frame index: PRNG seed / temporal dither phase
    struct pl_glsl_version glsl; // ignored if `gpu` is set
    bool dynamic_constants;
};

#define pl_shader_params(...) (&(struct pl_shader_params) { __VA_ARGS__ })

PL_API pl_shader pl_shader_alloc(pl_log log, const struct pl_shader_params *params);
PL_API void pl_shader_free(pl_shader *sh);
PL_API void pl_shader_reset(pl_shader sh, const struct pl_shader_params *params);
PL_API bool pl_shader_is_failed(const pl_shader sh);
PL_API bool pl_shader_is_compute(const pl_shader sh);
PL_API bool pl_shader_output_size(const pl_shader sh, int *w, int *h);

enum pl_shader_sig {
    PL_SHADER_SIG_NONE = 0, // no input / void output
    PL_SHADER_SIG_COLOR,    // vec4 color
    PL_SHADER_SIG_SAMPLER,  // unsupported on this backend
};

// Ref-counted description of a finished shader, shared with pl_dispatch_info so that it can
// outlive the shader (reference shaders.h:111-135)
typedef const struct pl_shader_info_t {
    struct pl_shader_params params;
    const char **steps;         // one entry per recorded stage, in order
    int num_steps;
    const char *description;    // the steps joined, e.g. "polar scaling + dithering"
} *pl_shader_info;

PL_API pl_shader_info pl_shader_info_ref(pl_shader_info info);
PL_API void pl_shader_info_deref(pl_shader_info *info);

// What pl_shader_finalize returns. Layout of the reference's struct (shaders.h:143-200).
// `glsl` holds the *serialised op list* the kernels are driven by (a "#pl_hip" text block,
// see INTEGRATION.md) -- the same string pl_pass_create accepts as `glsl_shader`. The
// resource arrays describe what that program binds: `descriptors` = the textures / LUT
// buffers it reads in binding order; this backend bakes constants into the op list, so
// `variables`, `constants` and `vertex_attribs` are empty.
struct pl_shader_res {
    pl_shader_info info;
    const char *glsl;
    const char *name;
    enum pl_shader_sig input;
    enum pl_shader_sig output;
    int compute_group_size[2];
    size_t compute_shmem;
    const struct pl_shader_va *vertex_attribs;
    int num_vertex_attribs;
    const struct pl_shader_var *variables;
    int num_variables;
    const struct pl_shader_desc *descriptors;
    int num_descriptors;
    const struct pl_shader_const *constants;
    int num_constants;
    // deprecated since v6.266: duplicates of `info`
    struct pl_shader_params params;
    const char **steps;
    int num_steps;
    const char *description;
};

struct pl_shader_va {
    struct pl_vertex_attrib attr;
    const void *data[4];
};

struct pl_shader_var {
    struct pl_var var;
    const void *data;
    bool dynamic;
};

struct pl_buffer_var {
    struct pl_var var;
    struct pl_var_layout layout;
};

typedef uint16_t pl_memory_qualifiers;
enum {
    PL_MEMORY_COHERENT = 1 << 0,
    PL_MEMORY_VOLATILE = 1 << 1,
};

struct pl_shader_desc {
    struct pl_desc desc;
    struct pl_desc_binding binding;
    struct pl_buffer_var *buffer_vars;
    int num_buffer_vars;
    pl_memory_qualifiers memory;
};

struct pl_shader_const {
    enum pl_var_type type;
    const char *name;
    const void *data;
    bool compile_time;
};

// The returned struct stays valid until the shader is reset / freed.
PL_API const struct pl_shader_res *pl_shader_finalize(pl_shader sh);

// Persistent state that outlives individual shaders (LUT device buffers,
// generated filters, peak-detection buffers ...). Ref-counted like the
// reference's (src/shaders.c:909-963).
typedef struct pl_shader_obj_t *pl_shader_obj;
PL_API void pl_shader_obj_destroy(pl_shader_obj *obj);

PL_API_END

#endif // LIBPLACEBO_SHADERS_H_
