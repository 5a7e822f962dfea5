/*
 * libplacebo-hip: the pl_shader builder.
 *
 * Same role and entry points as the reference's
 * src/include/libplacebo/shaders.h: a pl_shader accumulates a sampling stage
 * plus a chain of colour stages through pl_shader_* calls and is then run by
 * pl_dispatch_finish. The difference is what gets accumulated: typed ops that
 * select and parameterise precompiled HIP kernels, not GLSL text (SURVEY.md
 * §8b "The GLSL problem"). pl_shader_finalize().glsl is a human-readable
 * listing of the recorded ops.
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

struct pl_shader_res {
    const char *glsl;           // textual listing of the recorded ops
    const char *name;
    const char *description;
    enum pl_shader_sig input;
    enum pl_shader_sig output;
    int compute_group_size[2];
    size_t compute_shmem;
    int num_ops;
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
