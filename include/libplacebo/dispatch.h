/*
 * libplacebo-hip: pl_dispatch — turns finished shaders into kernel launches.
 * API-compatible with the reference's src/include/libplacebo/dispatch.h
 * (pl_dispatch_begin :44, pl_dispatch_info :50-80, pl_dispatch_finish
 * :100-135, pl_dispatch_compute :140-175, pl_dispatch_abort :245).
 */
#ifndef LIBPLACEBO_DISPATCH_H_
#define LIBPLACEBO_DISPATCH_H_

#include <libplacebo/gpu.h>
#include <libplacebo/shaders.h>

PL_API_BEGIN

typedef struct pl_dispatch_t *pl_dispatch;

PL_API pl_dispatch pl_dispatch_create(pl_log log, pl_gpu gpu);
PL_API void pl_dispatch_destroy(pl_dispatch *dp);

// Call once per frame: advances the frame index that seeds PRNGs / temporal
// dithering (an 8-bit counter, like the reference's dp->current_index).
PL_API void pl_dispatch_reset_frame(pl_dispatch dp);

// Returns a blank shader owned by the dispatch; hand it back with
// pl_dispatch_finish / pl_dispatch_compute / pl_dispatch_abort.
PL_API pl_shader pl_dispatch_begin(pl_dispatch dp);

struct pl_dispatch_info {
    pl_shader_info shader;      // of the shader that ran (->description names the pass)
    uint64_t signature;
    uint64_t samples[256];      // nanoseconds
    int num_samples;
    uint64_t last;
    uint64_t peak;
    uint64_t average;
};

// Take over `src` (keeps the shader description alive for as long as `dst` lives)
static inline void pl_dispatch_info_move(struct pl_dispatch_info *dst,
                                         const struct pl_dispatch_info *src)
{
    pl_shader_info_deref(&dst->shader);
    *dst = *src;
    dst->shader = pl_shader_info_ref(src->shader);
}

// Per-pass timing callback (requires a pl_timer per pass, created internally)
PL_API void pl_dispatch_callback(pl_dispatch dp, void *priv,
                                 void (*cb)(void *priv, const struct pl_dispatch_info *));

struct pl_dispatch_params {
    pl_shader *shader;      // consumed (set to NULL)
    pl_tex target;          // must be storable
    pl_rect2d rect;         // target region (0 = whole texture); may be flipped
    const struct pl_blend_params *blend_params;
    pl_timer timer;
};

#define pl_dispatch_params(...) (&(struct pl_dispatch_params) { __VA_ARGS__ })

PL_API bool pl_dispatch_finish(pl_dispatch dp, const struct pl_dispatch_params *params);

struct pl_dispatch_compute_params {
    pl_shader *shader;
    int dispatch_size[3];   // workgroups (0 = derive from width/height)
    int width, height;
    pl_timer timer;
};

#define pl_dispatch_compute_params(...) (&(struct pl_dispatch_compute_params) { __VA_ARGS__ })

PL_API bool pl_dispatch_compute(pl_dispatch dp, const struct pl_dispatch_compute_params *params);

PL_API void pl_dispatch_abort(pl_dispatch dp, pl_shader *sh);

// Deprecated: the contents of the gpu's pl_cache (src/include/libplacebo/dispatch.h:231-239)
PL_API size_t pl_dispatch_save(pl_dispatch dp, uint8_t *out_cache);
PL_API void pl_dispatch_load(pl_dispatch dp, const uint8_t *cache);

PL_API_END

#endif // LIBPLACEBO_DISPATCH_H_
