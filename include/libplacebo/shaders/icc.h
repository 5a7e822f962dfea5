/*
 * libplacebo-hip: ICC profile handles referred to by pl_frame / pl_render_params.
 * Types of the reference's src/include/libplacebo/shaders/icc.h (:29-95). ICC colour
 * management needs lcms2, which this build does not have (the reference compiles the same
 * way without PL_HAVE_LCMS: pl_icc_open fails); a frame carrying an `icc` object or a
 * `profile` is rendered from its pl_color_space alone, and a warning is logged once.
 */
#ifndef LIBPLACEBO_SHADERS_ICC_H_
#define LIBPLACEBO_SHADERS_ICC_H_

#include <libplacebo/cache.h>
#include <libplacebo/colorspace.h>
#include <libplacebo/shaders.h>

PL_API_BEGIN

struct pl_icc_params {
    enum pl_rendering_intent intent;
    int size_r, size_g, size_b;
    float max_luma;
    bool force_bpc;
    pl_cache cache;
    // deprecated since v6.321
    void *cache_priv;
    void (*cache_save)(void *priv, uint64_t sig, const uint8_t *cache, size_t size);
    bool (*cache_load)(void *priv, uint64_t sig, uint8_t *cache, size_t size);
};

typedef const struct pl_icc_object_t {
    struct pl_icc_params params;
    uint64_t signature;
    struct pl_color_space csp;
    float gamma;
    enum pl_color_primaries containing_primaries;
} *pl_icc_object;

#define PL_ICC_DEFAULTS                         \
    .intent = PL_INTENT_RELATIVE_COLORIMETRIC,  \
    .max_luma = PL_COLOR_SDR_WHITE,

#define pl_icc_params(...) (&(struct pl_icc_params) { PL_ICC_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_icc_params pl_icc_default_params;

// The entry points (icc.h:97-131), with the behaviour of a libplacebo built without lcms2
// (src/shaders/icc.c:802-836): open and update fail and log why, nothing else is reachable.
PL_API pl_icc_object pl_icc_open(pl_log log, const struct pl_icc_profile *profile,
                                 const struct pl_icc_params *params);
PL_API void pl_icc_close(pl_icc_object *icc);
PL_API bool pl_icc_update(pl_log log, pl_icc_object *obj,
                          const struct pl_icc_profile *profile,
                          const struct pl_icc_params *params);
PL_API void pl_icc_decode(pl_shader sh, pl_icc_object profile, pl_shader_obj *lut,
                          struct pl_color_space *out_csp);
PL_API void pl_icc_encode(pl_shader sh, pl_icc_object profile, pl_shader_obj *lut);

PL_API_END

#endif // LIBPLACEBO_SHADERS_ICC_H_
