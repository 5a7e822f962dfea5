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

PL_API_END

#endif // LIBPLACEBO_SHADERS_ICC_H_
