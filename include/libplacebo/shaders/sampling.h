/*
 * libplacebo-hip: sampling stages (K1-K6).
 * API-compatible with the reference's
 * src/include/libplacebo/shaders/sampling.h:34-61 (pl_sample_src), :64-103
 * (deband), :110-175 (samplers), :180-230 (filter params).
 */
#ifndef LIBPLACEBO_SHADERS_SAMPLING_H_
#define LIBPLACEBO_SHADERS_SAMPLING_H_

#include <libplacebo/colorspace.h>
#include <libplacebo/filters.h>
#include <libplacebo/shaders.h>

PL_API_BEGIN

struct pl_sample_src {
    pl_tex tex;             // texture to sample (required on this backend)
    pl_rect2df rect;        // sub-rect to sample from (0 = whole texture)
    enum pl_tex_address_mode address_mode;

    // Accepted for source compatibility with the reference's "external
    // sampler" mode; not supported here (tex must be set).
    int tex_w, tex_h;
    enum pl_fmt_type format;
    enum pl_sampler_type sampler;
    enum pl_tex_sample_mode mode;
    float sampled_w, sampled_h;

    int components;         // number of components to sample (0 = 4)
    uint8_t component_mask; // overrides `components` if set
    int new_w, new_h;       // output size (0 = round(|rect|))
    float scale;            // multiplied into the result (0 = 1)
};

#define pl_sample_src(...) (&(struct pl_sample_src) { __VA_ARGS__ })

struct pl_deband_params {
    int iterations;
    float threshold;
    float radius;
    float grain;
    float grain_neutral[3];
};

#define PL_DEBAND_DEFAULTS  \
    .iterations = 1,        \
    .threshold  = 3.0,      \
    .radius     = 16.0,     \
    .grain      = 4.0,

#define pl_deband_params(...) (&(struct pl_deband_params) {PL_DEBAND_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_deband_params pl_deband_default_params;

PL_API void pl_shader_deband(pl_shader sh, const struct pl_sample_src *src,
                             const struct pl_deband_params *params);

PL_API bool pl_shader_sample_direct(pl_shader sh, const struct pl_sample_src *src);
PL_API bool pl_shader_sample_nearest(pl_shader sh, const struct pl_sample_src *src);
PL_API bool pl_shader_sample_bilinear(pl_shader sh, const struct pl_sample_src *src);
PL_API bool pl_shader_sample_bicubic(pl_shader sh, const struct pl_sample_src *src);
PL_API bool pl_shader_sample_hermite(pl_shader sh, const struct pl_sample_src *src);
PL_API bool pl_shader_sample_gaussian(pl_shader sh, const struct pl_sample_src *src);
PL_API bool pl_shader_sample_oversample(pl_shader sh, const struct pl_sample_src *src,
                                        float threshold);

struct pl_sample_filter_params {
    struct pl_filter_config filter;
    float antiring;
    bool no_compute;    // evaluate taps in the reference's gather/fragment order
    bool no_widening;
    pl_shader_obj *lut; // required: persistent LUT / filter state
    int lut_entries;    // deprecated since v6.335, ignored: LUTs always have 256 entries
    float cutoff;       // deprecated since v6.335, ignored: 1e-3 for polar kernels
};

#define pl_sample_filter_params(...) (&(struct pl_sample_filter_params) { __VA_ARGS__ })

PL_API bool pl_shader_sample_polar(pl_shader sh, const struct pl_sample_src *src,
                                   const struct pl_sample_filter_params *params);
PL_API bool pl_shader_sample_ortho2(pl_shader sh, const struct pl_sample_src *src,
                                    const struct pl_sample_filter_params *params);

// An affine transformation of the image inside a canvas of `out_w` x `out_h` (reference
// shaders/sampling.h:204-244, src/shaders/sampling.c:1106-1217). The image is centred and scaled so
// that its longer side spans [-1, 1] (y up) before `transform` applies.
struct pl_distort_params {
    pl_transform2x2 transform;
    bool unscaled;      // place the image at its own size instead of stretching it over the canvas
    bool constrain;     // scale the result down so that it fits the canvas
    bool bicubic;       // bicubic instead of bilinear interpolation
    enum pl_tex_address_mode address_mode;  // what lies outside the image ...
    enum pl_alpha_mode alpha_mode;          // ... or, if set: transparent, in this alpha mode
};

#define PL_DISTORT_DEFAULTS \
    .transform.mat.m = {{ 1, 0 }, {0, 1}},

#define pl_distort_params(...) (&(struct pl_distort_params) {PL_DISTORT_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_distort_params pl_distort_default_params;

// A sampling stage (the first of a shader); what leaves the canvas is cut off.
PL_API void pl_shader_distort(pl_shader sh, pl_tex tex, int out_w, int out_h,
                              const struct pl_distort_params *params);

PL_API_END

#endif // LIBPLACEBO_SHADERS_SAMPLING_H_
