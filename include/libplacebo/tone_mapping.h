/*
 * libplacebo-hip: tone-mapping curves (Tier-0 host maths).
 * API-compatible with the reference's src/include/libplacebo/tone_mapping.h
 * (pl_tone_map_function :33-70, constants :73-150, params :152-190).
 */
#ifndef LIBPLACEBO_TONE_MAPPING_H_
#define LIBPLACEBO_TONE_MAPPING_H_

#include <stdbool.h>
#include <stddef.h>

#include <libplacebo/colorspace.h>
#include <libplacebo/common.h>

PL_API_BEGIN

struct pl_tone_map_params;

struct pl_tone_map_function {
    const char *name;
    const char *description;
    enum pl_hdr_scaling scaling;    // scale the curve is defined in
    // Map a LUT of input values (in `scaling`) in place
    void (*map)(float *lut, const struct pl_tone_map_params *params);
    void (*map_inverse)(float *lut, const struct pl_tone_map_params *params);
    void *priv;
    // deprecated since v6.311 (the curves read pl_tone_map_constants instead); kept filled in
    // with the reference's values for programs that still display them
    const char *param_desc;
    float param_min;
    float param_def;
    float param_max;
};

struct pl_tone_map_constants {
    float knee_adaptation;      // [0,1]
    float knee_minimum;         // (0, 0.5)
    float knee_maximum;         // (0.5, 1)
    float knee_default;
    float knee_offset;          // bt2390, [0.5, 2]
    float slope_tuning;         // spline, [0, 10]
    float slope_offset;         // spline, [0, 1]
    float spline_contrast;      // [0, 1.5]
    float reinhard_contrast;
    float linear_knee;          // mobius / gamma
    float exposure;             // linear
};

#define PL_TONE_MAP_CONSTANTS  \
    .knee_adaptation   = 0.4f, \
    .knee_minimum      = 0.1f, \
    .knee_maximum      = 0.8f, \
    .knee_default      = 0.4f, \
    .knee_offset       = 1.0f, \
    .slope_tuning      = 1.5f, \
    .slope_offset      = 0.2f, \
    .spline_contrast   = 0.5f, \
    .reinhard_contrast = 0.5f, \
    .linear_knee       = 0.3f, \
    .exposure          = 1.0f,

struct pl_tone_map_params {
    const struct pl_tone_map_function *function;
    struct pl_tone_map_constants constants;
    enum pl_hdr_scaling input_scaling;
    enum pl_hdr_scaling output_scaling;
    size_t lut_size;
    float input_min;
    float input_max;
    float input_avg;    // 0 if unknown
    float output_min;
    float output_max;
    struct pl_hdr_metadata hdr;
    float param;        // legacy single parameter (kept for layout compatibility)
};

#define pl_tone_map_params(...) (&(struct pl_tone_map_params) { __VA_ARGS__ });

PL_API bool pl_tone_map_params_equal(const struct pl_tone_map_params *a,
                                     const struct pl_tone_map_params *b);
PL_API void pl_tone_map_params_infer(struct pl_tone_map_params *params);
PL_API bool pl_tone_map_params_noop(const struct pl_tone_map_params *params);

// Fill out[lut_size] with the curve sampled evenly over [input_min, input_max]
PL_API void pl_tone_map_generate(float *out, const struct pl_tone_map_params *params);
PL_API float pl_tone_map_sample(float x, const struct pl_tone_map_params *params);

PL_API extern const struct pl_tone_map_function pl_tone_map_clip;
PL_API extern const struct pl_tone_map_function pl_tone_map_st2094_40;
PL_API extern const struct pl_tone_map_function pl_tone_map_st2094_10;
PL_API extern const struct pl_tone_map_function pl_tone_map_bt2390;
PL_API extern const struct pl_tone_map_function pl_tone_map_bt2446a;
PL_API extern const struct pl_tone_map_function pl_tone_map_spline;
PL_API extern const struct pl_tone_map_function pl_tone_map_reinhard;
PL_API extern const struct pl_tone_map_function pl_tone_map_mobius;
PL_API extern const struct pl_tone_map_function pl_tone_map_hable;
PL_API extern const struct pl_tone_map_function pl_tone_map_gamma;
PL_API extern const struct pl_tone_map_function pl_tone_map_linear;
PL_API extern const struct pl_tone_map_function pl_tone_map_linear_light;

PL_API extern const struct pl_tone_map_function * const pl_tone_map_functions[];
PL_API extern const int pl_num_tone_map_functions;
PL_API const struct pl_tone_map_function *pl_find_tone_map_function(const char *name);

#define pl_tone_map_auto pl_tone_map_spline

PL_API_END

#endif // LIBPLACEBO_TONE_MAPPING_H_
