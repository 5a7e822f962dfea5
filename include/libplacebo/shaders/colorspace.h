/*
 * libplacebo-hip: colour stages (K7-K11).
 * API-compatible with the reference's
 * src/include/libplacebo/shaders/colorspace.h: set_alpha :30, decode/encode
 * :52-60, linearize/delinearize :66-72, sigmoid :74-96, peak detection
 * :98-190, colour mapping :200-390.
 * Not provided (out of scope): LUT / clipping visualisation.
 */
#ifndef LIBPLACEBO_SHADERS_COLORSPACE_H_
#define LIBPLACEBO_SHADERS_COLORSPACE_H_

#include <libplacebo/colorspace.h>
#include <libplacebo/gamut_mapping.h>
#include <libplacebo/tone_mapping.h>
#include <libplacebo/shaders.h>
#include <libplacebo/shaders/dithering.h>

PL_API_BEGIN

// Convert between alpha modes; updates repr->alpha
PL_API void pl_shader_set_alpha(pl_shader sh, struct pl_color_repr *repr,
                                enum pl_alpha_mode mode);

// Decode `repr` to normalised RGB (updates `repr`) / encode RGB to `repr`
// Dolby Vision reshaping on its own (pl_shader_decode_color does it for PL_COLOR_SYSTEM_DOLBYVISION)
PL_API void pl_shader_dovi_reshape(pl_shader sh, const struct pl_dovi_metadata *data);

PL_API void pl_shader_decode_color(pl_shader sh, struct pl_color_repr *repr,
                                   const struct pl_color_adjustment *params);
PL_API void pl_shader_encode_color(pl_shader sh, const struct pl_color_repr *repr);

// Transfer function <-> linear light (1.0 = PL_COLOR_SDR_WHITE)
PL_API void pl_shader_linearize(pl_shader sh, const struct pl_color_space *csp);
PL_API void pl_shader_delinearize(pl_shader sh, const struct pl_color_space *csp);

struct pl_sigmoid_params {
    float center;
    float slope;
};

#define PL_SIGMOID_DEFAULTS \
    .center = 0.75,         \
    .slope  = 6.50,

#define pl_sigmoid_params(...) (&(struct pl_sigmoid_params) { PL_SIGMOID_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_sigmoid_params pl_sigmoid_default_params;

PL_API void pl_shader_sigmoidize(pl_shader sh, const struct pl_sigmoid_params *params);
PL_API void pl_shader_unsigmoidize(pl_shader sh, const struct pl_sigmoid_params *params);

struct pl_peak_detect_params {
    float smoothing_period;     // frames (IIR), 0 = none
    float scene_threshold_low;  // scene change hysteresis
    float scene_threshold_high;
    float percentile;           // 100 = true maximum; < 100 uses the histogram
    float black_cutoff;         // % PQ below which pixels are ignored
    bool allow_delayed;         // may use the previous frame's result
    float minimum_peak;         // legacy, unused (layout compatibility)
};

#define PL_PEAK_DETECT_DEFAULTS         \
    .smoothing_period       = 20.0f,    \
    .scene_threshold_low    = 1.0f,     \
    .scene_threshold_high   = 3.0f,     \
    .percentile             = 100.0f,   \
    .black_cutoff           = 1.0f,

#define PL_PEAK_DETECT_HQ_DEFAULTS      \
    PL_PEAK_DETECT_DEFAULTS             \
    .percentile             = 99.995f,

#define pl_peak_detect_params(...) \
    (&(struct pl_peak_detect_params) { PL_PEAK_DETECT_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_peak_detect_params pl_peak_detect_default_params;
PL_API extern const struct pl_peak_detect_params pl_peak_detect_high_quality_params;

// Measure the frame's luminance statistics into `state` (16x16 workgroups,
// 12-slice atomics buffer). The colour itself is unchanged.
PL_API bool pl_shader_detect_peak(pl_shader sh, struct pl_color_space csp,
                                  pl_shader_obj *state,
                                  const struct pl_peak_detect_params *params);

// Read back (blocking unless allow_delayed) and smooth the measurement
PL_API bool pl_get_detected_hdr_metadata(const pl_shader_obj state,
                                         struct pl_hdr_metadata *metadata);
PL_API void pl_reset_detected_peak(pl_shader_obj state);

// HIP extension (multi-GPU): raw access to the pending 816 x u32
// measurement buffer so that ranks rendering tiles / frames of one scene can
// all-reduce it (SUM, and MAX for frame_max_pq) before it is consumed.
// Returns the device pointer, or NULL if no measurement is pending.
PL_API void *pl_hip_peak_buffer(const pl_shader_obj state, size_t *out_size);

// Deprecated since v6.269; only here because pl_color_map_params still carries the members
enum pl_tone_map_mode {
    PL_TONE_MAP_AUTO,
    PL_TONE_MAP_RGB,
    PL_TONE_MAP_MAX,
    PL_TONE_MAP_HYBRID,
    PL_TONE_MAP_LUMA,
    PL_TONE_MAP_MODE_COUNT,
};

enum pl_gamut_mode {
    PL_GAMUT_CLIP,
    PL_GAMUT_WARN,
    PL_GAMUT_DARKEN,
    PL_GAMUT_DESATURATE,
    PL_GAMUT_MODE_COUNT,
};

struct pl_color_map_params {
    const struct pl_gamut_map_function *gamut_mapping;
    struct pl_gamut_map_constants gamut_constants;
    int lut3d_size[3];
    bool lut3d_tricubic;        // cubic B-spline instead of trilinear 3-D LUT lookup
    bool gamut_expansion;

    const struct pl_tone_map_function *tone_mapping_function;
    struct pl_tone_map_constants tone_constants;
    bool inverse_tone_mapping;
    enum pl_hdr_metadata_type metadata;
    int lut_size;
    float contrast_recovery;
    float contrast_smoothness;

    bool force_tone_mapping_lut;
    bool visualize_lut;         // unsupported (ignored)
    pl_rect2df visualize_rect;
    float visualize_hue;
    float visualize_theta;
    bool show_clipping;         // unsupported (ignored)

    // Members of older API levels, same positions as in the reference (:306-311).
    enum pl_tone_map_mode tone_mapping_mode;    // ignored (removed in v6.269)
    float tone_mapping_param;                   // forwarded to pl_tone_map_params.param
    float tone_mapping_crosstalk;               // ignored (fixed at 0.04)
    enum pl_rendering_intent intent;            // with gamut_mode: selects a gamut mapping
    enum pl_gamut_mode gamut_mode;              // function the way the reference does (:1717)
    float hybrid_mix;                           // ignored
};

#define PL_COLOR_MAP_DEFAULTS                                   \
    .gamut_mapping          = &pl_gamut_map_perceptual,         \
    .tone_mapping_function  = &pl_tone_map_spline,              \
    .gamut_constants        = { PL_GAMUT_MAP_CONSTANTS },       \
    .tone_constants         = { PL_TONE_MAP_CONSTANTS },        \
    .metadata               = PL_HDR_METADATA_ANY,              \
    .lut3d_size             = {48, 32, 256},                    \
    .lut_size               = 256,                              \
    .visualize_rect         = {0, 0, 1, 1},                     \
    .contrast_smoothness    = 3.5f,

#define PL_COLOR_MAP_HQ_DEFAULTS                                \
    PL_COLOR_MAP_DEFAULTS                                       \
    .contrast_recovery      = 0.30f,

#define pl_color_map_params(...) (&(struct pl_color_map_params) { PL_COLOR_MAP_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_color_map_params pl_color_map_default_params;
PL_API extern const struct pl_color_map_params pl_color_map_high_quality_params;

struct pl_color_map_args {
    struct pl_color_space src;
    struct pl_color_space dst;
    bool prelinearized;
    pl_shader_obj *state;   // tone/gamut LUTs + detected peak
    pl_tex feature_map;     // contrast recovery: low-resolution r16hf map of
                            // pl_shader_extract_features output (NULL = none)
};

#define pl_color_map_args(...) (&(struct pl_color_map_args) { __VA_ARGS__ })

// Replaces the colour by the feature the contrast-recovery stage works on: (I of IPT, 0, 0, 1)
// Colour blindness simulation / correction in `csp`: linearize -> cone matrix -> delinearize
// (reference shaders/colorspace.h, src/shaders/colorspace.c:2040-2064)
PL_API void pl_shader_cone_distort(pl_shader sh, struct pl_color_space csp,
                                   const struct pl_cone_params *params);

PL_API void pl_shader_extract_features(pl_shader sh, struct pl_color_space csp);

PL_API void pl_shader_color_map_ex(pl_shader sh, const struct pl_color_map_params *params,
                                   const struct pl_color_map_args *args);

PL_API void pl_shader_color_map(pl_shader sh, const struct pl_color_map_params *params,
                                struct pl_color_space src, struct pl_color_space dst,
                                pl_shader_obj *state, bool prelinearized);

PL_API_END

#endif // LIBPLACEBO_SHADERS_COLORSPACE_H_
