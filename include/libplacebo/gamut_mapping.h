/*
 * libplacebo-hip: gamut-mapping functions (Tier-0 host maths) generating the
 * IPT/ICh 3-D LUT consumed by the colour-mapping kernel.
 * API-compatible with the reference's src/include/libplacebo/gamut_mapping.h
 * (pl_gamut_map_function :30-50, constants :52-90, params :92-125).
 */
#ifndef LIBPLACEBO_GAMUT_MAPPING_H_
#define LIBPLACEBO_GAMUT_MAPPING_H_

#include <libplacebo/colorspace.h>
#include <libplacebo/common.h>

PL_API_BEGIN

struct pl_gamut_map_params;

struct pl_gamut_map_function {
    const char *name;
    const char *description;
    // Maps a LUT of IPT triples (stride `lut_stride` floats) in place
    void (*map)(float *lut, const struct pl_gamut_map_params *params);
    bool bidirectional; // also meaningful when expanding the gamut
    void *priv;
};

struct pl_gamut_map_constants {
    float perceptual_deadzone;  // [0,1]
    float perceptual_strength;  // [0,1]
    float colorimetric_gamma;   // [0,10]
    float softclip_knee;        // [0,1]
    float softclip_desat;       // [0,1]
};

#define PL_GAMUT_MAP_CONSTANTS    \
    .colorimetric_gamma  = 1.80f, \
    .softclip_knee       = 0.70f, \
    .softclip_desat      = 0.35f, \
    .perceptual_deadzone = 0.30f, \
    .perceptual_strength = 0.80f,

struct pl_gamut_map_params {
    const struct pl_gamut_map_function *function;
    struct pl_raw_primaries input_gamut;
    struct pl_raw_primaries output_gamut;
    float min_luma;     // PQ
    float max_luma;     // PQ
    struct pl_gamut_map_constants constants;
    int lut_size_I;
    int lut_size_C;
    int lut_size_h;
    int lut_stride;     // floats between LUT entries (>= 3)
    float chroma_margin; // legacy, unused (layout compatibility)
};

#define pl_gamut_map_params(...) (&(struct pl_gamut_map_params) {   \
    .constants = { PL_GAMUT_MAP_CONSTANTS },                        \
    __VA_ARGS__                                                     \
})

PL_API bool pl_gamut_map_params_equal(const struct pl_gamut_map_params *a,
                                      const struct pl_gamut_map_params *b);
PL_API bool pl_gamut_map_params_noop(const struct pl_gamut_map_params *params);

// Fill out[lut_size_h][lut_size_C][lut_size_I][lut_stride] with the mapped IPT
// of the lattice I in [min,max], C in [0,0.5], h in [-pi,pi]
PL_API void pl_gamut_map_generate(float *out, const struct pl_gamut_map_params *params);
PL_API void pl_gamut_map_sample(float x[3], const struct pl_gamut_map_params *params);

PL_API extern const struct pl_gamut_map_function pl_gamut_map_clip;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_perceptual;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_softclip;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_relative;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_saturation;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_absolute;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_desaturate;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_darken;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_highlight;
PL_API extern const struct pl_gamut_map_function pl_gamut_map_linear;

PL_API extern const struct pl_gamut_map_function * const pl_gamut_map_functions[];
PL_API extern const int pl_num_gamut_map_functions;
PL_API const struct pl_gamut_map_function *pl_find_gamut_map_function(const char *name);

PL_API_END

#endif // LIBPLACEBO_GAMUT_MAPPING_H_
