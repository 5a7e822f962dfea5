/*
 * libplacebo-hip: film grain metadata carried by pl_frame.
 * Data layouts of the reference's src/include/libplacebo/shaders/film_grain.h (:29-110) so
 * that `struct pl_frame` has the reference's size and member offsets. Grain synthesis is not
 * part of this backend: a frame whose film_grain.type is not PL_FILM_GRAIN_NONE is rendered
 * without grain and PL_RENDER_ERR_FILM_GRAIN is raised.
 */
#ifndef LIBPLACEBO_SHADERS_FILM_GRAIN_H_
#define LIBPLACEBO_SHADERS_FILM_GRAIN_H_

#include <stdbool.h>
#include <stdint.h>

#include <libplacebo/colorspace.h>
#include <libplacebo/shaders.h>

PL_API_BEGIN

enum pl_film_grain_type {
    PL_FILM_GRAIN_NONE = 0,
    PL_FILM_GRAIN_AV1,
    PL_FILM_GRAIN_H274,
    PL_FILM_GRAIN_COUNT,
};

// AV1 film grain parameters (AV1 spec section 6.8.20)
struct pl_av1_grain_data {
    int num_points_y;
    uint8_t points_y[14][2];            // [n][0] = value, [n][1] = scaling
    bool chroma_scaling_from_luma;
    int num_points_uv[2];               // Cb, Cr
    uint8_t points_uv[2][10][2];
    int scaling_shift;
    int ar_coeff_lag;
    int8_t ar_coeffs_y[24];
    int8_t ar_coeffs_uv[2][25];
    int ar_coeff_shift;
    int grain_scale_shift;
    int8_t uv_mult[2];
    int8_t uv_mult_luma[2];
    int16_t uv_offset[2];
    bool overlap;
};

// ITU-T H.274 film grain characteristics SEI
struct pl_h274_grain_data {
    int model_id;
    int blending_mode_id;
    int log2_scale_factor;
    bool component_model_present[3];
    uint16_t num_intensity_intervals[3];
    uint8_t num_model_values[3];
    const uint8_t *intensity_interval_lower_bound[3];
    const uint8_t *intensity_interval_upper_bound[3];
    const int16_t (*comp_model_value[3])[6];
};

struct pl_film_grain_data {
    enum pl_film_grain_type type;
    uint64_t seed;
    union {
        struct pl_av1_grain_data av1;
        struct pl_h274_grain_data h274;
    } params;
};

PL_API_END

#endif // LIBPLACEBO_SHADERS_FILM_GRAIN_H_
