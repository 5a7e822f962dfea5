/*
 * libplacebo-hip: custom 1D / 3D colour LUTs.
 * API-compatible with the reference's src/include/libplacebo/shaders/lut.h
 * (pl_custom_lut :32-58, pl_lut_parse_cube :61, pl_lut_free :64, pl_shader_custom_lut :75).
 */
#ifndef LIBPLACEBO_SHADERS_LUT_H_
#define LIBPLACEBO_SHADERS_LUT_H_

#include <libplacebo/colorspace.h>
#include <libplacebo/shaders.h>

PL_API_BEGIN

struct pl_custom_lut {
    // identifies the contents (cache invalidation); pl_lut_parse_* hash the file
    uint64_t signature;

    // size of each dimension, R G B; 1D LUTs only set size[0]
    int size[3];

    // RGB triples in [0, 1] scale; 3D: R is the innermost dimension, B the outermost
    const float *data;

    // optional matrices applied before / after the lookup (ignored if all zero)
    pl_matrix3x3 shaper_in, shaper_out;

    // nominal metadata of the LUT's input / output ({0} = unknown; informative for
    // pl_shader_custom_lut, used by the renderer to place the LUT in the pipeline)
    struct pl_color_repr repr_in, repr_out;
    struct pl_color_space color_in, color_out;
};

// Parse a LUT in .cube format (1D and 3D, DOMAIN_MIN / DOMAIN_MAX). NULL on failure.
PL_API struct pl_custom_lut *pl_lut_parse_cube(pl_log log, const char *str, size_t str_len);
PL_API void pl_lut_free(struct pl_custom_lut **lut);

// color.rgb = shaper_out * LUT(shaper_in * color.rgb): 1D LUTs per channel with linear
// interpolation, 3D LUTs with tetrahedral interpolation. `lut_state` holds the device copy.
PL_API void pl_shader_custom_lut(pl_shader sh, const struct pl_custom_lut *lut,
                                 pl_shader_obj *lut_state);

PL_API_END

#endif // LIBPLACEBO_SHADERS_LUT_H_
