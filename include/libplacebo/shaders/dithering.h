/*
 * libplacebo-hip: dithering stages (K13, K14).
 * API-compatible with the reference's
 * src/include/libplacebo/shaders/dithering.h:29-80 (pl_dither_params),
 * :95-140 (error diffusion).
 */
#ifndef LIBPLACEBO_SHADERS_DITHERING_H_
#define LIBPLACEBO_SHADERS_DITHERING_H_

#include <libplacebo/colorspace.h>
#include <libplacebo/dither.h>
#include <libplacebo/shaders.h>

PL_API_BEGIN

enum pl_dither_method {
    PL_DITHER_BLUE_NOISE,       // 64x64 void-and-cluster LUT (default)
    PL_DITHER_ORDERED_LUT,      // Bayer LUT
    PL_DITHER_ORDERED_FIXED,    // 16x16 Bayer, computed with integer bit tricks
    PL_DITHER_WHITE_NOISE,      // per-pixel PRNG
    PL_DITHER_METHOD_COUNT,
};

struct pl_dither_params {
    enum pl_dither_method method;
    int lut_size;               // log2 of the LUT size (0 = 6)
    bool temporal;              // rotate/mirror the matrix per frame
    enum pl_color_transfer transfer; // for gamma-aware dithering at <= 4 bits
};

#define PL_DITHER_DEFAULTS                              \
    .method     = PL_DITHER_BLUE_NOISE,                 \
    .lut_size   = 6,                                    \
    .transfer   = PL_COLOR_TRC_LINEAR,

#define pl_dither_params(...) (&(struct pl_dither_params) { PL_DITHER_DEFAULTS __VA_ARGS__ })
PL_API extern const struct pl_dither_params pl_dither_default_params;

// Dither the colour to `new_depth` bits. `dither_state` holds the LUT across
// frames (required for the LUT based methods).
PL_API void pl_shader_dither(pl_shader sh, int new_depth, pl_shader_obj *dither_state,
                             const struct pl_dither_params *params);

struct pl_error_diffusion_params {
    pl_tex input_tex;
    pl_tex output_tex;
    int new_depth;
    const struct pl_error_diffusion_kernel *kernel;
};

#define pl_error_diffusion_params(...) (&(struct pl_error_diffusion_params) { __VA_ARGS__ })

// Whole-image error diffusion (one workgroup, LDS ring buffer). Run with
// pl_dispatch_compute.
PL_API bool pl_shader_error_diffusion(pl_shader sh, const struct pl_error_diffusion_params *params);

// LDS bytes the error diffusion kernel needs for an image of this height.
PL_API size_t pl_error_diffusion_shmem_req(const struct pl_error_diffusion_kernel *kernel,
                                           int height);

PL_API_END

#endif // LIBPLACEBO_SHADERS_DITHERING_H_
