/*
 * libplacebo-hip: dither matrix generators and error-diffusion kernels.
 * API-compatible with the reference's src/include/libplacebo/dither.h:26-82.
 */
#ifndef LIBPLACEBO_DITHER_H_
#define LIBPLACEBO_DITHER_H_

#include <libplacebo/common.h>

PL_API_BEGIN

// Fill a size×size matrix (row-major floats in [0,1)) with a Bayer ordered
// dither pattern. `size` must be a power of two.
PL_API void pl_generate_bayer_matrix(float *data, int size);

// Fill a size×size matrix with void-and-cluster blue noise. `size` must be a
// power of two (<= 256). Ties are broken with libc rand(), like the reference
// (src/dither.c:162): seed with srand() for a reproducible matrix.
PL_API void pl_generate_blue_noise(float *data, int size);

#define PL_EDF_MIN_DX (-2)
#define PL_EDF_MAX_DX  (2)
#define PL_EDF_MAX_DY  (2)

struct pl_error_diffusion_kernel {
    const char *name;
    const char *description;
    int shift; // column shift per row so that diffusion only goes "forward"
    int pattern[PL_EDF_MAX_DY + 1][PL_EDF_MAX_DX - PL_EDF_MIN_DX + 1];
    int divisor;
};

PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_simple;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_false_fs;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_sierra_lite;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_floyd_steinberg;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_atkinson;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_jarvis_judice_ninke;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_stucki;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_burkes;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_sierra2;
PL_API extern const struct pl_error_diffusion_kernel pl_error_diffusion_sierra3;

PL_API extern const struct pl_error_diffusion_kernel * const pl_error_diffusion_kernels[];
PL_API extern const int pl_num_error_diffusion_kernels;
PL_API const struct pl_error_diffusion_kernel *pl_find_error_diffusion_kernel(const char *name);

PL_API_END

#endif // LIBPLACEBO_DITHER_H_
