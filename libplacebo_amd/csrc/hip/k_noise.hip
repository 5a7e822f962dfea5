/*
 * libplacebo-hip -- PL_DITHER_WHITE_NOISE as a plane.
 *
 * The reference evaluates the PRNG inside the dithering shader (dithering.c:204-207:
 * bias = pcg3d(uvec3(gl_FragCoord.xy, seed)).x). Inlined into the op interpreter that costs
 * every kernel that carries the interpreter eight more VGPRs -- one wave per SIMD on the
 * polar kernels (240 -> 313 us on the tone-mapping launch) -- for a method nobody uses in
 * production. (Even a three-line extra case in dither_bias does: 121 -> 129.) So the dispatch
 * evaluates it once per pass into a plane that is laid out as a dither MATRIX -- square, side
 * S = the power of two that covers the pass, of which the width x height corner is filled --
 * and the op becomes an ordinary non-temporal LUT dither (i1 = 0, size S): the kernels index
 * it with (x & (S - 1), y & (S - 1)) like any other matrix, without a line of new device code.
 * Same values, bit for bit; 4 bytes per pixel of extra traffic only when the method is selected.
 */
#include "prng.hiph"

__global__ __launch_bounds__(256)
void k_white_noise(float *plane, int stride, int w, int h, int x0, int y0, uint32_t seed)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h)
        return;
    prng3 st = { (uint32_t) (x + x0), (uint32_t) (y + y0), seed };
    float rnd[3];
    pcg3d(st, rnd);
    plane[(size_t) y * stride + x] = rnd[0];
}

extern "C" int plh_launch_white_noise(hipStream_t stream, float *plane, int stride, int w, int h,
                                      int x0, int y0, uint32_t seed)
{
    hipLaunchKernelGGL(k_white_noise, dim3((w + 63) / 64, (h + 3) / 4), dim3(256), 0, stream,
                       plane, stride, w, h, x0, y0, seed);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
