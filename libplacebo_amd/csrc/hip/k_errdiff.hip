/*
 * libplacebo-hip — error diffusion dithering kernel (K14).
 *
 * Device half of pl_shader_error_diffusion (src/shaders/dithering.c:326-527).
 * The algorithm is the reference's: after the shear (y, x) -> (y, x + y*shift)
 * every pixel of a sheared column only receives error from earlier columns, so
 * a column is processed in parallel; columns are consumed in order by ONE
 * workgroup, `block_size` pixels per step with a barrier between steps. Errors
 * travel through a ring buffer of (height + 2) x ring_cols packed words in LDS,
 *
 *     | R8 | 0000 | G8 | 0000 | B8 |     bits 31-24, 19-12, 7-0
 *
 * added with atomics (integer, order independent -> deterministic).
 *
 * MI355X: the ring buffer of a 2160-row frame with the 3-row kernels is 78 KB
 * and 156 KB at 4320 rows; both fit the 160 KB LDS of one CU, where the
 * reference (32-64 KB of shared memory on other GPUs) has to fall back to
 * ordered dithering. The launch is a single 1024-lane workgroup by construction
 * of the algorithm; throughput is bounded by its height*(width + height*shift) /
 * 1024 sequential steps, not by memory.
 *
 * round() is round-half-even here and in the oracle (v_rndne_f32 / rintf).
 */
#include "devmath.hiph"
#include "backend.h"

__global__ __launch_bounds__(1024)
void k_errdiff(const plh_errdiff_args a)
{
    extern __shared__ uint32_t err_rgb8[];
    const uint32_t ring_size = (uint32_t) a.ring_rows * (uint32_t) a.ring_cols;
    const uint32_t tid = threadIdx.x, bs = blockDim.x;
    if (blockIdx.x != 0)
        return; // safeguard against accidental over-execution

    for (uint32_t i = tid; i < ring_size; i += bs)
        err_rgb8[i] = 0u;

    const float quant = (float) a.quant;
    const uint32_t height = (uint32_t) a.height;
    for (uint32_t block_id = 0; block_id < (uint32_t) a.blocks; block_id++) {
        __syncthreads();
        const uint32_t id = block_id * bs + tid;
        const int y = (int) (id % height), x_shifted = (int) (id / height);
        const int x = x_shifted - y * a.shift;
        if (x < 0 || x >= a.width)
            continue;

        const uint32_t idx = (uint32_t) (x_shifted * a.ring_rows + y) % ring_size;
        const float4_t pix_orig = plh_fetch(a.src, x, y);

        // add the error previously propagated into this pixel, clear its slot
        const uint32_t err_u32 = atomicExch(&err_rgb8[idx], 0u) +
                                 ((128u << 24) | (128u << 12) | 128u);
        float pix[3] = { pix_orig.x, pix_orig.y, pix_orig.z };
        const int e[3] = { (int) ((err_u32 >> 24) & 0xFFu) - 128,
                           (int) ((err_u32 >> 12) & 0xFFu) - 128,
                           (int) (err_u32 & 0xFFu) - 128 };
        float dithered[3], err_div[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            pix[c] = pix[c] * quant + (float) e[c] / 254.0f;
            dithered[c] = __builtin_rintf(pix[c]);
            err_div[c] = (pix[c] - dithered[c]) * 254.0f / (float) a.divisor;
        }
        const float4_t out = { dithered[0] / quant, dithered[1] / quant, dithered[2] / quant,
                               pix_orig.w };
        plh_store(a.dst, x, y, out);

        // propagate, grouped by weight (dithering.c:480-521)
        for (int dividend = 1; dividend <= a.divisor; dividend++) {
            bool assigned = false;
            uint32_t packed = 0;
            for (int dy = 0; dy <= 2; dy++) {
                for (int dx = -2; dx <= 2; dx++) {
                    if (a.pattern[dy][dx + 2] != dividend)
                        continue;
                    if (!assigned) {
                        assigned = true;
                        const int tr = (int) __builtin_rintf(err_div[0] * (float) dividend);
                        const int tg = (int) __builtin_rintf(err_div[1] * (float) dividend);
                        const int tb = (int) __builtin_rintf(err_div[2] * (float) dividend);
                        packed = ((uint32_t) (tr & 0xFF) << 24) | ((uint32_t) (tg & 0xFF) << 12) |
                                 (uint32_t) (tb & 0xFF);
                    }
                    // errors leaving through the left border stay in the ring buffer in the
                    // reference unless guarded like this (dithering.c:508-513)
                    if (dx < 0 && x < -dx)
                        continue;
                    const int shifted_x = dx + dy * a.shift;
                    const uint32_t delta = (uint32_t) (shifted_x * a.ring_rows + dy);
                    atomicAdd(&err_rgb8[(idx + delta) % ring_size], packed);
                }
            }
        }
    }
}

extern "C" int plh_launch_errdiff(plh_stream stream, const plh_errdiff_args *args)
{
    const size_t shmem = (size_t) args->ring_rows * args->ring_cols * sizeof(uint32_t);
    if (shmem > 160 * 1024)
        return -1000;
    // opt in to more than the default 64 KiB of dynamic LDS, once per device
    static uint64_t lds_done;
    if (shmem > 64 * 1024)
        (void) plh_kernel_needs_lds((const void *) k_errdiff, stream, 160 * 1024, &lds_done);
    hipLaunchKernelGGL(k_errdiff, dim3(1), dim3(args->block_size), shmem, (hipStream_t) stream,
                       *args);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
