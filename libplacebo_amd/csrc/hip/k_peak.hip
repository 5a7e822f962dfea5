/*
 * libplacebo-hip — HDR peak detection (K10).
 *
 * Device half of pl_shader_detect_peak (src/shaders/colorspace.c:1155-1353).
 * A pass that contains a PEAK_DETECT op is launched with this kernel: the same
 * sampler + colour-op interpreter as k_pass_generic, but with the reference's
 * 16x16 workgroup tiling, because the measurement is tiling-dependent:
 *   - every workgroup contributes wg_sum / (256 - wg_black) (integer division)
 *     to frame_sum_pq[slice], slice = (wg.y * numWG.x + wg.x) % 12      :1259-1263,1340-1347
 *   - lanes outside the image still measure their (clamped) sample      (dispatch pads groups)
 *   - luma -> 14-bit PQ -> LDS atomics (sum / max / black count / 64-bin
 *     histogram) -> one set of global atomics per workgroup             :1279-1348
 * Wavefront reductions (__shfl / ballot on 64 lanes) fold the per-lane values
 * before touching LDS, exactly like the reference's subgroup path; integer
 * arithmetic makes the result order-independent.
 */
#include "colorops.hiph"
#include "samplers.hiph"

#define PEAK_BW 16
#define PEAK_BH 16
#define PEAK_SLICES 12
#define PEAK_HIST_BINS 64
#define PQ_BITS 14
#define HIST_BITS 7
#define HIST_BIAS (1 << (HIST_BITS - 1))

// struct peak_buf_data (colorspace.c:936-942): 816 x u32
struct peak_buf {
    uint32_t frame_wg_count[PEAK_SLICES];
    uint32_t frame_wg_active[PEAK_SLICES];
    uint32_t frame_sum_pq[PEAK_SLICES];
    uint32_t frame_max_pq[PEAK_SLICES];
    uint32_t frame_hist[PEAK_SLICES][PEAK_HIST_BINS];
};

DEV uint32_t wave_sum(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

DEV uint32_t wave_max(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1)
        v = max(v, (uint32_t) __shfl_xor((int) v, off, 64));
    return v;
}

// op: i0 = transfer (already inferred), i1 = TRC flags, i2 = use_histogram;
// f[0..11] = linearize params (as PLH_OP_LINEARIZE); ptr2 -> extra block:
//   e[0..2] = luma coeffs, e[3] = 203/10000, e[4] = m1, e[5..7] = c1 c2 c3, e[8] = m2,
//   e[9] = cutoff (0 = none)
DEV void op_peak_detect(const float4_t &c_in, const plh_op &op, const peak_ctx &pk)
{
    const float *e = (const float *) op.ptr2;
    float4_t c = c_in;
    if (op.i0 != TRC_LINEAR)
        op_linearize(c, op);

    float luma = e[0] * c.x + e[1] * c.y + e[2] * c.z;
    luma *= e[3];
    luma = plh_powf(plh_clamp(luma, 0.0f, 1.0f), e[4]);
    luma = (e[5] + e[6] * luma) / (1.0f + e[7] * luma);
    luma = plh_powf(luma, e[8]);
    const float cutoff = e[9];
    if (cutoff != 0.0f) {
        // luma *= smoothstep(0, cutoff, luma)
        const float t = plh_clamp(luma / cutoff, 0.0f, 1.0f);
        luma *= t * t * (3.0f - 2.0f * t);
    }
    const uint32_t y_pq = (uint32_t) (16383.0f * luma);

    const int lane = (threadIdx.y * PEAK_BW + threadIdx.x) & 63;
    if (op.i2) {
        int bin = (int) y_pq >> (PQ_BITS - HIST_BITS);
        bin -= HIST_BIAS;
        bin = min(max(bin, 0), PEAK_HIST_BINS - 1);
        const int first = __shfl(bin, 0, 64);
        if (__all(bin == first)) {
            if (lane == 0)
                atomicAdd(&pk.wg_hist[bin], 64u);
        } else {
            atomicAdd(&pk.wg_hist[bin], 1u);
        }
    }

    const uint32_t group_sum = wave_sum(y_pq);
    const uint32_t group_max = wave_max(y_pq);
    const unsigned long long black = cutoff != 0.0f ? __ballot(y_pq == 0u) : 0ull;
    if (lane == 0) {
        atomicAdd(pk.wg_sum, group_sum);
        atomicMax(pk.wg_max, group_max);
        if (cutoff != 0.0f)
            atomicAdd(pk.wg_black, (uint32_t) __popcll(black));
    }
    __syncthreads();

    const uint32_t local_idx = threadIdx.y * PEAK_BW + threadIdx.x;
    const uint32_t wg_idx = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t slice = wg_idx % PEAK_SLICES;
    // All workgroups of a frame would hammer the same 48 words (two cache lines) of the
    // measurement buffer with atomics, which serialises them (~4.5 ns each, 0.6 ms at 4K).
    // They go to one of PLH_PEAK_COPIES scratch copies instead; k_peak_fold adds the copies
    // into the real buffer afterwards. Integer sums / maxima: the result is identical.
    peak_buf *frame = (peak_buf *) pk.frame + (wg_idx / PEAK_SLICES) % PLH_PEAK_COPIES;
    if (op.i2) {
        if (cutoff != 0.0f && local_idx == 0)
            pk.wg_hist[0] -= *pk.wg_black;
        __syncthreads();
        // (a workgroup typically populates 1-3 of the 64 bins: adding the zeros of the others
        // would only queue ~20x more atomics on the 12 x 64 hot words)
        for (uint32_t i = local_idx; i < PEAK_HIST_BINS; i += PEAK_BW * PEAK_BH) {
            const uint32_t n = pk.wg_hist[i];
            if (n)
                atomicAdd(&frame->frame_hist[slice][i], n);
        }
    }

    if (local_idx == 0) {
        const uint32_t num = PEAK_BW * PEAK_BH - *pk.wg_black;
        atomicAdd(&frame->frame_wg_count[slice], 1u);
        atomicAdd(&frame->frame_wg_active[slice], min(num, 1u));
        if (num > 0u) {
            atomicAdd(&frame->frame_sum_pq[slice], *pk.wg_sum / num);
            atomicMax(&frame->frame_max_pq[slice], *pk.wg_max);
        }
    }
}

DEV float4_t run_sampler_pk(const plh_sampler_args &s, float px, float py)
{
    switch (s.type) {
    case PLH_SAMPLE_NEAREST:
        return scale4(tex_nearest(s.src, s.address_mode, px, py), s.scale);
    case PLH_SAMPLE_BILINEAR:
        return scale4(tex_linear(s.src, s.address_mode, px, py), s.scale);
    case PLH_SAMPLE_BICUBIC:
        return sample_bicubic(s, px, py);
    case PLH_SAMPLE_HERMITE:
        return sample_hermite(s, px, py);
    case PLH_SAMPLE_GAUSSIAN:
        return sample_gaussian(s, px, py);
    case PLH_SAMPLE_OVERSAMPLE:
        return sample_oversample(s, px, py);
    }
    float4_t c = {0.0f, 0.0f, 0.0f, 1.0f};
    return c;
}

__global__ __launch_bounds__(PEAK_BW * PEAK_BH)
void k_pass_peak(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    __shared__ uint32_t wg_state[4 + PEAK_HIST_BINS];
    const uint32_t local_idx = threadIdx.y * PEAK_BW + threadIdx.x;
    if (local_idx < 4 + PEAK_HIST_BINS)
        wg_state[local_idx] = 0u;
    __syncthreads();

    const peak_ctx pk = { &wg_state[0], &wg_state[1], &wg_state[2], &wg_state[4], p.peak_scratch };

    const int idx = blockIdx.x * PEAK_BW + threadIdx.x;
    const int idy = blockIdx.y * PEAK_BH + threadIdx.y;
    const float mx = p.out_scale[0] * ((float) idx + 0.5f);
    const float my = p.out_scale[1] * ((float) idy + 0.5f);

    float4_t c = {0.0f, 0.0f, 0.0f, 1.0f};
    if (p.s.type != PLH_SAMPLE_NONE) {
        const float px = plh_attr(p.s.pos, 0, mx, my);
        const float py = plh_attr(p.s.pos, 1, mx, my);
        c = run_sampler_pk(p.s, px, py);
    }

    const frag_t fc = { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f, 0.0f, 0,
                        mx, my };
    apply_ops<true>(c, p.ops, 0, p.num_ops, fc, &pk);

    const float fx = p.out_scale[0] * (float) idx, fy = p.out_scale[1] * (float) idy;
    if (fx < 1.0f && fy < 1.0f) {
        const int ox = p.base_x + p.dir_x * (p.transpose ? idy : idx);
        const int oy = p.base_y + p.dir_y * (p.transpose ? idx : idy);
        if (ox >= 0 && oy >= 0 && ox < p.dst.w && oy < p.dst.h)
            plh_store(p.dst, ox, oy, c);
    }
}

__global__ void k_peak_fold(uint32_t *dst, uint32_t *scratch)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= PLH_PEAK_WORDS)
        return;
    const bool is_max = t >= 3 * PEAK_SLICES && t < 4 * PEAK_SLICES;   // frame_max_pq
    uint32_t acc = 0;
    for (int c = 0; c < PLH_PEAK_COPIES; c++) {
        const uint32_t v = scratch[c * PLH_PEAK_WORDS + t];
        acc = is_max ? max(acc, v) : acc + v;
        scratch[c * PLH_PEAK_WORDS + t] = 0u;
    }
    if (is_max)
        atomicMax(&dst[t], acc);
    else if (acc)
        atomicAdd(&dst[t], acc);
}

int plh_launch_peak(hipStream_t stream, const plh_pass *pass)
{
    const dim3 block(PEAK_BW, PEAK_BH);
    const dim3 grid((pass->width + PEAK_BW - 1) / PEAK_BW, (pass->height + PEAK_BH - 1) / PEAK_BH);
    if (!pass->peak_buf || !pass->peak_scratch)
        return -1002;
    hipLaunchKernelGGL(k_pass_peak, grid, block, 0, stream, *pass);
    hipLaunchKernelGGL(k_peak_fold, dim3((PLH_PEAK_WORDS + 255) / 256), dim3(256), 0, stream,
                       (uint32_t *) pass->peak_buf, (uint32_t *) pass->peak_scratch);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
