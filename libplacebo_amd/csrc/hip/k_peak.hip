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

// The same over the DPP network (no LDS round trips): a prefix sum along each row of 16 lanes, the
// rows' totals broadcast into the next row; lane 63 holds the wave's total. Returns it as a scalar.
#define PEAK_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (int) (v), ctrl, rmask, 0xf, false)
DEV uint32_t wave_sum_dpp(uint32_t v)
{
    v += (uint32_t) PEAK_DPP(v, 0x111, 0xf);    // row_shr:1
    v += (uint32_t) PEAK_DPP(v, 0x112, 0xf);    // row_shr:2
    v += (uint32_t) PEAK_DPP(v, 0x114, 0xf);    // row_shr:4
    v += (uint32_t) PEAK_DPP(v, 0x118, 0xf);    // row_shr:8
    v += (uint32_t) PEAK_DPP(v, 0x142, 0xa);    // row_bcast:15 into rows 1, 3
    v += (uint32_t) PEAK_DPP(v, 0x143, 0xc);    // row_bcast:31 into rows 2, 3
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}

DEV uint32_t wave_max_dpp(uint32_t v)
{
    v = max(v, (uint32_t) PEAK_DPP(v, 0x111, 0xf));
    v = max(v, (uint32_t) PEAK_DPP(v, 0x112, 0xf));
    v = max(v, (uint32_t) PEAK_DPP(v, 0x114, 0xf));
    v = max(v, (uint32_t) PEAK_DPP(v, 0x118, 0xf));
    v = max(v, (uint32_t) PEAK_DPP(v, 0x142, 0xa));
    v = max(v, (uint32_t) PEAK_DPP(v, 0x143, 0xc));
    return (uint32_t) __builtin_amdgcn_readlane((int) v, 63);
}

// One wavefront owns one of the reference's 16x16 workgroups: lane l measures the pixels
// (l & 15, (l >> 4) + 4k), k = 0..3. Nothing crosses waves, so there is no __syncthreads and
// no LDS state besides the wave's 64-bin histogram; a block is PEAK_WAVES independent tiles.
// (The 256-lane / 1 px per lane version spent half of its time in barriers and LDS atomics:
// 4K 127 us -> see DESIGN.md section 7.)
#define PEAK_WAVES 4
#define PL_MIN_INT(a, b) ((a) < (b) ? (a) : (b))
#define PEAK_NPX 4

// op: i0 = transfer (already inferred), i1 = TRC flags, i2 = use_histogram;
// f[0..11] = linearize params (as PLH_OP_LINEARIZE); ptr2 -> extra block:
//   e[0..2] = luma coeffs, e[3] = 203/10000, e[4] = m1, e[5..7] = c1 c2 c3, e[8] = m2,
//   e[9] = cutoff (0 = none)
// Linearisation of the measured copy. The result is quantised to 14 bits of PQ, for which the
// straightforward PQ EOTF -- two native pows and a Newton-refined reciprocal, 14 instructions per
// channel -- is as good as the well-conditioned one the image path uses (35): a relative error of
// 1e-4 in linear light is a tenth of a 14-bit PQ code.
struct peak_pq_consts {
    float inv_m2, c1, c2, c3, inv_m1, gain, out_scale, out_add;
    int flags;
};

DEV peak_pq_consts peak_load_pq(const plh_op &op)
{
    // f[2] = 1/m2, f[3..5] = c1 c2 c3, f[6] = 1/m1, f[7] = 10000/203 (as lin1, transfer.hiph)
    const float *f = op.f;
    return { f[2], f[3], f[4], f[5], f[6], f[7], f[0], f[1], op.i1 };
}

DEV void peak_linearize_pq(float4_t &c, const peak_pq_consts &k)
{
    float v[3] = { c.x, c.y, c.z };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (k.flags & PLH_TRC_CLAMP0)
            v[i] = fmaxf(v[i], 0.0f);
        const float p = plh_powf(v[i], k.inv_m2);
        const float r = div1(fmaxf(p - k.c1, 0.0f), __builtin_fmaf(-k.c3, p, k.c2));
        v[i] = plh_powf(r, k.inv_m1) * k.gain;
        if (k.flags & PLH_TRC_RESCALE)
            v[i] = k.out_scale * v[i] + k.out_add;
    }
    c.x = v[0]; c.y = v[1]; c.z = v[2];
}

DEV void peak_linearize(float4_t &c, const plh_op &op)
{
    if (op.i0 != TRC_PQ) {
        op_linearize(c, op);
        return;
    }
    peak_linearize_pq(c, peak_load_pq(op));
}

// the constants of the detect stage (the block behind op.ptr2), read once per kernel
struct peak_consts {
    float luma[3], white, m1, c1, c2, c3, m2, cutoff;
};

DEV peak_consts peak_load_consts(const plh_op &op)
{
    const float *e = (const float *) op.ptr2;
    return { { e[0], e[1], e[2] }, e[3], e[4], e[5], e[6], e[7], e[8], e[9] };
}

// luma of a linear colour -> 14 bits of PQ
DEV uint32_t peak_luma_pq14(const float4_t &c, const peak_consts &e)
{
    float luma = e.luma[0] * c.x + e.luma[1] * c.y + e.luma[2] * c.z;
    luma *= e.white;
    luma = plh_powf(plh_clamp(luma, 0.0f, 1.0f), e.m1);
    luma = div1(e.c1 + e.c2 * luma, 1.0f + e.c3 * luma);
    luma = plh_powf(luma, e.m2);
    const float cutoff = e.cutoff;
    if (cutoff != 0.0f) {
        // luma *= smoothstep(0, cutoff, luma)
        const float t = plh_clamp(div1(luma, cutoff), 0.0f, 1.0f);
        luma *= t * t * (3.0f - 2.0f * t);
    }
    return (uint32_t) (16383.0f * luma);
}

DEV uint32_t peak_pq14(const float4_t &c_in, const plh_op &op, const peak_consts &e)
{
    float4_t c = c_in;
    if (op.i0 != TRC_LINEAR)
        peak_linearize(c, op);
    return peak_luma_pq14(c, e);
}

DEV uint32_t peak_pq14(const float4_t &c_in, const plh_op &op)
{
    return peak_pq14(c_in, op, peak_load_consts(op));
}

DEV void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// the workgroup-level part of the measurement (colorspace.c:1279-1348) for one tile
DEV void peak_measure(const float4_t (&c)[PEAK_NPX], const plh_op &op, uint32_t *hist,
                      uint32_t wg_idx, void *scratch)
{
    const int lane = threadIdx.x & 63;
    const float cutoff = ((const float *) op.ptr2)[9];
    uint32_t y_pq[PEAK_NPX];
#pragma unroll
    for (int k = 0; k < PEAK_NPX; k++)
        y_pq[k] = peak_pq14(c[k], op);

    if (op.i2) {
        hist[lane] = 0u;
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < PEAK_NPX; k++) {
            int bin = (int) y_pq[k] >> (PQ_BITS - HIST_BITS);
            bin -= HIST_BIAS;
            bin = min(max(bin, 0), PEAK_HIST_BINS - 1);
            const int first = __shfl(bin, 0, 64);
            if (__all(bin == first)) {
                if (lane == 0)
                    atomicAdd(&hist[bin], 64u);
            } else {
                atomicAdd(&hist[bin], 1u);
            }
        }
    }

    uint32_t lane_sum = 0, lane_max = 0, nblack = 0;
#pragma unroll
    for (int k = 0; k < PEAK_NPX; k++) {
        lane_sum += y_pq[k];
        lane_max = max(lane_max, y_pq[k]);
        if (cutoff != 0.0f)
            nblack += (uint32_t) __popcll(__ballot(y_pq[k] == 0u));
    }
    const uint32_t wg_sum = wave_sum(lane_sum);
    const uint32_t wg_max = wave_max(lane_max);

    const uint32_t slice = wg_idx % PEAK_SLICES;
    // All workgroups of a frame would hammer the same 48 words (two cache lines) of the
    // measurement buffer with atomics, which serialises them (~4.5 ns each, 0.6 ms at 4K).
    // They go to one of PLH_PEAK_COPIES scratch copies instead; k_peak_fold adds the copies
    // into the real buffer afterwards. Integer sums / maxima: the result is identical.
    peak_buf *frame = (peak_buf *) scratch + (wg_idx / PEAK_SLICES) % PLH_PEAK_COPIES;
    if (op.i2) {
        wave_lds_fence();
        if (cutoff != 0.0f && lane == 0)
            hist[0] -= nblack;
        wave_lds_fence();
        // (a workgroup typically populates 1-3 of the 64 bins: adding the zeros of the others
        // would only queue ~20x more atomics on the 12 x 64 hot words)
        const uint32_t n = hist[lane];  // 64 lanes = 64 bins
        if (n)
            atomicAdd(&frame->frame_hist[slice][lane], n);
    }

    if (lane == 0) {
        const uint32_t num = PEAK_BW * PEAK_BH - nblack;
        atomicAdd(&frame->frame_wg_count[slice], 1u);
        atomicAdd(&frame->frame_wg_active[slice], min(num, 1u));
        if (num > 0u) {
            atomicAdd(&frame->frame_sum_pq[slice], wg_sum / num);
            atomicMax(&frame->frame_max_pq[slice], wg_max);
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

// LITE: the ops around the measurement only use the cheap cases (plh_ops_lite): the usual
// "plane -> FBO + measurement" pass then needs < 128 VGPRs (4 waves per SIMD instead of 3)
template <bool LITE>
__global__ __launch_bounds__(64 * PEAK_WAVES)
void k_pass_peak(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    __shared__ uint32_t hists[PEAK_WAVES][PEAK_HIST_BINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles_x = (p.width + PEAK_BW - 1) / PEAK_BW;
    const int tiles_y = (p.height + PEAK_BH - 1) / PEAK_BH;
    // the reference's gl_WorkGroupID.y * gl_NumWorkGroups.x + gl_WorkGroupID.x
    const uint32_t wg_idx = blockIdx.x * PEAK_WAVES + wave;
    if (wg_idx >= (uint32_t) (tiles_x * tiles_y))
        return;     // whole wave
    const int tx = wg_idx % tiles_x, ty = wg_idx / tiles_x;

    float4_t c[PEAK_NPX];
    frag_t fcs[PEAK_NPX];
    int idx[PEAK_NPX], idy[PEAK_NPX];
    float px[PEAK_NPX], py[PEAK_NPX];
#pragma unroll
    for (int k = 0; k < PEAK_NPX; k++) {
        idx[k] = tx * PEAK_BW + (lane & 15);
        idy[k] = ty * PEAK_BH + (lane >> 4) + 4 * k;
        const float mx = p.out_scale[0] * ((float) idx[k] + 0.5f);
        const float my = p.out_scale[1] * ((float) idy[k] + 0.5f);
        fcs[k] = { (float) (idx[k] + p.frag_x0) + 0.5f, (float) (idy[k] + p.frag_y0) + 0.5f, 0.0f, 0,
                   mx, my };
        px[k] = plh_attr(s.pos, 0, mx, my);
        py[k] = plh_attr(s.pos, 1, mx, my);
        c[k] = {0.0f, 0.0f, 0.0f, 1.0f};
    }
    if (s.type == PLH_SAMPLE_NEAREST) {
        int txl[PEAK_NPX], tyl[PEAK_NPX];
#pragma unroll
        for (int k = 0; k < PEAK_NPX; k++) {
            txl[k] = plh_wrap((int) __builtin_floorf(px[k] * (float) s.src.w), s.src.w, s.address_mode);
            tyl[k] = plh_wrap((int) __builtin_floorf(py[k] * (float) s.src.h), s.src.h, s.address_mode);
        }
        plh_fetch_n<PEAK_NPX>(s.src, txl, tyl, c);
#pragma unroll
        for (int k = 0; k < PEAK_NPX; k++)
            c[k] = scale4(c[k], s.scale);
    } else if (s.type != PLH_SAMPLE_NONE) {
#pragma unroll
        for (int k = 0; k < PEAK_NPX; k++)
            c[k] = run_sampler_pk(s, px[k], py[k]);
    }

    // ops before the measurement, the measurement, ops after it
    int pk_op = p.num_ops;
    for (int i = 0; i < p.num_ops; i++) {
        if (p.ops[i].kind == PLH_OP_PEAK_DETECT) {
            pk_op = i;
            break;
        }
    }
    apply_ops_n<PEAK_NPX, false, LITE>(c, p.ops, 0, pk_op, fcs);
    if (pk_op < p.num_ops) {
        peak_measure(c, p.ops[pk_op], hists[wave], wg_idx, p.peak_scratch);
        apply_ops_n<PEAK_NPX, false, LITE>(c, p.ops, pk_op + 1, p.num_ops, fcs);
    }

    if (!p.dst.ptr)
        return;     // target-less measurement pass
    int sx[PEAK_NPX], sy[PEAK_NPX];
    bool ok[PEAK_NPX];
#pragma unroll
    for (int k = 0; k < PEAK_NPX; k++) {
        sx[k] = p.base_x + p.dir_x * (p.transpose ? idy[k] : idx[k]);
        sy[k] = p.base_y + p.dir_y * (p.transpose ? idx[k] : idy[k]);
        ok[k] = p.out_scale[0] * (float) idx[k] < 1.0f && p.out_scale[1] * (float) idy[k] < 1.0f &&
                sx[k] >= 0 && sy[k] >= 0 && sx[k] < p.dst.w && sy[k] < p.dst.h;
    }
    plh_store_n<PEAK_NPX>(p.dst, sx, sy, ok, c, p.nt_store);
}

/*
 * k_peak_fast: the renderer's measuring pass as its own kernel -- a plane read texel for texel
 * (identity rect, rgba16 or rgba16hf), the optional identity PLANE_MAP, the measurement, and the
 * rgba16hf intermediate (or no target: the measurement of an existing FBO). Same measurement
 * code (peak_measure: integer sums / maxima / histogram per 16x16 tile, order-free) and the same
 * f16 codes as k_pass_peak, without the sampler switch, the attribute interpolation, the op
 * interpreter and the generic store: a lane owns two horizontally adjacent pixels on two rows
 * (one 16-byte load and store per row) instead of four single pixels.
 */
// STORE: 0 = no target (the measurement of an existing intermediate), 1 = the rgba16hf
// intermediate, 2 = ops are [PEAK_DETECT] [FEATURES] and the target is the r16hf feature plane of
// contrast recovery (pl_shader_extract_features on the measured colours: the renderer merges the
// two passes that read the same intermediate, renderer.c: measure_peak)
template <bool F16SRC, int STORE>
__global__ __launch_bounds__(64 * PEAK_WAVES)
void k_peak_fast(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    __shared__ uint32_t hists[PEAK_WAVES][PEAK_HIST_BINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles_x = (p.width + PEAK_BW - 1) / PEAK_BW;
    const int tiles_y = (p.height + PEAK_BH - 1) / PEAK_BH;
    const uint32_t wg_idx = blockIdx.x * PEAK_WAVES + wave;
    if (wg_idx >= (uint32_t) (tiles_x * tiles_y))
        return;     // whole wave
    const int tx = wg_idx % tiles_x, ty = wg_idx / tiles_x;
    const int w = p.width, h = p.height;
    const bool has_map = STORE != 2 && p.num_ops == 2;  // [identity PLANE_MAP] PEAK_DETECT
    const plh_op &o_map = p.ops[0], &o_pk = p.ops[STORE == 2 ? 0 : p.num_ops - 1];

    // pixels (x0, y), (x0 + 1, y) for y = y0, y0 + 8: lanes beyond the image measure the clamped
    // edge texel, as the padding invocations of the reference's workgroups do
    const int x0 = tx * PEAK_BW + 2 * (lane & 7), y0 = ty * PEAK_BH + (lane >> 3);
    const char *sp = (const char *) s.src.ptr;
    float4_t c[PEAK_NPX];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int y = min(y0 + 8 * k, h - 1);
        const char *row = sp + (size_t) y * s.src.pitch;
        uint4 v;
        if (x0 + 1 < w) {
            v = *(const uint4 *) (row + (size_t) x0 * 8);
        } else {
            const uint2 e = *(const uint2 *) (row + (size_t) min(x0, w - 1) * 8);
            v = make_uint4(e.x, e.y, e.x, e.y);
        }
        const uint32_t q[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const uint32_t lo = q[2 * i], hi = q[2 * i + 1];
            float4_t t;
            if (F16SRC)
                t = { plh_h2f(lo & 0xffff), plh_h2f(lo >> 16), plh_h2f(hi & 0xffff), plh_h2f(hi >> 16) };
            else
                t = { plh_un16(lo & 0xffff), plh_un16(lo >> 16), plh_un16(hi & 0xffff), plh_un16(hi >> 16) };
            // identity PLANE_MAP of the first i1 components: the others take their neutral values
            if (has_map) {
                if (o_map.i1 < 4) t.w = o_map.f[3];
                if (o_map.i1 < 3) t.z = o_map.f[2];
                if (o_map.i1 < 2) t.y = o_map.f[1];
            }
            c[2 * k + i] = t;
        }
    }
    if constexpr (STORE == 2) {
        // the feature plane: I of IPT of every pixel (op_features itself: bit-identical to the pass
        // of its own, k_pass_features), two f16 per row and lane
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int y = y0 + 8 * k;
            if (y >= h || x0 >= w)
                continue;
            uint32_t o[2];
            float4_t t[2] = { c[2 * k], c[2 * k + 1] };
            // (an image that is not linear yet -- HDR10 without a scaler in front -- is linearised
            // for the features: [PEAK_DETECT] LINEARIZE FEATURES)
            if (p.num_ops == 3)
                op_linearize_px(t, p.ops[1]);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                op_features(t[i], p.ops[p.num_ops - 1]);
                o[i] = plh_f2h(t[i].x);
            }
            char *d = (char *) p.dst.ptr + (size_t) y * p.dst.pitch + (size_t) x0 * 2;
            if (x0 + 1 < w)
                *(uint32_t *) d = o[0] | (o[1] << 16);
            else
                *(uint16_t *) d = (uint16_t) o[0];
        }
    }
    // (the intermediate goes out first: its stores are in flight while the measurement computes)
    if constexpr (STORE == 1) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int y = y0 + 8 * k;
            if (y >= h || x0 >= w)
                continue;
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const float4_t &t = c[2 * k + i];
                o[2 * i] = (uint32_t) plh_f2h(t.x) | ((uint32_t) plh_f2h(t.y) << 16);
                o[2 * i + 1] = (uint32_t) plh_f2h(t.z) | ((uint32_t) plh_f2h(t.w) << 16);
            }
            char *d = (char *) p.dst.ptr + (size_t) y * p.dst.pitch + (size_t) x0 * 8;
            if (x0 + 1 < w)
                *(uint4 *) d = make_uint4(o[0], o[1], o[2], o[3]);
            else
                *(uint2 *) d = make_uint2(o[0], o[1]);
        }
    }
    peak_measure(c, o_pk, hists[wave], wg_idx, p.peak_scratch);
}

/*
 * k_peak_tiles: k_peak_fast with the measurement kept on chip. One wave per 16x16 tile and five
 * global atomics per tile meant 64 scratch copies of the buffer (all tiles of a frame would
 * otherwise queue on the two cache lines that hold its 48 scalar words) and a second kernel to
 * add them up. The slice of a tile is `index % 12`: here every workgroup belongs to ONE slice and
 * its four waves walk that slice's tiles (index = 12 j + slice), so that
 *   - count / lit count / sum of means / maximum live in registers for the whole walk and the
 *     histogram in one 64-word LDS array per workgroup (black pixels taken out of bin 0 once, at
 *     the end: the sums are integer, their order is free);
 *   - it needs 30 registers (one row of a lane's two at a time): one of its waves fits on a SIMD
 *     beside the four waves of the metric's scaler, whose launch the pass then runs inside;
 *   - a workgroup leaves 4 + (bins it touched) global atomics behind, into a layout in which the
 *     scalar words of a slice have a 128-byte line each (at most 170 workgroups per line and frame);
 *   - the workgroup that finishes last gathers the 816 words into the result buffer and the
 *     host's mailbox, zeroes the scratch words and publishes the ticket: no second kernel.
 * Same per-pixel code, same tiles, same integer sums as k_peak_fast / k_pass_peak: the 816 words
 * are identical (tests/test_gpu_kernel_variants.py::test_peak_fast_equals_generic).
 */
#define PEAK_PAD 32     // words between the scalar accumulators of the scratch layout (one line each)
#define PEAK_TICKETS 4096    // word offset of the 12 + 1 ticket counters (a line each), behind the 2304 data words
DEV uint32_t *peak_pad_word(uint32_t *scratch, uint32_t i)
{
    // word i of struct peak_buf in the padded scratch layout: the 48 scalars first, then the
    // histograms (a slice's 64 bins = two lines of their own)
    return i < 4 * PEAK_SLICES ? scratch + i * PEAK_PAD : scratch + 4 * PEAK_SLICES * PEAK_PAD + (i - 4 * PEAK_SLICES);
}

// PQ: the measured copy is linearised from PQ (HDR10 sources); else it is linear already (the
// measurement of a scaler's linear intermediate). STORE: 0 = no target, 1 = the rgba16hf
// intermediate. Every uniform of the walk is read ONCE in front of it and pinned in SGPRs: inside
// the loop a kernel-argument field is re-loaded at each use, every load behind a full scalar wait
// (the first version of this kernel: 174 of them per tile, 108 us for 1080p instead of 13).
template <bool F16SRC, int STORE, bool PQ>
__global__ __launch_bounds__(64 * PEAK_WAVES)
void k_peak_tiles(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    __shared__ uint32_t blk[4 + PEAK_HIST_BINS];    // count, lit count, sum, max, histogram
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int w = p.width, h = p.height;
    const int tiles_x = (w + PEAK_BW - 1) / PEAK_BW;
    const int tiles_y = (h + PEAK_BH - 1) / PEAK_BH;
    const uint32_t ntiles = (uint32_t) (tiles_x * tiles_y);
    const uint32_t slice = blockIdx.x % PEAK_SLICES, bs = blockIdx.x / PEAK_SLICES;
    const uint32_t stride = (gridDim.x / PEAK_SLICES) * PEAK_WAVES;
    const uint32_t per_slice = (ntiles + PEAK_SLICES - 1 - slice) / PEAK_SLICES;    // tiles 12 j + slice < ntiles
    // STORE 0 / 1: [identity PLANE_MAP] PEAK_DETECT; STORE 2: PEAK_DETECT [LINEARIZE (PQ)] FEATURES
    const bool has_map = STORE != 2 && p.num_ops == 2;
    const plh_op &o_map = p.ops[0], &o_pk = p.ops[STORE == 2 ? 0 : p.num_ops - 1];
    // (STORE 2: the fields op_features / lin_values<TRC_PQ> read, copied once and pinned)
    plh_op feat, flin;
    bool feat_lin = false;
    if constexpr (STORE == 2) {
        const plh_op &of = p.ops[p.num_ops - 1], &ol = p.ops[1];
        feat_lin = p.num_ops == 3;
#pragma unroll
        for (int k = 0; k < 14; k++)
            feat.f[k] = of.f[k];
        flin.i1 = ol.i1;
#pragma unroll
        for (int k = 0; k < 8; k++)
            flin.f[k] = ol.f[k];
        asm volatile("" : "+s"(feat.f[0]), "+s"(feat.f[1]), "+s"(feat.f[2]), "+s"(feat.f[3]), "+s"(feat.f[4]),
                          "+s"(feat.f[5]), "+s"(feat.f[6]), "+s"(feat.f[7]), "+s"(feat.f[8]), "+s"(feat.f[9]),
                          "+s"(feat.f[10]), "+s"(feat.f[11]), "+s"(feat.f[12]), "+s"(feat.f[13]));
        asm volatile("" : "+s"(flin.i1), "+s"(flin.f[0]), "+s"(flin.f[1]), "+s"(flin.f[2]), "+s"(flin.f[3]),
                          "+s"(flin.f[4]), "+s"(flin.f[5]), "+s"(flin.f[6]), "+s"(flin.f[7]));
    }
    // (the detect stage's block is read through a global pointer: vector loads of one address)
    auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const peak_consts pcv = peak_load_consts(o_pk);
    peak_consts pc = { { uni(pcv.luma[0]), uni(pcv.luma[1]), uni(pcv.luma[2]) }, uni(pcv.white), uni(pcv.m1),
                       uni(pcv.c1), uni(pcv.c2), uni(pcv.c3), uni(pcv.m2), uni(pcv.cutoff) };
    peak_pq_consts pq = peak_load_pq(o_pk);
    int use_hist = o_pk.i2;
    const float rtx = 1.0f / (float) tiles_x;     // (uniform, in a vector register)
    uintptr_t sp = (uintptr_t) s.src.ptr, dp = (uintptr_t) p.dst.ptr;
    int spitch = s.src.pitch, dpitch = p.dst.pitch;
    int map_n = 4;
    float nt1 = 0.0f, nt2 = 0.0f, nt3 = 1.0f;
    if (has_map) {
        map_n = o_map.i1;
        nt1 = o_map.f[1]; nt2 = o_map.f[2]; nt3 = o_map.f[3];
    }
    asm volatile("" : "+s"(w), "+s"(h), "+s"(use_hist), "+s"(sp), "+s"(dp), "+s"(spitch),
                      "+s"(dpitch), "+s"(map_n), "+s"(nt1), "+s"(nt2), "+s"(nt3));
    asm volatile("" : "+s"(pc.luma[0]), "+s"(pc.luma[1]), "+s"(pc.luma[2]), "+s"(pc.white), "+s"(pc.m1),
                      "+s"(pc.c1), "+s"(pc.c2), "+s"(pc.c3), "+s"(pc.m2), "+s"(pc.cutoff));
    asm volatile("" : "+s"(pq.inv_m2), "+s"(pq.c1), "+s"(pq.c2), "+s"(pq.c3), "+s"(pq.inv_m1), "+s"(pq.gain),
                      "+s"(pq.out_scale), "+s"(pq.out_add), "+s"(pq.flags));
    typedef __attribute__((address_space(1))) const plh_u32x4 g_u32x4;
    typedef __attribute__((address_space(1))) plh_u32x4 g_u32x4_w;
    typedef __attribute__((address_space(1))) plh_u32x2 g_u32x2_w;

    if (threadIdx.x < 4 + PEAK_HIST_BINS)
        blk[threadIdx.x] = 0u;
    __syncthreads();

    // tile j of the slice: origin of the lane's pixels (x0, y0), (x0 + 1, y0), and the same on row
    // y0 + 8; lanes beyond the image measure the clamped edge texel, as the padding invocations
    // of the reference's workgroups do
    auto origin = [&](uint32_t j, int &x0, int &y0) {
        const uint32_t t = PEAK_SLICES * j + slice;
        const int ty = (int) (((float) t + 0.5f) * rtx);    // exact: t < 2^22
        const int tx = (int) t - ty * tiles_x;
        x0 = tx * PEAK_BW + 2 * (lane & 7);
        y0 = ty * PEAK_BH + (lane >> 3);
    };
    uint32_t n_wg = 0, n_lit = 0, sum_pq = 0, lane_max = 0, black = 0;     // (all but lane_max uniform)
    // One row of the lane's two at a time, in a rolled loop: the kernel is held to 32 registers so
    // that one of its waves fits on a SIMD BESIDE the four 120-register waves of the metric's
    // scaler (k_polar_mx<3, true, 3, 8>) -- the measuring pass of the next frame then runs inside
    // the scaler's launch instead of between two of them.
    for (uint32_t j = bs * PEAK_WAVES + (uint32_t) wave; j < per_slice; j += stride) {
        int x0, y0;
        origin(j, x0, y0);
        const int px = min(x0, w - 2);
        uint32_t lane_sum = 0, nblack = 0;
        auto fetch = [&](int k) {
            return *(g_u32x4 *) (sp + (size_t) min(y0 + 8 * k, h - 1) * (size_t) spitch + (size_t) px * 8);
        };
        auto row = [&](int k, plh_u32x4 v) {
            const int y = y0 + 8 * k;
            if (x0 + 1 >= w) {      // both texels are the row's last one
                v.x = v.z; v.y = v.w;
            }
            const uint32_t q[4] = { v.x, v.y, v.z, v.w };
            float4_t c[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const uint32_t lo = q[2 * i], hi = q[2 * i + 1];
                float4_t t;
                if (F16SRC)
                    t = { plh_h2f(lo & 0xffff), plh_h2f(lo >> 16), plh_h2f(hi & 0xffff), plh_h2f(hi >> 16) };
                else
                    t = { plh_un16(lo & 0xffff), plh_un16(lo >> 16), plh_un16(hi & 0xffff), plh_un16(hi >> 16) };
                // identity PLANE_MAP of the first i1 components: the others take their neutral values
                if (map_n < 4) t.w = nt3;
                if (map_n < 3) t.z = nt2;
                if (map_n < 2) t.y = nt1;
                c[i] = t;
            }
            if constexpr (STORE == 2) {
                // the feature plane: I of IPT of every pixel (op_features itself: bit-identical to the
                // pass of its own, k_pass_features), two f16 per row and lane; an image that is not
                // linear yet -- HDR10 without a scaler in front -- is linearised for the features
                float4_t t[2] = { c[0], c[1] };
                if (feat_lin) {
                    float v6[6] = { t[0].x, t[0].y, t[0].z, t[1].x, t[1].y, t[1].z };
                    lin_values<TRC_PQ>(v6, flin);
                    t[0].x = v6[0]; t[0].y = v6[1]; t[0].z = v6[2];
                    t[1].x = v6[3]; t[1].y = v6[4]; t[1].z = v6[5];
                }
                uint32_t o[2];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    op_features(t[i], feat);
                    o[i] = plh_f2h(t[i].x);
                }
                const uintptr_t d = dp + (size_t) y * (size_t) dpitch + (size_t) x0 * 2;
                if (y < h && x0 + 1 < w)
                    *(__attribute__((address_space(1))) uint32_t *) d = o[0] | (o[1] << 16);
                else if (y < h && x0 < w)
                    *(__attribute__((address_space(1))) uint16_t *) d = (uint16_t) o[0];
            }
            // (the intermediate goes out first: its stores are in flight while the measurement computes)
            if constexpr (STORE == 1) {
                uint32_t o[4];
#pragma unroll
                for (int i = 0; i < 2; i++) {
                    o[2 * i] = (uint32_t) plh_f2h(c[i].x) | ((uint32_t) plh_f2h(c[i].y) << 16);
                    o[2 * i + 1] = (uint32_t) plh_f2h(c[i].z) | ((uint32_t) plh_f2h(c[i].w) << 16);
                }
                const uintptr_t d = dp + (size_t) y * (size_t) dpitch + (size_t) x0 * 8;
                if (y < h && x0 + 1 < w)
                    *(g_u32x4_w *) d = (plh_u32x4) { o[0], o[1], o[2], o[3] };
                else if (y < h && x0 < w)
                    *(g_u32x2_w *) d = (plh_u32x2) { o[0], o[1] };
            }
            // the measurement (colorspace.c:1279-1348)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if (PQ)
                    peak_linearize_pq(c[i], pq);
                const uint32_t y_pq = peak_luma_pq14(c[i], pc);
                if (use_hist) {
                    int bin = (int) y_pq >> (PQ_BITS - HIST_BITS);
                    bin -= HIST_BIAS;
                    bin = min(max(bin, 0), PEAK_HIST_BINS - 1);
                    const int first = __builtin_amdgcn_readfirstlane(bin);
                    if (__all(bin == first)) {
                        if (lane == 0)
                            atomicAdd(&blk[4 + bin], 64u);
                    } else {
                        atomicAdd(&blk[4 + bin], 1u);
                    }
                }
                lane_sum += y_pq;
                lane_max = max(lane_max, y_pq);
                if (pc.cutoff != 0.0f)
                    nblack += (uint32_t) __popcll(__ballot(y_pq == 0u));
            }
        };
        if constexpr (STORE == 2) {
            // (the variant that also writes the feature plane runs alone on the main stream: both
            // rows at once, their loads in flight together -- it need not fit beside anything)
            const plh_u32x4 v0 = fetch(0), v1 = fetch(1);
            row(0, v0);
            row(1, v1);
        } else {
#pragma unroll 1
            for (int k = 0; k < 2; k++)
                row(k, fetch(k));
        }
        const uint32_t wg_sum = wave_sum_dpp(lane_sum);
        const uint32_t num = PEAK_BW * PEAK_BH - nblack;
        n_wg += 1u;
        n_lit += min(num, 1u);
        if (nblack == 0u)
            sum_pq += wg_sum / (PEAK_BW * PEAK_BH);
        else if (num > 0u)
            sum_pq += wg_sum / num;
        black += nblack;
    }

    // the wave's totals into the workgroup's, the workgroup's into the frame's
    const uint32_t wmax = wave_max_dpp(lane_max);   // (an all-black tile's maximum is 0: no effect)
    if (lane == 0 && n_wg) {
        atomicAdd(&blk[0], n_wg);
        atomicAdd(&blk[1], n_lit);
        atomicAdd(&blk[2], sum_pq);
        atomicMax(&blk[3], wmax);
        if (use_hist && black)
            atomicSub(&blk[4], black);
    }
    __syncthreads();
    if (wave != 0)
        return;     // (their registers are free for the next workgroup; wave 0 carries the result out)
    // No agent-scope fence anywhere below: on this chip that is a write-back of the XCD's whole L2
    // (buffer_wbl2), and with one per workgroup the kernel took 103 us instead of 13. What the
    // last workgroup reads are only words that were written by device-scope atomics, and it reads
    // them with atomics as well (exchange with zero: the read and the clean-up in one): every
    // access to those words is performed at the point where the XCDs' atomics meet. A workgroup
    // takes its ticket after its own atomics have RETURNED (their results are consumed here).
    uint32_t *scratch = (uint32_t *) p.peak_scratch;
    uint32_t seen = 0;
    if (lane < 4) {
        const uint32_t v = blk[lane];
        uint32_t *d = peak_pad_word(scratch, (uint32_t) lane * PEAK_SLICES + slice);
        if (lane == 3)
            seen += atomicMax(d, v);
        else if (v)
            seen += atomicAdd(d, v);
    }
    const uint32_t n = blk[4 + lane];
    if (use_hist && n)
        seen += atomicAdd(peak_pad_word(scratch, 4 * PEAK_SLICES + slice * PEAK_HIST_BINS + (uint32_t) lane), n);
    asm volatile("" :: "v"(seen));
    // tickets in two levels -- per slice, then one for the slices -- so that no word sees more than
    // gridDim.x / 12 of these returning atomics (2040 on one word were 9 us of the kernel)
    uint32_t *tickets = scratch + PEAK_TICKETS;
    const uint32_t groups = gridDim.x / PEAK_SLICES;
    uint32_t ticket = 0;
    if (lane == 0) {
        ticket = atomicAdd(tickets + slice * PEAK_PAD, 1u);
        if (ticket == groups - 1) {
            atomicExch(tickets + slice * PEAK_PAD, 0u);
            ticket = atomicAdd(tickets + PEAK_SLICES * PEAK_PAD, 1u) + groups;     // (last of all: groups + 11)
        }
    }
    if ((uint32_t) __builtin_amdgcn_readfirstlane((int) ticket) != groups + PEAK_SLICES - 1)
        return;
    uint32_t *counter = tickets + PEAK_SLICES * PEAK_PAD;
    uint32_t *dst = (uint32_t *) p.peak_buf, *mailbox = (uint32_t *) p.peak_mailbox;
    for (uint32_t i = lane; i < PLH_PEAK_WORDS; i += 64) {
        const uint32_t v = atomicExch(peak_pad_word(scratch, i), 0u);
        dst[i] = v;         // the whole buffer is rewritten: the host never has to clear it
        if (mailbox)
            __hip_atomic_store(&mailbox[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (lane == 0)
        atomicExch(counter, 0u);
    if (!mailbox)
        return;
    // The mailbox words were written through to host memory (system-scope stores); when this
    // wave's count of outstanding memory operations is back to zero they have arrived, and the
    // ticket may follow. (A system-scope release would write the XCD's L2 back first -- megabytes
    // of the intermediate's dirty lines that the host does not read.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) expcnt(0) lgkmcnt(0)
    if (lane == 0)
        __hip_atomic_store(&mailbox[PLH_PEAK_WORDS], p.peak_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the shape k_peak_fast is written for
static bool peak_fast_applies(const plh_pass *pass)
{
    const plh_sampler_args &s = pass->s;
    const char *env = getenv("PL_HIP_PEAK_FAST");
    if (env && env[0] == '0')
        return false;
    const bool native = pass->width == s.src.w && pass->height == s.src.h &&
        s.pos[0][0] == 0.0f && s.pos[0][1] == 0.0f && s.pos[3][0] == 1.0f && s.pos[3][1] == 1.0f &&
        s.pos[1][0] == 1.0f && s.pos[1][1] == 0.0f && s.pos[2][0] == 0.0f && s.pos[2][1] == 1.0f;
    const bool target = pass->dst.ptr != NULL;
    const bool plain_target = pass->base_x == 0 && pass->base_y == 0 && pass->dir_x == 1 && pass->dir_y == 1 &&
                              pass->dst.w >= pass->width && pass->dst.h >= pass->height;
    // PEAK_DETECT [LINEARIZE] FEATURES into the r16hf feature plane (STORE = 2)
    if (native && s.type == PLH_SAMPLE_NEAREST && s.scale == 1.0f &&
        (s.src.fmt == PLH_FMT_RGBA16 || s.src.fmt == PLH_FMT_RGBA16F) &&
        s.address_mode == PLH_ADDRESS_CLAMP && !pass->transpose && !pass->num_pre_ops && target &&
        pass->dst.fmt == PLH_FMT_R16F && plain_target && pass->ops[0].kind == PLH_OP_PEAK_DETECT &&
        ((pass->num_ops == 2 && pass->ops[1].kind == PLH_OP_FEATURES) ||
         (pass->num_ops == 3 && pass->ops[1].kind == PLH_OP_LINEARIZE && pass->ops[2].kind == PLH_OP_FEATURES)))
        return true;
    return native && s.type == PLH_SAMPLE_NEAREST && s.scale == 1.0f &&
           (s.src.fmt == PLH_FMT_RGBA16 || s.src.fmt == PLH_FMT_RGBA16F) &&
           s.address_mode == PLH_ADDRESS_CLAMP && !pass->transpose && !pass->num_pre_ops &&
           (!target || (pass->dst.fmt == PLH_FMT_RGBA16F && pass->base_x == 0 && pass->base_y == 0 &&
                        pass->dir_x == 1 && pass->dir_y == 1 && pass->dst.w >= pass->width &&
                        pass->dst.h >= pass->height)) &&
           ((pass->num_ops == 1 && pass->ops[0].kind == PLH_OP_PEAK_DETECT) ||
            (pass->num_ops == 2 && pass->ops[0].kind == PLH_OP_PLANE_MAP && pass->ops[0].i2 &&
             pass->ops[0].i1 >= 1 && pass->ops[1].kind == PLH_OP_PEAK_DETECT));
}

// Sum (maximum for frame_max_pq) of the PLH_PEAK_COPIES partial buffers. 13 blocks of 64 words x
// 4 copy groups: every lane has 16 independent loads in flight, one memory round trip for the
// whole fold (a single block takes 12 us: one CU cannot pull 200 KiB any faster). The block
// that finishes last copies the 816 words into the host mailbox (if any) and publishes the
// ticket behind them with system-scope ordering; `scratch[COPIES * WORDS]` counts the blocks.
#define FOLD_WORDS 64
#define FOLD_GROUPS 4
__global__ __launch_bounds__(FOLD_WORDS * FOLD_GROUPS)
void k_peak_fold(uint32_t *dst, uint32_t *scratch, uint32_t *mailbox, uint32_t ticket)
{
    __shared__ uint32_t part[FOLD_GROUPS][FOLD_WORDS];
    __shared__ bool last;
    const uint32_t lane = threadIdx.x & (FOLD_WORDS - 1), g = threadIdx.x / FOLD_WORDS;
    const uint32_t t = blockIdx.x * FOLD_WORDS + lane;
    const bool is_max = t >= 3 * PEAK_SLICES && t < 4 * PEAK_SLICES;   // frame_max_pq
    constexpr int PER = PLH_PEAK_COPIES / FOLD_GROUPS;
    uint32_t acc = 0;
    if (t < PLH_PEAK_WORDS) {
        uint32_t v[PER];
#pragma unroll
        for (int c = 0; c < PER; c++)
            v[c] = scratch[(g * PER + c) * PLH_PEAK_WORDS + t];
#pragma unroll
        for (int c = 0; c < PER; c++) {
            acc = is_max ? max(acc, v[c]) : acc + v[c];
            scratch[(g * PER + c) * PLH_PEAK_WORDS + t] = 0u;
        }
    }
    part[g][lane] = acc;
    __syncthreads();
    if (g == 0 && t < PLH_PEAK_WORDS) {
#pragma unroll
        for (int k = 1; k < FOLD_GROUPS; k++)
            acc = is_max ? max(acc, part[k][lane]) : acc + part[k][lane];
        dst[t] = acc;       // the whole buffer is rewritten: the host never has to clear it
    }
    if (!mailbox)
        return;
    __threadfence();
    __syncthreads();
    uint32_t *counter = scratch + PLH_PEAK_COPIES * PLH_PEAK_WORDS;
    if (threadIdx.x == 0)
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last)
        return;
    __threadfence();
    for (uint32_t i = threadIdx.x; i < PLH_PEAK_WORDS; i += blockDim.x) {
        const uint32_t v = __hip_atomic_load(&dst[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mailbox[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *counter = 0u;
        __hip_atomic_store(&mailbox[PLH_PEAK_WORDS], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int plh_launch_peak(hipStream_t stream, const plh_pass *pass)
{
    const int tiles = ((pass->width + PEAK_BW - 1) / PEAK_BW) * ((pass->height + PEAK_BH - 1) / PEAK_BH);
    const dim3 block(64 * PEAK_WAVES);
    const dim3 grid((tiles + PEAK_WAVES - 1) / PEAK_WAVES);
    if (!pass->peak_buf || !pass->peak_scratch)
        return -1002;
    int pk_op = pass->num_ops;
    for (int i = 0; i < pass->num_ops; i++) {
        if (pass->ops[i].kind == PLH_OP_PEAK_DETECT) {
            pk_op = i;
            break;
        }
    }
    const char *v2 = getenv("PL_HIP_PEAK_TILES");
    const int trc = pk_op < pass->num_ops ? pass->ops[pk_op].i0 : -1;
    const bool feat_target = pass->dst.ptr && pass->dst.fmt == PLH_FMT_R16F;
    // (the feature plane: PEAK_DETECT FEATURES of a linear image, or PEAK_DETECT LINEARIZE FEATURES of PQ)
    const bool feat_ok = !feat_target || pass->num_ops == 2 || pass->ops[1].i0 == TRC_PQ;
    if (peak_fast_applies(pass) && !(v2 && v2[0] == '0') && feat_ok && pass->width >= 2 &&
        (trc == TRC_PQ || trc == TRC_LINEAR)) {
        // k_peak_tiles folds its own result (no k_peak_fold behind it)
        const bool f16 = pass->s.src.fmt == PLH_FMT_RGBA16F, store = pass->dst.ptr != NULL, pq = trc == TRC_PQ;
        const int per_slice = (tiles + PEAK_SLICES - 1) / PEAK_SLICES;
        // (two tiles per wave, at most 8 workgroups per CU. Measured beside the metric's scaler and
        // alone, 1080p and 4K, profiles/r05_06_peak_groups.txt: fewer, longer-lived workgroups suit
        // the former, more the latter; this is the setting that loses neither)
        const char *genv = getenv("PL_HIP_PEAK_GROUPS");
        const int gmax = genv ? atoi(genv) : 170;
        int groups = (per_slice + 2 * PEAK_WAVES - 1) / (2 * PEAK_WAVES);
        groups = groups < 1 ? 1 : groups > gmax ? gmax : groups;
        const dim3 tgrid(PEAK_SLICES * groups);
#define PEAK_TILES_GO(F, S, Q) PLH_LAUNCH_LAST((k_peak_tiles<F, S, Q>), tgrid, block, 0, stream, *pass)
        if (f16 && feat_target) { if (pq) PEAK_TILES_GO(true, 2, true); else PEAK_TILES_GO(true, 2, false); }
        else if (feat_target)   { if (pq) PEAK_TILES_GO(false, 2, true); else PEAK_TILES_GO(false, 2, false); }
        else if (f16 && store)  { if (pq) PEAK_TILES_GO(true, 1, true); else PEAK_TILES_GO(true, 1, false); }
        else if (f16)           { if (pq) PEAK_TILES_GO(true, 0, true); else PEAK_TILES_GO(true, 0, false); }
        else if (store)         { if (pq) PEAK_TILES_GO(false, 1, true); else PEAK_TILES_GO(false, 1, false); }
        else                    { if (pq) PEAK_TILES_GO(false, 0, true); else PEAK_TILES_GO(false, 0, false); }
#undef PEAK_TILES_GO
        const hipError_t terr = hipGetLastError();
        return terr == hipSuccess ? 0 : -(int) terr;
    }
    if (peak_fast_applies(pass)) {
        const bool f16 = pass->s.src.fmt == PLH_FMT_RGBA16F, store = pass->dst.ptr != NULL;
        const bool feat = store && pass->dst.fmt == PLH_FMT_R16F;
        if (f16 && feat)        hipLaunchKernelGGL((k_peak_fast<true, 2>), grid, block, 0, stream, *pass);
        else if (feat)          hipLaunchKernelGGL((k_peak_fast<false, 2>), grid, block, 0, stream, *pass);
        else if (f16 && store)  hipLaunchKernelGGL((k_peak_fast<true, 1>), grid, block, 0, stream, *pass);
        else if (f16)           hipLaunchKernelGGL((k_peak_fast<true, 0>), grid, block, 0, stream, *pass);
        else if (store)         hipLaunchKernelGGL((k_peak_fast<false, 1>), grid, block, 0, stream, *pass);
        else                    hipLaunchKernelGGL((k_peak_fast<false, 0>), grid, block, 0, stream, *pass);
    } else if (plh_ops_lite(pass, 0, pk_op) && plh_ops_lite(pass, PL_MIN_INT(pk_op + 1, pass->num_ops), pass->num_ops))
        hipLaunchKernelGGL(k_pass_peak<true>, grid, block, 0, stream, *pass);
    else
        hipLaunchKernelGGL(k_pass_peak<false>, grid, block, 0, stream, *pass);
    static_assert(PLH_PEAK_COPIES % FOLD_GROUPS == 0, "copy groups");
    PLH_LAUNCH_LAST(k_peak_fold, dim3((PLH_PEAK_WORDS + FOLD_WORDS - 1) / FOLD_WORDS),
                       dim3(FOLD_WORDS * FOLD_GROUPS), 0, stream, (uint32_t *) pass->peak_buf,
                       (uint32_t *) pass->peak_scratch, (uint32_t *) pass->peak_mailbox,
                       pass->peak_ticket);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
