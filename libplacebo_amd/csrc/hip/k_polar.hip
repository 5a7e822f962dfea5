/*
 * libplacebo-hip — polar (EWA) resampling kernel (K2/K3).
 *
 * Device half of pl_shader_sample_polar (src/shaders/sampling.c:587-912) and
 * polar_sample (:503-558), compute-shader formulation (:723-783):
 *
 *   fcoord = fract(pos*size - 0.5);  base texel = floor(pos*size - 0.5)
 *   for every tap (x, y) of the statically pruned list (host: tap order and
 *   flags are produced exactly like the reference's generation-time loops):
 *       d = length(vec2(x, y) - fcoord);   [if skippable: if (d < R)]
 *       w = LUT(d / R);  wsum += w;  color += w * texel(base + (x, y))
 *       [anti-ringing: weighted soft-min/max of taps with d <= radius_zero]
 *   color = scale / wsum * color;  [AR clamp];  alpha = 1 if not sampled
 *
 * MI355X mapping: a 256-thread workgroup (4 waves, 32x8 lanes) produces a
 * 32 x 8*rows output tile (rows = 4 unless the footprint would not fit). The source footprint of the tile
 * (ceil(32/ratio) + 2*ceil(R) texels square) is staged once in LDS as half4
 * (exactly the precision of the reference's rgba16hf FBO) or float4, together
 * with the 256-entry weight LUT stored as {L[i], L[i+1]} pairs so a tap costs
 * two ds_read_b64. Source texels are read from HBM once per tile with
 * row-contiguous lanes; the optional `pre-ops` (the reference's separate
 * "PASS A": normalise / decode / linearize / sigmoidize / FBO rounding) run on
 * the texels while they are being staged, which removes a full-frame FBO
 * write+read from the frame.
 */
#include "polar_common.hiph"

struct ar_state {
    float ar[4][2], wwsum[4][2];
};

template <typename T, uint32_t MASK, bool USE_AR>
__global__ __launch_bounds__(POLAR_BW * POLAR_BH)
void k_polar(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float2 *lut = (float2 *) smem;                              // 256 pairs = 2 KiB
    tile_px<T> *tile = (tile_px<T> *) (smem + 256 * sizeof(float2));

    const plh_sampler_args &s = p.s;
    const int tid = threadIdx.y * POLAR_BW + threadIdx.x;
    const float sw = (float) s.src.w, sh = (float) s.src.h;

    // ---- workgroup footprint: base texel of the four corner lanes ----------
    const int rows = s.tile_rows;
    const int gx0 = blockIdx.x * POLAR_BW, gy0 = blockIdx.y * POLAR_BH * rows;
    int ox, oy;
    {
        int bx[2], by[2];
        for (int k = 0; k < 2; k++) {
            const float mx = p.out_scale[0] * ((float) (gx0 + k * (POLAR_BW - 1)) + 0.5f);
            const float my = p.out_scale[1] *
                ((float) (gy0 + k * (POLAR_BH * rows - 1)) + 0.5f);
            // pos.x only depends on (mx) when the quad is axis aligned, but we
            // evaluate the full bilinear attribute like every lane does
            const float cx = plh_attr(s.pos, 0, mx, p.out_scale[1] * ((float) gy0 + 0.5f));
            const float cy = plh_attr(s.pos, 1, p.out_scale[0] * ((float) gx0 + 0.5f), my);
            bx[k] = (int) __builtin_floorf(cx * sw - 0.5f);
            by[k] = (int) __builtin_floorf(cy * sh - 0.5f);
        }
        // one texel of slack on the low side (the host adds two to the tile
        // size): the corner estimate may be off by one when pos*size - 0.5
        // lands within rounding noise of an integer
        const int off = s.bound - 1;
        ox = min(bx[0], bx[1]) - off - 1;
        oy = min(by[0], by[1]) - off - 1;
    }

    // ---- stage LUT pairs + source tile in LDS ----------------------------------
    for (int i = tid; i < 256; i += POLAR_BW * POLAR_BH)
        lut[i] = ((const float2 *) s.lut)[i];

    const int tw = s.tile_w, th = s.tile_h;
    // an rgba16hf source without pre-ops is copied bit for bit (the f16 -> f32 -> f16 round
    // trip of the generic path is the identity)
    const bool raw16 = sizeof(tile_px<T>) == 8 && s.src.fmt == PLH_FMT_RGBA16F && !p.num_pre_ops;
    const float rcp_tw = 1.0f / (float) tw;
    for (int i = tid; i < tw * th; i += POLAR_BW * POLAR_BH) {
        const int ty = (int) (((float) i + 0.5f) * rcp_tw), tx = i - ty * tw;  // exact: i < 2^22
        const int sx = plh_wrap(ox + tx, s.src.w, s.address_mode);
        const int sy = plh_wrap(oy + ty, s.src.h, s.address_mode);
        if (raw16) {
            *(uint2 *) &tile[i] = *(const uint2 *) ((const char *) s.src.ptr +
                                                    (size_t) sy * s.src.pitch + (size_t) sx * 8);
            continue;
        }
        float4_t c = plh_fetch(s.src, sx, sy);
        if (p.num_pre_ops) {
            const frag_t fc = { (float) sx + 0.5f, (float) sy + 0.5f };
            apply_ops(c, p.ops, 0, p.num_pre_ops, fc);
        }
        tile_put(tile[i], c);
    }
    __syncthreads();

    // ---- taps -----------------------------------------------------------------------
    const int idx = gx0 + threadIdx.x;
    const float mx = p.out_scale[0] * ((float) idx + 0.5f);

#pragma unroll 1
    for (int r = 0; r < rows; r++) {
        const int idy = gy0 + threadIdx.y + r * POLAR_BH;
        const float my = p.out_scale[1] * ((float) idy + 0.5f);
        const float px = plh_attr(s.pos, 0, mx, my);
        const float py = plh_attr(s.pos, 1, mx, my);

        const float tx = px * sw - 0.5f, ty = py * sh - 0.5f;
        const float flx = __builtin_floorf(tx), fly = __builtin_floorf(ty);
        const float fcx = tx - flx, fcy = ty - fly;         // fcoord
        // tile index of tap (0,0); lanes outside the image footprint (padding
        // lanes of edge groups) are clamped so their LDS reads stay in range
        int relx = (int) flx - ox, rely = (int) fly - oy;
        relx = min(max(relx, s.bound - 1), tw - s.bound - 1);
        rely = min(max(rely, s.bound - 1), th - s.bound - 1);
        const tile_px<T> *tp = tile + rely * tw + relx;

        float col[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float wsum = 0.0f;
        ar_state ars;
        if (USE_AR) {
            for (int c = 0; c < 4; c++)
                ars.ar[c][0] = ars.ar[c][1] = ars.wwsum[c][0] = ars.wwsum[c][1] = 0.0f;
        }

        for (int t = 0; t < s.num_taps; t++) {
            const uint32_t tap = s.taps[t];
            const int x = (int8_t) (tap & 0xff), y = (int8_t) ((tap >> 8) & 0xff);
            const uint32_t fl = tap >> 16;

            float d;
            bool live;
            const float w = polar_weight(s, lut, tap, fcx, fcy, d, live);
            wsum += w;

            const float4_t c = tile_get(tp[y * tw + x]);
            const float cv[4] = { c.x, c.y, c.z, c.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (MASK & (1u << k))
                    col[k] = __builtin_fmaf(w, cv[k], col[k]);
            }

            if (USE_AR) {
                if ((fl & PLH_TAP_AR) && live && d <= s.radius_zero) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (!(MASK & (1u << k)))
                            continue;
                        float cc[2] = { 1.0f - s.scale * cv[k], s.scale * cv[k] };
                        for (int j = 0; j < 2; j++) {
                            float ww = cc[j] + 0.10f;
                            ww = ww * ww; ww = ww * ww; ww = ww * ww; ww = ww * ww; ww = ww * ww;
                            ww = w * ww;
                            ars.ar[k][j] = __builtin_fmaf(ww, cc[j], ars.ar[k][j]);
                            ars.wwsum[k][j] += ww;
                        }
                    }
                }
            }
        }

        // color = scale / wsum * color                               sampling.c:897
        const float norm = s.scale / wsum;
        float4_t out;
        float *o[4] = { &out.x, &out.y, &out.z, &out.w };
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float v = norm * col[k];
            if (USE_AR && (MASK & (1u << k))) {
                float lo = ars.ar[k][0] / ars.wwsum[k][0];
                const float hi = ars.ar[k][1] / ars.wwsum[k][1];
                lo = 1.0f - lo;
                float w = fminf(fmaxf(v, lo), hi);
                w = lo > hi ? (lo * 0.5f + hi * 0.5f) : w;
                v = plh_mix(v, w, s.antiring);
            }
            *o[k] = v;
        }
        if (!(MASK & 8u))
            out.w = 1.0f;

        const frag_t fc = { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f,
                            0.0f, 0, p.out_scale[0] * ((float) idx + 0.5f),
                            p.out_scale[1] * ((float) idy + 0.5f) };
        apply_ops(out, p.ops, p.num_pre_ops, p.num_ops, fc);

        // guarded store (dispatch.c:1126-1142)
        const float gx = p.out_scale[0] * (float) idx, gy = p.out_scale[1] * (float) idy;
        if (gx < 1.0f && gy < 1.0f) {
            const int oxp = p.base_x + p.dir_x * (p.transpose ? idy : idx);
            const int oyp = p.base_y + p.dir_y * (p.transpose ? idx : idy);
            if (oxp >= 0 && oyp >= 0 && oxp < p.dst.w && oyp < p.dst.h)
                plh_store(p.dst, oxp, oyp, out);
        }
    }
}


/* ------------------------------------------------------------------------------------
 * Phase-class formulation (struct plh_polar_pp, plh_device.h)
 *
 * The per-pixel path above spends ~35 VALU instructions per tap on the weight
 * (IEEE sqrt, LUT address, lerp) and 5 on the accumulation, which makes it
 * ALU-bound at ~3 % of the HBM roofline. But the weight of a tap is a pure
 * function of fcoord, fcoord.x only depends on the output column and fcoord.y
 * on the row, and only a few dozen distinct fp32 values of each occur per
 * frame. So the weights are tabulated once per (class pair, tap) — by
 * k_polar_weights, with the *same* device arithmetic (polar_weight) — and the
 * frame kernel keeps only the FMAs: a lane owns an n x n block of output pixels
 * that share one base texel (n = 2 for a 2x upscale), reads each source texel of
 * the footprint once from LDS and feeds it to the n*n accumulators with weights
 * fetched from an LDS copy of the tile's slice of the table.
 *
 * Exactness: every lane still evaluates its own fcoord/base (polar_coord) and
 * compares them with the tables; a pixel that disagrees (a rounding tie in the
 * degenerate bilinear attribute interpolation) takes the per-pixel path
 * inline. Summation order, fma usage and the final `scale / wsum` are those of
 * k_polar, so results are bit-identical to it.
 */
__global__ void k_polar_classify(const plh_pass p_, float *out)
{
    const plh_pass &p = plh_kernarg_pass();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float *colfc = out;
    int *colbase = (int *) (out + p.width);
    float *rowfc = out + 2 * p.width;
    int *rowbase = (int *) (out + 2 * p.width + p.height);
    float fcx, fcy;
    int bx, by;
    if (i < p.width) {
        polar_coord(p, i, 0, fcx, fcy, bx, by);
        colfc[i] = fcx;
        colbase[i] = bx;
    }
    if (i < p.height) {
        polar_coord(p, 0, i, fcx, fcy, bx, by);
        rowfc[i] = fcy;
        rowbase[i] = by;
    }
}

__global__ void k_polar_weights(const plh_pass p_, const float *clsx, int ncx,
                                const float *clsy, int ncy, float *weights)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncx * ncy)
        return;
    const int cy = i / ncx, cx = i - cy * ncx;
    const float fcx = clsx[cx], fcy = clsy[cy];
    float *w = weights + (size_t) i * (s.num_taps + 1);
    float wsum = 0.0f;
    for (int t = 0; t < s.num_taps; t++) {
        float d;
        bool live;
        const float wt = polar_weight(s, (const float2 *) s.lut, s.taps[t], fcx, fcy, d, live);
        wsum += wt;
        w[t] = wt;
    }
    w[s.num_taps] = s.scale / wsum;     // color = scale / wsum * color, sampling.c:897
}

// Uniform read-only tables are read through the constant address space so that
// they become s_load_* (SGPR) instead of per-lane flat loads in the tap loop.
// k_polar_pp_f16.hip / k_polar_pp_f32.hip
int plh_launch_polar_pp_f16(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block,
                            size_t shmem, int n);
int plh_launch_polar_pp_f32(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block,
                            size_t shmem, int n);
int plh_launch_polar_pp_f16_c12(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block,
                                size_t shmem, int n);
int plh_launch_polar_pp_f32_c12(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block,
                                size_t shmem, int n);
// k_polar_mx.hip
int plh_launch_polar_mx(hipStream_t stream, const plh_pass *pass);
// k_polar_mxp.hip: the same on persistent workgroups, for the commonest shapes
bool plh_polar_mxp_applies(const plh_pass *pass);
int plh_launch_polar_mxp(hipStream_t stream, const plh_pass *pass);
bool plh_polar_mxd_applies(plh_pass *pass);
int plh_launch_polar_mxd(hipStream_t stream, const plh_pass *pass);
bool plh_polar_mxr_applies(plh_pass *pass);
int plh_launch_polar_mxr(hipStream_t stream, const plh_pass *pass);

int plh_launch_polar_classify(plh_stream stream, const plh_pass *pass, void *out)
{
    const int n = pass->width > pass->height ? pass->width : pass->height;
    hipLaunchKernelGGL(k_polar_classify, dim3((n + 255) / 256), dim3(256), 0,
                       (hipStream_t) stream, *pass, (float *) out);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}

int plh_launch_polar_weights(plh_stream stream, const plh_pass *pass, const float *clsx, int ncx,
                             const float *clsy, int ncy, float *weights)
{
    hipLaunchKernelGGL(k_polar_weights, dim3((ncx * ncy + 63) / 64), dim3(64), 0,
                       (hipStream_t) stream, *pass, clsx, ncx, clsy, ncy, weights);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}

template <typename T, uint32_t MASK>
static int launch_ar(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block, size_t shmem)
{
    if (pass->s.antiring > 0.0f)
        PLH_LAUNCH_LAST((k_polar<T, MASK, true>), grid, block, shmem, stream, *pass);
    else
        PLH_LAUNCH_LAST((k_polar<T, MASK, false>), grid, block, shmem, stream, *pass);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}

template <typename T>
static int launch_mask(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block, size_t shmem)
{
    switch (pass->s.comp_mask & 0xf) {
    case 0x1: return launch_ar<T, 0x1>(stream, pass, grid, block, shmem);
    case 0x3: return launch_ar<T, 0x3>(stream, pass, grid, block, shmem);
    case 0x7: return launch_ar<T, 0x7>(stream, pass, grid, block, shmem);
    default:  return launch_ar<T, 0xf>(stream, pass, grid, block, shmem);
    }
}

int plh_launch_polar(hipStream_t stream, const plh_pass *pass_in)
{
    plh_pass local = *pass_in;
    const plh_pass *pass = &local;
    const dim3 block(POLAR_BW, POLAR_BH);
    const uint32_t cm = pass->s.comp_mask & 0xf;
    if (pass->s.pp && plh_polar_mxd_applies(&local))
        return plh_launch_polar_mxd(stream, pass);     // the 2 : 1 downscale on the matrix pipe
    if (pass->s.pp && pass->s.mx.enabled == 3) {
        // an integer upscale by 3 or 4 on the matrix pipe, where the pass has the kernel's shape
        plh_pass probe = local;
        if (plh_polar_mxr_applies(&probe))
            return plh_launch_polar_mxr(stream, &probe);
    }
    if (pass->s.pp && pass->s.mx.enabled == 1 && (cm == 0x7 || cm == 0xf)) {
        // (the matrix-pipe kernel has a variant for the map chain of an HDR pass)
        if (cm == 0x7)
            plh_match_map_chain(&local, true);
        if (!local.chain.enabled)
            plh_match_fast_epilogue(&local);
        if (plh_polar_mxp_applies(pass))
            return plh_launch_polar_mxp(stream, pass);
        return plh_launch_polar_mx(stream, pass);
    }
    // the phase-class kernels have a CHAIN variant for RGB / RGBA tiles (k_polar_pp.hiph)
    if (pass->s.pp && (cm == 0x7 || cm == 0xf))
        plh_match_map_chain(&local);
    if (!local.chain.enabled)
        plh_match_fast_epilogue(&local);
    if (pass->s.pp && (cm == 0x7 || cm == 0xf || cm == 0x1 || cm == 0x3)) {
        const int n = pass->s.pp_n, cw = pass->s.pp_cells_w, ch = pass->s.pp_cells_h;
        const int cth = POLAR_BH * pass->s.tile_rows;
        const dim3 grid((cw + POLAR_BW - 1) / POLAR_BW, (ch + cth - 1) / cth);
        const size_t px = pass->s.tile_fp32 ? sizeof(float4) : sizeof(uint2);
        const size_t shmem = PP_LDS_FIXED + pass->s.pp_lds_weights +
                             (size_t) pass->s.tile_w * pass->s.tile_h * px;
        if (shmem > 160 * 1024)
            return -1000;
        const bool planes = cm == 0x1 || cm == 0x3;
        if (pass->s.tile_fp32)
            return planes ? plh_launch_polar_pp_f32_c12(stream, pass, grid, block, shmem, n)
                          : plh_launch_polar_pp_f32(stream, pass, grid, block, shmem, n);
        return planes ? plh_launch_polar_pp_f16_c12(stream, pass, grid, block, shmem, n)
                      : plh_launch_polar_pp_f16(stream, pass, grid, block, shmem, n);
    }
    const int th = POLAR_BH * pass->s.tile_rows;
    const dim3 grid((pass->width + POLAR_BW - 1) / POLAR_BW, (pass->height + th - 1) / th);
    const size_t px = pass->s.tile_fp32 ? sizeof(float4) : sizeof(uint2);
    const size_t shmem = 256 * sizeof(float2) + (size_t) pass->s.tile_w * pass->s.tile_h * px;
    if (shmem > 160 * 1024)
        return -1000; // host picks tile sizes that fit; see shader_sampling.c
    if (pass->s.tile_fp32)
        return launch_mask<float>(stream, pass, grid, block, shmem);
    return launch_mask<__half>(stream, pass, grid, block, shmem);
}
