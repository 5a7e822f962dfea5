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
#include "colorops.hiph"
#include "fastepi.hiph"

#ifndef POLAR_BW
#define POLAR_BW 32
#define POLAR_BH 8
#endif

template <typename T> struct tile_px;
template <> struct tile_px<__half> { uint2 v; };    // 4 x f16
template <> struct tile_px<float>  { float4 v; };   // 4 x f32

DEV void tile_put(tile_px<__half> &t, const float4_t &c)
{
    t.v.x = (uint32_t) plh_f2h(c.x) | ((uint32_t) plh_f2h(c.y) << 16);
    t.v.y = (uint32_t) plh_f2h(c.z) | ((uint32_t) plh_f2h(c.w) << 16);
}

DEV void tile_put(tile_px<float> &t, const float4_t &c)
{
    t.v = make_float4(c.x, c.y, c.z, c.w);
}

DEV float4_t tile_get(const tile_px<__half> &t)
{
    const uint2 v = t.v;
    float4_t c = { plh_h2f(v.x & 0xffff), plh_h2f(v.x >> 16),
                   plh_h2f(v.y & 0xffff), plh_h2f(v.y >> 16) };
    return c;
}

DEV float4_t tile_get(const tile_px<float> &t)
{
    const float4 v = t.v;
    float4_t c = { v.x, v.y, v.z, v.w };
    return c;
}


// fcoord / base texel of output pixel (idx, idy) — the one place this is spelled
DEV void polar_coord(const plh_pass &p, int idx, int idy, float &fcx, float &fcy,
                     int &bx, int &by)
{
    const plh_sampler_args &s = p.s;
    const float mx = p.out_scale[0] * ((float) idx + 0.5f);
    const float my = p.out_scale[1] * ((float) idy + 0.5f);
    const float px = plh_attr(s.pos, 0, mx, my);
    const float py = plh_attr(s.pos, 1, mx, my);
    const float tx = px * (float) s.src.w - 0.5f, ty = py * (float) s.src.h - 0.5f;
    const float flx = __builtin_floorf(tx), fly = __builtin_floorf(ty);
    fcx = tx - flx;
    fcy = ty - fly;
    bx = (int) flx;
    by = (int) fly;
}

// weight of one tap for a given fcoord (0 when the tap is skipped)
template <typename LUT>
DEV float polar_weight(const plh_sampler_args &s, const LUT lut, uint32_t tap,
                       float fcx, float fcy, float &d, bool &live)
{
    const int x = (int8_t) (tap & 0xff), y = (int8_t) ((tap >> 8) & 0xff);
    const uint32_t fl = tap >> 16;
    const float dx = (float) x - fcx, dy = (float) y - fcy;
    d = __builtin_sqrtf(dx * dx + dy * dy);                     // length()
    live = !(fl & PLH_TAP_SKIPPABLE) || d < s.radius;
    // w = lut(d / R): linear LUT lookup, lut.c:700-715 semantics
    const float fpos = plh_clamp(d * s.rcp_radius, 0.0f, 1.0f) * 255.0f;
    const float fbase = __builtin_floorf(fpos);
    const float2 l = lut[(int) fbase];
    const float w = plh_mix(l.x, l.y, fpos - fbase);
    return live ? w : 0.0f;     // adding zeros == skipping the tap
}

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

        const frag_t fc = { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f };
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
#define PLH_CONST(T, ptr) ((const T __attribute__((address_space(4))) *) (uintptr_t) (ptr))

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float floatv4_t __attribute__((ext_vector_type(4)));

DEV floatv4_t tile_vec(const tile_px<__half> &t)
{
    const half4_t h = *(const half4_t *) &t;
    // written as conversions feeding fmas so that they fold into v_fma_mix_f32
    return __builtin_convertvector(h, floatv4_t);
}

DEV floatv4_t tile_vec(const tile_px<float> &t)
{
    return *(const floatv4_t *) &t;
}

// the per-pixel weights path for one pixel, on the staged tile (no anti-ringing)
template <typename T, uint32_t MASK>
DEV void polar_pixel_generic(const plh_sampler_args &s, const float2 *lut, const tile_px<T> *tp,
                             int tw, float fcx, float fcy, float col[4], float &norm)
{
    float wsum = 0.0f;
    col[0] = col[1] = col[2] = col[3] = 0.0f;
    for (int t = 0; t < s.num_taps; t++) {
        const uint32_t tap = s.taps[t];
        const int x = (int8_t) (tap & 0xff), y = (int8_t) ((tap >> 8) & 0xff);
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
    }
    norm = s.scale / wsum;
}

// per output row of the workgroup, staged in LDS: class value, base texel, offset of the
// row class in the weight sub-table
struct pp_rowinfo { float fc; int32_t base; int32_t woff; int32_t pad; };
#define PP_LDS_FIXED (2048 + 1024 + 64)   // lut pairs, row info, class lists

// LITE: the recorded ops only use the cheap cases (plh_ops_lite). FAST (implies LITE): the
// post-ops are the fused epilogue described by p.epi and the target is rgba16.
// profiling switches (PL_HIP_PP_DEBUG bits: 1 no taps, 2 no verification, 4 no stores, 8 no tile
// staging, 16 no weight staging, 32 no epilogue, 64 no rows, 128 no weight reads, 256 one texel) only exist in -DPLH_PP_DEBUG builds:
// each one is a scalar load + branch inside the row loop otherwise
#ifdef PLH_PP_DEBUG
#define PP_DBG(bit) (s.pp_debug & (bit))
#else
#define PP_DBG(bit) false
#endif

// (two 4-tap steps in flight: 8K -> 4K downscale 334 -> 311 us, no change for the 2x upscale)
#if defined(PP_TAP_UNROLL_N) && PP_TAP_UNROLL_N == 1
#define PP_TAP_UNROLL _Pragma("unroll 1")
#else
#define PP_TAP_UNROLL _Pragma("unroll 2")
#endif

// texels per lane and staging batch (one memory round trip per batch; a 40x32 tile is 5 per lane)
#ifndef PP_SB
#define PP_SB 6       // decode + pre-ops path (measured: 4 -> 61.8 us, 6 -> 60.6, 8 -> 68)
#endif
#ifndef PP_SB_RAW
#define PP_SB_RAW 8   // bit-copy path (8K -> 4K: 6 -> 286 us, 8 -> 279)
#endif

#ifdef PLH_PP_WAVES6
#define PP_WAVES __attribute__((amdgpu_waves_per_eu(6, 8)))
#else
#define PP_WAVES
#endif

template <typename T, uint32_t MASK, int N, bool LITE, bool FAST>
__global__ __launch_bounds__(POLAR_BW * POLAR_BH) PP_WAVES
void k_polar_pp(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const plh_polar_pp &pp = s.ppv;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float2 *lut = (const float2 *) s.lut;                 // 256 pairs (global)
    pp_rowinfo *rinfo = (pp_rowinfo *) (smem + 2048);           // <= 64 output rows, 1 KiB
    float *ws = (float *) (smem + PP_LDS_FIXED);                // weight sub-table
    tile_px<T> *tile = (tile_px<T> *) (smem + PP_LDS_FIXED + s.pp_lds_weights);

    const int tid = threadIdx.y * POLAR_BW + threadIdx.x;
    const int rows = s.tile_rows;
    const int tw = s.tile_w, th = s.tile_h;
    const int ox = pp.colorg[blockIdx.x], oy = pp.roworg[blockIdx.y];
    const int nx = pp.coln[blockIdx.x], ny = pp.rown[blockIdx.y];
    const int tp = pp.tp, ntaps = pp.ntaps;
    // (the fused epilogue is only matched for untransposed passes)
    const bool tr = FAST ? false : (bool) p.transpose;

    // ---- stage LUT pairs, the tile's slice of the weight table, the source tile ---------
    // Every staging step is written as "issue a batch of independent loads, then store":
    // a load -> wait -> store loop pays one memory round trip (~1 us) per iteration.
    // (the LUT is only read by the rare per-pixel fixups below: straight from global memory)
    // The tile's slice of the weight table (float4 units, 4 per lane in flight) is requested
    // first: its address chain (class lists -> weights) then overlaps the tile loads below.
    // (four named registers rather than an array: the compiler demotes a conditionally
    // consumed array to scratch memory, ~27 MB of spurious HBM writes per 4K frame)
    const int tp4 = tp >> 2, units = nx * ny * tp4;
    const int wstride = POLAR_BW * POLAR_BH;
    float4 wv0, wv1, wv2, wv3;
    {
        const float rcp_tp4 = 1.0f / (float) tp4, rcp_nx = 1.0f / (float) nx;
        const uint16_t *cl = pp.collist + blockIdx.x * PLH_PP_LMAX;
        const uint16_t *rl = pp.rowlist + blockIdx.y * PLH_PP_LMAX;
        auto wload = [&](int u) {
            u = min(u, units - 1);
            const int pair = (int) (((float) u + 0.5f) * rcp_tp4), t4 = u - pair * tp4;
            const int ly = (int) (((float) pair + 0.5f) * rcp_nx), lx = pair - ly * nx;
            const size_t g = (size_t) rl[ly] * pp.ncx + cl[lx];
            return *(const float4 *) (pp.weights + g * tp + t4 * 4);
        };
        wv0 = wload(tid); wv1 = wload(tid + wstride);
        wv2 = wload(tid + 2 * wstride); wv3 = wload(tid + 3 * wstride);
    }
    {
        int32_t *toff = (int32_t *) (ws + (s.pp_lds_weights >> 2)) - ((ntaps + 3) & ~3);
        for (int t = tid; t < ntaps; t += POLAR_BW * POLAR_BH)
            toff[t] = pp.tapoff[t];     // tail of the weights area (reserved by the host)
        const int y0 = N * (blockIdx.y * rows * POLAR_BH) - pp.pady;
        for (int j = tid; j < N * rows * POLAR_BH; j += POLAR_BW * POLAR_BH) {
            const int yc = min(max(y0 + j, 0), p.height - 1);
            pp_rowinfo ri = { pp.rowfc[yc], pp.rowbase[yc], pp.rowloc[yc] * nx * tp, 0 };
            rinfo[j] = ri;
        }
    }
    // an rgba16hf source without pre-ops is copied bit for bit (the f16 -> f32 -> f16 round
    // trip of the generic path is the identity)
    const bool raw16 = sizeof(tile_px<T>) == 8 && s.src.fmt == PLH_FMT_RGBA16F && !p.num_pre_ops;
    const float rcp_tw = 1.0f / (float) tw;
    if (PP_DBG(8)) {
    } else if (raw16) {
        // batches of PP_SB_RAW independent 8-byte loads per lane, so that a tile costs one memory
        // round trip instead of one per texel
        for (int i0 = tid; i0 < tw * th; i0 += PP_SB_RAW * POLAR_BW * POLAR_BH) {
            uint2 v[PP_SB_RAW];
#pragma unroll
            for (int u = 0; u < PP_SB_RAW; u++) {
                const int i = min(i0 + u * POLAR_BW * POLAR_BH, tw * th - 1);
                const int ty = (int) (((float) i + 0.5f) * rcp_tw), tx = i - ty * tw;
                const int sx = plh_wrap(ox + tx, s.src.w, s.address_mode);
                const int sy = plh_wrap(oy + ty, s.src.h, s.address_mode);
                v[u] = *(const uint2 *) ((const char *) s.src.ptr + (size_t) sy * s.src.pitch +
                                         (size_t) sx * 8);
            }
#pragma unroll
            for (int u = 0; u < PP_SB_RAW; u++) {
                const int i = i0 + u * POLAR_BW * POLAR_BH;
                if (i < tw * th)
                    *(uint2 *) &tile[i] = v[u];
            }
        }
    } else if (s.src.fmt == PLH_FMT_RGBA16) {
        // packed unorm16 source (the fused-PASS-A case): same batching, decode + pre-ops after
        for (int i0 = tid; i0 < tw * th; i0 += PP_SB * POLAR_BW * POLAR_BH) {
            uint2 v[PP_SB];
            int px[PP_SB], py[PP_SB];
#pragma unroll
            for (int u = 0; u < PP_SB; u++) {
                const int i = min(i0 + u * POLAR_BW * POLAR_BH, tw * th - 1);
                const int ty = (int) (((float) i + 0.5f) * rcp_tw), tx = i - ty * tw;
                px[u] = plh_wrap(ox + tx, s.src.w, s.address_mode);
                py[u] = plh_wrap(oy + ty, s.src.h, s.address_mode);
                v[u] = *(const uint2 *) ((const char *) s.src.ptr + (size_t) py[u] * s.src.pitch +
                                         (size_t) px[u] * 8);
            }
            float4_t c[PP_SB];
            frag_t fcs[PP_SB];
#pragma unroll
            for (int u = 0; u < PP_SB; u++) {
                c[u] = { plh_un16(v[u].x & 0xffff), plh_un16(v[u].x >> 16),
                         plh_un16(v[u].y & 0xffff), plh_un16(v[u].y >> 16) };
                fcs[u] = { (float) px[u] + 0.5f, (float) py[u] + 0.5f, 0.0f, 0 };
            }
            if (p.num_pre_ops)
                apply_ops_n<PP_SB, false, LITE>(c, p.ops, 0, p.num_pre_ops, fcs);
#pragma unroll
            for (int u = 0; u < PP_SB; u++) {
                const int i = i0 + u * POLAR_BW * POLAR_BH;
                if (i < tw * th)
                    tile_put(tile[i], c[u]);
            }
        }
    } else {
        for (int i = tid; i < tw * th; i += POLAR_BW * POLAR_BH) {
            const int ty = (int) (((float) i + 0.5f) * rcp_tw), tx = i - ty * tw;  // exact: i < 2^22
            const int sx = plh_wrap(ox + tx, s.src.w, s.address_mode);
            const int sy = plh_wrap(oy + ty, s.src.h, s.address_mode);
            float4_t c = plh_fetch(s.src, sx, sy);
            if (p.num_pre_ops) {
                // fused "PASS A": the ops the reference runs in a separate pass before the scaler
                const frag_t fc = { (float) sx + 0.5f, (float) sy + 0.5f };
                apply_ops<false, LITE>(c, p.ops, 0, p.num_pre_ops, fc);
            }
            tile_put(tile[i], c);
        }
    }
    {
        // weight slice: the first four units of every lane were requested above
        if (!PP_DBG(16)) {
            if (tid < units)
                *(float4 *) (ws + tid * 4) = wv0;
            if (tid + wstride < units)
                *(float4 *) (ws + (tid + wstride) * 4) = wv1;
            if (tid + 2 * wstride < units)
                *(float4 *) (ws + (tid + 2 * wstride) * 4) = wv2;
            if (tid + 3 * wstride < units)
                *(float4 *) (ws + (tid + 3 * wstride) * 4) = wv3;
            // (slices beyond 4 units per lane: many classes per tile, irrational ratios)
            const float rcp_tp4 = 1.0f / (float) tp4, rcp_nx = 1.0f / (float) nx;
            for (int u = tid + 4 * wstride; u < units; u += wstride) {
                const int pair = (int) (((float) u + 0.5f) * rcp_tp4), t4 = u - pair * tp4;
                const int ly = (int) (((float) pair + 0.5f) * rcp_nx), lx = pair - ly * nx;
                const size_t g = (size_t) pp.rowlist[blockIdx.y * PLH_PP_LMAX + ly] * pp.ncx +
                                 pp.collist[blockIdx.x * PLH_PP_LMAX + lx];
                *(float4 *) (ws + u * 4) = *(const float4 *) (pp.weights + g * tp + t4 * 4);
            }
        }
    }
    __syncthreads();

    int dither_op = -1;     // uniform
    for (int i = p.num_pre_ops; i < p.num_ops; i++) {
        if (p.ops[i].kind == PLH_OP_DITHER)
            dither_op = i;
    }

    // ---- per-lane column state ------------------------------------------------------------
    const int cellx = blockIdx.x * POLAR_BW + threadIdx.x;
    int colx[N];            // output columns of this lane (may lie outside the image)
    int cwoff[N];           // offset of the column's class in the weight sub-table
    float cfc[N];
    float attr[N][4];       // the fx halves of the attribute interpolation (plh_attr)
    float refx[N];          // pos.x of the column as the tables saw it (row 0)
    bool cgood[N];          // refx reproduces the column's tabulated fcoord/base, and the
                            // column's fy halves are those the row tables were built with
    bool cok[N];            // column passes the store guards
    int cpos[N];            // target coordinate contributed by the column
    float fragx[N];
    int cbase;
    const float sw = (float) s.src.w, sh = (float) s.src.h;
    // attribute halves of column 0 (the row tables were evaluated there) and fy of row 0
    const float mx0 = p.out_scale[0] * 0.5f, my0 = p.out_scale[1] * 0.5f;
    const float y0a = plh_mix(s.pos[0][1], s.pos[1][1], mx0);
    const float y0b = plh_mix(s.pos[2][1], s.pos[3][1], mx0);
    {
        int b = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            colx[i] = N * cellx - pp.padx + i;
            const float mx = p.out_scale[0] * ((float) colx[i] + 0.5f);
            attr[i][0] = plh_mix(s.pos[0][0], s.pos[1][0], mx);
            attr[i][1] = plh_mix(s.pos[2][0], s.pos[3][0], mx);
            attr[i][2] = plh_mix(s.pos[0][1], s.pos[1][1], mx);
            attr[i][3] = plh_mix(s.pos[2][1], s.pos[3][1], mx);
            const int xc = min(max(colx[i], 0), p.width - 1);
            cwoff[i] = pp.colloc[xc] * tp;
            cfc[i] = pp.colfc[xc];
            // the lane's base texel: that of its first in-range column
            if (i == 0 || colx[i - 1] < 0)
                b = pp.colbase[xc];
        }
        cbase = b;
#pragma unroll
        for (int i = 0; i < N; i++) {
            refx[i] = plh_mix(attr[i][0], attr[i][1], my0);
            const float tx_ = refx[i] * sw - 0.5f, flx = __builtin_floorf(tx_);
            cgood[i] = __float_as_uint(tx_ - flx) == __float_as_uint(cfc[i]) &&
                       (int) flx == cbase &&
                       __float_as_uint(attr[i][2]) == __float_as_uint(y0a) &&
                       __float_as_uint(attr[i][3]) == __float_as_uint(y0b);
            const int idx = colx[i];
            cpos[i] = tr ? p.base_y + p.dir_y * idx : p.base_x + p.dir_x * idx;
            cok[i] = idx >= 0 && idx < p.width && p.out_scale[0] * (float) idx < 1.0f &&
                     cpos[i] >= 0 && cpos[i] < (tr ? p.dst.h : p.dst.w) &&
                     !(PP_DBG(4));
            fragx[i] = (float) (idx + p.frag_x0) + 0.5f;
        }
    }

#pragma unroll 1
    for (int r = 0; r < ((PP_DBG(64)) ? 0 : rows); r++) {
        const int celly = (blockIdx.y * rows + r) * POLAR_BH + threadIdx.y;
        int rowy[N], rwoff[N];
        float rfc[N];
        int rbase = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            rowy[j] = N * celly - pp.pady + j;
            const pp_rowinfo ri = rinfo[N * (r * POLAR_BH + threadIdx.y) + j];
            rwoff[j] = ri.woff;
            rfc[j] = ri.fc;
            if (j == 0 || rowy[j - 1] < 0)
                rbase = ri.base;
        }

        // fetch the dither values now; they are consumed after the tap loop
        float bias[N][N];
#pragma unroll
        for (int j = 0; j < N; j++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                if constexpr (FAST) {
                    const int ix = (colx[i] + p.frag_x0) & p.epi.mask;
                    const int iy = (rowy[j] + p.frag_y0) & p.epi.mask;
                    bias[j][i] = p.epi.has_dither ? p.epi.matrix[iy * p.epi.size + ix] : 0.0f;
                } else {
                    const frag_t fc = { (float) (colx[i] + p.frag_x0) + 0.5f,
                                        (float) (rowy[j] + p.frag_y0) + 0.5f };
                    bias[j][i] = dither_op >= 0 ? dither_bias(p.ops[dither_op], fc) : 0.0f;
                }
            }
        }

        // lanes of padding cells are clamped so their LDS reads stay inside the tile
        const int relx = min(max(cbase - ox, s.bound - 1), tw - s.bound - 1);
        const int rely = min(max(rbase - oy, s.bound - 1), th - s.bound - 1);
        const tile_px<T> *tp0 = tile + rely * tw + relx;

        floatv4_t acc[N][N];
        const float *wp[N][N];
#pragma unroll
        for (int j = 0; j < N; j++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                acc[j][i] = (floatv4_t) (0.0f);
                wp[j][i] = ws + rwoff[j] + cwoff[i];
            }
        }

        const int32_t *tapoff = (const int32_t *) (ws + (s.pp_lds_weights >> 2)) -
                                ((ntaps + 3) & ~3);
        const int nt_run = (PP_DBG(1)) ? 0 : ntaps;
        auto tap = [&](int off, const float (&w)[N][N]) {
            // byte offset of the tap inside the tile (host: (y * tile_w + x) * sizeof(texel))
            const floatv4_t c = tile_vec(*(const tile_px<T> *) ((const char *) tp0 + off));
#pragma unroll
            for (int j = 0; j < N; j++) {
#pragma unroll
                for (int i = 0; i < N; i++) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        if (MASK & (1u << k))
                            acc[j][i][k] = __builtin_fmaf(w[j][i], c[k], acc[j][i][k]);
                    }
                }
            }
        };
        // four taps per step: one 16-byte LDS read per cell for the weights (rows of the weight
        // table are 16-byte aligned) and one broadcast read for the tap offsets
        int t = 0;
PP_TAP_UNROLL
        for (; t + 4 <= nt_run; t += 4) {
            int4 off = *(const int4 *) (tapoff + t);
            if (PP_DBG(256))    // (profiling: every tap reads the same texel)
                off = make_int4(0, 0, 0, 0);
            float4 w4[N][N];
#pragma unroll
            for (int j = 0; j < N; j++) {
#pragma unroll
                for (int i = 0; i < N; i++) {
                    // (profiling bit 128: weights from registers instead of LDS)
                    w4[j][i] = PP_DBG(128) ? make_float4(cfc[i], rfc[j], cfc[i], rfc[j])
                                           : *(const float4 *) (wp[j][i] + t);
                }
            }
            float w[N][N];
#define PP_TAP(o, m) \
            _Pragma("unroll") for (int j = 0; j < N; j++) \
                _Pragma("unroll") for (int i = 0; i < N; i++) w[j][i] = w4[j][i].m; \
            tap(o, w)
            PP_TAP(off.x, x); PP_TAP(off.y, y); PP_TAP(off.z, z); PP_TAP(off.w, w);
#undef PP_TAP
        }
        for (; t < nt_run; t++) {
            float w[N][N];
#pragma unroll
            for (int j = 0; j < N; j++) {
#pragma unroll
                for (int i = 0; i < N; i++)
                    w[j][i] = wp[j][i][t];
            }
            tap(tapoff[t], w);
        }

        // ---- normalise, verify, post-ops, store ------------------------------------------
        if (PP_DBG(32))
            continue;
        float4_t outs[N * N];
        frag_t fcs[N * N];
        int sx[N * N], sy[N * N];
        bool ok[N * N];
        uint32_t redo = 0;
#pragma unroll
        for (int j = 0; j < N; j++) {
            const int idy = rowy[j];
            const float my = p.out_scale[1] * ((float) idy + 0.5f);
            // the row as the tables saw it (column 0): pos.y -> fcoord.y / base
            const float ty_ = plh_mix(y0a, y0b, my) * sh - 0.5f, fly = __builtin_floorf(ty_);
            const bool rgood = __float_as_uint(ty_ - fly) == __float_as_uint(rfc[j]) &&
                               (int) fly == rbase;
            const int rpos = tr ? p.base_x + p.dir_x * idy : p.base_y + p.dir_y * idy;
            const bool rok = idy >= 0 && idy < p.height && p.out_scale[1] * (float) idy < 1.0f &&
                             rpos >= 0 && rpos < (tr ? p.dst.w : p.dst.h);
            const float fragy = (float) (idy + p.frag_y0) + 0.5f;
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int q = j * N + i;
                float col[4] = { acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3] };
                float norm = wp[j][i][ntaps];

                // Is this pixel's own fcoord/base the tabulated one? pos.x must equal the
                // column's reference (the fy interpolation of two equal halves is the identity
                // except for rounding ties); pos.y equals the row's reference by construction
                // when the column's fy halves are those of column 0 (cgood).
                const float px = plh_mix(attr[i][0], attr[i][1], my);
                const bool same = __float_as_uint(px) == __float_as_uint(refx[i]) && cgood[i] &&
                                  rgood;
                // no -> recomputed with per-pixel weights after the regular stores (rare: a
                // rounding tie in the attribute interpolation)
                if (cok[i] && rok && !same && !(PP_DBG(2)))
                    redo |= 1u << q;

                outs[q] = { norm * col[0], norm * col[1], norm * col[2], norm * col[3] };
                if (!(MASK & 8u))
                    outs[q].w = 1.0f;
                if constexpr (FAST) {
                    // op_dither (non-gamma path) and the SCALE op, parameters in SGPRs
                    float4_t &o = outs[q];
                    if (p.epi.has_dither) {
                        const float b = bias[j][i], ds = p.epi.dscale, di = p.epi.dinv;
                        o.x = __builtin_floorf(ds * o.x + b) * di;
                        o.y = __builtin_floorf(ds * o.y + b) * di;
                        o.z = __builtin_floorf(ds * o.z + b) * di;
                        // alpha: when it is not sampled it is 1.0, and floor(ds * 1 + b) == ds for
                        // every bias in [0, 1) (ds = 2^depth - 1 is an integer): no per-pixel work
                        o.w = (MASK & 8u) ? __builtin_floorf(ds * o.w + b) * di : ds * di;
                    }
                    if (p.epi.has_scale) {
                        o.x *= p.epi.scale; o.y *= p.epi.scale; o.z *= p.epi.scale; o.w *= p.epi.scale;
                    }
                }
                fcs[q] = { fragx[i], fragy, bias[j][i], dither_op >= 0 };
                // guarded store (dispatch.c:1126-1142), guards hoisted per column / row
                sx[q] = tr ? rpos : cpos[i];
                sy[q] = tr ? cpos[i] : rpos;
                ok[q] = cok[i] && rok;
            }
        }
        if constexpr (FAST && (N * N) % 2 == 0) {
            plh_store_rgba16_n<N * N>(p.dst, sx, sy, ok, outs, p.nt_store);
        } else if constexpr (FAST) {
            plh_store_n<N * N>(p.dst, sx, sy, ok, outs, p.nt_store);
        } else {
            apply_ops_n<N * N, false, LITE>(outs, p.ops, p.num_pre_ops, p.num_ops, fcs);
            plh_store_n<N * N>(p.dst, sx, sy, ok, outs, p.nt_store);
        }

        // ---- pixels whose own phase is not the tabulated one: the per-pixel path, one inlined
        // copy for all of the lane's pixels (a function call would cost scratch traffic) ------
        while (redo) {
            const int q = __builtin_ctz(redo);
            redo &= redo - 1;
            const int i = q % N, j = q / N;
            int idx = colx[0], idy = rowy[0];
#pragma unroll
            for (int k = 1; k < N; k++) {
                idx = i == k ? colx[k] : idx;
                idy = j == k ? rowy[k] : idy;
            }
            float fcx, fcy, col[4], norm;
            int bx, by;
            polar_coord(p, idx, idy, fcx, fcy, bx, by);
            const int rx = min(max(bx - ox, s.bound - 1), tw - s.bound - 1);
            const int ry = min(max(by - oy, s.bound - 1), th - s.bound - 1);
            polar_pixel_generic<T, MASK>(s, lut, tile + ry * tw + rx, tw, fcx, fcy, col, norm);
            float4_t o1[1] = { { norm * col[0], norm * col[1], norm * col[2], norm * col[3] } };
            if (!(MASK & 8u))
                o1[0].w = 1.0f;
            const frag_t f1[1] = { { (float) (idx + p.frag_x0) + 0.5f,
                                     (float) (idy + p.frag_y0) + 0.5f, 0.0f, 0 } };
            apply_ops_n<1, false, LITE>(o1, p.ops, p.num_pre_ops, p.num_ops, f1);
            const int x1[1] = { p.base_x + p.dir_x * (tr ? idy : idx) };
            const int y1[1] = { p.base_y + p.dir_y * (tr ? idx : idy) };
            const bool k1[1] = { true };    // (only pixels that passed the store guards get here)
            plh_store_n<1>(p.dst, x1, y1, k1, o1);
        }
    }
}

template <typename T, uint32_t MASK>
static int launch_pp(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block, size_t shmem,
                     int n)
{
    const bool lite = plh_ops_lite(pass, 0, pass->num_ops);
    const bool fast = lite && pass->epi.enabled;
#define LAUNCH(N, L, F) \
    hipLaunchKernelGGL((k_polar_pp<T, MASK, N, L, F>), grid, block, shmem, stream, *pass)
    if (n == 2) {
        if (fast)      LAUNCH(2, true, true);
        else if (lite) LAUNCH(2, true, false);
        else           LAUNCH(2, false, false);
    } else {
        if (fast)      LAUNCH(1, true, true);
        else if (lite) LAUNCH(1, true, false);
        else           LAUNCH(1, false, false);
    }
#undef LAUNCH
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}

template <typename T>
static int launch_pp_mask(hipStream_t stream, const plh_pass *pass, dim3 grid, dim3 block,
                          size_t shmem, int n)
{
    // (the host only builds phase classes for 3- and 4-component passes; 1- and 2-component
    // planes use the per-pixel kernel, which keeps the number of variants down)
    if ((pass->s.comp_mask & 0xf) == 0x7)
        return launch_pp<T, 0x7>(stream, pass, grid, block, shmem, n);
    if ((pass->s.comp_mask & 0xf) == 0xf)
        return launch_pp<T, 0xf>(stream, pass, grid, block, shmem, n);
    return -1001;
}

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
        hipLaunchKernelGGL((k_polar<T, MASK, true>), grid, block, shmem, stream, *pass);
    else
        hipLaunchKernelGGL((k_polar<T, MASK, false>), grid, block, shmem, stream, *pass);
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
    plh_match_fast_epilogue(&local);
    const plh_pass *pass = &local;
    const dim3 block(POLAR_BW, POLAR_BH);
    const uint32_t cm = pass->s.comp_mask & 0xf;
    if (pass->s.pp && (cm == 0x7 || cm == 0xf)) {
        const int n = pass->s.pp_n, cw = pass->s.pp_cells_w, ch = pass->s.pp_cells_h;
        const int cth = POLAR_BH * pass->s.tile_rows;
        const dim3 grid((cw + POLAR_BW - 1) / POLAR_BW, (ch + cth - 1) / cth);
        const size_t px = pass->s.tile_fp32 ? sizeof(float4) : sizeof(uint2);
        const size_t shmem = PP_LDS_FIXED + pass->s.pp_lds_weights +
                             (size_t) pass->s.tile_w * pass->s.tile_h * px;
        if (shmem > 160 * 1024)
            return -1000;
        if (pass->s.tile_fp32)
            return launch_pp_mask<float>(stream, pass, grid, block, shmem, n);
        return launch_pp_mask<__half>(stream, pass, grid, block, shmem, n);
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
