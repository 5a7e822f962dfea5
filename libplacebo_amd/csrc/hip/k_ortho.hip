/*
 * libplacebo-hip — separable (orthogonal) resampling kernel (K4).
 *
 * Device half of pl_shader_sample_ortho2 (src/shaders/sampling.c:950-1104): one
 * 1-D convolution along `dir` with N = row_size taps,
 *
 *   fcoord = fract(pos*size - 0.5)[dir];  first tap = floor(pos*size - 0.5)[dir] - (N/2 - 1)
 *   ws     = LUT row at fcoord (256 rows, linearly interpolated between rows)
 *   ca    += ws[n] * texel(first + n)                      n = 0 .. N-1
 *   "linear trick" filters (radius == radius_zero): taps come in pairs that one
 *   bilinear fetch evaluates: ca += (w0+w1) * mix(t[n], t[n+1], w1/(w0+w1))
 *   anti-ringing: clamp to [min, max] of the two centre taps, mixed by strength
 *   color = scale * ca
 *
 * Texture-unit emulation: every tap sits on a texel centre along `dir`, where a
 * TMU returns the texel itself; across `dir` the host tells us whether the pass
 * is on the texel grid (nearest) or needs the exact-fp32 bilinear blend.
 *
 * Launch shape: 64x4 lanes, one output pixel per lane. Consecutive lanes read
 * consecutive texels for every tap (vertical pass) or overlapping windows
 * (horizontal pass, served by L1), so HBM sees each source row once.
 */
#include "colorops.hiph"
#include "samplers.hiph"
#include "fastepi.hiph"
#include <stdlib.h>

#define ORTHO_BW 64
#define ORTHO_BH 4

// texel (i along dir, `o` across it) with the across-axis filtering rule
DEV float4_t ortho_fetch(const plh_sampler_args &s, int i, int o0, int o1, float ofrac)
{
    const int n = s.dir ? s.src.h : s.src.w;
    const int iw = plh_wrap(i, n, s.address_mode);
    if (!s.linear)
        return s.dir ? plh_fetch(s.src, o0, iw) : plh_fetch(s.src, iw, o0);
    const float4_t a = s.dir ? plh_fetch(s.src, o0, iw) : plh_fetch(s.src, iw, o0);
    const float4_t b = s.dir ? plh_fetch(s.src, o1, iw) : plh_fetch(s.src, iw, o1);
    return mix4(a, b, ofrac);
}

template <bool LITE>
__global__ __launch_bounds__(ORTHO_BW * ORTHO_BH)
void k_ortho(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const int idx = blockIdx.x * ORTHO_BW + threadIdx.x;
    const int idy = blockIdx.y * ORTHO_BH + threadIdx.y;
    const float mx = p.out_scale[0] * ((float) idx + 0.5f);
    const float my = p.out_scale[1] * ((float) idy + 0.5f);
    const float px = plh_attr(s.pos, 0, mx, my), py = plh_attr(s.pos, 1, mx, my);

    const float pa = s.dir ? py : px, po = s.dir ? px : py;
    const int na = s.dir ? s.src.h : s.src.w, no = s.dir ? s.src.w : s.src.h;
    const float ta = pa * (float) na - 0.5f;
    const float fla = __builtin_floorf(ta);
    const float fcoord = ta - fla;
    const int N = s.row_size;
    const int first = (int) fla - (N / 2 - 1);

    // across the filtered axis
    int o0, o1 = 0;
    float ofrac = 0.0f;
    if (!s.linear) {
        o0 = plh_wrap((int) __builtin_floorf(po * (float) no), no, s.address_mode);
    } else {
        const float to = po * (float) no - 0.5f, flo = __builtin_floorf(to);
        ofrac = to - flo;
        o0 = plh_wrap((int) flo, no, s.address_mode);
        o1 = plh_wrap((int) flo + 1, no, s.address_mode);
    }

    // LUT rows bracketing fcoord (linear LUT, lut.c:700-715 semantics)
    const float fpos = plh_clamp(fcoord, 0.0f, 1.0f) * 255.0f;
    const float fbase = __builtin_floorf(fpos);
    const float fr = fpos - fbase;
    const float *r0 = s.weights + (size_t) (int) fbase * s.row_stride;
    const float *r1 = s.weights + (size_t) min((int) fbase + 1, 255) * s.row_stride;

    float ca[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float lo[4] = {1e9f, 1e9f, 1e9f, 1e9f}, hi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (!s.use_linear && !s.linear) {
        // common case: one texel per tap. Eight taps per batch so that their loads overlap
        // (a per-tap plh_fetch costs one memory round trip per tap).
        const int n_axis = s.dir ? s.src.h : s.src.w;
        for (int n0 = 0; n0 < N; n0 += 8) {
            int tx[8], ty[8];
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int n = min(n0 + u, N - 1);
                const int iw = plh_wrap(first + n, n_axis, s.address_mode);
                tx[u] = s.dir ? o0 : iw;
                ty[u] = s.dir ? iw : o0;
                w[u] = plh_mix(r0[n], r1[n], fr);
            }
            float4_t t[8];
            plh_fetch_n<8>(s.src, tx, ty, t);
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int n = n0 + u;
                if (n >= N)
                    continue;
                const float cv[4] = { t[u].x, t[u].y, t[u].z, t[u].w };
                if (s.use_ar && (n == N / 2 - 1 || n == N / 2)) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        lo[k] = fminf(lo[k], cv[k]);
                        hi[k] = fmaxf(hi[k], cv[k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; k++)
                    ca[k] = __builtin_fmaf(w[u], cv[k], ca[k]);
            }
        }
    } else {
        // linear-trick filters and passes that are off the texel grid across the axis
        const int step = s.use_linear ? 2 : 1;
        for (int n = 0; n < N; n += step) {
            const float w = plh_mix(r0[n], r1[n], fr);
            float4_t c;
            if (s.use_linear) {
                // off = n + ws[n % 4 + 1]: one bilinear fetch between taps n and n + 1
                const float f = plh_mix(r0[n + 1], r1[n + 1], fr);
                c = mix4(ortho_fetch(s, first + n, o0, o1, ofrac),
                         ortho_fetch(s, first + n + 1, o0, o1, ofrac), f);
            } else {
                c = ortho_fetch(s, first + n, o0, o1, ofrac);
            }
            const float cv[4] = { c.x, c.y, c.z, c.w };
            if (s.use_ar && (n == N / 2 - 1 || n == N / 2)) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lo[k] = fminf(lo[k], cv[k]);
                    hi[k] = fmaxf(hi[k], cv[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                ca[k] = __builtin_fmaf(w, cv[k], ca[k]);
        }
    }
    if (s.use_ar) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            ca[k] = plh_mix(ca[k], plh_clamp(ca[k], lo[k], hi[k]), s.antiring);
    }

    // vec4 color = vec4(0, 0, 0, 1); color.<comps> = scale * ca
    float4_t out = { 0.0f, 0.0f, 0.0f, 1.0f };
    if (s.comp_mask & 1u) out.x = s.scale * ca[0];
    if (s.comp_mask & 2u) out.y = s.scale * ca[1];
    if (s.comp_mask & 4u) out.z = s.scale * ca[2];
    if (s.comp_mask & 8u) out.w = s.scale * ca[3];

    float4_t outs[1] = { out };
    const frag_t fcs[1] = { { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f,
                              0.0f, 0, mx, my } };
    apply_ops_n<1, false, LITE>(outs, p.ops, 0, p.num_ops, fcs);

    // guarded store (dispatch.c:1126-1142)
    const int sx[1] = { p.base_x + p.dir_x * (p.transpose ? idy : idx) };
    const int sy[1] = { p.base_y + p.dir_y * (p.transpose ? idx : idy) };
    const bool ok[1] = { p.out_scale[0] * (float) idx < 1.0f && p.out_scale[1] * (float) idy < 1.0f &&
                         sx[0] >= 0 && sy[0] >= 0 && sx[0] < p.dst.w && sy[0] < p.dst.h };
    plh_store_n<1>(p.dst, sx, sy, ok, outs);
}


/* ------------------------------------------------------------------------ */
/*
 * k_ortho_fast: the same convolution for the hot configurations -- an 8-byte texel source
 * (rgba16 / rgba16hf: the plane or the first pass' FBO) or a one- / two-component plane of
 * planar video (r8, rg8, r16, rg16 and their f16 FBOs), 4 / 6 / 8 taps, one texel per tap
 * (no linear trick), on the texel grid across the axis. Same fma order as k_ortho, so both
 * are bit-identical (tests run both: PL_HIP_ORTHO_FAST=0).
 *   - two horizontally adjacent pixels per lane: one 16-byte store per lane and row
 *   - weights: the two LUT rows bracketing fcoord as 16-byte loads (rows are 16-byte aligned)
 *   - all texel loads of both pixels in flight before the first fma
 *   - EPI 0: no colour ops; 1: fused epilogue (fastepi.hiph) -> rgba16; 2 / 3: LITE / full op
 *     interpreter; 4: the map chain (struct plh_map_chain) + fused epilogue as straight-line code
 *     (pl_render_default_params: unsigmoidize + delinearize + dither)
 */
// SRC = the source's plh format: the 8-byte RGBA formats (plane / FBO of packed frames) and the
// one- / two-component planes of planar video (their passes carry no colour ops: EPI 0 only)
// (`base` comes through the asm that pins it in SGPRs, after which it is a generic pointer to
// the compiler and every access a flat_ instruction: it is a device allocation, say so)
#define OF_GLOBAL __attribute__((address_space(1)))

template <int SRC>
DEV uint2 of_load(const char *base, int pitch, int x, int y)
{
    const OF_GLOBAL char *row = (const OF_GLOBAL char *) (uintptr_t) base + (size_t) y * pitch;
    if constexpr (SRC == PLH_FMT_RGBA16 || SRC == PLH_FMT_RGBA16F) {
        const plh_u32x2 v = *(const OF_GLOBAL plh_u32x2 *) (row + (size_t) x * 8);
        return make_uint2(v.x, v.y);
    } else if constexpr (SRC == PLH_FMT_RG16 || SRC == PLH_FMT_RG16F) {
        return make_uint2(*(const OF_GLOBAL uint32_t *) (row + (size_t) x * 4), 0);
    } else if constexpr (SRC == PLH_FMT_R16 || SRC == PLH_FMT_R16F || SRC == PLH_FMT_RG8) {
        return make_uint2(*(const OF_GLOBAL uint16_t *) (row + (size_t) x * 2), 0);
    } else {
        return make_uint2(*(const OF_GLOBAL uint8_t *) (row + x), 0);
    }
}

template <int SRC>
DEV float4_t of_decode(const uint2 v)
{
    float4_t c = { 0.0f, 0.0f, 0.0f, 1.0f };    // (plh_fetch's defaults for absent components)
    if constexpr (SRC == PLH_FMT_RGBA16F) {
        c = { plh_h2f(v.x & 0xffff), plh_h2f(v.x >> 16), plh_h2f(v.y & 0xffff), plh_h2f(v.y >> 16) };
    } else if constexpr (SRC == PLH_FMT_RGBA16) {
        c = { plh_un16(v.x & 0xffff), plh_un16(v.x >> 16), plh_un16(v.y & 0xffff), plh_un16(v.y >> 16) };
    } else if constexpr (SRC == PLH_FMT_RG16F) {
        c.x = plh_h2f(v.x & 0xffff); c.y = plh_h2f(v.x >> 16);
    } else if constexpr (SRC == PLH_FMT_RG16) {
        c.x = plh_un16(v.x & 0xffff); c.y = plh_un16(v.x >> 16);
    } else if constexpr (SRC == PLH_FMT_R16F) {
        c.x = plh_h2f(v.x & 0xffff);
    } else if constexpr (SRC == PLH_FMT_R16) {
        c.x = plh_un16(v.x & 0xffff);
    } else if constexpr (SRC == PLH_FMT_RG8) {
        c.x = plh_un8(v.x & 0xff); c.y = plh_un8((v.x >> 8) & 0xff);
    } else {
        c.x = plh_un8(v.x & 0xff);
    }
    return c;
}

DEV int of_clamp(int i, int n) { return min(max(i, 0), n - 1); }

// tap address along the filtered axis: clamp, or GL_MIRRORED_REPEAT for taps that overshoot by
// less than n (one reflection; the launcher checks n against the tap count)
DEV int of_tap(int i, int n, bool mirror)
{
    if (!mirror)
        return of_clamp(i, n);
    return i < 0 ? -1 - i : (i >= n ? 2 * n - 1 - i : i);
}

// (clamp or single-reflection mirror addressing; repeat costs an integer modulo per tap ->
// generic kernel)
// NT: number of taps (row_size), 4 / 6 / 8, or 16 = run-time count in [10, 16]
// LIN: "linear trick" LUTs of all-positive filters (fill_ortho_lut, sampling.c:919-936): taps come
// in pairs {w0 + w1, w1 / (w0 + w1)}, one blended fetch per pair
template <int SRC, int EPI, int DIR, int NT, bool LIN = false>
__global__ __launch_bounds__(ORTHO_BW * ORTHO_BH)
void k_ortho_fast(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const int cx = blockIdx.x * ORTHO_BW + threadIdx.x;
    const int idy = blockIdx.y * ORTHO_BH + threadIdx.y;
    // NT = 16: any even tap count from 10 to 16 (downscales), N read from the pass
    const int N = NT == 16 ? s.row_size : NT;
    // source base / pitch pinned in SGPRs: left alone the compiler re-loads them from the
    // kernel arguments in front of every texel load, each time with a full scalar wait
    const char *sp = (const char *) s.src.ptr;
    int spitch = s.src.pitch;
    asm volatile("" : "+s"(sp), "+s"(spitch));
    const int na = DIR ? s.src.h : s.src.w, no = DIR ? s.src.w : s.src.h;
    const float my = p.out_scale[1] * ((float) idy + 0.5f);
    const bool mirror = s.address_mode == PLH_ADDRESS_MIRROR;

    // The weight table (256 phases x the padded tap count: 4-8 KiB for 4 / 6 / 8 taps) is staged in
    // LDS once per workgroup -- one row per thread -- and the two rows that bracket a pixel's phase
    // are read from there: a horizontal pass otherwise issues 8 more vector-memory loads per lane
    // (2 pixels x 2 rows x 2 float4) than it has texel loads, and the pass is not free of its load
    // count (with all of them switched off it ran 52 -> 44 us, profiles/r05_25).
    // (horizontal passes only: a vertical pass's two pixels share their rows -- 4 loads -- and the
    // staging + barrier cost it 1 us where they save the horizontal pass 1-5, profiles/r05_28)
    constexpr bool LDS_LUT = NT != 16 && DIR == 0;
    constexpr int LUT_STRIDE = (NT + 3) / 4 * 4;
    __shared__ float4 lut_s[LDS_LUT ? 256 * LUT_STRIDE / 4 : 1];
    if constexpr (LDS_LUT) {
        const int row = threadIdx.y * ORTHO_BW + threadIdx.x;      // 0 .. 255
        const float4 *g = (const float4 *) (s.weights + (size_t) row * LUT_STRIDE);
#pragma unroll
        for (int j = 0; j < LUT_STRIDE / 4; j++)
            lut_s[row * (LUT_STRIDE / 4) + j] = g[j];
    }

    uint2 raw[2][NT];
    float w[2][NT];
    int first[2], o0[2];
    float fcoord[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int idx = 2 * cx + q;
        const float mx = p.out_scale[0] * ((float) idx + 0.5f);
        const float px = plh_attr(s.pos, 0, mx, my), py = plh_attr(s.pos, 1, mx, my);
        const float pa = DIR ? py : px, po = DIR ? px : py;
        const float ta = pa * (float) na - 0.5f;
        const float fla = __builtin_floorf(ta);
        fcoord[q] = ta - fla;
        first[q] = (int) fla - (N / 2 - 1);
        o0[q] = of_tap((int) __builtin_floorf(po * (float) no), no, mirror);
    }

    // texels. Horizontal 2x upscales: the two windows are the same or one texel apart, so
    // the second pixel's taps are the first one's shifted -> N + 1 loads instead of 2N.
    const int shift = first[1] - first[0];
    const bool overlap = !DIR && o0[1] == o0[0] && (shift == 0 || shift == 1);
    // Bytes of a texel of this source; the raw form of texel i of a run of bytes that was loaded in
    // 16-byte pieces (what of_load returns for it: the texel's bits in the low end of .x, .y for
    // the second half of an 8-byte one)
    constexpr int TB = (SRC == PLH_FMT_RGBA16 || SRC == PLH_FMT_RGBA16F) ? 8 :
                       (SRC == PLH_FMT_RG16 || SRC == PLH_FMT_RG16F) ? 4 :
                       (SRC == PLH_FMT_R16 || SRC == PLH_FMT_R16F || SRC == PLH_FMT_RG8) ? 2 : 1;
    auto texel_of = [](const uint32_t *wds, int i) {
        if constexpr (TB == 8)
            return make_uint2(wds[2 * i], wds[2 * i + 1]);
        else if constexpr (TB == 4)
            return make_uint2(wds[i], 0);
        else if constexpr (TB == 2)
            return make_uint2((wds[i >> 1] >> (16 * (i & 1))) & 0xffffu, 0);
        else
            return make_uint2((wds[i >> 2] >> (8 * (i & 3))) & 0xffu, 0);
    };
    // Vertical pass: the lane's two pixels sit on adjacent columns of the same rows, so ONE load of
    // two texels per tap row serves both -- 6 loads instead of 12 for a 6-tap filter (a pass is not
    // free of the NUMBER of its loads: each is 64 addresses for the texture addresser, whatever it
    // fetches; profiles/r05_summary.md).
    const bool pair_v = DIR && shift == 0 && o0[1] == o0[0] + 1;
    // Horizontal pass whose two windows overlap (a 2x upscale: every chroma plane of 4:2:0 video,
    // every default-preset upscale), away from the sides: the N + 1 consecutive texels as 16-byte
    // pieces -- 4 loads instead of 7 for 8-byte texels, ONE for the 14 bytes of an rg8 plane
    constexpr int HB = (NT + 1) * TB, HL = (HB + 15) / 16;      // bytes wanted, 16-byte loads
    const bool wide_h = !DIR && NT != 16 && overlap && first[0] >= 0 && first[0] * TB + HL * 16 <= na * TB;
    uint2 extra = make_uint2(0, 0);
    if (pair_v) {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int iw = of_tap(first[0] + min(n, N - 1), na, mirror);
            const OF_GLOBAL char *row = (const OF_GLOBAL char *) (uintptr_t) sp + (size_t) iw * spitch;
            // (aligned to the texel only: assembled from the bytes, which is one load instruction)
            uint32_t wds[4] = { 0, 0, 0, 0 };
            __builtin_memcpy(wds, (const void *) (row + (size_t) o0[0] * TB), 2 * TB);
            raw[0][n] = texel_of(wds, 0);
            raw[1][n] = texel_of(wds, 1);
        }
    } else if (wide_h) {
        uint32_t wds[4 * HL];
        const OF_GLOBAL char *row = (const OF_GLOBAL char *) (uintptr_t) sp + (size_t) o0[0] * spitch +
                                    (size_t) first[0] * TB;
#pragma unroll
        for (int k = 0; k < HL; k++)
            __builtin_memcpy(wds + 4 * k, (const void *) (row + 16 * k), 16);
#pragma unroll
        for (int n = 0; n < NT; n++)
            raw[0][n] = texel_of(wds, n);
        extra = texel_of(wds, NT);
    } else {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int iw = of_tap(first[0] + min(n, N - 1), na, mirror);
            raw[0][n] = DIR ? of_load<SRC>(sp, spitch, o0[0], iw) : of_load<SRC>(sp, spitch, iw, o0[0]);
        }
    }
    if (pair_v) {
        // (both pixels' taps are in)
    } else if (overlap) {
        if (!wide_h)
            extra = of_load<SRC>(sp, spitch, of_tap(first[0] + N, na, mirror), o0[0]);
    } else {
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const int iw = of_tap(first[1] + min(n, N - 1), na, mirror);
            raw[1][n] = DIR ? of_load<SRC>(sp, spitch, o0[1], iw) : of_load<SRC>(sp, spitch, iw, o0[1]);
        }
    }

    // (the dither values are requested now, not after the convolution: one round trip less)
    float bias[2] = { 0.0f, 0.0f };
    if constexpr (EPI == 1 || EPI == 4) {
        if (p.epi.has_dither) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int ix = (2 * cx + q + p.frag_x0) & p.epi.mask;
                const int iy = (idy + p.frag_y0) & p.epi.mask;
                bias[q] = p.epi.matrix[iy * p.epi.size + ix];
            }
        }
    }

    // weights: LUT rows bracketing fcoord (linear LUT, lut.c:700-715 semantics). Pixels with
    // the same fcoord (every pair of a vertical pass, bar rounding ties) share them.
    const bool same_w = __float_as_uint(fcoord[1]) == __float_as_uint(fcoord[0]);
    if constexpr (LDS_LUT)
        __syncthreads();    // (the staged table; the texel loads above are in flight behind it)
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (q == 1 && same_w)
            break;
        const float fpos = plh_clamp(fcoord[q], 0.0f, 1.0f) * 255.0f;
        const float fbase = __builtin_floorf(fpos);
        const float fr = fpos - fbase;
        const float4 *r0, *r1;
        if constexpr (LDS_LUT) {
            r0 = lut_s + (int) fbase * (LUT_STRIDE / 4);
            r1 = lut_s + min((int) fbase + 1, 255) * (LUT_STRIDE / 4);
        } else {
            r0 = (const float4 *) (s.weights + (size_t) (int) fbase * s.row_stride);
            r1 = (const float4 *) (s.weights + (size_t) min((int) fbase + 1, 255) * s.row_stride);
        }
        float ra[NT], rb[NT];
#pragma unroll
        for (int j = 0; j < NT / 4 + (NT % 4 != 0); j++) {
            float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), b = a;
            if (4 * j < N) {    // (rows are padded to a multiple of four floats)
                a = r0[j];
                b = r1[j];
            }
            const float av[4] = { a.x, a.y, a.z, a.w }, bv[4] = { b.x, b.y, b.z, b.w };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (4 * j + k < NT) {
                    ra[4 * j + k] = av[k];
                    rb[4 * j + k] = bv[k];
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NT; n++)
            w[q][n] = plh_mix(ra[n], rb[n], fr);
    }
    if (same_w) {
#pragma unroll
        for (int n = 0; n < NT; n++)
            w[1][n] = w[0][n];
    }
    if (overlap) {
        // raw[1][n] = texel first[0] + shift + n
#pragma unroll
        for (int n = 0; n < NT; n++) {
            const uint2 nxt = n + 1 < NT ? raw[0][n + 1] : extra;
            raw[1][n] = shift ? (n + 1 == N ? extra : nxt) : raw[0][n];
        }
    }

    float4_t outs[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        float ca[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float lo[4] = {1e9f, 1e9f, 1e9f, 1e9f}, hi[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int n = 0; n < NT; n += LIN ? 2 : 1) {
            if (NT == 16 && n >= N)
                continue;
            float4_t t = of_decode<SRC>(raw[q][n]);
            if constexpr (LIN) {
                // off = n + ws[n + 1]: the blend a bilinear fetch between taps n and n + 1 returns
                t = mix4(t, of_decode<SRC>(raw[q][n + 1 < NT ? n + 1 : n]), w[q][n + 1 < NT ? n + 1 : n]);
            }
            const float cv[4] = { t.x, t.y, t.z, t.w };
            if (s.use_ar && (n == N / 2 - 1 || n == N / 2)) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    lo[k] = fminf(lo[k], cv[k]);
                    hi[k] = fmaxf(hi[k], cv[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                ca[k] = __builtin_fmaf(w[q][n], cv[k], ca[k]);
        }
        if (s.use_ar) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                ca[k] = plh_mix(ca[k], plh_clamp(ca[k], lo[k], hi[k]), s.antiring);
        }
        // vec4 color = vec4(0, 0, 0, 1); color.<comps> = scale * ca
        float4_t out = { 0.0f, 0.0f, 0.0f, 1.0f };
        if (s.comp_mask & 1u) out.x = s.scale * ca[0];
        if (s.comp_mask & 2u) out.y = s.scale * ca[1];
        if (s.comp_mask & 4u) out.z = s.scale * ca[2];
        if (s.comp_mask & 8u) out.w = s.scale * ca[3];
        outs[q] = out;
    }

    int sx[2], sy[2];
    bool ok[2];
    frag_t fcs[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int idx = 2 * cx + q;
        fcs[q] = { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f, 0.0f, 0,
                   p.out_scale[0] * ((float) idx + 0.5f), my };
        sx[q] = p.base_x + p.dir_x * (p.transpose ? idy : idx);
        sy[q] = p.base_y + p.dir_y * (p.transpose ? idx : idy);
        ok[q] = p.out_scale[0] * (float) idx < 1.0f && p.out_scale[1] * (float) idy < 1.0f &&
                sx[q] >= 0 && sy[q] >= 0 && sx[q] < p.dst.w && sy[q] < p.dst.h;
    }
    if constexpr (EPI == 1 || EPI == 4) {
        // EPI 4: the recorded chain in front of the fused epilogue as straight-line code
        // (struct plh_map_chain: the default preset's unsigmoidize + delinearize, an HDR map)
        if constexpr (EPI == 4)
            run_map_chain<2>(outs, p);
#pragma unroll
        for (int q = 0; q < 2; q++) {
            float4_t &o = outs[q];
            if (p.epi.has_alpha)
                o.w = p.epi.alpha;
            // op_dither (non-gamma path) and the SCALE op (colorops.hiph)
            if (p.epi.has_dither) {
                const float b = bias[q];
                const float ds = p.epi.dscale, di = p.epi.dinv;
                o.x = __builtin_floorf(ds * o.x + b) * di;
                o.y = __builtin_floorf(ds * o.y + b) * di;
                o.z = __builtin_floorf(ds * o.z + b) * di;
                o.w = __builtin_floorf(ds * o.w + b) * di;
            }
            if (p.epi.has_scale)
                o = scale4(o, p.epi.scale);
        }
        plh_store_rgba16_n<2>(p.dst, sx, sy, ok, outs, p.nt_store);
    } else {
        if constexpr (EPI == 2)
            apply_ops_n<2, false, true>(outs, p.ops, 0, p.num_ops, fcs);
        if constexpr (EPI == 3)
            apply_ops_n<2, false, false>(outs, p.ops, 0, p.num_ops, fcs);
        plh_store_n<2>(p.dst, sx, sy, ok, outs, p.nt_store);
    }
}

template <int SRC>
static void launch_ortho_fast(hipStream_t stream, const plh_pass *pass, int epi)
{
    const dim3 block(ORTHO_BW, ORTHO_BH);
    const dim3 grid(((pass->width + 1) / 2 + ORTHO_BW - 1) / ORTHO_BW,
                    (pass->height + ORTHO_BH - 1) / ORTHO_BH);
#define LAUNCH_N(E, NT) do { \
        if (pass->s.dir) PLH_LAUNCH_LAST((k_ortho_fast<SRC, E, 1, NT>), grid, block, 0, stream, *pass); \
        else             PLH_LAUNCH_LAST((k_ortho_fast<SRC, E, 0, NT>), grid, block, 0, stream, *pass); \
    } while (0)
    // (the variants with the tap count at compile time stage the weight table in LDS and take its
    // row pitch from the tap count: rows padded to a multiple of four floats, as the host makes them)
    const bool packed_rows = pass->s.row_stride == (pass->s.row_size + 3) / 4 * 4;
#define LAUNCH(E) do { \
        if (pass->s.row_size == 4 && packed_rows)      LAUNCH_N(E, 4); \
        else if (pass->s.row_size == 6 && packed_rows) LAUNCH_N(E, 6); \
        else if (pass->s.row_size == 8 && packed_rows) LAUNCH_N(E, 8); \
        else                                           LAUNCH_N(E, 16); \
    } while (0)
    if (pass->s.use_linear) {
        // linear-trick filters (bicubic / gaussian / hermite ... low-pass), any tap count; colour
        // ops behind them only as the fused epilogue / the map chain, packed sources
        // (the tap count at compile time where it is 4 or 8 -- a halving with hermite / bicubic: the
        // run-time form issues all 16 loads whatever the count)
#define LAUNCH_LIN_N(E, NT) do { \
            if (pass->s.dir) PLH_LAUNCH_LAST((k_ortho_fast<SRC, E, 1, NT, true>), grid, block, 0, stream, *pass); \
            else             PLH_LAUNCH_LAST((k_ortho_fast<SRC, E, 0, NT, true>), grid, block, 0, stream, *pass); \
        } while (0)
#define LAUNCH_LIN(E) do { \
            if (pass->s.row_size == 4 && packed_rows)      LAUNCH_LIN_N(E, 4); \
            else if (pass->s.row_size == 8 && packed_rows) LAUNCH_LIN_N(E, 8); \
            else                                           LAUNCH_LIN_N(E, 16); \
        } while (0)
        if constexpr (SRC == PLH_FMT_RGBA16 || SRC == PLH_FMT_RGBA16F) {
            if (epi == 1)      LAUNCH_LIN(1);
            else if (epi == 4) LAUNCH_LIN(4);
            else               LAUNCH_LIN(0);
        } else {
            LAUNCH_LIN(0);
        }
#undef LAUNCH_LIN
#undef LAUNCH_LIN_N
        return;
    }
    if constexpr (SRC == PLH_FMT_RGBA16 || SRC == PLH_FMT_RGBA16F) {
        if (epi == 0)      LAUNCH(0);
        else if (epi == 1) LAUNCH(1);
        else if (epi == 2) LAUNCH(2);
        else if (epi == 4) LAUNCH(4);
        else               LAUNCH(3);
    } else {
        LAUNCH(0);      // plane passes: no colour ops (ortho_fast_variant)
    }
#undef LAUNCH
#undef LAUNCH_N
}

static bool ortho_fast_packed(int fmt)
{
    return fmt == PLH_FMT_RGBA16 || fmt == PLH_FMT_RGBA16F;
}

static bool ortho_fast_plane(int fmt)
{
    return fmt == PLH_FMT_R8 || fmt == PLH_FMT_RG8 || fmt == PLH_FMT_R16 || fmt == PLH_FMT_RG16 ||
           fmt == PLH_FMT_R16F || fmt == PLH_FMT_RG16F;
}

// -1: not eligible, else the epilogue variant
static int ortho_fast_variant(plh_pass *pass)
{
    // PL_HIP_ORTHO_FAST=0: always the generic kernel (read per launch: tests switch kernels)
    const char *e = getenv("PL_HIP_ORTHO_FAST");
    const int enabled = e ? atoi(e) : 1;
    const plh_sampler_args &s = pass->s;
    const int n_axis = s.dir ? s.src.h : s.src.w;
    bool addr_ok = s.address_mode == PLH_ADDRESS_CLAMP;
    if (s.address_mode == PLH_ADDRESS_MIRROR && n_axis >= 2 * s.row_size) {
        // one reflection is enough if the rect stays within one texture size of the texture
        addr_ok = true;
        for (int c = 0; c < 4; c++) {
            for (int k = 0; k < 2; k++)
                addr_ok = addr_ok && s.pos[c][k] > -0.9f && s.pos[c][k] < 1.9f;
        }
    }
    if (s.use_linear) {
        // the LIN variant: run-time tap count (even, <= 16); no colour ops, or -- packed sources:
        // the last pass of a downscale in linear light, pl_render_default_params 4K -> 1080p -- the
        // fused epilogue / the map chain (DELINEARIZE + dither) behind it
        if (!enabled || !addr_ok || s.linear || (s.row_size & 1) || s.row_size < 2 ||
            s.row_size > 16 || (s.row_stride & 3) || pass->num_pre_ops ||
            (!ortho_fast_packed(s.src.fmt) && !ortho_fast_plane(s.src.fmt)))
            return -1;
        if (!pass->num_ops)
            return 0;
        if (!ortho_fast_packed(s.src.fmt))
            return -1;
        plh_match_fast_epilogue(pass, true);
        if (pass->epi.enabled)
            return 1;
        plh_match_map_chain(pass);
        return pass->chain.enabled ? 4 : -1;
    }
    if (!enabled || !addr_ok || s.linear ||
        (s.row_size != 4 && s.row_size != 6 && s.row_size != 8 &&
         !(s.row_size >= 10 && s.row_size <= 16 && !(s.row_size & 1))) ||
        (s.row_stride & 3) || pass->num_pre_ops ||
        (!ortho_fast_packed(s.src.fmt) && !ortho_fast_plane(s.src.fmt)))
        return -1;
    if (!pass->num_ops)
        return 0;
    if (!ortho_fast_packed(s.src.fmt))
        return -1;  // planes with colour ops: generic kernel
    plh_match_fast_epilogue(pass, true);
    if (pass->epi.enabled)
        return 1;
    plh_match_map_chain(pass);
    if (pass->chain.enabled)
        return 4;
    for (int i = 0; i < pass->num_ops; i++) {
        const int k = pass->ops[i].kind;
        if (k == PLH_OP_PEAK_DETECT || k == PLH_OP_MIX_ADD)
            return -1;  // need their own kernels
    }
    return plh_ops_lite(pass, 0, pass->num_ops) ? 2 : 3;
}

int plh_launch_ortho(hipStream_t stream, const plh_pass *pass)
{
    {
        plh_pass local = *pass;
        const int epi = ortho_fast_variant(&local);
        if (epi >= 0) {
            switch (local.s.src.fmt) {
            case PLH_FMT_RGBA16F: launch_ortho_fast<PLH_FMT_RGBA16F>(stream, &local, epi); break;
            case PLH_FMT_RGBA16:  launch_ortho_fast<PLH_FMT_RGBA16>(stream, &local, epi); break;
            case PLH_FMT_R8:      launch_ortho_fast<PLH_FMT_R8>(stream, &local, epi); break;
            case PLH_FMT_RG8:     launch_ortho_fast<PLH_FMT_RG8>(stream, &local, epi); break;
            case PLH_FMT_R16:     launch_ortho_fast<PLH_FMT_R16>(stream, &local, epi); break;
            case PLH_FMT_RG16:    launch_ortho_fast<PLH_FMT_RG16>(stream, &local, epi); break;
            case PLH_FMT_R16F:    launch_ortho_fast<PLH_FMT_R16F>(stream, &local, epi); break;
            default:              launch_ortho_fast<PLH_FMT_RG16F>(stream, &local, epi); break;
            }
            const hipError_t err = hipGetLastError();
            return err == hipSuccess ? 0 : -(int) err;
        }
    }
    const dim3 block(ORTHO_BW, ORTHO_BH);
    const dim3 grid((pass->width + ORTHO_BW - 1) / ORTHO_BW,
                    (pass->height + ORTHO_BH - 1) / ORTHO_BH);
    if (plh_ops_lite(pass, 0, pass->num_ops))
        PLH_LAUNCH_LAST(k_ortho<true>, grid, block, 0, stream, *pass);
    else
        PLH_LAUNCH_LAST(k_ortho<false>, grid, block, 0, stream, *pass);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}


/*
 * k_lowpass2 (round 6): both passes of a separable DOWNSCALE of a one-component r16hf plane in one
 * launch -- the low-pass behind the contrast-recovery feature map (4K -> 1097 x 617 with the widened
 * bicubic, 14 taps per axis: k_ortho_fast<R16F, 0, 1, 16> 24.6 us + <.., 0, 16> 16.7 us for a 16 MB
 * plane, each lane issuing its 14 - 16 tap loads per pixel; VERDICT r04 / r05 "the two low-pass
 * passes as one LDS-resident kernel").
 * A workgroup renders 32 x 16 output pixels. It finds the source columns its horizontal pass needs
 * and the source rows its vertical pass needs (from the passes' own per-pixel geometry, reduced in
 * LDS), stages that part of the plane ONCE (mirrored at the plane's edges as the taps would be),
 * runs the vertical pass for its 16 rows over those columns into an LDS image -- rounded to f16, as
 * the r16hf intermediate of the two-pass form rounds it -- and the horizontal pass from there.
 * Arithmetic: k_ortho_fast's statement for statement for either pass (geometry per pixel through
 * plh_attr, the two weight rows bracketing fcoord blended per pixel, taps accumulated in order by
 * fma, scale, f16), so the plane is bit-identical to the two-pass one
 * (tests/test_gpu_contrast_recovery.py::test_fused_lowpass_equals_the_two_passes).
 */
#define LP2_TW 32
#define LP2_TH 16
#define LP2_NT 256
#define LP2_KR 20       // rows of the tile per wave (4 waves): rows_cap <= 80

struct lp2_geo { int first; float fcoord; };

// minimum and maximum over the wave, then ONE pair of LDS atomics per wave (every lane on its own:
// 256 threads x 18 atomics on four addresses, which LDS takes one lane at a time -- the first
// version of the kernel spent 90 of its 132 us there)
DEV void lp2_range(int lo, int hi, int *slot_lo, int *slot_hi, int lane)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off));
        hi = max(hi, __shfl_xor(hi, off));
    }
    if (lane == 0) {
        atomicMin(slot_lo, lo);
        atomicMax(slot_hi, hi);
    }
}

// geometry of pixel (idx, idy) of a pass filtering along DIR (k_ortho_fast, lines "ta = ...")
template <int DIR>
DEV lp2_geo lp2_geometry(const float (&pos)[4][2], const float (&os)[2], int idx, int idy, int na, int N)
{
    const float mx = os[0] * ((float) idx + 0.5f), my = os[1] * ((float) idy + 0.5f);
    const float pa = plh_attr(pos, DIR ? 1 : 0, mx, my);
    const float ta = pa * (float) na - 0.5f;
    const float fla = __builtin_floorf(ta);
    lp2_geo g;
    g.fcoord = ta - fla;
    g.first = (int) fla - (N / 2 - 1);
    return g;
}

// the pixel's weights: the two table rows bracketing fcoord, blended (k_ortho_fast, NT = 16).
// UNIFORM: fcoord is the same in every lane of the wave (the caller has checked): the rows are then
// read by the scalar unit
template <bool UNIFORM>
DEV void lp2_weights(const float *table, int stride, int N, float fcoord, float (&w)[16])
{
    const float fpos = plh_clamp(fcoord, 0.0f, 1.0f) * 255.0f;
    const float fbase = __builtin_floorf(fpos);
    const float fr = fpos - fbase;
    int i0 = (int) fbase, i1 = min((int) fbase + 1, 255);
    if (UNIFORM) {
        i0 = __builtin_amdgcn_readfirstlane(i0);
        i1 = __builtin_amdgcn_readfirstlane(i1);
    }
    typedef float lp2_f32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const lp2_f32x4 gf4;
    typedef __attribute__((address_space(4))) const lp2_f32x4 cf4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        lp2_f32x4 a = { 0.0f, 0.0f, 0.0f, 0.0f }, b = a;
        if (4 * j < N) {
            if (UNIFORM) {
                a = ((cf4 *) (uintptr_t) (table + (size_t) i0 * stride))[j];
                b = ((cf4 *) (uintptr_t) (table + (size_t) i1 * stride))[j];
            } else {
                a = ((gf4 *) (uintptr_t) (table + (size_t) i0 * stride))[j];
                b = ((gf4 *) (uintptr_t) (table + (size_t) i1 * stride))[j];
            }
        }
        w[4 * j] = plh_mix(a.x, b.x, fr);
        w[4 * j + 1] = plh_mix(a.y, b.y, fr);
        w[4 * j + 2] = plh_mix(a.z, b.z, fr);
        w[4 * j + 3] = plh_mix(a.w, b.w, fr);
    }
}

// LIN: "linear trick" weight rows (all-positive filters, fill_ortho_lut: sampling.c:919-936; the
// bicubic B-spline of the feature map's low-pass is one): taps in pairs {w0 + w1, w1 / (w0 + w1)},
// the pair blended first -- k_ortho_fast<.., LIN>'s statements
template <bool LIN>
__global__ __launch_bounds__(LP2_NT)
void k_lowpass2(const plh_lowpass2 a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lp2_smem[];
    __shared__ int rng[4];      // first / last source column, first / last source row of the tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int X0 = blockIdx.x * LP2_TW, Y0 = blockIdx.y * LP2_TH;
    const int wB = a.dst.w, hB = a.dst.h, srcw = a.src.w, srch = a.src.h;
    const int NV = a.n_v, NH = a.n_h, cap_c = a.cols_cap, cap_r = a.rows_cap;   // (cap_c: a multiple of 4)
    const bool mirror = a.mirror;
    uint16_t *tile = (uint16_t *) lp2_smem;                                  // cap_r x cap_c f16 codes
    float *mid = (float *) (lp2_smem + (((size_t) cap_r * cap_c * 2 + 15) & ~(size_t) 15));    // 16 x cap_c
    if (tid == 0) {
        rng[0] = rng[2] = 0x7fffffff;
        rng[1] = rng[3] = -0x7fffffff;
    }
    __syncthreads();

    // ---- the horizontal pass's geometry of this thread's two outputs (rows 8 apart), and with it
    // the source columns the tile needs
    const int Xq = min(X0 + (tid & 31), wB - 1);
    lp2_geo gh[2];
#pragma unroll
    for (int q = 0; q < 2; q++)
        gh[q] = lp2_geometry<0>(a.pos_h, a.os_h, Xq, min(Y0 + (tid >> 5) + 8 * q, hB - 1), a.mid_w, NH);
    lp2_range(min(gh[0].first, gh[1].first), max(gh[0].first, gh[1].first) + NH - 1, &rng[0], &rng[1], lane);
    __syncthreads();
    const int cmin = rng[0], ncols = min(rng[1] - cmin + 1, cap_c);

    // ---- the vertical pass's geometry of this thread's items -- intermediate rows wave, wave + 4,
    // ... of the tile, the two logical columns 2 lane, 2 lane + 1 -- and the source rows the tile needs
    lp2_geo gv[4][2];
    int vlo = 0x7fffffff, vhi = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int Y = min(Y0 + wave + 4 * r, hB - 1);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const int j = min(2 * lane + c, ncols - 1);
            const int x = of_tap(cmin + j, a.mid_w, mirror);
            gv[r][c] = lp2_geometry<1>(a.pos_v, a.os_v, x, Y, srch, NV);
            vlo = min(vlo, gv[r][c].first);
            vhi = max(vhi, gv[r][c].first);
        }
    }
    lp2_range(vlo, vhi + NV - 1, &rng[2], &rng[3], lane);
    __syncthreads();
    const int rmin = rng[2], nrows = min(rng[3] - rmin + 1, cap_r);

    // ---- the tile: source rows rmin .., columns cmin .., mirrored (or clamped) like the taps
    // A tile whose columns lie inside the plane (all but the tiles on its left and right edge) is
    // moved as 32-bit column pairs with EVERY row's load in flight before the first is waited for: a
    // loop over the rows with the load inside is a memory round trip per row, 18 in a row per wave
    // (the first version: 43 us for what the two passes do in 43).
    if (cmin >= 0 && cmin + cap_c <= srcw) {
        typedef __attribute__((address_space(1))) const uint32_t gu32;
        uint32_t v[LP2_KR];
        const uintptr_t col = (uintptr_t) a.src.ptr + (size_t) (cmin + 2 * min(lane, cap_c / 2 - 1)) * 2;
#pragma unroll
        for (int k = 0; k < LP2_KR; k++) {
            const int y = of_tap(rmin + min(wave + 4 * k, nrows - 1), srch, mirror);
            // (2-byte aligned: assembled from the bytes, which the backend turns into one dword load)
            __builtin_memcpy(&v[k], (const void *) (gu32 *) (col + (size_t) y * (size_t) a.src.pitch), 4);
        }
        uint32_t *t32 = (uint32_t *) tile;
#pragma unroll
        for (int k = 0; k < LP2_KR; k++) {
            if (wave + 4 * k < nrows && 2 * lane < cap_c)
                t32[(wave + 4 * k) * (cap_c / 2) + lane] = v[k];
        }
    } else {
        typedef __attribute__((address_space(1))) const uint16_t gu16;
        for (int r = wave; r < nrows; r += LP2_NT / 64) {
            const int y = of_tap(rmin + r, srch, mirror);
            gu16 *row = (gu16 *) ((uintptr_t) a.src.ptr + (size_t) y * (size_t) a.src.pitch);
            for (int j = lane; j < cap_c; j += 64)
                tile[r * cap_c + j] = row[of_tap(cmin + min(j, ncols - 1), srcw, mirror)];
        }
    }
    __syncthreads();

    // ---- vertical pass into the LDS image (f16-rounded, as the r16hf intermediate would be): the
    // lane's two columns share a tap row's 32-bit read, and -- bar a rounding tie in the geometry,
    // which is checked -- their weights
    const uint32_t *tile32 = (const uint32_t *) tile;
    const int pitch32 = cap_c / 2;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        if (Y0 + wave + 4 * r >= hB || 2 * lane >= cap_c)
            continue;
        const lp2_geo g0 = gv[r][0], g1 = gv[r][1];
        float w[16];
        const bool uni = __builtin_amdgcn_ballot_w64(__float_as_uint(g0.fcoord) !=
                                                     (uint32_t) __builtin_amdgcn_readfirstlane((int) __float_as_uint(g0.fcoord))) == 0;
        if (uni)
            lp2_weights<true>(a.wgt_v, a.stride_v, NV, g0.fcoord, w);
        else
            lp2_weights<false>(a.wgt_v, a.stride_v, NV, g0.fcoord, w);
        // (the taps lie inside the tile by construction: one clamp for memory safety)
        const int t0 = min(max(g0.first - rmin, 0), max(nrows - NV, 0));
        float ca0 = 0.0f, ca1 = 0.0f;
#pragma unroll
        for (int n = 0; n < 16; n += LIN ? 2 : 1) {
            if (n >= NV)
                continue;
            const uint32_t v = tile32[(t0 + n) * pitch32 + lane];
            float c0 = plh_h2f(v & 0xffffu), c1 = plh_h2f(v >> 16);
            if (LIN) {
                // (n + 1 <= N - 1: the tap counts are even)
                const uint32_t v1 = tile32[(t0 + n + 1) * pitch32 + lane];
                c0 = plh_mix(c0, plh_h2f(v1 & 0xffffu), w[n + 1 < 16 ? n + 1 : n]);
                c1 = plh_mix(c1, plh_h2f(v1 >> 16), w[n + 1 < 16 ? n + 1 : n]);
            }
            ca0 = __builtin_fmaf(w[n], c0, ca0);
            ca1 = __builtin_fmaf(w[n], c1, ca1);
        }
        // a second column whose geometry rounded differently: on its own
        const bool same = __float_as_uint(g1.fcoord) == __float_as_uint(g0.fcoord) && g1.first == g0.first;
        if (__builtin_amdgcn_ballot_w64(!same) != 0) {
            float w1[16];
            lp2_weights<false>(a.wgt_v, a.stride_v, NV, g1.fcoord, w1);
            const int u0 = min(max(g1.first - rmin, 0), max(nrows - NV, 0));
            float cb = 0.0f;
#pragma unroll
            for (int n = 0; n < 16; n += LIN ? 2 : 1) {
                if (n >= NV)
                    continue;
                float c1 = plh_h2f(tile[(u0 + n) * cap_c + 2 * lane + 1]);
                if (LIN)
                    c1 = plh_mix(c1, plh_h2f(tile[(u0 + n + 1) * cap_c + 2 * lane + 1]), w1[n + 1 < 16 ? n + 1 : n]);
                cb = __builtin_fmaf(w1[n], c1, cb);
            }
            ca1 = same ? ca1 : cb;
        }
        float2 out = make_float2(plh_h2f(plh_f2h(a.scale_v * ca0)), plh_h2f(plh_f2h(a.scale_v * ca1)));
        *(float2 *) (mid + (wave + 4 * r) * cap_c + 2 * lane) = out;
    }
    __syncthreads();

    // ---- horizontal pass from the LDS image: the lane's two outputs (rows 8 apart) share their
    // column, and -- bar a rounding tie -- their weights
    {
        const int X = X0 + (tid & 31), Ya = Y0 + (tid >> 5), Yb = Ya + 8;
        float w[16];
        lp2_weights<false>(a.wgt_h, a.stride_h, NH, gh[0].fcoord, w);
        const int t0 = min(max(gh[0].first - cmin, 0), max(ncols - NH, 0));
        const float *ra = mid + (Ya - Y0) * cap_c + t0, *rb = ra + 8 * cap_c;
        float ca0 = 0.0f, ca1 = 0.0f;
#pragma unroll
        for (int n = 0; n < 16; n += LIN ? 2 : 1) {
            if (n >= NH)
                continue;
            float c0 = ra[n], c1 = rb[n];
            if (LIN) {
                c0 = plh_mix(c0, ra[n + 1], w[n + 1 < 16 ? n + 1 : n]);
                c1 = plh_mix(c1, rb[n + 1], w[n + 1 < 16 ? n + 1 : n]);
            }
            ca0 = __builtin_fmaf(w[n], c0, ca0);
            ca1 = __builtin_fmaf(w[n], c1, ca1);
        }
        const bool same = __float_as_uint(gh[1].fcoord) == __float_as_uint(gh[0].fcoord) && gh[1].first == gh[0].first;
        if (__builtin_amdgcn_ballot_w64(!same) != 0) {
            float w1[16];
            lp2_weights<false>(a.wgt_h, a.stride_h, NH, gh[1].fcoord, w1);
            const float *rc = mid + (Yb - Y0) * cap_c + min(max(gh[1].first - cmin, 0), max(ncols - NH, 0));
            float cb = 0.0f;
#pragma unroll
            for (int n = 0; n < 16; n += LIN ? 2 : 1) {
                if (n >= NH)
                    continue;
                float c1 = rc[n];
                if (LIN)
                    c1 = plh_mix(c1, rc[n + 1], w1[n + 1 < 16 ? n + 1 : n]);
                cb = __builtin_fmaf(w1[n], c1, cb);
            }
            ca1 = same ? ca1 : cb;
        }
        typedef __attribute__((address_space(1))) uint16_t gu16o;
        if (X < wB && Ya < hB)
            *(gu16o *) ((uintptr_t) a.dst.ptr + (size_t) Ya * (size_t) a.dst.pitch + (size_t) X * 2) =
                (uint16_t) plh_f2h(a.scale_h * ca0);
        if (X < wB && Yb < hB)
            *(gu16o *) ((uintptr_t) a.dst.ptr + (size_t) Yb * (size_t) a.dst.pitch + (size_t) X * 2) =
                (uint16_t) plh_f2h(a.scale_h * ca1);
    }
}

static size_t lowpass2_lds(const plh_lowpass2 &a)
{
    const size_t tile = ((size_t) a.rows_cap * a.cols_cap * 2 + 15) & ~(size_t) 15;
    return tile + (size_t) LP2_TH * a.cols_cap * 4;
}

// PL_HIP_LOWPASS_FUSED=0: the two passes
extern "C" int plh_lowpass2_applies(const struct plh_lowpass2 *args)
{
    const char *env = getenv("PL_HIP_LOWPASS_FUSED");
    if (env && env[0] == '0')
        return 0;
    const plh_lowpass2 &a = *args;
    if (a.src.fmt != PLH_FMT_R16F || a.dst.fmt != PLH_FMT_R16F || a.n_v > 16 || a.n_h > 16 || a.n_v < 2 ||
        a.n_h < 2 || (a.n_v & 1) || (a.n_h & 1) || a.mid_w != a.src.w || a.mid_h != a.dst.h ||
        a.dst.w < 1 || a.dst.h < 1)
        return 0;
    // one reflection only (of_tap), as k_ortho_fast's launcher requires
    if (a.mirror && (a.src.w < 16 || a.src.h < 16))
        return 0;
    // (a lane owns two columns of the tile: 128 at most -- ratios up to 3.5 with the widened bicubic)
    return a.rows_cap >= a.n_v && a.rows_cap <= 4 * LP2_KR && a.cols_cap >= a.n_h && a.cols_cap <= 128 && !(a.cols_cap & 3) &&
           lowpass2_lds(a) <= 60 * 1024;
}

extern "C" int plh_launch_lowpass2(plh_stream stream_, const struct plh_lowpass2 *args)
{
    hipStream_t stream = (hipStream_t) stream_;
    if (!plh_lowpass2_applies(args))
        return 1;
    const plh_lowpass2 &a = *args;
    const size_t shmem = lowpass2_lds(a);
    const dim3 grid((a.dst.w + LP2_TW - 1) / LP2_TW, (a.dst.h + LP2_TH - 1) / LP2_TH);
    if (a.linear_trick)
        PLH_LAUNCH_LAST(k_lowpass2<true>, grid, dim3(LP2_NT), shmem, stream, a);
    else
        PLH_LAUNCH_LAST(k_lowpass2<false>, grid, dim3(LP2_NT), shmem, stream, a);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
