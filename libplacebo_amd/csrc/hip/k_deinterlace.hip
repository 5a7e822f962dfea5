/*
 * libplacebo-hip -- deinterlacing (pl_shader_deinterlace, reference src/shaders/deinterlacing.c).
 *
 * An interlaced frame holds two fields woven row by row. The field being shown passes through;
 * the rows of the other one are rebuilt from their neighbours in space (the rows above and below)
 * and in time (the previous and the next frame): doubled (bob), by yadif's edge-directed predictor
 * clamped to what the temporal neighbours allow, or by bwdif's two cubic filters chosen per pixel
 * by motion. Every output is a function of a dozen to two dozen texels within +-3 columns and
 * +-4 rows of three frames: a stencil that moves a frame's bytes in and out once and is bound by
 * the number of loads a lane issues, not by their bytes (DESIGN 4.10's finding).
 *
 * The reference runs one invocation per output pixel and notes itself that half of them idle (the
 * kept rows). Here a lane owns a ROW PAIR -- one kept row, one rebuilt row of the same column --
 * so that every lane of a wave does the same work, and the rows of the current frame a pair needs
 * (two above, two below) are loaded once for both. The colour ops recorded behind the sampler run
 * through the interpreter, two pixels per lane.
 */
#include <stdlib.h>

#include "colorops.hiph"
#include "backend.h"

#define DEINT_BW 64
#define DEINT_BH 4

// GET(TEX, X, Y): nearest texel under the MIRROR address mode the reference binds with (:51)
DEV float4_t deint_get(const plh_view &v, int x, int y)
{
    return plh_fetch(v, plh_wrap(x, v.w, PLH_ADDRESS_MIRROR), plh_wrap(y, v.h, PLH_ADDRESS_MIRROR));
}

DEV float comp(const float4_t &c, int ch)
{
    return ch == 0 ? c.x : ch == 1 ? c.y : ch == 2 ? c.z : c.w;
}

// yadif's spatial predictor (:131-157): the direction, out of five, along which the rows above
// and below agree best
DEV float yadif_spatial(const float (&up)[7], const float (&dn)[7], float bias)
{
    float pred = (up[3] + dn[3]) / 2.0f;
    float best = __builtin_fabsf(up[2] - dn[2]) + __builtin_fabsf(up[3] - dn[3]) +
                 __builtin_fabsf(up[4] - dn[4]) - bias;
    // leaning left: one step, then two
    float score = __builtin_fabsf(up[1] - dn[3]) + __builtin_fabsf(up[2] - dn[4]) +
                  __builtin_fabsf(up[3] - dn[5]);
    if (score < best) {
        pred = (up[2] + dn[4]) / 2.0f;
        best = score;
        score = __builtin_fabsf(up[0] - dn[4]) + __builtin_fabsf(up[1] - dn[5]) +
                __builtin_fabsf(up[2] - dn[6]);
        if (score < best) {
            pred = (up[1] + dn[5]) / 2.0f;
            best = score;
        }
    }
    // leaning right
    score = __builtin_fabsf(up[3] - dn[1]) + __builtin_fabsf(up[4] - dn[2]) +
            __builtin_fabsf(up[5] - dn[3]);
    if (score < best) {
        pred = (up[4] + dn[2]) / 2.0f;
        best = score;
        score = __builtin_fabsf(up[4] - dn[0]) + __builtin_fabsf(up[5] - dn[1]) +
                __builtin_fabsf(up[6] - dn[2]);
        if (score < best) {
            pred = (up[5] + dn[1]) / 2.0f;
            best = score;
        }
    }
    return pred;
}

// Rows of one column around the rebuilt row y: the current frame at -3 -1 +1 +3, the nearest
// temporal neighbours at -1 +1, the second ones at -4 -2 0 +2 +4
struct deint_column {
    float cur[4];
    float prev[2], next[2];
    float prev2[5], next2[5];
};

// yadif's temporal predictor (:188-216): the spatial prediction clamped around the average of the
// second neighbours by as much as the fields around it changed
DEV float yadif_temporal(const deint_column &t, float pred, bool skip_spatial_check)
{
    const float F = t.cur[1], G = t.cur[2];
    const float p0 = (t.prev2[1] + t.next2[1]) / 2.0f, p2 = (t.prev2[2] + t.next2[2]) / 2.0f,
                p4 = (t.prev2[3] + t.next2[3]) / 2.0f;
    const float tdiff0 = __builtin_fabsf(t.prev2[2] - t.next2[2]) / 2.0f;
    const float tdiff1 = (__builtin_fabsf(t.prev[0] - F) + __builtin_fabsf(t.prev[1] - G)) / 2.0f;
    const float tdiff2 = (__builtin_fabsf(t.next[0] - F) + __builtin_fabsf(G - t.next[1])) / 2.0f;
    float diff = fmaxf(tdiff0, fmaxf(tdiff1, tdiff2));
    if (!skip_spatial_check) {
        const float maxi = fmaxf(p2 - fminf(G, F), fminf(p0 - F, p4 - G));
        const float mini = fminf(p2 - fmaxf(G, F), fmaxf(p0 - F, p4 - G));
        diff = fmaxf(diff, fmaxf(mini, -maxi));
    }
    if (pred > p2 + diff)
        pred = p2 + diff;
    if (pred < p2 - diff)
        pred = p2 - diff;
    return pred;
}

// bwdif (:270-322)
DEV float bwdif_intra(const float (&cur)[4])
{
    return (5077.0f / 8192.0f) * (cur[1] + cur[2]) - (981.0f / 8192.0f) * (cur[0] + cur[3]);
}

DEV float bwdif_process(const deint_column &t)
{
    const float s = t.prev2[2] + t.next2[2];
    const float d = s / 2.0f;
    const float c = t.cur[1], e = t.cur[2];

    const float tdiff0 = __builtin_fabsf(t.prev2[2] - t.next2[2]);
    const float tdiff1 = __builtin_fabsf(t.prev[0] - c) + __builtin_fabsf(t.prev[1] - e);
    const float tdiff2 = __builtin_fabsf(t.next[0] - c) + __builtin_fabsf(t.next[1] - e);
    float diff = fmaxf(tdiff0, fmaxf(tdiff1, tdiff2)) / 2.0f;
    const bool still = diff == 0.0f;

    const float bs = t.prev2[1] + t.next2[1], fs = t.prev2[3] + t.next2[3];
    const float b = (bs / 2.0f) - c, f = (fs / 2.0f) - c;
    const float dc = d - c, de = d - e;
    const float mmax = fmaxf(de, fmaxf(dc, fminf(b, f)));
    const float mmin = fminf(de, fminf(dc, fmaxf(b, f)));
    diff = fmaxf(diff, fmaxf(mmin, -mmax));

    const float edges = t.cur[0] + t.cur[3];
    const float single = (5077.0f / 8192.0f) * (c + e) - (981.0f / 8192.0f) * edges;
    float all = ((5570.0f / 8192.0f) * s - (3801.0f / 8192.0f) * (bs + fs) +
                 (1016.0f / 8192.0f) * (t.prev2[0] + t.next2[0] + t.prev2[4] + t.next2[4])) / 4.0f;
    all += (4309.0f / 8192.0f) * (c + e) - (213.0f / 8192.0f) * edges;

    float interpol = __builtin_fabsf(c - e) > tdiff0 ? all : single;
    interpol = fminf(fmaxf(interpol, d - diff), d + diff);
    return still ? d : interpol;
}

template <bool LITE>
__global__ __launch_bounds__(DEINT_BW * DEINT_BH)
void k_deinterlace(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_view &cur = p.s.src;
    const plh_deint_args &a = p.deint;
    const int x = blockIdx.x * DEINT_BW + (int) (threadIdx.x % DEINT_BW);
    const int pair = blockIdx.y * DEINT_BH + (int) (threadIdx.x / DEINT_BW);
    // the pair's kept row and the row to rebuild (the whole frame is kept without a field)
    const int yk = 2 * pair + a.keep, yr = 2 * pair + 1 - a.keep;
    const bool rebuild = a.algo != PLH_DEINT_WEAVE && a.keep >= 0;

    float4_t c[2];
    c[0] = deint_get(cur, x, a.keep >= 0 ? yk : 2 * pair);
    c[1] = deint_get(cur, x, a.keep >= 0 ? yr : 2 * pair + 1);
    if (rebuild) {
        const plh_view &prev2 = a.first ? a.prev : cur, &next2 = a.first ? cur : a.next;
        if (a.algo == PLH_DEINT_BOB) {
            // the kept row above (top field shown) or below
            c[1] = deint_get(cur, x, yr + (a.keep ? 1 : -1));
        } else if (a.algo == PLH_DEINT_BWDIF && a.intra_only) {
            const float4_t r[4] = { deint_get(cur, x, yr - 3), deint_get(cur, x, yr - 1),
                                    deint_get(cur, x, yr + 1), deint_get(cur, x, yr + 3) };
            float out[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                const float col[4] = { comp(r[0], ch), comp(r[1], ch), comp(r[2], ch), comp(r[3], ch) };
                out[ch] = bwdif_intra(col);
            }
            c[1] = { out[0], out[1], out[2], out[3] };
        } else {
            // the column's temporal neighbourhood, all four components at once
            float4_t tc[4], tp[2], tn[2], tp2[5], tn2[5];
#pragma unroll
            for (int k = 0; k < 4; k++)
                tc[k] = deint_get(cur, x, yr + 2 * k - 3);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                tp[k] = deint_get(a.prev, x, yr + 2 * k - 1);
                tn[k] = deint_get(a.next, x, yr + 2 * k - 1);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                tp2[k] = deint_get(prev2, x, yr + 2 * k - 4);
                tn2[k] = deint_get(next2, x, yr + 2 * k - 4);
            }
            float4_t up[7], dn[7];
            if (a.algo == PLH_DEINT_YADIF) {
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    up[k] = k == 3 ? tc[1] : deint_get(cur, x + k - 3, yr - 1);
                    dn[k] = k == 3 ? tc[2] : deint_get(cur, x + k - 3, yr + 1);
                }
            }
            float out[4];
#pragma unroll
            for (int ch = 0; ch < 4; ch++) {
                deint_column t;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    t.cur[k] = comp(tc[k], ch);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    t.prev[k] = comp(tp[k], ch);
                    t.next[k] = comp(tn[k], ch);
                }
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    t.prev2[k] = comp(tp2[k], ch);
                    t.next2[k] = comp(tn2[k], ch);
                }
                if (a.algo == PLH_DEINT_YADIF) {
                    float u[7], d[7];
#pragma unroll
                    for (int k = 0; k < 7; k++) {
                        u[k] = comp(up[k], ch);
                        d[k] = comp(dn[k], ch);
                    }
                    out[ch] = yadif_temporal(t, yadif_spatial(u, d, a.spatial_bias),
                                             a.skip_spatial_check != 0);
                } else {
                    out[ch] = bwdif_process(t);
                }
            }
            c[1] = { out[0], out[1], out[2], out[3] };
        }
    }

    // components outside the mask keep the shader's initial colour (0, 0, 0, 1) (:37, :364)
    frag_t fcs[2];
    int sx[2], sy[2];
    bool ok[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        if (!(p.s.comp_mask & 1u)) c[q].x = 0.0f;
        if (!(p.s.comp_mask & 2u)) c[q].y = 0.0f;
        if (!(p.s.comp_mask & 4u)) c[q].z = 0.0f;
        if (!(p.s.comp_mask & 8u)) c[q].w = 1.0f;
        const int idx = x, idy = a.keep >= 0 ? (q ? yr : yk) : 2 * pair + q;
        fcs[q] = { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f, 0.0f, 0,
                   p.out_scale[0] * ((float) idx + 0.5f), p.out_scale[1] * ((float) idy + 0.5f) };
        sx[q] = p.base_x + p.dir_x * (p.transpose ? idy : idx);
        sy[q] = p.base_y + p.dir_y * (p.transpose ? idx : idy);
        ok[q] = idx < p.width && idy < p.height && sx[q] >= 0 && sy[q] >= 0 &&
                sx[q] < p.dst.w && sy[q] < p.dst.h;
    }
    apply_ops_n<2, false, LITE>(c, p.ops, 0, p.num_ops, fcs);
    plh_store_n<2>(p.dst, sx, sy, ok, c, p.nt_store);
}

/* ---- the renderer's case: a whole video plane, nothing behind the sampler ----------------------
 * bob, weave and bwdif (the default) only look up and down, so a row of a plane is an array of
 * component slots that can be taken four bytes at a time whatever the texel is -- r8: four pixels,
 * rg8 / r16: two, rg16: one -- and a lane owns one such dword of a row pair: bwdif's 18 rows are 18
 * dword loads for up to four pixels instead of 18 single bytes per pixel. The plane comes out in
 * its own format (what the renderer stores when a scaler follows: rounded like any store) or as
 * floats of the same layout (when the plane is the image's reference grid and continues
 * unrounded: renderer.c deinterlace_plane). Same arithmetic, same order: bit-identical to
 * k_deinterlace (tests/test_gpu_deinterlace.py runs both). yadif looks sideways as well and stays
 * on the general kernel. */
template <typename C> DEV float deint_slot(uint32_t w, int k);
template <> DEV float deint_slot<uint8_t>(uint32_t w, int k) { return plh_un8((w >> (8 * k)) & 0xffu); }
template <> DEV float deint_slot<uint16_t>(uint32_t w, int k) { return plh_un16((w >> (16 * k)) & 0xffffu); }

template <typename C>
DEV uint32_t deint_pack(const float (&v)[4 / sizeof(C)])
{
    if constexpr (sizeof(C) == 1) {
        return plh_unorm(v[0], 255.0f) | plh_unorm(v[1], 255.0f) << 8 |
               plh_unorm(v[2], 255.0f) << 16 | plh_unorm(v[3], 255.0f) << 24;
    } else {
        return plh_unorm16x2(v[0], v[1]);
    }
}

template <typename C, bool F32DST>
__global__ __launch_bounds__(DEINT_BW * DEINT_BH)
void k_deint_rows(const plh_pass p_)
{
    constexpr int N = 4 / sizeof(C);    // component slots per dword
    const plh_pass &p = plh_kernarg_pass();
    const plh_view &cur = p.s.src;
    const plh_deint_args &a = p.deint;
    const int seg = blockIdx.x * DEINT_BW + (int) (threadIdx.x % DEINT_BW);
    const int pair = blockIdx.y * DEINT_BH + (int) (threadIdx.x / DEINT_BW);
    const int nc = (cur.fmt - 1) % 3 == 2 ? 4 : (cur.fmt - 1) % 3 + 1;
    const int slots = cur.w * nc;       // of a row
    if (seg * N >= slots)
        return;
    const int yk = 2 * pair + a.keep, yr = 2 * pair + 1 - a.keep;

    auto row = [&](const plh_view &v, int y) {
        const int my = plh_wrap(y, v.h, PLH_ADDRESS_MIRROR);
        return *(const uint32_t *) ((const uint8_t *) v.ptr + (size_t) my * v.pitch + 4 * seg);
    };
    auto put = [&](int y, uint32_t raw, const float (&val)[N], bool have_raw) {
        if (y >= cur.h)
            return;
        uint8_t *dst = (uint8_t *) p.dst.ptr + (size_t) y * p.dst.pitch;
        if constexpr (F32DST) {
            float *o = (float *) dst + seg * N;
#pragma unroll
            for (int k = 0; k < N; k++) {
                if (seg * N + k < slots)
                    o[k] = have_raw ? deint_slot<C>(raw, k) : val[k];
            }
        } else {
            // (a row's last dword may reach into the pitch padding: both pitches are whole dwords)
            ((uint32_t *) dst)[seg] = have_raw ? raw : deint_pack<C>(val);
        }
    };

    const float none[N] = {};
    put(yk, row(cur, yk), none, true);
    if (yr >= cur.h)
        return;
    if (a.algo == PLH_DEINT_WEAVE) {
        put(yr, row(cur, yr), none, true);
    } else if (a.algo == PLH_DEINT_BOB) {
        put(yr, row(cur, yr + (a.keep ? 1 : -1)), none, true);
    } else if (a.intra_only) {
        const uint32_t r[4] = { row(cur, yr - 3), row(cur, yr - 1), row(cur, yr + 1), row(cur, yr + 3) };
        float out[N];
#pragma unroll
        for (int k = 0; k < N; k++) {
            const float col[4] = { deint_slot<C>(r[0], k), deint_slot<C>(r[1], k),
                                   deint_slot<C>(r[2], k), deint_slot<C>(r[3], k) };
            out[k] = bwdif_intra(col);
        }
        put(yr, 0, out, false);
    } else {
        const plh_view &prev2 = a.first ? a.prev : cur, &next2 = a.first ? cur : a.next;
        uint32_t rc[4], rp[2], rn[2], rp2[5], rn2[5];
#pragma unroll
        for (int k = 0; k < 4; k++)
            rc[k] = row(cur, yr + 2 * k - 3);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            rp[k] = row(a.prev, yr + 2 * k - 1);
            rn[k] = row(a.next, yr + 2 * k - 1);
        }
#pragma unroll
        for (int k = 0; k < 5; k++) {
            rp2[k] = row(prev2, yr + 2 * k - 4);
            rn2[k] = row(next2, yr + 2 * k - 4);
        }
        float out[N];
#pragma unroll
        for (int s = 0; s < N; s++) {
            deint_column t;
#pragma unroll
            for (int k = 0; k < 4; k++)
                t.cur[k] = deint_slot<C>(rc[k], s);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                t.prev[k] = deint_slot<C>(rp[k], s);
                t.next[k] = deint_slot<C>(rn[k], s);
            }
#pragma unroll
            for (int k = 0; k < 5; k++) {
                t.prev2[k] = deint_slot<C>(rp2[k], s);
                t.next2[k] = deint_slot<C>(rn2[k], s);
            }
            out[s] = bwdif_process(t);
        }
        put(yr, 0, out, false);
    }
}

// yadif on a whole plane: the same dword-of-a-row-pair scheme, with the rows above and below as
// WINDOWS of 2 R + 1 dwords (the spatial predictor looks 3 pixels = 3 NC slots to either side:
// R = ceil(3 NC / N) dwords; r8: 3 loads per row for four pixels where the general kernel issues
// 14 single-byte loads per pixel). Lanes whose window would leave the row -- the first and last R
// dwords, and a row's ragged tail -- take every tap through the mirrored per-slot fetch instead.
template <typename C>
DEV float deint_slot_at(const plh_view &v, int nc, int y, int slot)
{
    const int px = plh_wrap(slot >= 0 ? slot / nc : -((-slot + nc - 1) / nc), v.w, PLH_ADDRESS_MIRROR);
    const int c = ((slot % nc) + nc) % nc;
    const int my = plh_wrap(y, v.h, PLH_ADDRESS_MIRROR);
    const C raw = ((const C *) ((const uint8_t *) v.ptr + (size_t) my * v.pitch))[px * nc + c];
    if constexpr (sizeof(C) == 1)
        return plh_un8(raw);
    else
        return plh_un16(raw);
}

template <typename C, int NC, bool F32DST>
__global__ __launch_bounds__(DEINT_BW * DEINT_BH)
void k_deint_rows_yadif(const plh_pass p_)
{
    constexpr int N = 4 / sizeof(C);            // component slots per dword
    constexpr int R = (3 * NC + N - 1) / N;     // window radius in dwords
    const plh_pass &p = plh_kernarg_pass();
    const plh_view &cur = p.s.src;
    const plh_deint_args &a = p.deint;
    const int seg = blockIdx.x * DEINT_BW + (int) (threadIdx.x % DEINT_BW);
    const int pair = blockIdx.y * DEINT_BH + (int) (threadIdx.x / DEINT_BW);
    const int slots = cur.w * NC;
    if (seg * N >= slots)
        return;
    const int yk = 2 * pair + a.keep, yr = 2 * pair + 1 - a.keep;
    const plh_view &prev2 = a.first ? a.prev : cur, &next2 = a.first ? cur : a.next;

    auto row = [&](const plh_view &v, int y, int dw) {
        const int my = plh_wrap(y, v.h, PLH_ADDRESS_MIRROR);
        return *(const uint32_t *) ((const uint8_t *) v.ptr + (size_t) my * v.pitch + 4 * dw);
    };
    auto put = [&](int y, uint32_t raw, const float (&val)[N], bool have_raw) {
        if (y >= cur.h)
            return;
        uint8_t *dst = (uint8_t *) p.dst.ptr + (size_t) y * p.dst.pitch;
        if constexpr (F32DST) {
            float *o = (float *) dst + seg * N;
#pragma unroll
            for (int k = 0; k < N; k++) {
                if (seg * N + k < slots)
                    o[k] = have_raw ? deint_slot<C>(raw, k) : val[k];
            }
        } else {
            ((uint32_t *) dst)[seg] = have_raw ? raw : deint_pack<C>(val);
        }
    };

    const float none[N] = {};
    put(yk, row(cur, yk, seg), none, true);
    if (yr >= cur.h)
        return;

    float out[N];
    const int whole = slots / N;        // dwords that lie entirely inside the row
    if (seg - R >= 0 && seg + R < whole) {
        uint32_t up[2 * R + 1], dn[2 * R + 1];
#pragma unroll
        for (int k = 0; k < 2 * R + 1; k++) {
            up[k] = row(cur, yr - 1, seg - R + k);
            dn[k] = row(cur, yr + 1, seg - R + k);
        }
        const uint32_t rp[2] = { row(a.prev, yr - 1, seg), row(a.prev, yr + 1, seg) };
        const uint32_t rn[2] = { row(a.next, yr - 1, seg), row(a.next, yr + 1, seg) };
        const uint32_t rp2[3] = { row(prev2, yr - 2, seg), row(prev2, yr, seg), row(prev2, yr + 2, seg) };
        const uint32_t rn2[3] = { row(next2, yr - 2, seg), row(next2, yr, seg), row(next2, yr + 2, seg) };
#pragma unroll
        for (int s = 0; s < N; s++) {
            float u[7], d[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                constexpr int base = R * N;             // slot index of this lane's first slot in the window
                const int idx = base + s + (j - 3) * NC;
                u[j] = deint_slot<C>(up[idx / N], idx % N);
                d[j] = deint_slot<C>(dn[idx / N], idx % N);
            }
            deint_column t = {};
            t.cur[1] = u[3];
            t.cur[2] = d[3];
            t.prev[0] = deint_slot<C>(rp[0], s);  t.prev[1] = deint_slot<C>(rp[1], s);
            t.next[0] = deint_slot<C>(rn[0], s);  t.next[1] = deint_slot<C>(rn[1], s);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                t.prev2[k + 1] = deint_slot<C>(rp2[k], s);
                t.next2[k + 1] = deint_slot<C>(rn2[k], s);
            }
            out[s] = yadif_temporal(t, yadif_spatial(u, d, a.spatial_bias), a.skip_spatial_check != 0);
        }
    } else {
        // an edge lane: the sideways taps one by one through the mirror; the temporal taps only
        // look up and down, so they still come as dwords unless this is the row's ragged tail
        const bool tail = seg >= whole;
        uint32_t rp[2] = {}, rn[2] = {}, rp2[3] = {}, rn2[3] = {};
        if (!tail) {
#pragma unroll
            for (int k = 0; k < 2; k++) {
                rp[k] = row(a.prev, yr + 2 * k - 1, seg);
                rn[k] = row(a.next, yr + 2 * k - 1, seg);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                rp2[k] = row(prev2, yr + 2 * k - 2, seg);
                rn2[k] = row(next2, yr + 2 * k - 2, seg);
            }
        }
#pragma unroll
        for (int s = 0; s < N; s++) {
            const int slot = seg * N + s;
            float u[7], d[7];
#pragma unroll
            for (int j = 0; j < 7; j++) {
                u[j] = deint_slot_at<C>(cur, NC, yr - 1, slot + (j - 3) * NC);
                d[j] = deint_slot_at<C>(cur, NC, yr + 1, slot + (j - 3) * NC);
            }
            deint_column t = {};
            t.cur[1] = u[3];
            t.cur[2] = d[3];
#pragma unroll
            for (int k = 0; k < 2; k++) {
                t.prev[k] = tail ? deint_slot_at<C>(a.prev, NC, yr + 2 * k - 1, slot) : deint_slot<C>(rp[k], s);
                t.next[k] = tail ? deint_slot_at<C>(a.next, NC, yr + 2 * k - 1, slot) : deint_slot<C>(rn[k], s);
            }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                t.prev2[k + 1] = tail ? deint_slot_at<C>(prev2, NC, yr + 2 * k - 2, slot) : deint_slot<C>(rp2[k], s);
                t.next2[k + 1] = tail ? deint_slot_at<C>(next2, NC, yr + 2 * k - 2, slot) : deint_slot<C>(rn2[k], s);
            }
            out[s] = slot < slots
                   ? yadif_temporal(t, yadif_spatial(u, d, a.spatial_bias), a.skip_spatial_check != 0)
                   : 0.0f;
        }
    }
    put(yr, 0, out, false);
}

// whether the pass is what k_deint_rows does: a whole unorm plane into a texture of its own
// format, or of floats with the same components, every row a whole number of dwords
static int deint_rows_variant(const plh_pass *p)
{
    const plh_view &s = p->s.src, &d = p->dst;
    const plh_deint_args &a = p->deint;
    static int off = -1;
    if (off < 0)
        off = getenv("PL_HIP_DEINT_ROWS") && !atoi(getenv("PL_HIP_DEINT_ROWS"));
    if (off || p->num_ops || a.keep < 0 || s.fmt > PLH_FMT_RGBA16)
        return 0;
    if (a.algo == PLH_DEINT_YADIF && s.fmt > PLH_FMT_RG16)
        return 0;   // (rgba16: a window of 13 dwords per row for half a pixel: the general kernel)
    const int nc = (s.fmt - 1) % 3 == 2 ? 4 : (s.fmt - 1) % 3 + 1;
    const bool same = d.fmt == s.fmt, f32 = d.fmt == PLH_FMT_R32F + (s.fmt - 1) % 3;
    if ((!same && !f32) || p->s.comp_mask != (1u << nc) - 1u)
        return 0;
    if (p->base_x || p->base_y || p->dir_x != 1 || p->dir_y != 1 || p->transpose ||
        p->width != s.w || p->height != s.h || d.w != s.w || d.h != s.h)
        return 0;
    const plh_view *views[3] = { &s, &a.prev, &a.next };
    for (const plh_view *v : views) {
        if (v->fmt != s.fmt || v->w != s.w || v->h != s.h || v->pitch % 4 || (uintptr_t) v->ptr % 4)
            return 0;
    }
    if (d.pitch % 4 || (uintptr_t) d.ptr % 4)
        return 0;
    return (s.fmt <= PLH_FMT_RGBA8 ? 1 : 2) + (f32 ? 2 : 0);
}

extern "C" int plh_launch_deinterlace(plh_stream stream_, const struct plh_pass *pass)
{
    hipStream_t stream = (hipStream_t) stream_;
    const int pairs = (pass->height + 1) / 2;
    const dim3 block(DEINT_BW * DEINT_BH);
    if (const int variant = deint_rows_variant(pass)) {
        const int nc = (pass->s.src.fmt - 1) % 3 == 2 ? 4 : (pass->s.src.fmt - 1) % 3 + 1;
        const int dwords = (pass->s.src.w * nc * (variant & 1 ? 1 : 2) + 3) / 4;
        const dim3 grid((dwords + DEINT_BW - 1) / DEINT_BW, (pairs + DEINT_BH - 1) / DEINT_BH);
        if (pass->deint.algo == PLH_DEINT_YADIF) {
#define YADIF(C, NC) do { \
                if (variant > 2) PLH_LAUNCH_LAST((k_deint_rows_yadif<C, NC, true>), grid, block, 0, stream, *pass); \
                else PLH_LAUNCH_LAST((k_deint_rows_yadif<C, NC, false>), grid, block, 0, stream, *pass); \
            } while (0)
            if (variant & 1) {
                if (nc == 1) YADIF(uint8_t, 1); else if (nc == 2) YADIF(uint8_t, 2); else YADIF(uint8_t, 4);
            } else {
                if (nc == 1) YADIF(uint16_t, 1); else YADIF(uint16_t, 2);
            }
#undef YADIF
        } else switch (variant) {
        case 1: PLH_LAUNCH_LAST((k_deint_rows<uint8_t, false>), grid, block, 0, stream, *pass); break;
        case 2: PLH_LAUNCH_LAST((k_deint_rows<uint16_t, false>), grid, block, 0, stream, *pass); break;
        case 3: PLH_LAUNCH_LAST((k_deint_rows<uint8_t, true>), grid, block, 0, stream, *pass); break;
        default: PLH_LAUNCH_LAST((k_deint_rows<uint16_t, true>), grid, block, 0, stream, *pass); break;
        }
        const hipError_t err = hipGetLastError();
        return err == hipSuccess ? 0 : -(int) err;
    }
    const dim3 grid((pass->width + DEINT_BW - 1) / DEINT_BW, (pairs + DEINT_BH - 1) / DEINT_BH);
    if (plh_ops_lite(pass, 0, pass->num_ops))
        PLH_LAUNCH_LAST(k_deinterlace<true>, grid, block, 0, stream, *pass);
    else
        PLH_LAUNCH_LAST(k_deinterlace<false>, grid, block, 0, stream, *pass);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
