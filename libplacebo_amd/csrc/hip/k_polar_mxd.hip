/*
 * libplacebo-hip -- k_polar_mxd: the polar (EWA) 2 : 1 downscale as a tile contraction on the f16
 * matrix pipe (struct plh_polar_mx with enabled == 2, plh_device.h; the upscale is k_polar_mx.hiph,
 * whose header explains the numerics: f16 source tile = the reference's rgba16hf intermediate,
 * weights as f16 hi + lo halves, exact products, fp32 sums, first-order terms in the per-pixel
 * phase deviation).
 *
 * What the reference computes per output pixel (src/shaders/sampling.c:503-558, 723-783), for an
 * exact halving: the filter is widened by the ratio (radius 3.24 -> 6.48 texels: taps -6 .. 7 on
 * both axes, 148 inside the disc), the base texel of output (X, Y) is (2 X, 2 Y) + origin and
 * fcoord = 1/2 on both axes up to the fp32 rounding of pos * size (1e-3 at 8K):
 *     out[Y][X] = sum_j sum_i w'(j, i) S[2 Y + j][2 X + i]          (w' = w * scale / wsum)
 * With m = output row, n = output column of a 16 x 16 wave tile and k = source column:
 *     out[m][n] = sum_j sum_k S[2 m + j][k] * T_j[k - 2 n]
 * -- per source row offset j a GEMM whose A operand is 16 (every second) rows x 32 columns of the
 * tile and whose B operand is a constant banded matrix; the 16 outputs need k in [0, 44): two
 * 32-column blocks. Per j and block: hi, lo (+ dfx(X) * d/dx, folded in registers: v_pk_fma_f16)
 * into one accumulator set, dfy-term into a second one that the epilogue folds in with the row's
 * own dfy(Y). At fcoord = (1/2, 1/2) the weights of rows j and 13 - j are the same (d/dy: opposite),
 * so B is stored for j < 7 only: 56 fragments = 56 KiB of LDS.
 *
 * Shape. A workgroup of 8 waves renders 64 x 32 outputs (4 x 2 wave tiles) from a 140 x 76 source
 * tile, channel-planar f16 with a row pitch of 304 bytes (16-byte aligned rows; every second row
 * of 16 lanes + the 16-byte column step fall on distinct bank groups: conflict-free
 * ds_read_b128). LDS 56 + 67.7 KiB: one workgroup per CU, 2 waves per SIMD with the whole
 * register file. Per wave: 14 x 2 x (3 A reads + 9 MFMAs) + 56 B reads = 252 v_mfma_f32_16x16x32_f16.
 * Handled: an rgba16hf (BASELINE configs[4]) or rgba16 source without fused pre-ops or behind an
 * identity PLANE_MAP, RGB; an rgba16hf target without post-ops, or an rgba16 target behind the fused
 * epilogue (dither + scale). Round 4: the passes of a LINEAR-LIGHT downscale -- the reference
 * linearises in front of a downscaler (src/renderer.c:1997-2003) -- i.e. an rgba16 source with
 * PLANE_MAP + LINEARIZE as fused pre-ops (the reference's PASS A: plane -> linear rgba16hf
 * intermediate; HDR 8K -> 4K without debanding, into the intermediate the measurement reads) and a
 * DELINEARIZE in front of the fused epilogue (SDR 4K -> 1080p in linear light). Everything else
 * stays on k_polar_pp.
 */
#include "polar_common.hiph"
#include "transfer.hiph"

#define MXD_TW      64                      // output columns per workgroup tile
#define MXD_TH      32                      // output rows
#define MXD_SRC_W   (2 * MXD_TW + 12)       // 140 source columns
#define MXD_SRC_H   (2 * MXD_TH + 12)       // 76 source rows
#define MXD_PITCH   304                     // bytes per tile row of one channel (16 * 19)
#define MXD_PLANE   (MXD_SRC_H * MXD_PITCH)
#define MXD_B_BYTES (PLH_MXD_NFRAG * 64 * 16)
#define MXD_NT      512
#define MXD_HP      (MXD_SRC_W / 2)         // texel pairs per tile row
#define MXD_NPAIRS  (MXD_HP * MXD_SRC_H)
#define MXD_NV      ((MXD_NPAIRS + MXD_NT - 1) / MXD_NT)

typedef _Float16 mxd_f16x8 __attribute__((ext_vector_type(8)));
typedef float mxd_f32x4 __attribute__((ext_vector_type(4)));

// UNORM: an rgba16 source (decoded and rounded to f16 while staged: the reference's rgba16hf
// intermediate, fused) instead of an rgba16hf one; F16DST: an rgba16hf target, else rgba16 through
// the fused epilogue (dither + scale, fastepi.hiph).
// PRE: a LINEARIZE pre-op (the last one) on the staged texels -- UNORM sources only; DELIN: a
// DELINEARIZE as the first post-op.
template <bool UNORM, bool F16DST, bool PRE = false, bool DELIN = false>
__global__ __launch_bounds__(MXD_NT)
void k_polar_mxd(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const plh_polar_mx &mx = s.mx;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *bl = smem;
    unsigned char *tile = smem + MXD_B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // profiling aid (PL_HIP_PP_DEBUG: 1 = no contraction, 4 = no stores, 8 = no tile loads): only in a
    // library built with -DPLH_MX_DEBUG (k_polar_mx.hiph says what the switches cost)
#ifdef PLH_MX_DEBUG
    const int dbg = s.pp_debug;
#else
    constexpr int dbg = 0;
#endif
    int j0 = mx.row_first[0];       // first source row with a weight (uniform)
    asm volatile("" : "+s"(j0));

    // Persistent workgroups, one per CU: the B fragments are loaded once, and the loads of the next
    // tile are in flight (in registers) while this one is contracted. Workgroups go to the 8 XCDs
    // round-robin and every XCD has its own L2: XCD x works on the x-th contiguous eighth of the
    // tiles (row-major), its workgroups side by side on consecutive tiles -- neighbours share 12
    // source rows / columns of halo.
    const int tiles_x = (p.width + MXD_TW - 1) / MXD_TW;
    const int tiles_y = (p.height + MXD_TH - 1) / MXD_TH;
    int band_first, band_size, stride, first;
    {
        const uint32_t total = (uint32_t) tiles_x * (uint32_t) tiles_y, groups = gridDim.x, lin = blockIdx.x;
        const uint32_t q = total >> 3, r = total & 7u, xcd = lin & 7u;
        band_first = (int) (xcd * q + min(xcd, r));
        band_size = (int) (q + (xcd < r ? 1u : 0u));
        stride = (int) ((groups - xcd + 7u) >> 3);      // workgroups of this XCD
        first = (int) (lin >> 3);
    }

    // ---- B fragments: global (L2 resident) -> LDS, one global_load_lds_dwordx4 per fragment -------
#pragma unroll
    for (int f = wave; f < PLH_MXD_NFRAG; f += MXD_NT / 64) {
        const unsigned char *g = (const unsigned char *) mx.bfrag + ((size_t) f * 64 + lane) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                         (__attribute__((address_space(3))) void *) (bl + f * 1024), 16, 0, 0);
    }
    // columns 140 .. 151 of every tile row (and 16 bytes behind the tile) are read by the last
    // column block, where they only meet zero weights: make them zeros, not whatever LDS held
    for (int i = tid; i < 3 * MXD_SRC_H; i += MXD_NT) {
        uint32_t *d = (uint32_t *) (tile + i * MXD_PITCH + MXD_SRC_W * 2);
#pragma unroll
        for (int k = 0; k < (MXD_PITCH - MXD_SRC_W * 2) / 4; k++)
            d[k] = 0;
    }
    if (tid < 4)
        ((uint32_t *) (tile + 3 * MXD_PLANE))[tid] = 0;

    int sw = s.src.w, sh = s.src.h, u_spitch = s.src.pitch;
    uintptr_t u_sptr = (uintptr_t) s.src.ptr, u_dfx = (uintptr_t) mx.dfx, u_dfy = (uintptr_t) mx.dfy;
    asm volatile("" : "+s"(sw), "+s"(sh), "+s"(u_spitch), "+s"(u_sptr), "+s"(u_dfx), "+s"(u_dfy));

    // texel pair u of this lane: tile row, pair within the row (the same for every tile)
    int ty[MXD_NV], tp[MXD_NV];
#pragma unroll
    for (int u = 0; u < MXD_NV; u++) {
        const int i = min(tid + u * MXD_NT, MXD_NPAIRS - 1);
        ty[u] = (int) (((float) i + 0.5f) * (1.0f / (float) MXD_HP));   // exact: i < 2^22
        tp[u] = i - ty[u] * MXD_HP;
    }
    // the source tile of workgroup tile t: pairs of horizontally adjacent rgba16hf texels, one
    // 16-byte load each, all issued together
    const int ln = lane & 15, lg = lane >> 4;
    const int wc = wave & 3, wr = wave >> 2;
    // ... and with them the phase deviations of the lane's output column and four output rows:
    // everything a tile needs from memory arrives together, one wait at the top of its turn
    uint4 v[MXD_NV];
    float nx_dfx, nx_dfy[4];
    typedef __attribute__((address_space(1))) const float mxd_gfloat;
    auto tile_load = [&](int t) {
        const int by = (int) ((uint32_t) t / (uint32_t) tiles_x), bx = t - by * tiles_x;
        const int ox = mx.org_x + 2 * MXD_TW * bx, oy = mx.org_y + 2 * MXD_TH * by;
#pragma unroll
        for (int u = 0; u < MXD_NV; u++) {
            const int sy = min(max(oy + ty[u], 0), sh - 1);
            const int px = min(max(ox + 2 * tp[u], 0), sw - 2);
            const plh_u32x4 q = *(const __attribute__((address_space(1))) plh_u32x4 *)
                                    (u_sptr + (size_t) sy * (size_t) u_spitch + (size_t) px * 8);
            v[u] = make_uint4(q.x, q.y, q.z, q.w);
        }
        nx_dfx = ((mxd_gfloat *) u_dfx)[MXD_TW * bx + 16 * wc + ln];
#pragma unroll
        for (int r = 0; r < 4; r++)
            nx_dfy[r] = ((mxd_gfloat *) u_dfy)[MXD_TH * by + 16 * wr + 4 * lg + r];
    };
    // A fragment of lane l for source row offset j, column block kb: the 16 bytes at tile row
    // 2 * (16 wr + ln) + j, column 32 wc + 32 kb + 8 lg
    const unsigned char *ab = tile + (2 * (16 * wr + ln)) * MXD_PITCH + (32 * wc + 8 * lg) * 2;
    const unsigned char *bfl = bl + lane * 16;

    for (int u = 0; u < MXD_NV; u++)
        v[u] = make_uint4(tid, u, 0, 0);
    nx_dfx = 0.0f;
    for (int r = 0; r < 4; r++)
        nx_dfy[r] = 0.0f;
    if (first < band_size && !(dbg & 8))
        tile_load(band_first + first);
#pragma unroll 1
    for (int it = first; it < band_size; it += stride) {
        const int t = band_first + it;
        const int by = (int) ((uint32_t) t / (uint32_t) tiles_x), bx = t - by * tiles_x;
        const int ox = mx.org_x + 2 * MXD_TW * bx;

        // ---- registers -> LDS: the f16 codes move bit for bit into the channel planes ------------
        if (ox < 0 || ox + MXD_SRC_W > sw) {
            // a pair at clamped positions: beyond the left edge both texels are the pair's first,
            // beyond the right edge both its second
#pragma unroll
            for (int u = 0; u < MXD_NV; u++) {
                const uint4 w = v[u];
                const int sx = ox + 2 * tp[u];
                const bool ldup = sx < 0, hdup = sx > sw - 2;
                const uint32_t ax = hdup ? w.z : w.x, ay = hdup ? w.w : w.y;
                const uint32_t bx_ = ldup ? w.x : w.z, by_ = ldup ? w.y : w.w;
                v[u] = make_uint4(ax, ay, bx_, by_);
            }
        }
#pragma unroll
        for (int u = 0; u < MXD_NV; u++) {
            const uint4 w = v[u];
            if (tid + u * MXD_NT < MXD_NPAIRS) {
                unsigned char *d = tile + ty[u] * MXD_PITCH + tp[u] * 4;
                if (UNORM && PRE) {
                    // the raw codes, converted in place below
                    *(uint32_t *) d = (w.x & 0xffffu) | (w.z << 16);
                    *(uint32_t *) (d + MXD_PLANE) = (w.x >> 16) | (w.z & 0xffff0000u);
                    *(uint32_t *) (d + 2 * MXD_PLANE) = (w.y & 0xffffu) | (w.w << 16);
                } else if (UNORM) {
                    // decode to fp32, THEN round to f16 (what the rgba16hf store + load of the unfused
                    // pass does). The two roundings are kept apart on purpose: left alone the compiler
                    // folds the last fma of the decode into v_fma_mixlo_f16, which rounds once -- one
                    // f16 ulp off for the codes whose fp32 value is a tie (65519 -> 1 - 2^-12).
                    float t[6] = { plh_un16(w.x & 0xffffu), plh_un16(w.z & 0xffffu), plh_un16(w.x >> 16),
                                   plh_un16(w.z >> 16), plh_un16(w.y & 0xffffu), plh_un16(w.w & 0xffffu) };
#pragma unroll
                    for (int k = 0; k < 6; k++)
                        asm volatile("" : "+v"(t[k]));
                    *(uint32_t *) d = (uint32_t) plh_f2h(t[0]) | ((uint32_t) plh_f2h(t[1]) << 16);
                    *(uint32_t *) (d + MXD_PLANE) = (uint32_t) plh_f2h(t[2]) | ((uint32_t) plh_f2h(t[3]) << 16);
                    *(uint32_t *) (d + 2 * MXD_PLANE) = (uint32_t) plh_f2h(t[4]) | ((uint32_t) plh_f2h(t[5]) << 16);
                } else {
                    *(uint32_t *) d = (w.x & 0xffffu) | (w.z << 16);
                    *(uint32_t *) (d + MXD_PLANE) = (w.x >> 16) | (w.z & 0xffff0000u);
                    *(uint32_t *) (d + 2 * MXD_PLANE) = (w.y & 0xffffu) | (w.w << 16);
                }
            }
        }

        if constexpr (PRE) {
            // decode + linearize + f16 rounding (= the reference's PASS A into its rgba16hf
            // intermediate) of the texel pairs this lane has just stored: a rolled loop, so that the
            // curve's code exists once (LDS keeps a wave's own accesses in order: no barrier)
            const plh_op &lop = p.ops[p.num_pre_ops - 1];
#pragma unroll 1
            for (int u = 0; u < MXD_NV; u++) {
                const int i = tid + u * MXD_NT;
                if (i >= MXD_NPAIRS)
                    break;
                const int y = (int) (((float) i + 0.5f) * (1.0f / (float) MXD_HP));
                unsigned char *d = tile + y * MXD_PITCH + (i - y * MXD_HP) * 4;
                const uint32_t q0 = *(const uint32_t *) d, q1 = *(const uint32_t *) (d + MXD_PLANE),
                               q2 = *(const uint32_t *) (d + 2 * MXD_PLANE);
                float c[6] = { plh_un16(q0 & 0xffffu), plh_un16(q1 & 0xffffu), plh_un16(q2 & 0xffffu),
                               plh_un16(q0 >> 16), plh_un16(q1 >> 16), plh_un16(q2 >> 16) };
                op_linearize_values(c, lop);
                // (fp32 result, THEN the f16 rounding: see the decode above)
                asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]));
                *(uint32_t *) d = (uint32_t) plh_f2h(c[0]) | ((uint32_t) plh_f2h(c[3]) << 16);
                *(uint32_t *) (d + MXD_PLANE) = (uint32_t) plh_f2h(c[1]) | ((uint32_t) plh_f2h(c[4]) << 16);
                *(uint32_t *) (d + 2 * MXD_PLANE) = (uint32_t) plh_f2h(c[2]) | ((uint32_t) plh_f2h(c[5]) << 16);
            }
        }

        // ---- the wave's 16 x 16 outputs ----------------------------------------------------------
        const int X = MXD_TW * bx + 16 * wc + ln;           // the lane's output column (B / D operand: n = ln)
        const int Y0 = MXD_TH * by + 16 * wr + 4 * lg;      // its four output rows Y0 .. Y0 + 3 (D: m = 4 lg + r)
        const _Float16 dxh = (_Float16) nx_dfx;
        const mxd_f16x8 dx8 = { dxh, dxh, dxh, dxh, dxh, dxh, dxh, dxh };
        float dfy[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            dfy[r] = nx_dfy[r];
        __syncthreads();
        // the next tile's texels: asked for now, used after this tile's contraction
        if (it + stride < band_size && !(dbg & 8))
            tile_load(t + stride);

        // Accumulators: per channel, one set for source rows j < 7 and one for their mirror rows
        // 13 - j -- which share the B fragments, d/dy with the opposite sign (the epilogue subtracts)
        // -- so that consecutive MFMAs never wait for each other's result. The fragments of the
        // next (j, kb) step are read from LDS while this step's 18 MFMAs run.
        mxd_f32x4 acc[2][3], accy[2][3];
        // (zeroed inside the contraction's branch: the zeros are then the first step's inline
        // constant instead of 48 v_mov per tile and lane -- k_polar_mx.hiph says how the compiler
        // hoists them otherwise)
        auto init_acc = [&](float v) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++) {
                    acc[h][ch] = (mxd_f32x4) (v);
                    accy[h][ch] = (mxd_f32x4) (v);
                }
            }
        };
        struct fragset { mxd_f16x8 bhi, blo, bdy, a[2][3]; };
        auto read_frags = [&](fragset &f, int j, int kb) {
            const unsigned char *bf = bfl + 4 * (2 * j + kb) * 1024;
            f.bhi = *(const mxd_f16x8 *) bf;
            // (the column-phase term rides on the lo half, k_polar_mx.hiph)
            f.blo = __builtin_elementwise_fma(*(const mxd_f16x8 *) (bf + 2048), dx8, *(const mxd_f16x8 *) (bf + 1024));
            f.bdy = *(const mxd_f16x8 *) (bf + 3072);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const unsigned char *a0 = ab + (h ? PLH_MXD_TAPS - 1 - j : j) * MXD_PITCH + 64 * kb;
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    f.a[h][ch] = *(const mxd_f16x8 *) (a0 + ch * MXD_PLANE);
            }
        };
        auto contract = [&](const fragset &f) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    acc[h][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[h][ch], f.bhi, acc[h][ch], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    accy[h][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[h][ch], f.bdy, accy[h][ch], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    acc[h][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.a[h][ch], f.blo, acc[h][ch], 0, 0, 0);
            }
        };
        if (dbg & 1) {
            init_acc((float) (tid & 1) * 0.25f);
        } else {
            init_acc(0.0f);
            fragset f0, f1;
            // (from the first source row that carries a weight -- mx.row_first[0]: row 1 for every
            // radius <= 3.25, six row steps = 216 MFMAs per wave instead of seven = 252)
            read_frags(f0, j0, 0);
            // (first step peeled: its accumulators are the constant 0)
            read_frags(f1, j0, 1);
            contract(f0);
            read_frags(f0, min(j0 + 1, PLH_MXD_TAPS / 2 - 1), 0);
            contract(f1);
#pragma unroll 1
            for (int j = j0 + 1; j < PLH_MXD_TAPS / 2; j++) {
                read_frags(f1, j, 1);
                contract(f0);
                read_frags(f0, min(j + 1, PLH_MXD_TAPS / 2 - 1), 0);    // (the last one is not used)
                contract(f1);
            }
        }

        // ---- epilogue: rgba16hf store, guarded (dispatch.c:1126-1142) ----------------------------
        const int cpos = p.base_x + p.dir_x * X;
        const bool cok = X < p.width && p.out_scale[0] * (float) X < 1.0f && cpos >= 0 && cpos < p.dst.w;
        float oall[12];
#pragma unroll
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++)
                oall[3 * r + ch] = __builtin_fmaf(dfy[r], accy[0][ch][r] - accy[1][ch][r], acc[0][ch][r] + acc[1][ch][r]);
        }
        if constexpr (DELIN)
            op_delinearize_values(oall, p.ops[p.num_pre_ops]);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int Y = Y0 + r;
            const int rpos = p.base_y + p.dir_y * Y;
            const bool ok = cok && Y < p.height && p.out_scale[1] * (float) Y < 1.0f && rpos >= 0 && rpos < p.dst.h;
            float o[3] = { oall[3 * r], oall[3 * r + 1], oall[3 * r + 2] };
            plh_u32x2 px;
            if (F16DST) {
                px.x = (uint32_t) plh_f2h(o[0]) | ((uint32_t) plh_f2h(o[1]) << 16);
                px.y = (uint32_t) plh_f2h(o[2]) | 0x3c000000u;      // alpha = 1 (not sampled)
            } else {
                // op_dither (plain path) and the SCALE op, as k_polar_mx's fast epilogue
                float a = 1.0f;
                if (p.epi.has_dither) {
                    const int ix = (X + p.frag_x0) & p.epi.mask, iy = (Y + p.frag_y0) & p.epi.mask;
                    const float b = p.epi.matrix[iy * p.epi.size + ix], ds = p.epi.dscale, di = p.epi.dinv;
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                        o[ch] = __builtin_floorf(ds * o[ch] + b) * di;
                    a = __builtin_floorf(ds * a + b) * di;
                }
                if (p.epi.has_scale) {
#pragma unroll
                    for (int ch = 0; ch < 3; ch++)
                        o[ch] *= p.epi.scale;
                    a *= p.epi.scale;
                }
                px.x = plh_unorm16x2(o[0], o[1]);
                px.y = plh_unorm16x2(o[2], a);
            }
            if (ok && !(dbg & 4)) {
                plh_u32x2 *d = (plh_u32x2 *) ((char *) p.dst.ptr + (size_t) rpos * p.dst.pitch + (size_t) cpos * 8);
                // (an rgba16 target is a final frame: streamed; an rgba16hf one is the intermediate
                // the next pass reads: cached)
                if constexpr (F16DST)
                    *d = px;
                else
                    __builtin_nontemporal_store(px, d);
            }
        }
        __syncthreads();    // (everyone is done with this tile before the next one overwrites it)
    }
}

// the pass k_polar_mxd is written for (file header); fills pass->epi for an rgba16 target
bool plh_polar_mxd_applies(plh_pass *pass)
{
    const plh_sampler_args &s = pass->s;
    if (s.mx.enabled != 2 || (s.comp_mask & 0xf) != 0x7 || pass->transpose ||
        (s.src.fmt != PLH_FMT_RGBA16F && s.src.fmt != PLH_FMT_RGBA16))
        return false;
    // pre-ops: [an identity PLANE_MAP of a plane that carries r, g, b (changes none)] [LINEARIZE]
    int i = 0;
    if (i < pass->num_pre_ops && pass->ops[i].kind == PLH_OP_PLANE_MAP) {
        const plh_op &op = pass->ops[i];
        if (!op.i2 || op.i1 < 3)
            return false;
        i++;
    }
    if (i < pass->num_pre_ops && pass->ops[i].kind == PLH_OP_LINEARIZE && s.src.fmt == PLH_FMT_RGBA16)
        i++;
    if (i != pass->num_pre_ops)
        return false;
    // post-ops: [DELINEARIZE], then nothing / a SCALE by exactly one (what encoding into a float
    // target records) into an rgba16hf target, or the fused epilogue into an rgba16 one
    int first = pass->num_pre_ops;
    if (first < pass->num_ops && pass->ops[first].kind == PLH_OP_DELINEARIZE)
        first++;
    if (pass->dst.fmt == PLH_FMT_RGBA16F) {
        const int post = pass->num_ops - first;
        if (post > 1)
            return false;
        if (post == 1) {
            const plh_op &op = pass->ops[first];
            if (op.kind != PLH_OP_SCALE || op.f[0] != 1.0f || op.f[1] != 1.0f || op.f[2] != 1.0f || op.f[3] != 1.0f)
                return false;
        }
        return true;
    }
    if (pass->dst.fmt != PLH_FMT_RGBA16)
        return false;
    plh_match_fast_epilogue(pass, false, first);    // [DITHER] [SCALE] -> rgba16
    return pass->epi.enabled && !pass->epi.has_alpha;
}

int plh_launch_polar_mxd(hipStream_t stream, const plh_pass *pass)
{
    const int tiles_x = (pass->width + MXD_TW - 1) / MXD_TW;
    const int tiles_y = (pass->height + MXD_TH - 1) / MXD_TH;
    const size_t shmem = MXD_B_BYTES + (size_t) 3 * MXD_PLANE + 16;
    // one persistent workgroup per CU (LDS: 124 KiB each) of the device that owns the stream
    int cus = 256;
    (void) plh_stream_device((plh_stream) stream, &cus);
    const int groups = tiles_x * tiles_y < cus ? tiles_x * tiles_y : cus;
    const bool unorm = pass->s.src.fmt == PLH_FMT_RGBA16, f16dst = pass->dst.fmt == PLH_FMT_RGBA16F;
    const bool pre = pass->num_pre_ops && pass->ops[pass->num_pre_ops - 1].kind == PLH_OP_LINEARIZE;
    const bool delin = pass->num_pre_ops < pass->num_ops && pass->ops[pass->num_pre_ops].kind == PLH_OP_DELINEARIZE;
#define MXD_LAUNCH(U, F, P, D) do { \
        static uint64_t lds_done; \
        const int e = plh_kernel_needs_lds((const void *) k_polar_mxd<U, F, P, D>, (plh_stream) stream, shmem, &lds_done); \
        if (e) \
            return e; \
        PLH_LAUNCH_LAST((k_polar_mxd<U, F, P, D>), dim3(groups), dim3(MXD_NT), shmem, stream, *pass); \
    } while (0)
    if (pre || delin) {
        // the passes of a linear-light downscale
        if (unorm && pre && f16dst && !delin)           MXD_LAUNCH(true, true, true, false);
        else if (unorm && pre && f16dst)                MXD_LAUNCH(true, true, true, true);
        else if (unorm && pre && delin)                 MXD_LAUNCH(true, false, true, true);
        else if (unorm && pre)                          MXD_LAUNCH(true, false, true, false);
        else if (!unorm && !pre && !f16dst)             MXD_LAUNCH(false, false, false, true);
        else if (!unorm && !pre)                        MXD_LAUNCH(false, true, false, true);
        else
            return -1000;
    }
    else if (unorm && f16dst)  MXD_LAUNCH(true, true, false, false);
    else if (unorm)       MXD_LAUNCH(true, false, false, false);
    else if (f16dst)      MXD_LAUNCH(false, true, false, false);
    else                  MXD_LAUNCH(false, false, false, false);
#undef MXD_LAUNCH
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
