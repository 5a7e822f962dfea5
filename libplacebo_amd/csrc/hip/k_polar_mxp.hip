/*
 * libplacebo-hip -- k_polar_mxp: k_polar_mx (the polar 2x upscale as a tile contraction on the f16
 * matrix pipe: k_polar_mx.hiph explains the numerics, the fragment layout and the live-row pairing)
 * on PERSISTENT workgroups, for the shapes that carry the benchmark configurations: RGB tiles of 8
 * wave-tile columns, an rgba16 / rgba16hf source whose fused pre-ops need no transcendental, and
 * the fused epilogue (dither + scale) or the map chain of an HDR pass behind the contraction.
 *
 * Why. With the contraction on its live rows (54 MFMAs per wave tile) and 49 VALU instructions per
 * pixel, BASELINE configs[2] spent its time neither in the vector pipe nor in the matrix pipe but
 * in what is NOT overlapped (profiles/r06_05_mx_floor.txt, the debug-switch build: 28.7 us; without
 * the stores 22.0, without the tile loads 23.0, without the contraction 26.2, with none of the
 * three 15.0): a workgroup loads its tile, waits, computes, stores and ends -- its LDS and wave
 * slots are released when the last store has been acknowledged, and only then does the next
 * workgroup start loading. Two workgroups per CU do not hide that for each other.
 * Here 2 x (CUs) workgroups stay resident and walk over the tiles of their XCD's band:
 *   - the B fragments are copied to LDS once per workgroup, not once per tile;
 *   - the texels of the NEXT tile are requested (into registers: twelve) as soon as this tile's
 *     texels are in LDS, and arrive while this tile is contracted and stored;
 *   - the stores of a tile are never waited for: the wave goes on to the next tile.
 * The row-phase term is folded phase by phase (k_polar_mx.hiph: YPHASE) in every variant, which
 * frees the registers the prefetch needs and costs the epilogue nothing.
 *
 * Same arithmetic as k_polar_mx<3, true, POST, 8> with YPHASE: its CHAIN variants bit for bit, its
 * FAST variant up to where the row-phase term is added (before instead of inside the epilogue:
 * an fp32 ulp, the statement "one code of k_polar_pp" is unchanged). PL_HIP_MX_PERSIST=0 keeps
 * k_polar_mx (tests compare the two).
 */
#include "k_polar_mx.hiph"

template <int POST>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
void k_polar_mxp(const plh_pass p_)
{
    constexpr int NCH = 3, WTC = 8;
    using G = mx_geom<WTC>;
    constexpr bool FAST = POST == MX_POST_FAST;
    constexpr bool CHAIN = POST == MX_POST_CHAIN || POST == MX_POST_CHAIN_CR, CR = POST == MX_POST_CHAIN_CR;
    static_assert(FAST || CHAIN, "the fused epilogue or the map chain");
    constexpr int NT = G::threads, PITCH = G::pitch, PLANE = G::plane, NV = G::nv;
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const plh_polar_mx &mx = s.mx;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *bl = smem;
    unsigned char *tile = smem + MX_B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // uniforms, read once and pinned in SGPRs (k_polar_mx.hiph says why)
    int u_height = p.height, u_dst_h = p.dst.h, u_base_y = p.base_y, u_dir_y = p.dir_y;
    int u_dpitch = p.dst.pitch, u_nt = p.nt_store, u_fx0 = p.frag_x0, u_fy0 = p.frag_y0;
    int u_has_dither = p.epi.has_dither, u_has_scale = p.epi.has_scale, u_emask = p.epi.mask, u_esize = p.epi.size;
    float u_osy = p.out_scale[1], u_ds = p.epi.dscale, u_di = p.epi.dinv, u_sc = p.epi.scale;
    uintptr_t u_dptr = (uintptr_t) p.dst.ptr, u_matrix = (uintptr_t) p.epi.matrix, u_dfy = (uintptr_t) mx.dfy;
    uintptr_t u_matrix_t = (uintptr_t) p.epi.matrix_t, u_dfx = (uintptr_t) mx.dfx;
    asm volatile("" : "+s"(u_height), "+s"(u_dst_h), "+s"(u_base_y), "+s"(u_dir_y), "+s"(u_dpitch),
                      "+s"(u_nt), "+s"(u_fx0), "+s"(u_fy0), "+s"(u_has_dither), "+s"(u_has_scale),
                      "+s"(u_emask), "+s"(u_esize));
    asm volatile("" : "+s"(u_osy), "+s"(u_ds), "+s"(u_di), "+s"(u_sc), "+s"(u_dptr), "+s"(u_matrix), "+s"(u_dfy),
                      "+s"(u_matrix_t), "+s"(u_dfx));
    typedef __attribute__((address_space(1))) const float mx_gfloat;
    typedef __attribute__((address_space(1))) const mx_f32x4 mx_gf4;

    // Workgroups go to the 8 XCDs round-robin in launch order and every XCD has its own L2: XCD x
    // works on the x-th contiguous eighth of the tiles (row-major), its workgroups side by side on
    // consecutive tiles -- neighbours share 8 source rows / columns of halo.
    const int tiles_x = (p.width + G::tile_w - 1) / G::tile_w;
    const int tiles_y = (p.height + MX_TILE_H - 1) / MX_TILE_H;
    int band_first, band_size, stride, first;
    {
        const uint32_t total = (uint32_t) tiles_x * (uint32_t) tiles_y, groups = gridDim.x, lin = blockIdx.x;
        const uint32_t q = total >> 3, r = total & 7u, xcd = lin & 7u;
        band_first = (int) (xcd * q + min(xcd, r));
        band_size = (int) (q + (xcd < r ? 1u : 0u));
        stride = (int) ((groups - xcd + 7u) >> 3);      // workgroups of this XCD
        first = (int) (lin >> 3);
    }

    // ---- B fragments: global (L2 resident) -> LDS, once per workgroup ---------------------------
    const int nfrag = 8 * mx.npairs;
#pragma unroll
    for (int f = wave; f < PLH_MX_NFRAG; f += NT / 64) {
        if (f >= nfrag)
            break;
        const unsigned char *g = (const unsigned char *) mx.bfrag + ((size_t) f * 64 + lane) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                         (__attribute__((address_space(3))) void *) (bl + f * 1024), 16, 0, 0);
    }

    int sw = s.src.w, sh = s.src.h, u_sfmt = s.src.fmt, u_spitch = s.src.pitch, u_npre = p.num_pre_ops;
    uintptr_t u_sptr = (uintptr_t) s.src.ptr;
    int u_org_x = mx.org_x, u_org_y = mx.org_y;
    asm volatile("" : "+s"(sw), "+s"(sh), "+s"(u_sfmt), "+s"(u_spitch), "+s"(u_npre), "+s"(u_sptr),
                      "+s"(u_org_x), "+s"(u_org_y));
    // (the launcher only takes rgba16 / rgba16hf sources here)
    const bool raw16 = u_sfmt == PLH_FMT_RGBA16F && !u_npre;
    const bool unorm = u_sfmt == PLH_FMT_RGBA16;
    const bool simple = u_npre == 0 || (u_npre == 1 && p.ops[0].kind == PLH_OP_PLANE_MAP && p.ops[0].i2);

    // texel pair u of this lane: tile row, pair within the row -- the same for every tile
    // (one register per pair: row in the upper half -- they live across the whole tile loop)
    uint32_t tyx[NV];
#pragma unroll
    for (int u = 0; u < NV; u++) {
        const int i = min(tid + u * NT, G::npairs - 1);
        const int y = (int) (((float) i + 0.5f) * (1.0f / (float) G::hp));    // exact: i < 2^22
        tyx[u] = ((uint32_t) y << 16) | (uint32_t) (i - y * G::hp);
    }
#define ty(u) ((int) (tyx[u] >> 16))
#define tp(u) ((int) (tyx[u] & 0xffffu))
    const int ln = lane & 15, lg = lane >> 4;

    // what a tile needs from memory, requested together: its texel pairs (one 16-byte load each,
    // clamped addressing: sampling.c:45-181) and the lane's column-phase deviation
    uint4 v[NV];
    float nx_dfx;
    auto tile_load = [&](int t) {
        const int tby = (int) ((uint32_t) t / (uint32_t) tiles_x), tbx = t - tby * tiles_x;
        const int ox = u_org_x + 8 * WTC * tbx, oy = u_org_y + 16 * MX_WT_ROWS * tby;
#pragma unroll
        for (int u = 0; u < NV; u++) {
            const int sy = min(max(oy + ty(u), 0), sh - 1);
            const int px = min(max(ox + 2 * tp(u), 0), sw - 2);
            const plh_u32x4 q = *(const __attribute__((address_space(1))) plh_u32x4 *)
                                    (u_sptr + (size_t) sy * (size_t) u_spitch + (size_t) px * 8);
            v[u] = make_uint4(q.x, q.y, q.z, q.w);
        }
        // (the tables are padded to whole tiles)
        nx_dfx = ((mx_gfloat *) u_dfx)[G::tile_w * tbx + 16 * wave + ln];
    };
    auto pair_store = [&](int ty_, int tp_, uint32_t o0, uint32_t o1, uint32_t o2) {
        unsigned char *d = tile + ty_ * PITCH + tp_ * 4;
        *(uint32_t *) d = o0;
        *(uint32_t *) (d + PLANE) = o1;
        *(uint32_t *) (d + 2 * PLANE) = o2;
    };

    const unsigned char *bfl = bl + lane * 16;
    int u_row0 = mx.row_first[0], u_row1 = mx.row_first[1], u_npairs = mx.npairs;
    asm volatile("" : "+s"(u_row0), "+s"(u_row1), "+s"(u_npairs));
    const bool u_np4 = u_npairs > 3;
    const unsigned char *bfl1 = bfl + u_npairs * 4096;      // row phase 1's fragments
    const bool u_tcol = FAST && u_has_dither && u_matrix_t && (u_fy0 & 7) == 0 && u_esize >= 8;

#pragma unroll
    for (int u = 0; u < NV; u++)
        v[u] = make_uint4(0, 0, 0, 0);
    nx_dfx = 0.0f;
    if (first < band_size)
        tile_load(band_first + first);

#pragma unroll 1
    for (int it = first; it < band_size; it += stride) {
        const int t = band_first + it;
        const int by = (int) ((uint32_t) t / (uint32_t) tiles_x), bx = t - by * tiles_x;
        const int ox = u_org_x + 8 * WTC * bx, oy = u_org_y + 16 * MX_WT_ROWS * by;
        const bool edge = ox < 0 || ox + G::src_w > sw;

        // every wave is done with the previous tile's LDS image (first turn: nothing to wait for)
        __syncthreads();

        // ---- registers -> LDS: decode, the reference's "PASS A" per source texel (recorded pre-ops,
        // f16 rounding = what the rgba16hf FBO store + load would do), planar stores ---------------
        if (edge) {
            // a pair at clamped positions: beyond the left edge both texels are the pair's first,
            // beyond the right edge both its second
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint4 w = v[u];
                const int sx = ox + 2 * tp(u);
                const bool ldup = sx < 0, hdup = sx > sw - 2;
                const uint32_t ax = hdup ? w.z : w.x, ay = hdup ? w.w : w.y;
                const uint32_t bx_ = ldup ? w.x : w.z, by_ = ldup ? w.y : w.w;
                v[u] = make_uint4(ax, ay, bx_, by_);
            }
        }
        if (raw16) {
            // rgba16hf source, no pre-ops: the f16 codes are moved bit for bit into the planes
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint4 w = v[u];
                if (tid + u * NT < G::npairs)
                    pair_store(ty(u), tp(u), (w.x & 0xffffu) | (w.z << 16), (w.x >> 16) | (w.z & 0xffff0000u),
                               (w.y & 0xffffu) | (w.w << 16));
            }
        } else if (simple) {
            // the plane as it is, or behind an identity PLANE_MAP (components the plane does not
            // carry take their neutral values): decode, round to f16, store
            const plh_op &om = p.ops[0];
            const int present = u_npre ? om.i1 : 4;
            float neutral[NCH];
#pragma unroll
            for (int k = 0; k < NCH; k++)
                neutral[k] = u_npre ? om.f[k] : 0.0f;
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint32_t q[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
                uint32_t o[NCH];
#pragma unroll
                for (int k = 0; k < NCH; k++) {
                    const uint32_t a = (k & 1) ? q[k >> 1] >> 16 : q[k >> 1] & 0xffffu;
                    const uint32_t b = (k & 1) ? q[2 + (k >> 1)] >> 16 : q[2 + (k >> 1)] & 0xffffu;
                    float fa = unorm ? mx_un16_for_f16(a) : plh_h2f(a), fb = unorm ? mx_un16_for_f16(b) : plh_h2f(b);
                    if (k >= present) {
                        fa = neutral[k];
                        fb = neutral[k];
                    }
                    o[k] = mx_pack(fa, fb);
                }
                if (tid + u * NT < G::npairs)
                    pair_store(ty(u), tp(u), o[0], o[1], o[2]);
            }
        } else {
            // any other op list without transcendentals: one interpreter walk over the lane's texels
            float4_t c[2 * NV];
            frag_t fcs[2 * NV];
#pragma unroll
            for (int u = 0; u < NV; u++) {
                const uint4 w = v[u];
                float4_t &c0 = c[2 * u], &c1 = c[2 * u + 1];
                if (unorm) {
                    c0 = { plh_un16(w.x & 0xffff), plh_un16(w.x >> 16), plh_un16(w.y & 0xffff), plh_un16(w.y >> 16) };
                    c1 = { plh_un16(w.z & 0xffff), plh_un16(w.z >> 16), plh_un16(w.w & 0xffff), plh_un16(w.w >> 16) };
                } else {
                    c0 = { plh_h2f(w.x & 0xffff), plh_h2f(w.x >> 16), plh_h2f(w.y & 0xffff), plh_h2f(w.y >> 16) };
                    c1 = { plh_h2f(w.z & 0xffff), plh_h2f(w.z >> 16), plh_h2f(w.w & 0xffff), plh_h2f(w.w >> 16) };
                }
                // gl_FragCoord of the fused pass: the (clamped) source texel
                const int sx = ox + 2 * tp(u);
                const float cy = (float) min(max(oy + ty(u), 0), sh - 1) + 0.5f;
                fcs[2 * u] = { (float) min(max(sx, 0), sw - 1) + 0.5f, cy, 0.0f, 0 };
                fcs[2 * u + 1] = { (float) min(max(sx + 1, 0), sw - 1) + 0.5f, cy, 0.0f, 0 };
            }
            apply_ops_n<2 * NV, false, true>(c, p.ops, 0, u_npre, fcs);
#pragma unroll
            for (int u = 0; u < NV; u++) {
                if (tid + u * NT < G::npairs)
                    pair_store(ty(u), tp(u), mx_pack(c[2 * u].x, c[2 * u + 1].x), mx_pack(c[2 * u].y, c[2 * u + 1].y),
                               mx_pack(c[2 * u].z, c[2 * u + 1].z));
            }
        }

        // the lane's column and its phase deviation (x 2^11)
        const int X = G::tile_w * bx + 16 * wave + ln;
        const _Float16 dxh = (_Float16) nx_dfx;
        const mx_f16x8 dx8 = { dxh, dxh, dxh, dxh, dxh, dxh, dxh, dxh };
        __syncthreads();
        // the next tile's texels: asked for now, used after this tile's contraction and stores
        if (it + stride < band_size)
            tile_load(t + stride);

        const int cpos = p.base_x + p.dir_x * X;
        const bool cok = X < p.width && p.out_scale[0] * (float) X < 1.0f && cpos >= 0 && cpos < p.dst.w;
        const uint32_t tcol = (uint32_t) ((X + u_fx0) & u_emask) * (uint32_t) u_esize * 4u;

#pragma unroll 1
        for (int i = 0; i < MX_WT_ROWS; i++) {
            // output pixels of this lane: column X, rows Y0 + 2 * r + py (r < 4, py < 2)
            const int Y0 = MX_TILE_H * by + 32 * i + 8 * lg;
            float bias[8];
            if constexpr (FAST) {
                if (!u_has_dither) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        bias[q] = 0.0f;
                } else if (u_tcol) {
                    // eight consecutive entries of one COLUMN of the matrix: two 16-byte loads from
                    // its transposed copy (k_polar_mx.hiph)
                    const uint32_t iy0 = (uint32_t) (Y0 + u_fy0) & (uint32_t) u_emask;
                    const mx_gf4 *pb = (mx_gf4 *) (u_matrix_t + tcol + iy0 * 4u);
                    const mx_f32x4 b0 = pb[0], b1 = pb[1];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        bias[q] = b0[q];
                        bias[4 + q] = b1[q];
                    }
                } else {
                    const int ix = (X + u_fx0) & u_emask;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const int iy = (Y0 + q + u_fy0) & u_emask;
                        bias[q] = ((mx_gfloat *) u_matrix)[iy * u_esize + ix];
                    }
                }
            }
            // the row-phase deviations of the lane's rows (Y0 is a multiple of 8, the table starts on
            // a 16-byte boundary): two 16-byte loads per row phase, folded in when the phase is done
            const mx_gf4 *pd = (mx_gf4 *) ((mx_gfloat *) u_dfy + Y0);

            mx_f32x4 acc[2][NCH];
            const unsigned char *ab = tile + (16 * i + ln + (lg >> 1)) * PITCH + (8 * wave + 8 * (lg & 1)) * 2;
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int py = 0; py < 2; py++) {
                const unsigned char *ap = ab + (py ? u_row1 : u_row0) * PITCH;
                const unsigned char *bp = py ? bfl1 : bfl;
                mx_f32x4 ay[NCH];
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    acc[py][ch] = (mx_f32x4) (0.0f);
                    ay[ch] = (mx_f32x4) (0.0f);
                }
                const mx_f32x4 d0 = pd[0], d1 = pd[1];
                const float dfy[4] = { py ? d0[1] : d0[0], py ? d0[3] : d0[2], py ? d1[1] : d1[0], py ? d1[3] : d1[2] };
                auto pair = [&](int j) {
                    const unsigned char *bf = bp + 4 * j * 1024;
                    const mx_f16x8 bhi = *(const mx_f16x8 *) bf;
                    const mx_f16x8 blo = __builtin_elementwise_fma(*(const mx_f16x8 *) (bf + 2048), dx8, *(const mx_f16x8 *) (bf + 1024));
                    const mx_f16x8 bdy = *(const mx_f16x8 *) (bf + 3072);
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        const mx_f16x8 a = *(const mx_f16x8 *) (ap + ch * PLANE + 2 * j * PITCH);
                        acc[py][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhi, acc[py][ch], 0, 0, 0);
                        acc[py][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, blo, acc[py][ch], 0, 0, 0);
                        ay[ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bdy, ay[ch], 0, 0, 0);
                    }
                };
                pair(0);
                pair(1);
                pair(2);
                if (u_np4)
                    pair(3);
                // the lane's rows of phase py are Y0 + 2 r + py, r < 4
#pragma unroll
                for (int r = 0; r < 4; r++) {
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++)
                        acc[py][ch][r] = __builtin_fmaf(dfy[r], ay[ch][r], acc[py][ch][r]);
                }
            }
            __builtin_amdgcn_s_setprio(0);

            // ---- epilogue: post-ops, guarded store (dispatch.c:1126-1142) -----------------------
            typedef __attribute__((address_space(1))) plh_u32x2 mx_gpx;
            const float ds = u_ds, di = u_di;
            const float sc = u_has_scale ? u_sc : 1.0f;
            // alpha (not sampled: 1) behind dither and scale: floor(ds * 1 + b) == ds for b in [0, 1)
            float aw = 1.0f;
            if (u_has_dither)
                aw = ds * di;
            aw *= sc;
            if constexpr (FAST) {
                const int rpos0 = u_base_y + u_dir_y * Y0, rpos7 = u_base_y + u_dir_y * (Y0 + 7);
                const bool whole = (int) cok & (int) (Y0 + 7 < u_height) & (int) (u_osy * (float) (Y0 + 7) < 1.0f) &
                                   (int) (min(rpos0, rpos7) >= 0) & (int) (max(rpos0, rpos7) < u_dst_h);
                const bool all_whole = __builtin_amdgcn_ballot_w64(!whole) == 0;
                auto store = [&](uintptr_t d, const plh_u32x2 px) {
                    if (u_nt)
                        __builtin_nontemporal_store(px, (mx_gpx *) d);
                    else
                        *(mx_gpx *) d = px;
                };
                auto rows = [&](auto dith) {
                    constexpr bool DITHER = decltype(dith)::value;
                    auto pixel = [&](int q) {
                        const int r = q >> 1, py = q & 1;
                        float o[NCH];
#pragma unroll
                        for (int k = 0; k < NCH; k++) {
                            o[k] = acc[py][k][r];
                            if (DITHER)
                                o[k] = __builtin_floorf(__builtin_fmaf(ds, o[k], bias[q])) * di;
                            o[k] *= sc;
                        }
                        plh_u32x2 px;
                        px.x = plh_unorm16x2(o[0], o[1]);
                        px.y = plh_unorm16x2(o[2], aw);
                        return px;
                    };
                    if (all_whole) {
                        uintptr_t d = u_dptr + (size_t) rpos0 * (size_t) u_dpitch + (size_t) cpos * 8;
                        const ptrdiff_t step = (ptrdiff_t) u_dir_y * (ptrdiff_t) u_dpitch;
#pragma unroll
                        for (int q = 0; q < 8; q++, d += step)
                            store(d, pixel(q));
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; q++) {
                            const int Y = Y0 + q;
                            const int rpos = u_base_y + u_dir_y * Y;
                            const bool ok = cok && Y < u_height && u_osy * (float) Y < 1.0f &&
                                            rpos >= 0 && rpos < u_dst_h;
                            const plh_u32x2 px = pixel(q);
                            if (ok)
                                store(u_dptr + (size_t) rpos * (size_t) u_dpitch + (size_t) cpos * 8, px);
                        }
                    }
                };
                if (u_has_dither)
                    rows(std::true_type{});
                else
                    rows(std::false_type{});
            } else {
                // the map chain as straight-line code, two pixels at a time, then the fused tail: ONE
                // instance of the chain in a rolled loop over the lane's four row pairs (k_polar_mx.hiph)
                constexpr int NP = 2;
                int ylo, yhi;
                if (u_dir_y > 0) {
                    ylo = max(0, -u_base_y);
                    yhi = min(u_height, u_dst_h - u_base_y);
                } else {
                    ylo = max(0, u_base_y - u_dst_h + 1);
                    yhi = min(u_height, u_base_y + 1);
                }
                const uint32_t ny = (uint32_t) max(yhi - ylo, 0);
                const int eshift = __builtin_ctz((unsigned) max(u_esize, 1)) + 2;       // (bytes per matrix row)
                const uint32_t ix4 = (uint32_t) ((X + u_fx0) & u_emask) << 2;
                const ptrdiff_t step = (ptrdiff_t) u_dir_y * (ptrdiff_t) u_dpitch;
                uintptr_t drow = u_dptr + (size_t) (u_base_y + u_dir_y * Y0) * (size_t) u_dpitch + (size_t) cpos * 8;

                float cur[NCH][2], nx1[NCH][2], nx2[NCH][2], nx3[NCH][2];
#pragma unroll
                for (int k = 0; k < NCH; k++) {
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        cur[k][r] = acc[0][k][r];
                        nx1[k][r] = acc[1][k][r];
                        nx2[k][r] = acc[0][k][2 + r];
                        nx3[k][r] = acc[1][k][2 + r];
                    }
                }
#pragma unroll 1
                for (int part = 0; part < 4; part++) {
                    const int yoff = 4 * (part >> 1) + (part & 1);      // the pair's rows: Y0 + yoff + {0, 2}
                    float4_t outs[NP];
                    float bq[NP];
#pragma unroll
                    for (int r = 0; r < NP; r++) {
                        const uint32_t iy = (uint32_t) (Y0 + yoff + 2 * r + u_fy0) & (uint32_t) u_emask;
                        bq[r] = u_has_dither ? *(mx_gfloat *) (u_matrix + ((iy << eshift) | ix4)) : 0.0f;
                        outs[r] = { cur[0][r], cur[1][r], cur[2][r], 1.0f };
                    }
                    float pos[NP][2];
                    if (CR) {
#pragma unroll
                        for (int r = 0; r < NP; r++) {
                            pos[r][0] = p.out_scale[0] * ((float) X + 0.5f);
                            pos[r][1] = u_osy * ((float) (Y0 + yoff + 2 * r) + 0.5f);
                        }
                    }
                    run_map_chain<NP, CR>(outs, p, pos);
#pragma unroll
                    for (int r = 0; r < NP; r++) {
                        const bool ok = cok && (uint32_t) (Y0 + yoff + 2 * r - ylo) < ny;
                        float4_t c = outs[r];
                        if (u_has_dither) {
                            const float b = bq[r];
                            c.x = __builtin_floorf(ds * c.x + b) * di;
                            c.y = __builtin_floorf(ds * c.y + b) * di;
                            c.z = __builtin_floorf(ds * c.z + b) * di;
                        }
                        if (u_has_scale) {
                            c.x *= sc; c.y *= sc; c.z *= sc;
                        }
                        plh_u32x2 px;
                        px.x = plh_unorm16x2(c.x, c.y);
                        px.y = plh_unorm16x2(c.z, aw);
                        if (ok) {
                            const uintptr_t d = drow + (r ? 2 * step : 0);
                            if (u_nt)
                                __builtin_nontemporal_store(px, (mx_gpx *) d);
                            else
                                *(mx_gpx *) d = px;
                        }
                    }
                    // the next pair of rows; its start is 1, 3, 1 rows further down
                    drow += (part & 1) ? 3 * step : step;
#pragma unroll
                    for (int k = 0; k < NCH; k++) {
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            cur[k][r] = nx1[k][r];
                            nx1[k][r] = nx2[k][r];
                            nx2[k][r] = nx3[k][r];
                        }
                    }
                }
            }
        }
    }
}

#undef ty
#undef tp

// Does the persistent kernel take this pass? (Else: k_polar_mx, which handles every shape.)
bool plh_polar_mxp_applies(const plh_pass *pass)
{
    const char *env = getenv("PL_HIP_MX_PERSIST");
    if (env && env[0] == '0')
        return false;
    const int fmt = pass->s.src.fmt;
    if ((pass->s.comp_mask & 0xf) != 0x7 || (fmt != PLH_FMT_RGBA16 && fmt != PLH_FMT_RGBA16F))
        return false;
    if (!plh_ops_lite(pass, 0, pass->num_pre_ops))
        return false;
    const bool post_lite = plh_ops_lite(pass, pass->num_pre_ops, pass->num_ops);
    return pass->chain.enabled || (post_lite && pass->epi.enabled);
}

template <int POST>
static void launch_mxp_variant(hipStream_t stream, const plh_pass *pass)
{
    using G = mx_geom<8>;
    const int tiles_x = (pass->width + G::tile_w - 1) / G::tile_w;
    const int tiles_y = (pass->height + MX_TILE_H - 1) / MX_TILE_H;
    const size_t shmem = MX_B_BYTES + (size_t) 3 * G::plane;
    static uint64_t lds_done;
    (void) plh_kernel_needs_lds((const void *) k_polar_mxp<POST>, (plh_stream) stream, shmem, &lds_done);
    // two workgroups per CU (LDS: 51.7 KiB each; 8 waves at <= 128 registers: 4 waves per SIMD)
    int cus = 256;
    (void) plh_stream_device((plh_stream) stream, &cus);
    const int groups = min(tiles_x * tiles_y, 2 * cus);
    PLH_LAUNCH_LAST((k_polar_mxp<POST>), dim3(groups), dim3(G::threads), shmem, stream, *pass);
}

int plh_launch_polar_mxp(hipStream_t stream, const plh_pass *pass)
{
    if (pass->chain.enabled && pass->chain.contrast_recovery)
        launch_mxp_variant<MX_POST_CHAIN_CR>(stream, pass);
    else if (pass->chain.enabled)
        launch_mxp_variant<MX_POST_CHAIN>(stream, pass);
    else
        launch_mxp_variant<MX_POST_FAST>(stream, pass);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
