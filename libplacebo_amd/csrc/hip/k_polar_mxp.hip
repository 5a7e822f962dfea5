/*
 * libplacebo-hip -- k_polar_mxp: k_polar_mx (the polar 2x upscale as a tile contraction on the f16
 * matrix pipe: k_polar_mx.hiph explains the numerics, the fragment layout and the live-row pairing)
 * on PERSISTENT workgroups, for the shapes that carry the benchmark configurations: RGB tiles of 8
 * wave-tile columns, an rgba16 / rgba16hf source whose fused pre-ops need no transcendental, and
 * the fused epilogue (dither + scale) behind the contraction.
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
 * The row-phase term is folded phase by phase (k_polar_mx.hiph: YPHASE), which frees the registers
 * the prefetch needs and costs the epilogue nothing.
 * What makes the overlap real is the memory counter: gfx950 has ONE counter for vector loads and
 * stores, retired in order, so a wait for any load is a wait for everything issued before it. A
 * tile's compute phase therefore issues no load at all -- the dither matrix (transposed: a lane
 * owns a column of eight rows) lives in LDS for the workgroup's lifetime, the tile's 64 row-phase
 * deviations arrive with its texels and are parked in LDS too -- and nothing is spilled (a scratch
 * reload is a vector load like any other): between the request for the next tile's texels and the
 * top of the next turn the wave only issues stores, which it never waits for.
 * (The map-chain epilogue gathers from its lookup tables all the time: its variant measured
 * 111.5 -> 115.6 us on persistent workgroups and stays on k_polar_mx.)
 *
 * Same arithmetic as k_polar_mx<3, true, MX_POST_FAST, 8>, bit for bit (the row-phase term is the
 * same fma, a few instructions earlier). PL_HIP_MX_PERSIST=0 keeps k_polar_mx (tests compare the two).
 */
#include "k_polar_mx.hiph"

// A workgroup barrier that orders LDS traffic only. __syncthreads() is a release / acquire fence on
// ALL memory around s_barrier: it waits for every global store of the wave to be acknowledged
// (s_waitcnt vmcnt(0)) -- twice per tile, exactly the wait this kernel exists to avoid. The tile's
// hand-over between waves goes through LDS alone: the wave's own LDS operations complete
// (lgkmcnt(0)), then the barrier. (The "memory" clobber keeps the compiler from moving accesses
// across it; the waits for data it tracks itself -- the prefetched texels -- it still inserts.)
DEV void mxp_sync_lds()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#define MXP_DFY_BYTES   (MX_TILE_H * 4)
#define MXP_STORE_DEFAULT 1         // (profiles/r06_08_store_kinds.txt: plain 25.6 us, non-temporal 22.3, system scope 51.8)
#define MXP_DMAT_MAX    64          // largest dither matrix kept in LDS (64 x 64 floats = 16 KiB)

// STORE: how the target is written -- 0: plain stores (write-back cached in the XCD's L2), 1:
// non-temporal, 2: system-scope stores (sc0 sc1: written through the L2)
template <int STORE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
void k_polar_mxp(const plh_pass p_)
{
    constexpr int NCH = 3, WTC = 8;
    using G = mx_geom<WTC>;
    constexpr int NT = G::threads, PITCH = G::pitch, PLANE = G::plane, NV = G::nv;
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const plh_polar_mx &mx = s.mx;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *bl = smem;
    unsigned char *tile = smem + MX_B_BYTES;
    unsigned char *ldfy = tile + NCH * PLANE;           // the tile's 64 row-phase deviations
    unsigned char *ldm = ldfy + MXP_DFY_BYTES;          // the dither matrix, transposed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // uniforms, read once and pinned in SGPRs (k_polar_mx.hiph says why)
    int u_height = p.height, u_dst_h = p.dst.h, u_base_y = p.base_y, u_dir_y = p.dir_y;
    int u_dpitch = p.dst.pitch, u_nt = p.nt_store, u_fx0 = p.frag_x0, u_fy0 = p.frag_y0;
    int u_has_dither = p.epi.has_dither, u_has_scale = p.epi.has_scale, u_emask = p.epi.mask, u_esize = p.epi.size;
    float u_osy = p.out_scale[1], u_ds = p.epi.dscale, u_di = p.epi.dinv, u_sc = p.epi.scale;
    uintptr_t u_dptr = (uintptr_t) p.dst.ptr, u_dfy = (uintptr_t) mx.dfy, u_dfx = (uintptr_t) mx.dfx;
    asm volatile("" : "+s"(u_height), "+s"(u_dst_h), "+s"(u_base_y), "+s"(u_dir_y), "+s"(u_dpitch),
                      "+s"(u_nt), "+s"(u_fx0), "+s"(u_fy0), "+s"(u_has_dither), "+s"(u_has_scale),
                      "+s"(u_emask), "+s"(u_esize));
    asm volatile("" : "+s"(u_osy), "+s"(u_ds), "+s"(u_di), "+s"(u_sc), "+s"(u_dptr), "+s"(u_dfy), "+s"(u_dfx));
    typedef __attribute__((address_space(1))) const float mx_gfloat;

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

    // ---- once per workgroup: B fragments and the (transposed) dither matrix, global (L2 resident)
    // -> LDS without passing through registers (global_load_lds_dwordx4: wave-uniform LDS base +
    // lane * 16) -------------------------------------------------------------------------------------
    const int nfrag = 8 * mx.npairs;
#pragma unroll
    for (int f = wave; f < PLH_MX_NFRAG; f += NT / 64) {
        if (f >= nfrag)
            break;
        const unsigned char *g = (const unsigned char *) mx.bfrag + ((size_t) f * 64 + lane) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                         (__attribute__((address_space(3))) void *) (bl + f * 1024), 16, 0, 0);
    }
    if (u_has_dither) {
        const int kib = (u_esize * u_esize * 4) >> 10;      // (8 x 8 floats = 256 bytes: one partial piece)
        const unsigned char *mt = (const unsigned char *) p.epi.matrix_t;
        for (int f = wave; f < max(kib, 1); f += NT / 64) {
            // (a matrix smaller than a piece: the lanes beyond it re-read its last 16 bytes)
            const int off = min(f * 1024 + lane * 16, u_esize * u_esize * 4 - 16);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (mt + off),
                                             (__attribute__((address_space(3))) void *) (ldm + f * 1024), 16, 0, 0);
        }
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
    // clamped addressing: sampling.c:45-181), the lane's column-phase deviation and -- one per lane
    // of a wave -- the phase deviations of the tile's 64 rows (the tables are padded to whole tiles)
    uint4 v[NV];
    float nx_dfx, nx_dfy;
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
        nx_dfx = ((mx_gfloat *) u_dfx)[G::tile_w * tbx + 16 * wave + ln];
        nx_dfy = ((mx_gfloat *) u_dfy)[MX_TILE_H * tby + lane];
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

    // (the copies into LDS above are vector loads: complete, and visible to every wave, before the
    // first tile is touched)
    __syncthreads();
    // ---- registers -> LDS: decode, the reference's "PASS A" per source texel (recorded pre-ops,
    // f16 rounding = what the rgba16hf FBO store + load would do), planar stores; with them the
    // tile's row-phase deviations. Returns the lane's column-phase deviation for that tile. --------
    auto tile_to_lds = [&](int t) {
        const int tby = (int) ((uint32_t) t / (uint32_t) tiles_x), tbx = t - tby * tiles_x;
        const int ox = u_org_x + 8 * WTC * tbx, oy = u_org_y + 16 * MX_WT_ROWS * tby;
        const bool edge = ox < 0 || ox + G::src_w > sw;
        if (wave == 0)
            ((float *) ldfy)[lane] = nx_dfy;
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
        return nx_dfx;
    };

    // (the copies into LDS above are vector loads: complete, and visible to every wave, before the
    // first tile is touched)
    __syncthreads();
    // The loop is rotated: a turn computes tile t from LDS and, at its END, moves tile t + stride
    // from the registers into LDS. The request for those texels, the sixteen stores of tile t and
    // the first use of the texels then lie in ONE straight line of code, and the wait in front of
    // that use counts past the stores (s_waitcnt vmcnt(16)); with the request in one turn and the
    // use at the top of the next, the compiler has to assume the worst of the loop's entry and its
    // back edge -- vmcnt(0), every store acknowledged -- at every tile.
#pragma unroll
    for (int u = 0; u < NV; u++)
        v[u] = make_uint4(0, 0, 0, 0);
    nx_dfx = nx_dfy = 0.0f;
    float cur_dfx = 0.0f;
    if (first < band_size) {
        tile_load(band_first + first);
        cur_dfx = tile_to_lds(band_first + first);
    }

#pragma unroll 1
    for (int it = first; it < band_size; it += stride) {
        const int t = band_first + it;
        const int by = (int) ((uint32_t) t / (uint32_t) tiles_x), bx = t - by * tiles_x;
        const bool more = it + stride < band_size;

        // the lane's column and its phase deviation (x 2^11)
        const int X = G::tile_w * bx + 16 * wave + ln;
        const _Float16 dxh = (_Float16) cur_dfx;
        const mx_f16x8 dx8 = { dxh, dxh, dxh, dxh, dxh, dxh, dxh, dxh };
        // tile t is in LDS
        mxp_sync_lds();
        // the next tile's texels: asked for now, used behind this tile's contraction and stores.
        // From here to that use this wave issues NO other load.
        if (more)
            tile_load(t + stride);

        // (uniform) the whole tile lies inside the pass and the target: nearly every tile
        const int tx1 = G::tile_w * bx + G::tile_w - 1, ty0 = MX_TILE_H * by, ty1 = ty0 + MX_TILE_H - 1;
        const int cp0 = p.base_x + p.dir_x * (G::tile_w * bx), cp1 = p.base_x + p.dir_x * tx1;
        const int rp0 = u_base_y + u_dir_y * ty0, rp1 = u_base_y + u_dir_y * ty1;
        const bool inside = tx1 < p.width && p.out_scale[0] * (float) tx1 < 1.0f && min(cp0, cp1) >= 0 &&
                            max(cp0, cp1) < p.dst.w && ty1 < u_height && u_osy * (float) ty1 < 1.0f &&
                            min(rp0, rp1) >= 0 && max(rp0, rp1) < u_dst_h;
        const uintptr_t sink = (uintptr_t) mx.sink + (uint32_t) lane * 8u;
        const int cpos = p.base_x + p.dir_x * X;
        const bool cok = X < p.width && p.out_scale[0] * (float) X < 1.0f && cpos >= 0 && cpos < p.dst.w;
        // the lane's column of the transposed dither matrix (bytes)
        const uint32_t tcol = (uint32_t) ((X + u_fx0) & u_emask) * (uint32_t) u_esize * 4u;

        // (both wave tiles written out: the store count of a turn must be a constant of the code)
        static_assert(MX_WT_ROWS == 2, "two wave tiles per wave");
        auto wave_tile = [&](auto wt) {
            constexpr int i = decltype(wt)::value;
            // output pixels of this lane: column X, rows Y0 + 2 * r + py (r < 4, py < 2)
            const int Y0 = MX_TILE_H * by + 32 * i + 8 * lg;
            const unsigned char *ab = tile + (16 * i + ln + (lg >> 1)) * PITCH + (8 * wave + 8 * (lg & 1)) * 2;
            // the row-phase deviations of the lane's eight rows
            const mx_f32x4 *pd = (const mx_f32x4 *) (ldfy + (32 * i + 8 * lg) * 4);
            // op_dither (plain path) and the SCALE op with their parameters in SGPRs
            typedef __attribute__((address_space(1))) plh_u32x2 mx_gpx;
            const float ds = u_ds, di = u_di;
            const float sc = u_has_scale ? u_sc : 1.0f;
            // alpha (not sampled: 1) behind dither and scale: floor(ds * 1 + b) == ds for b in [0, 1)
            float aw = 1.0f;
            if (u_has_dither)
                aw = ds * di;
            aw *= sc;
            const int rpos0 = u_base_y + u_dir_y * Y0;
            const uintptr_t d0 = u_dptr + (size_t) rpos0 * (size_t) u_dpitch + (size_t) cpos * 8;
            const ptrdiff_t step = (ptrdiff_t) u_dir_y * (ptrdiff_t) u_dpitch;
            // eight consecutive entries of the lane's COLUMN of the dither matrix (no wrap: the
            // fragment offset is a multiple of 8, plh_polar_mxp_applies)
            const uint32_t iy0 = (uint32_t) (Y0 + u_fy0) & (uint32_t) u_emask;
            const mx_f32x4 *pb = (const mx_f32x4 *) (ldm + tcol + iy0 * 4u);
            // Row phase by row phase: contraction of the phase's live tap rows, the row-phase term,
            // then the epilogue and the stores of ITS four rows (Y0 + 2 r + py) -- twelve accumulator
            // registers live at a time instead of twenty-four, and the first phase's epilogue runs
            // in the shadow of other waves' contractions.
#pragma unroll
            for (int py = 0; py < 2; py++) {
                const unsigned char *ap = ab + (py ? u_row1 : u_row0) * PITCH;
                const unsigned char *bp = py ? bfl1 : bfl;
                mx_f32x4 acc[NCH], ay[NCH];
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    acc[ch] = (mx_f32x4) (0.0f);
                    ay[ch] = (mx_f32x4) (0.0f);
                }
                auto pair = [&](int j) {
                    const unsigned char *bf = bp + 4 * j * 1024;
                    const mx_f16x8 bhi = *(const mx_f16x8 *) bf;
                    const mx_f16x8 blo = __builtin_elementwise_fma(*(const mx_f16x8 *) (bf + 2048), dx8, *(const mx_f16x8 *) (bf + 1024));
                    const mx_f16x8 bdy = *(const mx_f16x8 *) (bf + 3072);
#pragma unroll
                    for (int ch = 0; ch < NCH; ch++) {
                        const mx_f16x8 a = *(const mx_f16x8 *) (ap + ch * PLANE + 2 * j * PITCH);
                        acc[ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhi, acc[ch], 0, 0, 0);
                        acc[ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, blo, acc[ch], 0, 0, 0);
                        ay[ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bdy, ay[ch], 0, 0, 0);
                    }
                };
                __builtin_amdgcn_s_setprio(3);
                pair(0);
                pair(1);
                pair(2);
                if (u_np4)
                    pair(3);
                __builtin_amdgcn_s_setprio(0);
                const mx_f32x4 dv0 = pd[0], dv1 = pd[1];
                const float dfy[4] = { py ? dv0[1] : dv0[0], py ? dv0[3] : dv0[2], py ? dv1[1] : dv1[0], py ? dv1[3] : dv1[2] };
                float bias[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                if (u_has_dither) {
                    const mx_f32x4 b0 = pb[0], b1 = pb[1];
                    bias[0] = py ? b0[1] : b0[0]; bias[1] = py ? b0[3] : b0[2];
                    bias[2] = py ? b1[1] : b1[0]; bias[3] = py ? b1[3] : b1[2];
                }
                // Four stores per phase and lane WHATEVER the tile -- no guard, no exec mask (the
                // compiler branches around a masked store when no lane is left, and a path without
                // the store makes the store count of a turn unknowable): a lane outside the target
                // (clipped edge tiles only) stores to a sink nobody reads. Between the request for
                // the next tile's texels and their use lie exactly sixteen stores per wave, which is
                // what lets the wait in front of that use count past them.
                auto rows = [&](auto dith, auto inside) {
                    constexpr bool DITHER = decltype(dith)::value, INSIDE = decltype(inside)::value;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int q = 2 * r + py;
                        float o[NCH];
#pragma unroll
                        for (int k = 0; k < NCH; k++) {
                            o[k] = __builtin_fmaf(dfy[r], ay[k][r], acc[k][r]);
                            if (DITHER)
                                o[k] = __builtin_floorf(__builtin_fmaf(ds, o[k], bias[r])) * di;
                            o[k] *= sc;
                        }
                        plh_u32x2 px;
                        px.x = plh_unorm16x2(o[0], o[1]);
                        px.y = plh_unorm16x2(o[2], aw);
                        uintptr_t d = d0 + q * step;
                        if (!INSIDE) {
                            const int Y = Y0 + q, rpos = rpos0 + u_dir_y * q;
                            const bool ok = (int) cok & (int) (Y < u_height) & (int) (u_osy * (float) Y < 1.0f) &
                                            (int) (rpos >= 0) & (int) (rpos < u_dst_h);
                            d = ok ? d : sink;
                        }
                        // (ONE kind of store per instantiation: `if (nt) non-temporal else plain`
                        // is merged by the compiler into the plain one)
                        if constexpr (STORE == 1) {
                            __builtin_nontemporal_store(px, (mx_gpx *) d);
                        } else if constexpr (STORE == 2) {
                            const uint64_t both = (uint64_t) px.x | ((uint64_t) px.y << 32);
                            __scoped_atomic_store_n((__attribute__((address_space(1))) uint64_t *) d, both,
                                                    __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
                        } else {
                            *(mx_gpx *) d = px;
                        }
                    }
                };
                if (inside) {
                    if (u_has_dither)
                        rows(std::true_type{}, std::true_type{});
                    else
                        rows(std::false_type{}, std::true_type{});
                } else {
                    if (u_has_dither)
                        rows(std::true_type{}, std::false_type{});
                    else
                        rows(std::false_type{}, std::false_type{});
                }
            }
        };
        wave_tile(std::integral_constant<int, 0>{});
        wave_tile(std::integral_constant<int, 1>{});
        // every wave is done with this tile's LDS image: the next one moves in
        mxp_sync_lds();
        if (more)
            cur_dfx = tile_to_lds(t + stride);
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
    if (!plh_ops_lite(pass, 0, pass->num_pre_ops) || pass->chain.enabled)
        return false;
    if (!plh_ops_lite(pass, pass->num_pre_ops, pass->num_ops) || !pass->epi.enabled)
        return false;
    // the dither matrix lives in LDS, transposed, and a lane reads eight consecutive rows of it
    const plh_fast_epi &e = pass->epi;
    if (e.has_dither && (!e.matrix_t || e.size < 8 || e.size > MXP_DMAT_MAX || (pass->frag_y0 & 7)))
        return false;
    return true;
}

template <int STORE>
static void launch_mxp(hipStream_t stream, const plh_pass *pass, int groups, size_t shmem)
{
    static uint64_t lds_done;
    (void) plh_kernel_needs_lds((const void *) k_polar_mxp<STORE>, (plh_stream) stream, shmem, &lds_done);
    PLH_LAUNCH_LAST(k_polar_mxp<STORE>, dim3(groups), dim3(mx_geom<8>::threads), shmem, stream, *pass);
}

int plh_launch_polar_mxp(hipStream_t stream, const plh_pass *pass)
{
    using G = mx_geom<8>;
    const int tiles_x = (pass->width + G::tile_w - 1) / G::tile_w;
    const int tiles_y = (pass->height + MX_TILE_H - 1) / MX_TILE_H;
    const size_t shmem = MX_B_BYTES + (size_t) 3 * G::plane + MXP_DFY_BYTES + MXP_DMAT_MAX * MXP_DMAT_MAX * 4;
    // two workgroups per CU (LDS: 68 KiB each; 8 waves at <= 128 registers: 4 waves per SIMD)
    int cus = 256;
    (void) plh_stream_device((plh_stream) stream, &cus);
    const int groups = min(tiles_x * tiles_y, 2 * cus);
    const char *env = getenv("PL_HIP_MXP_STORE");
    const int kind = env ? atoi(env) : (pass->nt_store ? MXP_STORE_DEFAULT : 0);
    if (kind == 1)
        launch_mxp<1>(stream, pass, groups, shmem);
    else if (kind == 2)
        launch_mxp<2>(stream, pass, groups, shmem);
    else
        launch_mxp<0>(stream, pass, groups, shmem);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
