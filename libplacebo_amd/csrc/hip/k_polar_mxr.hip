/*
 * libplacebo-hip -- k_polar_mxr: the polar (EWA) upscale by an exact INTEGER ratio R = 3 or 4 as a
 * dense tile contraction on the f16 matrix pipe (struct plh_polar_mx with enabled == 3,
 * plh_device.h). 720p -> 4K is 3x, 540p / 960x540 -> 4K is 4x; k_polar_mx.hiph is the 2x case and
 * explains the numerics (f16 hi + lo weight halves, first-order terms in the per-pixel phase),
 * which are the same here. VERDICT r03 "missing 1". What is computed is the reference's polar
 * sampler (src/shaders/sampling.c:503-558 polar_sample, :587-912 pl_shader_sample_polar: every tap
 * within the radius weighted by the LUT at its distance, normalised by the weight sum) -- here with
 * the weights of each phase tabulated once per geometry by the host (shader_sampling.c:
 * polar_mxr_build, from the same phase-class weights k_polar_pp uses).
 *
 * What changes with R. An axis has R phases; output X belongs to base index (X + sx) / R and phase
 * (X + sx) % R, and -- with the shift the host reads off the geometry -- every phase of a base index
 * shares the base texel. So:
 *   rows     one GEMM per row phase py over the SAME four row pairs of the source tile (M = 16 base
 *            rows of a workgroup tile = 16 R output rows), processed one phase after the other:
 *            contraction, then the epilogue of that phase's 8 pixels per lane;
 *   columns  a wave owns 8 base columns, whose 16-column K window is 16-byte aligned in LDS like
 *            k_polar_mx's. Its 8 R output columns are two N halves of 4 bases x R phases each
 *            (4 R <= 16 columns of the 16 the MFMA has: 12 for R = 3), i.e. two MFMAs per (row
 *            pair, kind, channel) sharing one A fragment;
 *   B        4 row pairs x 2 halves x {hi, lo, d/dx, d/dy} = 32 fragments (32 KiB) PER ROW PHASE:
 *            staged in LDS one row phase at a time (global_load_lds_dwordx4, L2-resident), two
 *            barriers per phase.
 * Per 16 x 16 base block (256 R^2 pixels): 72 R MFMAs, 0.28 / R per pixel (2x: 0.158).
 *
 * Shape: a workgroup of 8 waves renders 64 x 16 bases = 64 R x 16 R pixels (192 x 48, 256 x 64);
 * LDS 32 KiB of fragments + a 72 x 24 source tile as three f16 planes (row pitch 160 B: the A
 * fragment addressing of k_polar_mx<.., 8>). Sources: packed rgba16 / rgba16hf with at most an
 * identity PLANE_MAP in front (the fused PASS A of a plain plane); epilogues: FAST (dither + scale
 * into rgba16) and CHAIN (the map chain of an HDR pass in front of that tail). Everything else
 * keeps k_polar_pp (plh_polar_mxr_applies).
 *
 * 3 : 2 (720p -> 1080p, 1440p -> 4K; template parameter G = 2): a base index is a GROUP of two
 * source texels whose three outputs start at texel offsets the host has folded into the weights'
 * placement. M = 16 groups of rows = 32 source rows; the A fragment of M row m and row pair j is
 * source row 2 m + 2 j + (l >> 5), so the tile is kept as an even-row and an odd-row half and the
 * fragment addressing stays that of G = 1. A group's footprint is 10 rows (five row pairs); a
 * wave's 8 source columns are 4 groups x 3 phases = ONE half of 12 columns. 45 MFMAs per 16 x 12
 * outputs of a row phase (0.23 per pixel).
 */
#include "k_polar_mx.hiph"

#define MXR_WAVES   8
#define MXR_NT      (64 * MXR_WAVES)
#define MXR_TSX     (8 * MXR_WAVES)         // source columns per workgroup tile (without the halo)
#define MXR_TBY     16                      // base rows per workgroup tile
#define MXR_SRC_W   (MXR_TSX + 8)
#define MXR_PITCH   160
#define MXR_B_BYTES (PLH_MXR_FRAGS_PER_PHASE * 1024)
#define MXR_HP      (MXR_SRC_W / 2)

template <int R, int G, bool CHAIN>
__global__ __launch_bounds__(MXR_NT) __attribute__((amdgpu_waves_per_eu(4)))
void k_polar_mxr(const plh_pass p_)
{
    constexpr int MXR_TBX = MXR_TSX / G;            // base columns per workgroup tile
    constexpr int MXR_BW = 8 / G;                   // ... per wave
    constexpr int MXR_NH = MXR_BW / 4;              // halves of 4 bases x R phases
    constexpr int MXR_NJ = G == 1 ? 4 : 5;          // row pairs of a base's footprint
    constexpr int MXR_SRC_H = G * MXR_TBY + 8;      // (+1: the row that only meets zero weights)
    constexpr int MXR_PLANE = MXR_SRC_H * MXR_PITCH;
    constexpr int MXR_HALF = (MXR_SRC_H / 2) * MXR_PITCH;   // G = 2: the odd rows' half of a plane
    constexpr int MXR_NPAIRS = MXR_HP * MXR_SRC_H;
    constexpr int MXR_NV = (MXR_NPAIRS + MXR_NT - 1) / MXR_NT;
    constexpr int MXR_NFRAG = 4 * MXR_NJ * MXR_NH;  // fragments of a row phase
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const plh_polar_mx &mx = s.mx;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *bl = smem;
    unsigned char *tile = smem + MXR_B_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ln = lane & 15, lg = lane >> 4;
    // (CHAIN) the PQ pair's piecewise cubics behind the tile (pqseg.hiph), read from the first row
    // phase's epilogue on -- behind that phase's barriers
    unsigned char *segl = tile + 3 * MXR_PLANE;
    uintptr_t u_segp = 0;
    uint32_t u_segr = 0;
    if constexpr (CHAIN) {
        u_segp = (uintptr_t) p.chain.pq_seg_ptr;
        u_segr = (uint32_t) p.chain.pq_seg_rshift;
        asm volatile("" : "+s"(u_segp), "+s"(u_segr));
        if (u_segp)
            pq_seg_stage(segl, (const void *) u_segp, u_segr, (uint32_t) tid, (uint32_t) MXR_NT);
    }
    const pq_seg seg = pq_seg_view(segl, u_segr, (uint32_t) lane, u_segp != 0);
#ifdef PLH_MX_DEBUG
    const int dbg = s.pp_debug;     // (profiling aid: only in a -DPLH_MX_DEBUG build, k_polar_mx.hiph)
#else
    constexpr int dbg = 0;
#endif

    // uniforms of the per-phase code, read once and pinned in SGPRs (k_polar_mx.hiph says why)
    int u_w = p.width, u_h = p.height, u_dst_w = p.dst.w, u_dst_h = p.dst.h;
    int u_base_x = p.base_x, u_base_y = p.base_y, u_dir_x = p.dir_x, u_dir_y = p.dir_y;
    int u_dpitch = p.dst.pitch, u_nt = p.nt_store, u_fx0 = p.frag_x0, u_fy0 = p.frag_y0;
    int u_has_dither = p.epi.has_dither, u_has_scale = p.epi.has_scale, u_emask = p.epi.mask, u_esize = p.epi.size;
    int u_sx = mx.sx, u_sy = mx.sy;
    float u_ds = p.epi.dscale, u_di = p.epi.dinv, u_sc = p.epi.scale;
    uintptr_t u_dptr = (uintptr_t) p.dst.ptr, u_matrix = (uintptr_t) p.epi.matrix;
    uintptr_t u_dfx = (uintptr_t) mx.dfx, u_dfy = (uintptr_t) mx.dfy, u_bfrag = (uintptr_t) mx.bfrag;
    asm volatile("" : "+s"(u_w), "+s"(u_h), "+s"(u_dst_w), "+s"(u_dst_h), "+s"(u_base_x), "+s"(u_base_y),
                      "+s"(u_dir_x), "+s"(u_dir_y), "+s"(u_dpitch), "+s"(u_nt), "+s"(u_fx0), "+s"(u_fy0));
    asm volatile("" : "+s"(u_has_dither), "+s"(u_has_scale), "+s"(u_emask), "+s"(u_esize), "+s"(u_sx), "+s"(u_sy),
                      "+s"(u_ds), "+s"(u_di), "+s"(u_sc));
    asm volatile("" : "+s"(u_dptr), "+s"(u_matrix), "+s"(u_dfx), "+s"(u_dfy), "+s"(u_bfrag));
    typedef __attribute__((address_space(1))) const float gfloat;
    typedef __attribute__((address_space(1))) plh_u32x2 gpx;

    // XCD x works on the x-th contiguous eighth of the tiles (k_polar_mx.hiph)
    const int nbx = (u_w - 1 + u_sx) / R + 1;       // base indices of the frame
    const int tiles_x = (nbx + MXR_TBX - 1) / MXR_TBX;
    int bx, by;
    {
        const uint32_t total = gridDim.x, lin = blockIdx.x;
        const uint32_t q = total >> 3, r = total & 7u;
        const uint32_t xcd = lin & 7u, k = lin >> 3;
        const uint32_t t = xcd * q + min(xcd, r) + k;
        by = (int) (t / (uint32_t) tiles_x);
        bx = (int) (t - (uint32_t) by * (uint32_t) tiles_x);
    }
    // source texel of LDS (0, 0): base index 0 sits on source column org_x + 3
    const int ox = mx.org_x + MXR_TSX * bx, oy = mx.org_y + G * MXR_TBY * by;

    // ---- source tile -> LDS: pairs of horizontally adjacent texels, one 16-byte load each, all
    // issued together; decode (unorm sources: the reference's PASS A fused, rounded to f16 as the
    // rgba16hf intermediate would), channel-planar stores --------------------------------------
    int sw = s.src.w, sh = s.src.h, u_sfmt = s.src.fmt, u_spitch = s.src.pitch, u_npre = p.num_pre_ops;
    uintptr_t u_sptr = (uintptr_t) s.src.ptr;
    asm volatile("" : "+s"(sw), "+s"(sh), "+s"(u_sfmt), "+s"(u_spitch), "+s"(u_npre), "+s"(u_sptr));
    const bool unorm = u_sfmt == PLH_FMT_RGBA16;
    {
        uint4 v[MXR_NV];
        int ty[MXR_NV], tp[MXR_NV], sx[MXR_NV];
#pragma unroll
        for (int u = 0; u < MXR_NV; u++) {
            const int i = min(tid + u * MXR_NT, MXR_NPAIRS - 1);
            ty[u] = (int) (((float) i + 0.5f) * (1.0f / (float) MXR_HP));    // exact: i < 2^22
            tp[u] = i - ty[u] * MXR_HP;
            sx[u] = ox + 2 * tp[u];
            const int cy = min(max(oy + ty[u], 0), sh - 1), cx = min(max(sx[u], 0), sw - 2);
            const plh_u32x4 q = *(const __attribute__((address_space(1))) plh_u32x4 *)
                                    (u_sptr + (size_t) cy * (size_t) u_spitch + (size_t) cx * 8);
            v[u] = (dbg & 8) ? make_uint4(tid, u, 0, 0) : make_uint4(q.x, q.y, q.z, q.w);
        }
        // identity PLANE_MAP in front (components the plane does not carry: neutral values)
        const plh_op &om = p.ops[0];
        const int present = u_npre ? om.i1 : 4;
#pragma unroll
        for (int u = 0; u < MXR_NV; u++) {
            // a pair at clamped positions: beyond the left edge both texels are the pair's first,
            // beyond the right edge both its second
            const bool ldup = sx[u] < 0, hdup = sx[u] > sw - 2;
            const uint32_t q[4] = { hdup ? v[u].z : v[u].x, hdup ? v[u].w : v[u].y,
                                    ldup ? v[u].x : v[u].z, ldup ? v[u].y : v[u].w };
            uint32_t o[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const uint32_t a = (k & 1) ? q[k >> 1] >> 16 : q[k >> 1] & 0xffffu;
                const uint32_t b = (k & 1) ? q[2 + (k >> 1)] >> 16 : q[2 + (k >> 1)] & 0xffffu;
                if (unorm) {
                    float fa = plh_un16(a), fb = plh_un16(b);
                    if (k >= present) {
                        fa = om.f[k];
                        fb = om.f[k];
                    }
                    o[k] = mx_pack(fa, fb);
                } else {
                    o[k] = k >= present ? mx_pack(om.f[k], om.f[k]) : (a | (b << 16));
                }
            }
            if (tid + u * MXR_NT < MXR_NPAIRS) {
                const int trow = G == 1 ? ty[u] * MXR_PITCH : (ty[u] & 1) * MXR_HALF + (ty[u] >> 1) * MXR_PITCH;
                unsigned char *d = tile + trow + tp[u] * 4;
                *(uint32_t *) d = o[0];
                *(uint32_t *) (d + MXR_PLANE) = o[1];
                *(uint32_t *) (d + 2 * MXR_PLANE) = o[2];
            }
        }
    }

    // the lane's output columns: half h, column n = ln of it -> base 4 h + n / R, phase n % R
    // (n / R and n % R for n < 16 without a division: R is a template parameter)
    const int nb = ln / R, px = ln - nb * R;
    // A fragment of lane l for row pair j: the 16 bytes at tile row G (l & 15) + (l >> 5) + 2 j, column
    // 8 wave + 8 ((l >> 4) & 1) -- k_polar_mx's, with one wave tile per wave (G = 2: row (l >> 5) of
    // the pair picks the even / odd half, (l & 15) + j the row within it)
    const unsigned char *ab = tile + (G == 1 ? (ln + (lg >> 1)) * MXR_PITCH : (lg >> 1) * MXR_HALF + ln * MXR_PITCH) +
                              (8 * wave + 8 * (lg & 1)) * 2;
    constexpr int MXR_JSTEP = G == 1 ? 2 * MXR_PITCH : MXR_PITCH;
    const unsigned char *bfl = bl + lane * 16;

    // the rows that may be stored form one interval [ylo, ylo + ny) (k_polar_mx.hiph)
    int ylo, yhi;
    if (u_dir_y > 0) {
        ylo = max(0, -u_base_y);
        yhi = min(u_h, u_dst_h - u_base_y);
    } else {
        ylo = max(0, u_base_y - u_dst_h + 1);
        yhi = min(u_h, u_base_y + 1);
    }
    const uint32_t ny = (uint32_t) max(yhi - ylo, 0);
    const float ds = u_ds, di = u_di, sc = u_sc;
    float aw = 1.0f;        // alpha (not sampled: 1) behind dither and scale
    if (u_has_dither)
        aw = ds * di;
    if (u_has_scale)
        aw *= sc;
    const uint32_t awbits = plh_unorm16x2(0.0f, aw) & 0xffff0000u;
    const int eshift = __builtin_ctz((unsigned) max(u_esize, 1)) + 2;
    const ptrdiff_t step = (ptrdiff_t) u_dir_y * (ptrdiff_t) u_dpitch * R;     // one base row down

#pragma unroll 1
    for (int py = 0; py < R; py++) {
        // ---- this row phase's B fragments: global (L2) -> LDS, no registers --------------------
        __syncthreads();    // (everyone is done with the previous phase's fragments / the tile is complete)
#pragma unroll
        for (int f = wave; f < MXR_NFRAG; f += MXR_WAVES) {
            const uintptr_t g = u_bfrag + ((size_t) (py * PLH_MXR_FRAGS_PER_PHASE + f) * 64 + lane) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) g,
                                             (__attribute__((address_space(3))) void *) (bl + f * 1024), 16, 0, 0);
        }
        // the lane's four rows of this phase: base rows 4 lg + r, r < 4
        const int Y0 = R * (MXR_TBY * by + 4 * lg) + py - u_sy;
        float dfy[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
            dfy[r] = ((gfloat *) u_dfy)[min(max(Y0 + R * r, 0), u_h - 1)];
        __builtin_amdgcn_s_waitcnt(0);      // (the LDS-direct loads count on vmcnt)
        __syncthreads();

        // one half (4 bases x R phases of columns) at a time: contraction, then the epilogue of its
        // 4 rows -- 24 accumulator registers live instead of 48
#pragma unroll 1
        for (int h = 0; h < MXR_NH; h++) {
            const int X = R * (MXR_TBX * bx + MXR_BW * wave + 4 * h + nb) + px - u_sx;
            const int cpos = u_base_x + u_dir_x * X;
            const bool cok = ln < 4 * R && X >= 0 && X < u_w && cpos >= 0 && cpos < u_dst_w;
            const _Float16 dxh = (_Float16) ((gfloat *) u_dfx)[min(max(X, 0), u_w - 1)];
            const mx_f16x8 dx8 = { dxh, dxh, dxh, dxh, dxh, dxh, dxh, dxh };

            mx_f32x4 acc[3], ay[3];
            __builtin_amdgcn_s_setprio(3);
            if (dbg & 1) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    acc[ch] = ay[ch] = (mx_f32x4) ((float) (tid & 1) * 0.25f);
            } else {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    acc[ch] = ay[ch] = (mx_f32x4) (0.0f);
#pragma unroll
                for (int j = 0; j < MXR_NJ; j++) {
                    const unsigned char *bf = bfl + 4 * (MXR_NH * j + h) * 1024;
                    const mx_f16x8 bhi = *(const mx_f16x8 *) bf;
                    const mx_f16x8 blo = __builtin_elementwise_fma(*(const mx_f16x8 *) (bf + 2048), dx8,
                                                                   *(const mx_f16x8 *) (bf + 1024));
                    const mx_f16x8 bdy = *(const mx_f16x8 *) (bf + 3072);
#pragma unroll
                    for (int ch = 0; ch < 3; ch++) {
                        const mx_f16x8 a = *(const mx_f16x8 *) (ab + ch * MXR_PLANE + j * MXR_JSTEP);
                        acc[ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bhi, acc[ch], 0, 0, 0);
                        acc[ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, blo, acc[ch], 0, 0, 0);
                        ay[ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, bdy, ay[ch], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);

            // ---- epilogue: the lane's 4 rows of this phase and half -------------------------------
            const uint32_t ix4 = (uint32_t) ((X + u_fx0) & u_emask) << 2;
            const uintptr_t drow = u_dptr + (size_t) (u_base_y + u_dir_y * Y0) * (size_t) u_dpitch + (size_t) cpos * 8;
            float o[4][3];
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int ch = 0; ch < 3; ch++)
                    o[r][ch] = __builtin_fmaf(dfy[r], ay[ch][r], acc[ch][r]);
            }
            if constexpr (CHAIN) {
#pragma unroll 1
                for (int rp = 0; rp < 2; rp++) {
                    float4_t outs[2];
                    float bq[2];
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int Y = Y0 + R * (2 * rp + r);
                        const uint32_t iy = (uint32_t) (Y + u_fy0) & (uint32_t) u_emask;
                        bq[r] = u_has_dither ? *(gfloat *) (u_matrix + ((iy << eshift) | ix4)) : 0.0f;
                        outs[r] = { rp ? o[2 + r][0] : o[r][0], rp ? o[2 + r][1] : o[r][1],
                                    rp ? o[2 + r][2] : o[r][2], 1.0f };
                    }
                    run_map_chain<2, false, true, true>(outs, p, nullptr, &seg);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int Y = Y0 + R * (2 * rp + r);
                        const bool ok = cok && (uint32_t) (Y - ylo) < ny;
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
                        plh_u32x2 o2;
                        o2.x = plh_unorm16x2(c.x, c.y);
                        o2.y = (plh_unorm16x2(c.z, 0.0f) & 0xffffu) | awbits;
                        if (ok && !(dbg & 4)) {
                            const uintptr_t d = drow + (2 * rp + r) * step;
                            __builtin_nontemporal_store(o2, (gpx *) d);     // (always: a final rgba16 frame; k_polar_mx.hiph)
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int Y = Y0 + R * r;
                    const bool ok = cok && (uint32_t) (Y - ylo) < ny;
                    if (u_has_dither) {
                        const uint32_t iy = (uint32_t) (Y + u_fy0) & (uint32_t) u_emask;
                        const float b = *(gfloat *) (u_matrix + ((iy << eshift) | ix4));
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                            o[r][ch] = __builtin_floorf(ds * o[r][ch] + b) * di;
                    }
                    if (u_has_scale) {
#pragma unroll
                        for (int ch = 0; ch < 3; ch++)
                            o[r][ch] *= sc;
                    }
                    plh_u32x2 o2;
                    o2.x = plh_unorm16x2(o[r][0], o[r][1]);
                    o2.y = (plh_unorm16x2(o[r][2], 0.0f) & 0xffffu) | awbits;
                    if (ok && !(dbg & 4)) {
                        const uintptr_t d = drow + r * step;
                        __builtin_nontemporal_store(o2, (gpx *) d);     // (always: a final rgba16 frame; k_polar_mx.hiph)
                    }
                }
            }
        }
    }
}

// the pass k_polar_mxr is written for (file header); fills pass->chain / pass->epi
bool plh_polar_mxr_applies(plh_pass *pass)
{
    const plh_sampler_args &s = pass->s;
    if (!s.pp || s.mx.enabled != 3 || (s.comp_mask & 0xf) != 0x7 || pass->transpose ||
        s.address_mode != PLH_ADDRESS_CLAMP || pass->dst.fmt != PLH_FMT_RGBA16)
        return false;
    if (s.src.fmt != PLH_FMT_RGBA16 && s.src.fmt != PLH_FMT_RGBA16F)
        return false;
    // pre-ops: none, or a fused identity PLANE_MAP (a plain plane in front: decode only)
    if (pass->num_pre_ops > 1)
        return false;
    if (pass->num_pre_ops == 1) {
        const plh_op &op = pass->ops[0];
        if (op.kind != PLH_OP_PLANE_MAP || !op.i2)
            return false;
    }
    plh_match_map_chain(pass);
    if (pass->chain.enabled)
        return !pass->chain.contrast_recovery && !pass->epi.has_alpha;
    plh_match_fast_epilogue(pass);
    return pass->epi.enabled && !pass->epi.has_alpha;
}

int plh_launch_polar_mxr(hipStream_t stream, const plh_pass *pass)
{
    const int R = pass->s.mx.ratio, G = pass->s.mx.group;
    if (G != 1 && G != 2)
        return -1000;
    const int nbx = (pass->width - 1 + pass->s.mx.sx) / R + 1, nby = (pass->height - 1 + pass->s.mx.sy) / R + 1;
    const int tbx = MXR_TSX / G;
    const int tiles = ((nbx + tbx - 1) / tbx) * ((nby + MXR_TBY - 1) / MXR_TBY);
    size_t shmem = MXR_B_BYTES + (size_t) 3 * (G * MXR_TBY + 8) * MXR_PITCH;
    const bool chain = pass->chain.enabled;
    const plh_pass *arg = pass;
    plh_pass local;
    if (chain && pass->chain.pq_seg) {
        // the PQ pair as piecewise cubics in LDS (pqseg.hiph)
        local = *pass;
        local.chain.pq_seg_rshift = 0;
        local.chain.pq_seg_ptr = plh_pqseg_tables((plh_stream) stream, pass->chain.pq_seg_consts);
        if (local.chain.pq_seg_ptr)
            shmem += PQSEG_BYTES;
        arg = &local;
    }
#define MXR_LAUNCH(RR, GG, CH) PLH_LAUNCH_LAST((k_polar_mxr<RR, GG, CH>), dim3(tiles), dim3(MXR_NT), shmem, stream, *arg)
    if (G == 2 && R == 3 && chain)  MXR_LAUNCH(3, 2, true);
    else if (G == 2 && R == 3)      MXR_LAUNCH(3, 2, false);
    else if (G == 2)
        return -1000;
    else if (R == 3 && chain)       MXR_LAUNCH(3, 1, true);
    else if (R == 3)                MXR_LAUNCH(3, 1, false);
    else if (R == 4 && chain)       MXR_LAUNCH(4, 1, true);
    else if (R == 4)                MXR_LAUNCH(4, 1, false);
    else
        return -1000;
#undef MXR_LAUNCH
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
