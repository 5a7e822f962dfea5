/*
 * libplacebo-hip — debanding kernel (K6).
 *
 * Device half of pl_shader_deband (src/shaders/sampling.c:183-275) with the
 * pcg3d PRNG of sh_prng (src/shaders.c:965-998):
 *
 *   color = texel(pos);  res = color.<mask>
 *   for i = 1 .. iterations:
 *       d   = rand().xy * (i * radius, 2*pi);  d = d.x * (cos d.y, sin d.y)
 *       avg = 0.25 * (T(+dx,+dy) + T(-dx,+dy) + T(-dx,-dy) + T(+dx,-dy))   (nearest fetches)
 *       res = |res - avg| > threshold / i ? res : avg
 *   res += min(|res - neutral|, grain) * (rand() - 0.5)
 *   color.<mask> = res;  color *= scale
 *
 * The PRNG state is uvec3(gl_FragCoord.xy, frame index); integer arithmetic
 * mod 2^32, i.e. bit-exact. sin/cos are the native v_sin_f32 / v_cos_f32 (as a Vulkan driver
 * would emit); sample positions agree with the libm oracle except within ~1e-5 px of a texel
 * boundary, where the neighbouring texel may be picked (tests: >= 99.9 % identical pixels).
 *
 * Launch shape: 64x4 lanes, one pixel per lane. 4*iterations + 1 data-dependent
 * 8..16-byte gathers per pixel within `radius` texels of it: served by L2/MALL,
 * HBM traffic stays one read + one write of the plane.
 *
 * The kernel is VALU-bound, not gather-bound (8K plane + PQ linearize: 408 VALU instructions
 * per pixel, VALU busy 86 % of the 487 us; profiles/r02_*). Tried and dropped: a 64x16 block
 * staging its texels + a 17-texel halo in LDS (38 KiB) and gathering from there, four pixels
 * per lane sharing one walk through the op interpreter -- 652 us: the window bookkeeping costs
 * more VALU than the gathers cost anything, and occupancy drops from 5 to 4 waves.
 */
#include "colorops.hiph"
#include "prng.hiph"
#include "samplers.hiph"
#include "backend.h"

#define DEBAND_BW 64
#define DEBAND_BH 4

template <bool LITE>
__global__ __launch_bounds__(DEBAND_BW * DEBAND_BH)
void k_deband(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    const int idx = blockIdx.x * DEBAND_BW + threadIdx.x;
    const int idy = blockIdx.y * DEBAND_BH + threadIdx.y;
    const float mx = p.out_scale[0] * ((float) idx + 0.5f);
    const float my = p.out_scale[1] * ((float) idy + 0.5f);
    const float px = plh_attr(s.pos, 0, mx, my), py = plh_attr(s.pos, 1, mx, my);
    const frag_t fc = { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f, 0.0f, 0,
                        mx, my };

    float4_t color = tex_nearest(s.src, s.address_mode, px, py);
    float res[3] = { color.x, color.y, color.z };
    const uint32_t mask = s.comp_mask & 7u;

    prng3 st = { (uint32_t) fc.x, (uint32_t) fc.y, s.prng_seed };
    float rnd[3];
    for (int i = 1; i <= s.iterations; i++) {
        pcg3d(st, rnd);
        float dx = rnd[0] * ((float) i * s.db_radius);
        const float ang = rnd[1] * 6.283185f;       // "%f" of 2*pi
#ifdef PLH_DEBAND_OCML_SINCOS
        const float dy = dx * sinf(ang);
        dx = dx * cosf(ang);
#else
        // v_sin_f32 / v_cos_f32 take revolutions: what a GLSL sin()/cos() compiles to on this
        // hardware (mul by 1/2pi + the native instruction)
        const float rev = ang * 0.15915494309189532f;
        const float dy = dx * __builtin_amdgcn_sinf(rev);
        dx = dx * __builtin_amdgcn_cosf(rev);
#endif
        float avg[3] = {0.0f, 0.0f, 0.0f};
        const float ox[4] = { dx, -dx, -dx, dx }, oy[4] = { dy, dy, -dy, -dy };
        int tx[4], ty[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // textureLod(tex, pos + pt * vec2(X, Y)) with NEAREST filtering
            const float qx = px + s.pt[0] * ox[k], qy = py + s.pt[1] * oy[k];
            tx[k] = plh_wrap((int) __builtin_floorf(qx * (float) s.src.w), s.src.w, s.address_mode);
            ty[k] = plh_wrap((int) __builtin_floorf(qy * (float) s.src.h), s.src.h, s.address_mode);
        }
        float4_t t[4];
        plh_fetch_n<4>(s.src, tx, ty, t);   // the four gathers in flight together
#pragma unroll
        for (int k = 0; k < 4; k++) {
            avg[0] += t[k].x; avg[1] += t[k].y; avg[2] += t[k].z;
        }
        const float bound = s.db_threshold / (float) i;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float a = avg[c] * 0.25f;
            const float diff = __builtin_fabsf(res[c] - a);
            if (mask & (1u << c))
                res[c] = diff > bound ? res[c] : a;
        }
    }

    if (s.db_grain > 0.0f) {
        pcg3d(st, rnd);
        // T(rand): the first num_comps components of the vec3, in enabled-component order
        int k = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (!(mask & (1u << c)))
                continue;
            const float strength = fminf(__builtin_fabsf(res[c] - s.db_neutral[c]), s.db_grain);
            res[c] += strength * (rnd[k++] - 0.5f);
        }
    }

    if (mask & 1u) color.x = res[0];
    if (mask & 2u) color.y = res[1];
    if (mask & 4u) color.z = res[2];
    color.x *= s.scale; color.y *= s.scale; color.z *= s.scale; color.w *= s.scale;

    float4_t outs[1] = { color };
    const frag_t fcs[1] = { fc };
    apply_ops_n<1, false, LITE>(outs, p.ops, 0, p.num_ops, fcs);

    const int sx[1] = { p.base_x + p.dir_x * (p.transpose ? idy : idx) };
    const int sy[1] = { p.base_y + p.dir_y * (p.transpose ? idx : idy) };
    const bool ok[1] = { p.out_scale[0] * (float) idx < 1.0f && p.out_scale[1] * (float) idy < 1.0f &&
                         sx[0] >= 0 && sy[0] >= 0 && sx[0] < p.dst.w && sy[0] < p.dst.h };
    plh_store_n<1>(p.dst, sx, sy, ok, outs);
}

/*
 * k_deband_fast: the renderer's debanding pass as one small kernel -- a whole rgba16 plane at
 * native resolution (output pixel (x, y) sits on texel (x, y)), clamp addressing, RGB mask,
 * ops = [identity PLANE_MAP] [LINEARIZE] [SIGMOIDIZE] (each optional), rgba16hf target. Same arithmetic as k_deband (the PRNG,
 * the tap positions from the interpolated attribute, the comparison, op_linearize itself) with
 * one exception: the four taps of a channel are summed AS INTEGERS and decoded once
 * (S / 262140, one rounding) where the general kernel decodes each tap (v / 65535) and adds the
 * floats (four roundings) -- 21 instead of 57 instructions per pixel, an average that is closer to
 * the exact one, and an rgba16hf result that differs from the general kernel's by one f16 ulp on
 * a fraction of a percent of the samples (tests/test_gpu_ortho_deband.py renders with both). The
 * DECISION between a sample's value and the average is the reference's in every kernel (round 6:
 * deband_compare below; before that a sample within an ulp of the threshold could keep its value
 * here and take the average there) -- the general kernel keeps the reference's four-float sum for
 * the value as well, because it is the one held to the oracle bit for bit on fp32 targets. What
 * else goes is what the general kernel pays for being general: the op interpreter, format and
 * address-mode switches, 64-bit address arithmetic (the taps: one 24-bit multiply-add against a
 * uniform base), the alpha channel of the four taps and of a plane that has none, the IEEE
 * division by the iteration count when there is one iteration, half of the store instructions.
 * Two horizontally adjacent pixels per lane: one 16-byte load for the two centre texels, one
 * 16-byte store.
 */
#define DBF_BW 64
#define DBF_BH 4
// the debanded plane's store (an rgba16hf intermediate that the next pass reads)
#ifdef PLH_DEBAND_NT
#define DEBAND_STORE16(dp, a, b, c, d) do { const plh_u32x4 v_ = { a, b, c, d }; \
        __builtin_nontemporal_store(v_, (plh_u32x4 *) (dp)); } while (0)
#else
#define DEBAND_STORE16(dp, a, b, c, d) (*(uint4 *) (dp) = make_uint4(a, b, c, d))
#endif

// The fast kernels' threshold decision is the REFERENCE's. They average the four taps of a channel
// as an integer sum (one rounding; the reference and the general kernel decode each tap and add
// four floats: three more roundings), so |res - avg| can land an ulp or two on the other side of
// the threshold -- and a sample that keeps its value in one kernel and takes the average in the
// other differs by the threshold itself, hundreds of 16-bit codes.
// In exact arithmetic res and avg are multiples of 1 / 262140, the computed differences of either
// formulation lie within 2e-7 of that lattice (spacing 3.8e-6), so the two can only disagree when
// the threshold itself lies within 2e-7 of a lattice point and the sample sits on it. Those samples
// are recognised (|diff - bound| <= 1e-6: none at all unless the threshold is so aligned) and take
// the decision from the reference's own four-float average; where no lane of a wave has one -- the
// presets' threshold on full-range planes: always -- the branch is not taken.
#define DEBAND_NEAR 1e-6f
DEV float deband_avg_ref(const plh_u32x2 *r, int c)
{
    float a = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t code = c == 0 ? (r[k].x & 0xffffu) : c == 1 ? (r[k].x >> 16) : (r[k].y & 0xffffu);
        a += plh_un16(code);
    }
    return a * 0.25f;
}

// one iteration's comparison for one pixel: res = |res - avg| > bound ? res : avg (avg: the integer
// sum's; the decision: see above)
DEV void deband_compare(float (&res)[3], const plh_u32x2 *r, const float (&avg)[3], float bound)
{
    bool keep[3], near = false;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float diff = __builtin_fabsf(res[c] - avg[c]);
        keep[c] = diff > bound;
        near |= __builtin_fabsf(diff - bound) <= DEBAND_NEAR;
    }
    if (__builtin_amdgcn_ballot_w64(near) != 0) {
#pragma unroll
        for (int c = 0; c < 3; c++)
            keep[c] = __builtin_fabsf(res[c] - deband_avg_ref(r, c)) > bound;
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
        res[c] = keep[c] ? res[c] : avg[c];
}

__global__ __launch_bounds__(DBF_BW * DBF_BH)
void k_deband_fast(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    // (Tried: an XCD-aware launch order -- XCD x working through the x-th contiguous eighth of the
    // tiles, so that the +-16 rows a block's taps reach stay in that XCD's L2. HBM fetch fell from
    // 661 MB to 277 MB per 8K plane (265 MB algorithmic) and the kernel got slower, 398 -> 477 us,
    // A/B in one process: it is VALU-bound, the extra traffic was hidden, and 32 CUs gathering
    // from the same 2 MB band queue on the same L2 channels.)
    const int bx = blockIdx.x, by = blockIdx.y;
    const int idx0 = 2 * (bx * DBF_BW + threadIdx.x);
    const int idy = by * DBF_BH + threadIdx.y;
    if (idx0 >= p.width || idy >= p.height)
        return;
    typedef __attribute__((address_space(1))) const unsigned char gbyte;
    gbyte *sp = (gbyte *) (uintptr_t) s.src.ptr;
    const uint32_t spitch = s.src.pitch;
    const int srcw = s.src.w, srch = s.src.h;
    const float sw = (float) srcw, sh = (float) srch;
    // ops: [identity PLANE_MAP] [LINEARIZE] [SIGMOIDIZE] (deband_fast_applies)
    const bool has_map = p.num_ops > 0 && p.ops[0].kind == PLH_OP_PLANE_MAP;
    const int i_lin = has_map ? 1 : 0;
    const bool has_lin = i_lin < p.num_ops && p.ops[i_lin].kind == PLH_OP_LINEARIZE;
    const bool has_sig = p.num_ops > 0 && p.ops[p.num_ops - 1].kind == PLH_OP_SIGMOIDIZE;
    const plh_op &o_map = p.ops[0], &o_lin = p.ops[i_lin], &o_sig = p.ops[p.num_ops > 0 ? p.num_ops - 1 : 0];
    const bool has_alpha = !has_map || o_map.i1 >= 4;   // else alpha is the PLANE_MAP's neutral value
    const float my = p.out_scale[1] * ((float) idy + 0.5f);

    // the two centre texels (idx0 is even and rows are 256-byte aligned: 16-byte aligned)
    const bool two = idx0 + 1 < p.width;
    const uint32_t coff = __umul24((uint32_t) idy, spitch) + ((uint32_t) idx0 << 3);
    uint4 centre;
    if (two) {
        const plh_u32x4 c4 = *(const __attribute__((address_space(1))) plh_u32x4 *) (sp + coff);
        centre = make_uint4(c4.x, c4.y, c4.z, c4.w);
    } else {
        const plh_u32x2 c0 = *(const __attribute__((address_space(1))) plh_u32x2 *) (sp + coff);
        centre = make_uint4(c0.x, c0.y, c0.x, c0.y);
    }

    // The lane's two pixels side by side, stage by stage: both pixels' random offsets, then all
    // eight gathers, then the comparisons -- one memory round trip per iteration and wave instead
    // of two (the kernel is bound by each wave's latency chain PRNG -> sin / cos -> address ->
    // gather -> compare, not by any pipe: DESIGN.md 9), and the six linearisations as independent
    // chains under one switch (transfer.hiph: op_linearize_values).
    float px[2], py[2], res[2][3], alpha[2];
    prng3 st[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int idx = idx0 + q;
        const float mx = p.out_scale[0] * ((float) idx + 0.5f);
        px[q] = plh_attr(s.pos, 0, mx, my);
        py[q] = plh_attr(s.pos, 1, mx, my);
        const uint32_t cx = q ? centre.z : centre.x, cy = q ? centre.w : centre.y;
        res[q][0] = plh_un16(cx & 0xffffu);
        res[q][1] = plh_un16(cx >> 16);
        res[q][2] = plh_un16(cy & 0xffffu);
        alpha[q] = has_alpha ? plh_un16(cy >> 16) : 1.0f;
        st[q] = { (uint32_t) ((float) (idx + p.frag_x0) + 0.5f),
                  (uint32_t) ((float) (idy + p.frag_y0) + 0.5f), s.prng_seed };
    }
    float rnd[2][3];
    for (int i = 1; i <= s.iterations; i++) {
        plh_u32x2 raw[2][4];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            pcg3d(st[q], rnd[q]);
            float dx = rnd[q][0] * ((float) i * s.db_radius);
            const float rev = (rnd[q][1] * 6.283185f) * 0.15915494309189532f;
            const float dy = dx * __builtin_amdgcn_sinf(rev);
            dx = dx * __builtin_amdgcn_cosf(rev);
            const float ox[4] = { dx, -dx, -dx, dx }, oy[4] = { dy, dy, -dy, -dy };
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float qx = px[q] + s.pt[0] * ox[k], qy = py[q] + s.pt[1] * oy[k];
                // (clamp(floor(v), 0, n - 1): the conversion truncates toward zero, which differs
                // from floor only for negative v -- where both end up clamped to 0)
                const int tx = min(max((int) (qx * sw), 0), srcw - 1);
                const int ty = min(max((int) (qy * sh), 0), srch - 1);
                // (pitch < 2^24 and rows < 2^24: one v_mad_u32_u24; the plane is < 4 GiB)
                const uint32_t off = __umul24((uint32_t) ty, spitch) + ((uint32_t) tx << 3);
                raw[q][k] = *(const __attribute__((address_space(1))) plh_u32x2 *) (sp + off);
            }
        }
        // (i = 1, the presets' one iteration: no IEEE division)
        const float bound = i == 1 ? s.db_threshold : s.db_threshold / (float) i;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const plh_u32x2 *r = raw[q];
            // the four taps of a channel: integer sum (< 2^18), one conversion, one product
            const uint32_t s0 = (r[0].x & 0xffffu) + (r[1].x & 0xffffu) + (r[2].x & 0xffffu) + (r[3].x & 0xffffu);
            const uint32_t s1 = (r[0].x >> 16) + (r[1].x >> 16) + (r[2].x >> 16) + (r[3].x >> 16);
            const uint32_t s2 = (r[0].y & 0xffffu) + (r[1].y & 0xffffu) + (r[2].y & 0xffffu) + (r[3].y & 0xffffu);
            const float avg[3] = { (float) s0 * (1.0f / 262140.0f), (float) s1 * (1.0f / 262140.0f),
                                   (float) s2 * (1.0f / 262140.0f) };
            deband_compare(res[q], r, avg, bound);
        }
    }
    if (s.db_grain > 0.0f) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            pcg3d(st[q], rnd[q]);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float strength = fminf(__builtin_fabsf(res[q][c] - s.db_neutral[c]), s.db_grain);
                res[q][c] += strength * (rnd[q][c] - 0.5f);
            }
        }
    }
    float lin[6];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        lin[3 * q] = res[q][0] * s.scale;
        lin[3 * q + 1] = res[q][1] * s.scale;
        lin[3 * q + 2] = res[q][2] * s.scale;
        alpha[q] *= s.scale;
        // identity PLANE_MAP of the first i1 components: the others take their neutral values
        if (has_map) {
            if (o_map.i1 < 4) alpha[q] = o_map.f[3];
            if (o_map.i1 < 3) lin[3 * q + 2] = o_map.f[2];
            if (o_map.i1 < 2) lin[3 * q + 1] = o_map.f[1];
        }
    }
    if (has_lin)
        op_linearize_values(lin, o_lin);
    if (has_sig) {
#pragma unroll
        for (int k = 0; k < 6; k++)
            lin[k] = sigmoid1(lin[k], o_sig.f);
    }
    uint32_t packed[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        packed[q][0] = (uint32_t) plh_f2h(lin[3 * q]) | ((uint32_t) plh_f2h(lin[3 * q + 1]) << 16);
        packed[q][1] = (uint32_t) plh_f2h(lin[3 * q + 2]) | ((uint32_t) plh_f2h(alpha[q]) << 16);
    }

    char *dp = (char *) p.dst.ptr + (size_t) idy * p.dst.pitch + (size_t) idx0 * 8;
    if (two)
        DEBAND_STORE16(dp, packed[0][0], packed[0][1], packed[1][0], packed[1][1]);
    else
        *(uint2 *) dp = make_uint2(packed[0][0], packed[0][1]);
}

/*
 * k_deband_lds: k_deband_fast with the taps gathered from an LDS window instead of from memory.
 * k_deband_fast's four taps per pixel are 8-byte gathers at random offsets within +-16 texels:
 * every lane touches its own cache line (the vector L1 looks them up one by one), a wave's taps
 * cover ~330 lines (42 KiB: more than a CU's L1). Measured per 8K plane: 152.5 M L1 line accesses
 * (596 k per CU: 250 us at one per clock) and 46.5 M read requests to L2 (3-6 GB for 265 MB of
 * image) -- that, not instruction issue or HBM, is the 385 us the
 * kernel takes (and explains what rounds 2-4 measured: fewer instructions bought nothing, and
 * concentrating each XCD on one band made it slower). Here a workgroup of 8 waves stages the
 * 98 x 66 texels around its 64 x 32 pixels once (coalesced 8-byte loads: 49.7 M L1 accesses and
 * 9.5 M L2 requests per plane), and the taps are ds_read_b64. Same arithmetic as k_deband_fast, bit for
 * bit (same PRNG, positions, integer tap sums); for radius * iterations <= 16.
 */
#define DBL_TW 64
#define DBL_HALO 17         // 16 + one texel of rounding slack
#define DBL_WW (DBL_TW + 2 * DBL_HALO)
#define DBL_NT 512
#ifndef DBL_NP
#define DBL_NP 2         // pixels a lane works on at a time (2, 4 or 8)
#endif

// DBL_TH: rows of a workgroup's tile -- 32 (three workgroups per CU), or 16 for frames so small
// that 32-row tiles would not fill the machine twice (four per CU, twice the tiles)
template <int DBL_TH>
__global__ __launch_bounds__(DBL_NT)
void k_deband_lds(const plh_pass p_)
{
    constexpr int DBL_WH = DBL_TH + 2 * DBL_HALO;
    const plh_pass &p = plh_kernarg_pass();
    const plh_sampler_args &s = p.s;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // (an LDS pointer type: 32-bit index arithmetic -- through a generic pointer every tap's address
    // was a v_mad_u64_u32)
    typedef __attribute__((address_space(3))) plh_u32x2 lds_px;
    lds_px *win = (lds_px *) (__attribute__((address_space(3))) unsigned char *) smem;
    const int tid = threadIdx.x;
    // Workgroups go to the 8 XCDs round-robin and every XCD has its own L2: XCD x works down the
    // x-th vertical band of tile columns, row-major within the band, so that the 34 of 98 window
    // columns AND rows a tile shares with its neighbours are found in that L2 (a band's tile row
    // is < 1 MB) instead of being fetched from HBM once per tile (2.35 x the plane).
    int bx, by;
    {
        const uint32_t tiles_x = (uint32_t) (p.width + DBL_TW - 1) / DBL_TW;
        const uint32_t tiles_y = (uint32_t) (p.height + DBL_TH - 1) / DBL_TH;
        const uint32_t lin = blockIdx.x, xcd = lin & 7u, k = lin >> 3;
        const uint32_t q = tiles_x >> 3, r = tiles_x & 7u;
        const uint32_t bw = q + (xcd < r ? 1u : 0u), bstart = xcd * q + min(xcd, r);
        if (k >= bw * tiles_y)
            return;
        by = (int) (k / bw);
        bx = (int) (bstart + (k - (uint32_t) by * bw));
    }
    const int x0 = bx * DBL_TW, y0 = by * DBL_TH;
    const int wx0 = x0 - DBL_HALO, wy0 = y0 - DBL_HALO;
    typedef __attribute__((address_space(1))) const unsigned char gbyte;
    gbyte *sp = (gbyte *) (uintptr_t) s.src.ptr;
    const uint32_t spitch = s.src.pitch;
    const int srcw = s.src.w, srch = s.src.h;
    const float sw = (float) srcw, sh = (float) srch;

    // ---- the window: every texel once, clamped at the frame's edges (the taps clamp the same way,
    // so a position outside the frame is never addressed). Wave w takes rows w, w + 8, ...: the row
    // (clamp, pitch) is scalar arithmetic, a lane's two columns (l and 64 + l) are computed once,
    // and every load / store is base + constant -- 2 vector instructions per texel instead of 17 --
    {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const uint32_t colA = (uint32_t) min(max(wx0 + lane, 0), srcw - 1) << 3;
        const uint32_t colB = (uint32_t) min(max(wx0 + 64 + lane, 0), srcw - 1) << 3;
        const bool hasB = lane < DBL_WW - 64;
        constexpr int ROWS = (DBL_WH + 7) / 8;
        plh_u32x2 va[ROWS], vb[ROWS];
#pragma unroll
        for (int u = 0; u < ROWS; u++) {
            const int r = min(wave + 8 * u, DBL_WH - 1);
            gbyte *row = sp + (size_t) min(max(wy0 + r, 0), srch - 1) * (size_t) spitch;
            va[u] = *(const __attribute__((address_space(1))) plh_u32x2 *) (row + colA);
            vb[u] = *(const __attribute__((address_space(1))) plh_u32x2 *) (row + colB);
        }
        lds_px *wa = win + wave * DBL_WW + lane;
#pragma unroll
        for (int u = 0; u < ROWS; u++) {
            if (wave + 8 * u < DBL_WH) {
                wa[8 * u * DBL_WW] = va[u];
                if (hasB)
                    wa[8 * u * DBL_WW + 64] = vb[u];
            }
        }
    }
    __syncthreads();

    // ops: [identity PLANE_MAP] [LINEARIZE] [SIGMOIDIZE] (deband_fast_applies)
    const bool has_map = p.num_ops > 0 && p.ops[0].kind == PLH_OP_PLANE_MAP;
    const int i_lin = has_map ? 1 : 0;
    const bool has_lin = i_lin < p.num_ops && p.ops[i_lin].kind == PLH_OP_LINEARIZE;
    const bool has_sig = p.num_ops > 0 && p.ops[p.num_ops - 1].kind == PLH_OP_SIGMOIDIZE;
    const plh_op &o_map = p.ops[0], &o_lin = p.ops[i_lin], &o_sig = p.ops[p.num_ops > 0 ? p.num_ops - 1 : 0];
    const bool has_alpha = !has_map || o_map.i1 >= 4;   // else alpha is the PLANE_MAP's neutral value

    // 32 lanes cover a row of the tile (two pixels each), a wave two rows, the workgroup 16 rows; a
    // lane works on DBL_NP pixels at a time -- pairs 16 rows apart -- stage by stage (all their
    // random offsets, all their taps, all their comparisons): the workgroup's LDS caps the CU at 4
    // waves per SIMD, so the independent chains have to come from inside the wave
    constexpr int NP = DBL_NP, PAIRS = NP / 2;
#pragma unroll 1
    for (int step = 0; step < DBL_TH / (16 * PAIRS); step++) {
        const int lx = 2 * (tid & 31), idx0 = x0 + lx;
        const bool two = idx0 + 1 < p.width;
        int idy[PAIRS];
        float px[NP], py[NP], res[NP][3], alpha[NP];
        prng3 st[NP];
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int ly = (step * PAIRS + (q >> 1)) * 16 + (tid >> 5);
            idy[q >> 1] = y0 + ly;
            const int idx = idx0 + (q & 1);
            const float mx = p.out_scale[0] * ((float) idx + 0.5f);
            const float my = p.out_scale[1] * ((float) (y0 + ly) + 0.5f);
            px[q] = plh_attr(s.pos, 0, mx, my);
            py[q] = plh_attr(s.pos, 1, mx, my);
            // (pixels beyond the frame are computed from the window's clamped texels and dropped at
            // the store)
            const plh_u32x2 c = win[(ly + DBL_HALO) * DBL_WW + lx + DBL_HALO + ((q & 1) && two ? 1 : 0)];
            res[q][0] = plh_un16(c.x & 0xffffu);
            res[q][1] = plh_un16(c.x >> 16);
            res[q][2] = plh_un16(c.y & 0xffffu);
            alpha[q] = has_alpha ? plh_un16(c.y >> 16) : 1.0f;
            st[q] = { (uint32_t) ((float) (idx + p.frag_x0) + 0.5f),
                      (uint32_t) ((float) (y0 + ly + p.frag_y0) + 0.5f), s.prng_seed };
        }
        float rnd[NP][3];
        for (int i = 1; i <= s.iterations; i++) {
            plh_u32x2 raw[NP][4];
#pragma unroll
            for (int q = 0; q < NP; q++) {
                pcg3d(st[q], rnd[q]);
                float dx = rnd[q][0] * ((float) i * s.db_radius);
                const float rev = (rnd[q][1] * 6.283185f) * 0.15915494309189532f;
                const float dy = dx * __builtin_amdgcn_sinf(rev);
                dx = dx * __builtin_amdgcn_cosf(rev);
                const float ox[4] = { dx, -dx, -dx, dx }, oy[4] = { dy, dy, -dy, -dy };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float qx = px[q] + s.pt[0] * ox[k], qy = py[q] + s.pt[1] * oy[k];
                    const int tx = min(max((int) (qx * sw), 0), srcw - 1);
                    const int ty = min(max((int) (qy * sh), 0), srch - 1);
                    // (within the window by construction: the launcher checks radius * iterations
                    // <= 16, the halo has one texel of slack, and clamping to the frame only moves a
                    // tap towards the pixel)
                    raw[q][k] = win[(ty - wy0) * DBL_WW + (tx - wx0)];
                }
            }
            const float bound = i == 1 ? s.db_threshold : s.db_threshold / (float) i;
#pragma unroll
            for (int q = 0; q < NP; q++) {
                const plh_u32x2 *r = raw[q];
                const uint32_t s0 = (r[0].x & 0xffffu) + (r[1].x & 0xffffu) + (r[2].x & 0xffffu) + (r[3].x & 0xffffu);
                const uint32_t s1 = (r[0].x >> 16) + (r[1].x >> 16) + (r[2].x >> 16) + (r[3].x >> 16);
                const uint32_t s2 = (r[0].y & 0xffffu) + (r[1].y & 0xffffu) + (r[2].y & 0xffffu) + (r[3].y & 0xffffu);
                const float avg[3] = { (float) s0 * (1.0f / 262140.0f), (float) s1 * (1.0f / 262140.0f),
                                       (float) s2 * (1.0f / 262140.0f) };
                deband_compare(res[q], r, avg, bound);
            }
        }
        if (s.db_grain > 0.0f) {
#pragma unroll
            for (int q = 0; q < NP; q++) {
                pcg3d(st[q], rnd[q]);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const float strength = fminf(__builtin_fabsf(res[q][c] - s.db_neutral[c]), s.db_grain);
                    res[q][c] += strength * (rnd[q][c] - 0.5f);
                }
            }
        }
        float lin[3 * NP];
#pragma unroll
        for (int q = 0; q < NP; q++) {
            lin[3 * q] = res[q][0] * s.scale;
            lin[3 * q + 1] = res[q][1] * s.scale;
            lin[3 * q + 2] = res[q][2] * s.scale;
            alpha[q] *= s.scale;
            if (has_map) {
                if (o_map.i1 < 4) alpha[q] = o_map.f[3];
                if (o_map.i1 < 3) lin[3 * q + 2] = o_map.f[2];
                if (o_map.i1 < 2) lin[3 * q + 1] = o_map.f[1];
            }
        }
        if (has_lin)
            op_linearize_values(lin, o_lin);
        if (has_sig) {
#pragma unroll
            for (int k = 0; k < 3 * NP; k++)
                lin[k] = sigmoid1(lin[k], o_sig.f);
        }
#pragma unroll
        for (int r = 0; r < PAIRS; r++) {
            uint32_t packed[2][2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const float *v = lin + 3 * (2 * r + q);
                packed[q][0] = (uint32_t) plh_f2h(v[0]) | ((uint32_t) plh_f2h(v[1]) << 16);
                packed[q][1] = (uint32_t) plh_f2h(v[2]) | ((uint32_t) plh_f2h(alpha[2 * r + q]) << 16);
            }
            const bool inside = idx0 < p.width && idy[r] < p.height;
            char *dp = (char *) p.dst.ptr + (size_t) idy[r] * p.dst.pitch + (size_t) idx0 * 8;
            if (inside && two)
                DEBAND_STORE16(dp, packed[0][0], packed[0][1], packed[1][0], packed[1][1]);
            else if (inside)
                *(uint2 *) dp = make_uint2(packed[0][0], packed[0][1]);
        }
    }
}

// [identity PLANE_MAP] [LINEARIZE] [SIGMOIDIZE], each optional, in this order, nothing else: the
// debanding pass of a plain plane in gamma light, in front of a downscaler (linear light) and in
// front of an upscaler (sigmoidized linear light: pl_render_high_quality_params on SDR video)
static bool deband_fast_ops(const plh_pass *pass)
{
    int i = 0;
    if (i < pass->num_ops && pass->ops[i].kind == PLH_OP_PLANE_MAP) {
        if (!pass->ops[i].i2 || pass->ops[i].i1 < 1)
            return false;
        i++;
    }
    if (i < pass->num_ops && pass->ops[i].kind == PLH_OP_LINEARIZE)
        i++;
    if (i < pass->num_ops && pass->ops[i].kind == PLH_OP_SIGMOIDIZE)
        i++;
    return i == pass->num_ops;
}

// the shape k_deband_fast is written for
static bool deband_fast_applies(const plh_pass *pass)
{
    const plh_sampler_args &s = pass->s;
    const char *env = getenv("PL_HIP_DEBAND_FAST");
    if (env && env[0] == '0')
        return false;
    const bool native = pass->width == s.src.w && pass->height == s.src.h &&
        s.pos[0][0] == 0.0f && s.pos[0][1] == 0.0f && s.pos[3][0] == 1.0f && s.pos[3][1] == 1.0f &&
        s.pos[1][0] == 1.0f && s.pos[1][1] == 0.0f && s.pos[2][0] == 0.0f && s.pos[2][1] == 1.0f;
    return native && s.src.fmt == PLH_FMT_RGBA16 && pass->dst.fmt == PLH_FMT_RGBA16F &&
           s.address_mode == PLH_ADDRESS_CLAMP && (s.comp_mask & 7u) == 7u && !pass->transpose &&
           pass->base_x == 0 && pass->base_y == 0 && pass->dir_x == 1 && pass->dir_y == 1 &&
           pass->dst.w >= pass->width && pass->dst.h >= pass->height &&
           (size_t) s.src.pitch * s.src.h < (1ull << 32) &&
           !pass->num_pre_ops && deband_fast_ops(pass);
}

int plh_launch_deband(hipStream_t stream, const plh_pass *pass)
{
    if (deband_fast_applies(pass)) {
        const char *lds = getenv("PL_HIP_DEBAND_LDS");
        const plh_sampler_args &s = pass->s;
        if (!(lds && lds[0] == '0') && s.db_lds && s.iterations >= 1 && s.db_radius * (float) s.iterations <= 16.0f) {
            int cus = 256;
            (void) plh_stream_device((plh_stream) stream, &cus);
            const int tiles_x = (pass->width + DBL_TW - 1) / DBL_TW;
            // (8 bands of tile columns, padded to the widest: k_deband_lds)
#define DBL_LAUNCH(TH) do { \
                const size_t shmem = (size_t) DBL_WW * (TH + 2 * DBL_HALO) * 8; \
                static uint64_t lds_done; \
                const int e = plh_kernel_needs_lds((const void *) k_deband_lds<TH>, (plh_stream) stream, shmem, &lds_done); \
                if (e) \
                    return e; \
                const int tiles_y = (pass->height + TH - 1) / TH; \
                const dim3 grid(8 * ((tiles_x + 7) / 8) * tiles_y); \
                PLH_LAUNCH_LAST(k_deband_lds<TH>, grid, dim3(DBL_NT), shmem, stream, *pass); \
            } while (0)
            // fewer than two rounds of 32-row tiles (three workgroups per CU): 16-row tiles
            if (tiles_x * ((pass->height + 31) / 32) < 2 * 3 * cus)
                DBL_LAUNCH(16);
            else
                DBL_LAUNCH(32);
#undef DBL_LAUNCH
            const hipError_t err = hipGetLastError();
            return err == hipSuccess ? 0 : -(int) err;
        }
        const int nbx = (pass->width + 2 * DBF_BW - 1) / (2 * DBF_BW);
        const int nby = (pass->height + DBF_BH - 1) / DBF_BH;
        const dim3 grid(nbx, nby);
        PLH_LAUNCH_LAST(k_deband_fast, grid, dim3(DBF_BW, DBF_BH), 0, stream, *pass);
        const hipError_t err = hipGetLastError();
        return err == hipSuccess ? 0 : -(int) err;
    }
    const dim3 block(DEBAND_BW, DEBAND_BH);
    const dim3 grid((pass->width + DEBAND_BW - 1) / DEBAND_BW,
                    (pass->height + DEBAND_BH - 1) / DEBAND_BH);
    if (plh_ops_lite(pass, 0, pass->num_ops))
        PLH_LAUNCH_LAST(k_deband<true>, grid, block, 0, stream, *pass);
    else
        PLH_LAUNCH_LAST(k_deband<false>, grid, block, 0, stream, *pass);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
