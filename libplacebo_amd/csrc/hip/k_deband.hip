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

int plh_launch_deband(hipStream_t stream, const plh_pass *pass)
{
    const dim3 block(DEBAND_BW, DEBAND_BH);
    const dim3 grid((pass->width + DEBAND_BW - 1) / DEBAND_BW,
                    (pass->height + DEBAND_BH - 1) / DEBAND_BH);
    if (plh_ops_lite(pass, 0, pass->num_ops))
        hipLaunchKernelGGL(k_deband<true>, grid, block, 0, stream, *pass);
    else
        hipLaunchKernelGGL(k_deband<false>, grid, block, 0, stream, *pass);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
