/*
 * libplacebo-hip — generic fused pass kernel (K1, K5, K7-K9, K13, K15, K16).
 *
 * One launch = one reference pass for every sampler that needs no workgroup
 * cooperation: direct/nearest/bilinear (sampling.c:277-316), the fast
 * bicubic/hermite/gaussian/oversample samplers (:318-471), or no sampler at
 * all. The sampled colour then runs through the recorded colour-op chain and
 * is stored with the target format's conversion.
 *
 * Launch shape: 64x4 threads, one output pixel per lane; consecutive lanes
 * write consecutive texels (512 B per wave row at rgba16). Memory-bound: every
 * source texel and every target texel crosses HBM once (bilinear re-reads hit
 * L1/L2).
 */
#include "colorops.hiph"
#include "samplers.hiph"

#define PASS_BW 64
#define PASS_BH 4

DEV float4_t run_sampler(const plh_sampler_args &s, float px, float py)
{
    float4_t c = {0.0f, 0.0f, 0.0f, 1.0f};
    switch (s.type) {
    case PLH_SAMPLE_NEAREST:
        c = scale4(tex_nearest(s.src, s.address_mode, px, py), s.scale);
        break;
    case PLH_SAMPLE_BILINEAR:
        c = scale4(tex_linear(s.src, s.address_mode, px, py), s.scale);
        break;
    case PLH_SAMPLE_BICUBIC:
        c = sample_bicubic(s, px, py);
        break;
    case PLH_SAMPLE_HERMITE:
        c = sample_hermite(s, px, py);
        break;
    case PLH_SAMPLE_GAUSSIAN:
        c = sample_gaussian(s, px, py);
        break;
    case PLH_SAMPLE_OVERSAMPLE:
        c = sample_oversample(s, px, py);
        break;
    }
    return c;
}

// Guarded store of translate_compute_shader (dispatch.c:1126-1142)
DEV void pass_store(const plh_pass &p, int idx, int idy, const float4_t &c)
{
    const float fx = p.out_scale[0] * (float) idx, fy = p.out_scale[1] * (float) idy;
    if (!(fx < 1.0f && fy < 1.0f))
        return;
    const int ox = p.base_x + p.dir_x * (p.transpose ? idy : idx);
    const int oy = p.base_y + p.dir_y * (p.transpose ? idx : idy);
    if (ox < 0 || oy < 0 || ox >= p.dst.w || oy >= p.dst.h)
        return; // imageStore outside the image is a no-op
    plh_store(p.dst, ox, oy, c);
}

__global__ __launch_bounds__(PASS_BW * PASS_BH)
void k_pass_generic(const plh_pass p_)
{
    const plh_pass &p = plh_kernarg_pass();
    const int idx = blockIdx.x * PASS_BW + threadIdx.x;
    const int idy = blockIdx.y * PASS_BH + threadIdx.y;
    // whole groups are launched; lanes beyond the padded rect still run the
    // maths in the reference and are dropped by the store guard
    const float mx = p.out_scale[0] * ((float) idx + 0.5f);
    const float my = p.out_scale[1] * ((float) idy + 0.5f);

    float4_t c = {0.0f, 0.0f, 0.0f, 1.0f};
    if (p.s.type != PLH_SAMPLE_NONE) {
        const float px = plh_attr(p.s.pos, 0, mx, my);
        const float py = plh_attr(p.s.pos, 1, mx, my);
        c = run_sampler(p.s, px, py);
    }

    const frag_t fc = { (float) (idx + p.frag_x0) + 0.5f, (float) (idy + p.frag_y0) + 0.5f };
    apply_ops(c, p.ops, 0, p.num_ops, fc);
    pass_store(p, idx, idy, c);
}

/* ------------------------------------------------------------------------ */

int plh_launch_polar(hipStream_t stream, const plh_pass *pass);
int plh_launch_ortho(hipStream_t stream, const plh_pass *pass);
int plh_launch_deband(hipStream_t stream, const plh_pass *pass);
int plh_launch_peak(hipStream_t stream, const plh_pass *pass);

extern "C" int plh_launch_pass(plh_stream stream_, const struct plh_pass *pass)
{
    hipStream_t stream = (hipStream_t) stream_;
    if (pass->width <= 0 || pass->height <= 0)
        return 0;

    switch (pass->s.type) {
    case PLH_SAMPLE_POLAR:
        return plh_launch_polar(stream, pass);
    case PLH_SAMPLE_ORTHO:
        return plh_launch_ortho(stream, pass);
    case PLH_SAMPLE_DEBAND:
        return plh_launch_deband(stream, pass);
    default:
        break;
    }

    // a peak-detection stage needs the 16x16 tiling + LDS state of k_peak.hip
    for (int i = 0; i < pass->num_ops; i++) {
        if (pass->ops[i].kind == PLH_OP_PEAK_DETECT)
            return plh_launch_peak(stream, pass);
    }

    const dim3 block(PASS_BW, PASS_BH);
    const dim3 grid((pass->width + PASS_BW - 1) / PASS_BW,
                    (pass->height + PASS_BH - 1) / PASS_BH);
    hipLaunchKernelGGL(k_pass_generic, grid, block, 0, stream, *pass);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
