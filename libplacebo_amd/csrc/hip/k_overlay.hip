/*
 * libplacebo-hip -- overlays (subtitles, on-screen display) and blended stores.
 *
 * The reference draws every overlay part as two triangles through the rasteriser, with the
 * fixed-function blend unit combining the fragment with what the target holds
 * (draw_overlays, src/renderer.c:811-1020), and it hands pl_dispatch_params.blend_params of an
 * ordinary pass to the same unit (src/dispatch.c:1199, gpu.h pl_blend_params). There is neither
 * here. What the two have in common is a read-modify-write of target pixels IN DRAWING ORDER:
 * overlapping parts (kerned glyphs, a box under its text) blend one after the other, each seeing
 * the target as the previous one left it -- rounded to the target's format, because that is
 * what the blend unit reads back.
 *
 * So the work is cut by TARGET pixels, not by parts: the host bins the parts into 16 x 16 target
 * tiles (dispatch.c: plh_dispatch_overlay), a workgroup owns one non-empty tile, a lane owns one
 * pixel and walks the tile's parts in order with the pixel in registers: coverage test (pixel
 * centre inside the part, top-left rule), texture fetch, the recorded colour ops, blend, rounding
 * through the target format -- one load and one store of the target per pixel however many parts
 * cover it. Subtitles touch a few thousand tiles; the kernel is bound by the latency of its
 * dependent loads (tile list -> part -> texel), not by bandwidth, and nothing here is tuned.
 */
#include "colorops.hiph"
#include "backend.h"
#include "samplers.hiph"

// what a store to and a fetch from a texture of this format make of a value
DEV float overlay_requant(int fmt, float x)
{
    if (fmt <= PLH_FMT_RGBA8)
        return plh_un8(plh_unorm(x, 255.0f));
    if (fmt <= PLH_FMT_RGBA16)
        return plh_un16(plh_unorm16x2(x, 0.0f) & 0xffffu);
    if (fmt <= PLH_FMT_RGBA16F)
        return plh_round_f16(x);
    return x;
}

DEV float overlay_factor(int f, float src_alpha)
{
    switch (f) {
    case PLH_BLEND_ONE:                 return 1.0f;
    case PLH_BLEND_SRC_ALPHA:           return src_alpha;
    case PLH_BLEND_ONE_MINUS_SRC_ALPHA: return 1.0f - src_alpha;
    default:                            return 0.0f;
    }
}

__global__ __launch_bounds__(PLH_OVERLAY_TILE * PLH_OVERLAY_TILE)
void k_overlay(const plh_pass p_, const plh_overlay_args o_)
{
    // both arguments through the kernarg segment (devmath.hiph: plh_kernarg_pass): the op
    // interpreter needs its ops in scalar registers
    static_assert(sizeof(plh_pass) % alignof(plh_overlay_args) == 0, "kernarg layout");
    const plh_pass &p = plh_kernarg_pass();
    const plh_overlay_args &o = *(const plh_overlay_args *)
        ((const char *) __builtin_amdgcn_kernarg_segment_ptr() + sizeof(plh_pass));
    // (loads through pointers found in there are flat loads, which the compiler takes for
    // divergent: say that they are not)
    const uint32_t *tile = o.tiles + 3 * blockIdx.x;
    const uint32_t txy = __builtin_amdgcn_readfirstlane(tile[0]);
    const uint32_t first = __builtin_amdgcn_readfirstlane(tile[1]);
    const uint32_t count = __builtin_amdgcn_readfirstlane(tile[2]);
    const int x = (int) (txy & 0xffffu) * PLH_OVERLAY_TILE + (int) (threadIdx.x % PLH_OVERLAY_TILE);
    const int y = (int) (txy >> 16) * PLH_OVERLAY_TILE + (int) (threadIdx.x / PLH_OVERLAY_TILE);
    if (x >= p.dst.w || y >= p.dst.h)
        return;
    const float px = (float) x + 0.5f, py = (float) y + 0.5f;
    const bool fixed_point = p.dst.fmt <= PLH_FMT_RGBA16;

    float4_t d = plh_fetch(p.dst, x, y);
    bool touched = false;
    for (uint32_t i = 0; i < count; i++) {
        const plh_overlay_part &q = o.parts[__builtin_amdgcn_readfirstlane(o.order[first + i])];
        if (!(px >= q.x0 && px < q.x1 && py >= q.y0 && py < q.y1))
            continue;

        float4_t c;
        float coverage = 1.0f;
        if (o.mode == PLH_OVERLAY_TEXEL) {
            c = plh_fetch(p.s.src, x - (int) q.x0, y - (int) q.y0);
        } else {
            const float u = q.u0 + ((px - q.ox) * q.ux + (py - q.oy) * q.uy);
            const float v = q.v0 + ((px - q.ox) * q.vx + (py - q.oy) * q.vy);
            const float4_t t = o.linear ? tex_linear(p.s.src, PLH_ADDRESS_CLAMP, u, v)
                                        : tex_nearest(p.s.src, PLH_ADDRESS_CLAMP, u, v);
            if (o.mode == PLH_OVERLAY_MONOCHROME) {
                c = { q.color[0], q.color[1], q.color[2], q.color[3] };
                coverage = t.x;
            } else {
                c = t;
            }
        }

        // the recorded ops: [0, num_pre_ops) decode / map / encode the overlay's colour, the
        // glyph coverage goes in between (:987-991), the rest is the plane's swizzle
        const frag_t fc = { px, py, 0.0f, 0, 0.0f, 0.0f };
#pragma unroll 1
        for (int part = 0; part < 2; part++) {
            apply_ops<false, false>(c, p.ops, part ? p.num_pre_ops : 0,
                                    part ? p.num_ops : p.num_pre_ops, fc);
            if (part == 0 && o.mode == PLH_OVERLAY_MONOCHROME) {
                if (o.premultiplied) {
                    c.x *= coverage; c.y *= coverage; c.z *= coverage;
                }
                c.w *= coverage;
            }
        }

        if (fixed_point) {
            // a fixed-point target: the blend unit clamps the fragment first
            c.x = plh_clamp(c.x, 0.0f, 1.0f); c.y = plh_clamp(c.y, 0.0f, 1.0f);
            c.z = plh_clamp(c.z, 0.0f, 1.0f); c.w = plh_clamp(c.w, 0.0f, 1.0f);
        }
        if (o.blend) {
            const float fs = overlay_factor(o.src_rgb, c.w), fd = overlay_factor(o.dst_rgb, c.w);
            const float as = overlay_factor(o.src_alpha, c.w), ad = overlay_factor(o.dst_alpha, c.w);
            c.x = c.x * fs + d.x * fd;
            c.y = c.y * fs + d.y * fd;
            c.z = c.z * fs + d.z * fd;
            c.w = c.w * as + d.w * ad;
        }
        d.x = overlay_requant(p.dst.fmt, c.x);
        d.y = overlay_requant(p.dst.fmt, c.y);
        d.z = overlay_requant(p.dst.fmt, c.z);
        d.w = overlay_requant(p.dst.fmt, c.w);
        touched = true;
    }
    if (touched)
        plh_store(p.dst, x, y, d);
}

extern "C" int plh_launch_overlay(plh_stream stream_, const struct plh_pass *pass,
                                  const struct plh_overlay_args *args)
{
    if (args->num_tiles <= 0)
        return 0;
    hipLaunchKernelGGL(k_overlay, dim3(args->num_tiles), dim3(PLH_OVERLAY_TILE * PLH_OVERLAY_TILE),
                       0, (hipStream_t) stream_, *pass, *args);
    const hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : -(int) err;
}
