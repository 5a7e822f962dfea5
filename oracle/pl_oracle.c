/*
 * TEST INFRASTRUCTURE ONLY — CPU oracle for the pl_render_image hot path.
 *
 * Plain scalar C restatement of the reference's algorithm (haasn/libplacebo
 * v7.365.0), one function per GPU stage, each citing the reference file:line it
 * follows. Nothing in the product (libplacebo_amd/, include/) includes, links
 * or calls this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load the resulting oracle/libploracle.so.
 *
 * Pinning status:
 *   - Tier-0 restatements (filter sampling/LUT generation, PQ, ...) are checked
 *     against the reference's own known-answer tests (src/tests/filters.c:16-75,
 *     tone_mapping.c:14-23) and bit-for-bit against the real reference CPU code
 *     built by oracle/build_ref.sh (oracle/_ref/libplref.so) in
 *     tests/test_oracle_pinning.py.
 *   - The per-pixel GPU stages (sampling, polar EWA, dither, colour ops) have NO
 *     full-frame golden data in the reference (src/tests/gpu_tests.c:1216
 *     "TODO: embed a reference texture") and the reference's Vulkan execution
 *     cannot run here: for those stages parity is "unpinned" beyond the small
 *     gpu_tests.c vectors restated in tests/; see DESIGN.md §Oracle.
 *   - Overlay rasterisation and blending (orc_overlay_fragments, orc_blend): parity UNPINNED
 *     -- the reference's tests never draw an overlay on a backend that rasterises; the
 *     restatement is held to the graphics APIs' rules by hand-worked cases
 *     (tests/test_oracle_overlay.py).
 *   - Deinterlacing (orc_deinterlace): parity UNPINNED likewise -- upstream only dispatches and
 *     times the shader (src/tests/gpu_tests.c:875, bench.c:314-366); pinned by the algorithms'
 *     own promises on hand-worked cases (tests/test_oracle_deinterlace.py). The same holds for
 *     orc_distort (tests/test_oracle_distort.py).
 *
 * Float semantics. GLSL leaves contraction, mix() and texture filtering
 * precision open; this file fixes them (same choices as csrc/hip/devmath.hiph,
 * which is what makes HIP output bit-comparable for the transcendental-free
 * stages):
 *   mix(x,y,a) = fma(y, a, fma(-x, a, x));  fract(x) = x - floor(x)
 *   length(v)  = sqrtf(v.x*v.x + v.y*v.y);  accumulate: a = fma(w, c, a)
 *   hardware bilinear / linear LUT = exact fp32 lerp (lut.c:700-715 form)
 *   x / const  = x * (1.0f / const) only where noted
 * Compiled with -ffp-contract=off so nothing else fuses.
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline float mixf(float x, float y, float a) { return fmaf(y, a, fmaf(-x, a, x)); }
static inline float fractf(float x) { return x - floorf(x); }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* ======================================================================== */
/* texel formats                                                             */

enum { FMT_R8 = 1, FMT_RG8, FMT_RGBA8, FMT_R16, FMT_RG16, FMT_RGBA16,
       FMT_R16F, FMT_RG16F, FMT_RGBA16F, FMT_R32F, FMT_RG32F, FMT_RGBA32F };

static int fmt_comps(int fmt) { return ((fmt - 1) % 3) == 0 ? 1 : ((fmt - 1) % 3) == 1 ? 2 : 4; }
static int fmt_class(int fmt) { return (fmt - 1) / 3; } // 0 u8, 1 u16, 2 f16, 3 f32

// IEEE binary16 <-> binary32 (round-to-nearest-even), bit-level
static float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t) (h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
    if (exp == 0) {
        if (!man) {
            bits = sign;
        } else { // subnormal
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400));
            bits = sign | ((uint32_t) (127 - 15 - e) << 23) | ((man & 0x3ff) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static uint16_t float_to_half(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (x >> 16) & 0x8000;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) // inf / nan
        return sign | 0x7c00 | (x > 0x7f800000u ? 0x200 | ((x >> 13) & 0x3ff) : 0);
    if (x >= 0x477ff000u) // rounds to >= 65520 -> inf
        return sign | 0x7c00;
    if (x < 0x33000001u) // rounds to zero (<= 2^-25)
        return sign;
    int exp = (int) (x >> 23) - 127;
    uint32_t man = (x & 0x7fffffu) | 0x800000u;
    int shift;
    uint16_t base;
    if (exp < -14) { // subnormal half
        shift = 13 + (-14 - exp);
        base = 0;
    } else {
        shift = 13;
        base = (uint16_t) ((exp + 15) << 10);
        man &= 0x7fffffu;
    }
    uint32_t q = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1)))
        q++;
    return sign | (uint16_t) (base + q); // carry into exponent is correct by construction
}

ORC_API float orc_round_f16(float f) { return half_to_float(float_to_half(f)); }

// Texture fetch of every texel -> float RGBA; missing components (0,0,0,1)
ORC_API void orc_tex_decode(const void *src, int fmt, int w, int h, size_t pitch, float *out)
{
    const int nc = fmt_comps(fmt), cls = fmt_class(fmt);
    for (int y = 0; y < h; y++) {
        const uint8_t *row = (const uint8_t *) src + (size_t) y * pitch;
        for (int x = 0; x < w; x++) {
            float *o = out + ((size_t) y * w + x) * 4;
            o[0] = o[1] = o[2] = 0.0f; o[3] = 1.0f;
            for (int c = 0; c < nc; c++) {
                const int i = x * nc + c;
                switch (cls) {
                case 0: o[c] = row[i] / 255.0f; break;
                case 1: o[c] = ((const uint16_t *) row)[i] / 65535.0f; break;
                case 2: o[c] = half_to_float(((const uint16_t *) row)[i]); break;
                case 3: o[c] = ((const float *) row)[i]; break;
                }
            }
        }
    }
}

// Image store with the target format's conversion (unorm: clamp, scale, RNE)
// unorm encode: the exact product, rounded half-even once -- what the GPU's colour export
// (v_cvt_pknorm_u16_f32) computes; the double product of a float and 2^n - 1 is exact
static double unorm_rne(float v, double maxv)
{
    return rint((double) fminf(fmaxf(v, 0.0f), 1.0f) * maxv);
}

ORC_API void orc_tex_encode(const float *img, int w, int h, int fmt, void *dst, size_t pitch)
{
    const int nc = fmt_comps(fmt), cls = fmt_class(fmt);
    for (int y = 0; y < h; y++) {
        uint8_t *row = (uint8_t *) dst + (size_t) y * pitch;
        for (int x = 0; x < w; x++) {
            const float *p = img + ((size_t) y * w + x) * 4;
            for (int c = 0; c < nc; c++) {
                const int i = x * nc + c;
                const float v = p[c];
                switch (cls) {
                case 0: row[i] = (uint8_t) unorm_rne(v, 255.0); break;
                case 1: ((uint16_t *) row)[i] = (uint16_t) unorm_rne(v, 65535.0); break;
                case 2: ((uint16_t *) row)[i] = float_to_half(v); break;
                case 3: ((float *) row)[i] = v; break;
                }
            }
        }
    }
}

/* ======================================================================== */
/* texture unit + vertex attribute emulation                                 */

enum { ADDR_CLAMP = 0, ADDR_REPEAT, ADDR_MIRROR };

struct orc_src {
    const float *tex;   // decoded RGBA float texels
    int w, h;
    float rect[4];      // x0 y0 x1 y1 in texels (sh_bind `rect`)
    int address_mode;
    // Optional: `tex` holds only the texels [rx, rx+rw) x [ry, ry+rh) of the w x h texture
    // (rw == 0: all of it). Lets a full-size pass be checked on output windows without
    // materialising the whole intermediate image; touching a texel outside aborts.
    int rx, ry, rw, rh;
};

static int wrap(int i, int n, int mode)
{
    if (mode == ADDR_CLAMP)
        return i < 0 ? 0 : i > n - 1 ? n - 1 : i;
    if (mode == ADDR_REPEAT) {
        int m = i % n;
        return m < 0 ? m + n : m;
    }
    int m = i % (2 * n);
    if (m < 0)
        m += 2 * n;
    return m < n ? m : 2 * n - 1 - m;
}

static const float *texel(const struct orc_src *s, int x, int y)
{
    const int tx = wrap(x, s->w, s->address_mode), ty = wrap(y, s->h, s->address_mode);
    if (!s->rw)
        return s->tex + ((size_t) ty * s->w + tx) * 4;
    if (tx < s->rx || ty < s->ry || tx >= s->rx + s->rw || ty >= s->ry + s->rh) {
        fprintf(stderr, "oracle: texel (%d, %d) outside the provided region [%d,%d)x[%d,%d)\n",
                tx, ty, s->rx, s->rx + s->rw, s->ry, s->ry + s->rh);
        abort();
    }
    return s->tex + ((size_t) (ty - s->ry) * s->rw + (tx - s->rx)) * 4;
}

// Output window for the samplers below: with a window set, only the output pixels
// [x0, x0+w) x [y0, y0+h) of the out_w x out_h pass are evaluated (same arithmetic: the
// interpolated attribute still refers to the full pass) and `out` is w x h.
static struct { int x0, y0, w, h; } g_win;
ORC_API void orc_set_window(int x0, int y0, int w, int h)
{
    g_win.x0 = x0; g_win.y0 = y0; g_win.w = w; g_win.h = h;
}
#define WIN_SETUP(out_w, out_h)                                                     \
    const int wx0 = g_win.w ? g_win.x0 : 0, wy0 = g_win.w ? g_win.y0 : 0;           \
    const int wx1 = g_win.w ? g_win.x0 + g_win.w : (out_w);                         \
    const int wy1 = g_win.w ? g_win.y0 + g_win.h : (out_h);                         \
    const int wstride = wx1 - wx0
#define WIN_OUT(out, x, y) ((out) + ((size_t) ((y) - wy0) * wstride + ((x) - wx0)) * 4)


// tex_coord corners: rect / tex_size (sh_bind, src/shaders.c:541-561)
static void corners(const struct orc_src *s, float p[4][2], float pt[2])
{
    const float sx = 1.0 / s->w, sy = 1.0 / s->h;
    const float x0 = sx * s->rect[0], y0 = sy * s->rect[1];
    const float x1 = sx * s->rect[2], y1 = sy * s->rect[3];
    p[0][0] = x0; p[0][1] = y0;
    p[1][0] = x1; p[1][1] = y0;
    p[2][0] = x0; p[2][1] = y1;
    p[3][0] = x1; p[3][1] = y1;
    pt[0] = sx; pt[1] = sy;
}

// Compute-shader emulation of the interpolated attribute (dispatch.c:1038-1062)
static float attr(const float p[4][2], int c, float fx, float fy)
{
    return mixf(mixf(p[0][c], p[1][c], fx), mixf(p[2][c], p[3][c], fx), fy);
}

static void tex_nearest(const struct orc_src *s, float px, float py, float out[4])
{
    const int ix = (int) floorf(px * (float) s->w), iy = (int) floorf(py * (float) s->h);
    memcpy(out, texel(s, ix, iy), 16);
}

static void tex_linear(const struct orc_src *s, float px, float py, float out[4])
{
    const float u = px * (float) s->w - 0.5f, v = py * (float) s->h - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float ax = u - fu, ay = v - fv;
    const float *t00 = texel(s, (int) fu, (int) fv), *t10 = texel(s, (int) fu + 1, (int) fv);
    const float *t01 = texel(s, (int) fu, (int) fv + 1), *t11 = texel(s, (int) fu + 1, (int) fv + 1);
    for (int c = 0; c < 4; c++)
        out[c] = mixf(mixf(t00[c], t10[c], ax), mixf(t01[c], t11[c], ax), ay);
}

/* ======================================================================== */
/* K1 / K5: single-fetch samplers (src/shaders/sampling.c:277-471)            */

enum { S_NEAREST = 1, S_BILINEAR, S_BICUBIC, S_HERMITE, S_GAUSSIAN, S_OVERSAMPLE };

static float smoothstep01(float x)
{
    const float t = clampf(x, 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

static void fast4(const struct orc_src *s, const float pt[2], float px, float py,
                  const float g[4], const float h[4], const float off[2], float scale,
                  float out[4])
{
    // p = pos.xyxy + pt.xyxy * (h + off.xyxy); 4 bilinear taps, 3 mixes
    const float p0 = px + pt[0] * (h[0] + off[0]), p1 = py + pt[1] * (h[1] + off[1]);
    const float p2 = px + pt[0] * (h[2] + off[0]), p3 = py + pt[1] * (h[3] + off[1]);
    float c00[4], c01[4], c10[4], c11[4];
    tex_linear(s, p0, p1, c00);
    tex_linear(s, p0, p3, c01);
    tex_linear(s, p2, p1, c10);
    tex_linear(s, p2, p3, c11);
    for (int c = 0; c < 4; c++) {
        const float c0 = mixf(c01[c], c00[c], g[1]);
        const float c1 = mixf(c11[c], c10[c], g[1]);
        out[c] = scale * mixf(c1, c0, g[0]);
    }
}

ORC_API void orc_sample_simple(const struct orc_src *s, int type, float scale,
                               float rx, float ry, float threshold,
                               int out_w, int out_h, float *out)
{
    float p[4][2], pt[2];
    corners(s, p, pt);
    const float osx = 1.0 / out_w, osy = 1.0 / out_h;
    const float size[2] = { (float) s->w, (float) s->h };
    const float ratio[2] = { rx, ry };

    // Identity fetch (1:1 on the texel grid, e.g. the renderer's img_sh ->
    // pl_shader_sample_direct of an FBO/plane): hardware bilinear returns the
    // texel itself; an exact fp32 lerp would not (rounding noise in `pos`), so
    // both oracle and product treat it as a nearest fetch. Design decision,
    // see DESIGN.md "texture unit emulation".
    if (type == S_BILINEAR && fabsf(rx - 1.0f) < 1e-6f && fabsf(ry - 1.0f) < 1e-6f &&
        s->rect[0] == truncf(s->rect[0]) && s->rect[1] == truncf(s->rect[1]))
        type = S_NEAREST;

    WIN_SETUP(out_w, out_h);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int y = wy0; y < wy1; y++) {
        for (int x = wx0; x < wx1; x++) {
            const float fx = osx * ((float) x + 0.5f), fy = osy * ((float) y + 0.5f);
            float pos[2] = { attr(p, 0, fx, fy), attr(p, 1, fx, fy) };
            float *o = WIN_OUT(out, x, y);
            float c[4];
            switch (type) {
            case S_NEAREST: // sampling.c:290-302
                tex_nearest(s, pos[0], pos[1], c);
                for (int k = 0; k < 4; k++) o[k] = scale * c[k];
                break;
            case S_BILINEAR: // sampling.c:304-316
                tex_linear(s, pos[0], pos[1], c);
                for (int k = 0; k < 4; k++) o[k] = scale * c[k];
                break;
            case S_BICUBIC: { // sampling.c:335-361
                float g[4], h[4];
                const float off[2] = {0, 0};
                for (int k = 0; k < 2; k++) {
                    const float fr = fractf(pos[k] * size[k] + 0.5f);
                    const float fr2 = fr * fr, inv = 1.0f - fr, inv2 = inv * inv;
                    const float w0 = 1.0f / 6.0f * inv2 * inv;
                    const float w1 = 2.0f / 3.0f - 0.5f * fr2 * (2.0f - fr);
                    const float w2 = 2.0f / 3.0f - 0.5f * inv2 * (2.0f - inv);
                    const float w3 = 1.0f / 6.0f * fr2 * fr;
                    g[k] = w0 + w1;
                    g[k + 2] = w2 + w3;
                    h[k] = w1 / g[k] + inv - 2.0f;
                    h[k + 2] = w3 / g[k + 2] + inv;
                }
                fast4(s, pt, pos[0], pos[1], g, h, off, scale, o);
                break;
            }
            case S_HERMITE: { // sampling.c:379-387
                for (int k = 0; k < 2; k++) {
                    const float fr = fractf(pos[k] * size[k] + 0.5f);
                    pos[k] += pt[k] * (smoothstep01(fr) - fr);
                }
                tex_linear(s, pos[0], pos[1], c);
                for (int k = 0; k < 4; k++) o[k] = scale * c[k];
                break;
            }
            case S_GAUSSIAN: { // sampling.c:405-431
                float g[4], h[4], off[2];
                for (int k = 0; k < 2; k++) {
                    const float of = -fractf(pos[k] * size[k] + 0.5f);
                    const float o2 = -2.0f * of * of;
                    const float w0 = expf(o2 + 4.0f * of - 2.0f);
                    const float w1 = expf(o2);
                    const float w2 = expf(o2 - 4.0f * of - 2.0f);
                    const float w3 = expf(o2 - 8.0f * of - 8.0f);
                    g[k] = w0 + w1;
                    g[k + 2] = w2 + w3;
                    h[k] = w1 / g[k] - 1.0f;
                    h[k + 2] = w3 / g[k + 2] + 1.0f;
                    g[k] /= g[k] + g[k + 2];
                    off[k] = of;
                }
                fast4(s, pt, pos[0], pos[1], g, h, off, scale, o);
                break;
            }
            case S_OVERSAMPLE: { // sampling.c:446-468
                const float thr = clampf(threshold, 0.0f, 0.5f);
                for (int k = 0; k < 2; k++) {
                    const float fc = fractf(pos[k] * size[k] - 0.5f);
                    float coeff = (fc - 0.5f) * ratio[k];
                    coeff = clampf(coeff + 0.5f, 0.0f, 1.0f);
                    if (thr > 0) {
                        coeff = coeff < thr ? 0.0f : coeff;
                        coeff = coeff > 1.0f - thr ? 1.0f : coeff;
                    }
                    pos[k] += (coeff - fc) * pt[k];
                }
                tex_linear(s, pos[0], pos[1], c);
                for (int k = 0; k < 4; k++) o[k] = scale * c[k];
                break;
            }
            }
        }
    }
}

/* ======================================================================== */
/* K2 / K3: polar EWA (src/shaders/sampling.c:503-558, 587-912)               */

// Linear LUT lookup, the reference's own ALU form (src/shaders/lut.c:700-715)
static float lut_lin(const float *lut, int n, float fpos)
{
    fpos = clampf(fpos, 0.0f, 1.0f) * (float) (n - 1);
    const float fbase = floorf(fpos), fceil = ceilf(fpos);
    return mixf(lut[(int) fbase], lut[(int) fceil], fpos - fbase);
}

struct polar_acc {
    float color[4], wsum;
    float ar[4][2], wwsum[4][2];
};

struct polar_cfg {
    const float *lut;       // 256 radial weights
    float radius, radius_zero, antiring, scale;
    unsigned mask;
    int use_ar;
};

// One tap: polar_sample(), sampling.c:503-558. Returns 0 if statically pruned.
static void polar_tap(const struct orc_src *s, const struct polar_cfg *cfg, int bx, int by,
                      float fcx, float fcy, int x, int y, struct polar_acc *a)
{
    const int yy = y > 0 ? y - 1 : y;
    const int xx = x > 0 ? x - 1 : x;
    const float dmin = sqrt(xx * xx + yy * yy);
    if (dmin >= cfg->radius)
        return; // generation-time pruning (:510-515)
    const int maybe_skippable = dmin >= cfg->radius - M_SQRT2;
    const int use_ar = cfg->use_ar && dmin < cfg->radius_zero;

    const float dx = (float) x - fcx, dy = (float) y - fcy;
    const float d = sqrtf(dx * dx + dy * dy);
    if (maybe_skippable && !(d < cfg->radius))
        return;
    // w = lut(d * 1.0 / radius): division by a constant, folded the way an
    // AMD Vulkan driver lowers a 2.5-ULP OpFDiv by a constant: d * (1/R)
    const float w = lut_lin(cfg->lut, 256, d * (1.0f / cfg->radius));
    a->wsum += w;
    const float *c = texel(s, bx + x, by + y);
    for (int k = 0; k < 4; k++) {
        if (cfg->mask & (1u << k))
            a->color[k] = fmaf(w, c[k], a->color[k]);
    }
    if (use_ar && d <= cfg->radius_zero) {
        for (int k = 0; k < 4; k++) {
            if (!(cfg->mask & (1u << k)))
                continue;
            float cc[2] = { cfg->scale * c[k], cfg->scale * c[k] };
            cc[0] = 1.0f - cc[0];
            for (int j = 0; j < 2; j++) {
                float ww = cc[j] + 0.10f;
                ww = ww * ww; ww = ww * ww; ww = ww * ww; ww = ww * ww; ww = ww * ww;
                ww = w * ww;
                a->ar[k][j] = fmaf(ww, cc[j], a->ar[k][j]);
                a->wwsum[k][j] += ww;
            }
        }
    }
}

ORC_API void orc_sample_polar(const struct orc_src *s, const float *lut, float radius,
                              float radius_zero, float antiring, int gather_order,
                              float scale, unsigned mask, int out_w, int out_h, float *out)
{
    float p[4][2], pt[2];
    corners(s, p, pt);
    const float osx = 1.0 / out_w, osy = 1.0 / out_h;
    const struct polar_cfg cfg = { lut, radius, radius_zero, antiring, scale, mask,
                                   antiring > 0 };
    const int bound = ceil(radius);

    WIN_SETUP(out_w, out_h);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int oy = wy0; oy < wy1; oy++) {
        for (int ox = wx0; ox < wx1; ox++) {
            const float fx = osx * ((float) ox + 0.5f), fy = osy * ((float) oy + 0.5f);
            const float px = attr(p, 0, fx, fy), py = attr(p, 1, fx, fy);
            // fcoord = fract(pos*size - 0.5); base = texel floor(pos*size - 0.5) (:639-640)
            const float tx = px * (float) s->w - 0.5f, ty = py * (float) s->h - 0.5f;
            const float flx = floorf(tx), fly = floorf(ty);
            const float fcx = tx - flx, fcy = ty - fly;
            const int bx = (int) flx, by = (int) fly;

            struct polar_acc a;
            memset(&a, 0, sizeof(a));

            if (!gather_order) {
                // compute-shader order (:776-783)
                for (int y = 1 - bound; y <= bound; y++) {
                    for (int x = 1 - bound; x <= bound; x++)
                        polar_tap(s, &cfg, bx, by, fcx, fcy, x, y, &a);
                }
            } else {
                // textureGather order (:798-893)
                uint64_t gathered_cur = 0, gathered_next = 0;
                const float radius2 = radius * radius;
                const int base = bound - 1;
                for (int y = 1 - bound; y <= bound; y++) {
                    for (int x = 1 - bound; x <= bound; x++) {
                        const uint64_t bit = 1llu << (base + x);
                        if (gathered_cur & bit)
                            continue;
                        const int xx = x * x, xx1 = (x + 1) * (x + 1);
                        const int yy = y * y, yy1 = (y + 1) * (y + 1);
                        int use_gather = (xx > xx1 ? xx : xx1) + (yy > yy1 ? yy : yy1) < radius2;
                        use_gather &= (x > y ? x : y) <= 31;
                        use_gather &= (x < y ? x : y) >= -32;
                        if (!use_gather) {
                            polar_tap(s, &cfg, bx, by, fcx, fcy, x, y, &a);
                            continue;
                        }
                        static const int xo[4] = {0, 1, 1, 0}, yo[4] = {1, 1, 0, 0};
                        for (int q = 0; q < 4; q++) {
                            if (x + xo[q] > bound || y + yo[q] > bound)
                                continue;
                            if (!yo[q] && (gathered_cur & (bit << xo[q])))
                                continue;
                            polar_tap(s, &cfg, bx, by, fcx, fcy, x + xo[q], y + yo[q], &a);
                        }
                        gathered_next |= bit | (bit << 1);
                        x++;
                    }
                    gathered_cur = gathered_next;
                    gathered_next = 0;
                }
            }

            // color = scale / wsum * color; AR; alpha (:896-908)
            float *o = WIN_OUT(out, ox, oy);
            const float norm = scale / a.wsum;
            for (int k = 0; k < 4; k++) {
                float v = norm * a.color[k];
                if (cfg.use_ar && (mask & (1u << k))) {
                    float lo = a.ar[k][0] / a.wwsum[k][0];
                    const float hi = a.ar[k][1] / a.wwsum[k][1];
                    lo = 1.0f - lo;
                    float w = fminf(fmaxf(v, lo), hi);
                    w = lo > hi ? (lo * 0.5f + hi * 0.5f) : w;
                    v = mixf(v, w, antiring);
                }
                o[k] = v;
            }
            if (!(mask & 8u))
                o[3] = 1.0f;
        }
    }
}

/* ======================================================================== */
/* colour ops on whole images                                                */

ORC_API void orc_op_scale(float *img, size_t npix, const float s[4])
{
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        for (int c = 0; c < 4; c++)
            img[i * 4 + c] *= s[c];
    }
}

ORC_API void orc_op_quant_f16(float *img, size_t npix)
{
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix * 4; i++)
        img[i] = orc_round_f16(img[i]);
}

ORC_API void orc_op_affine(float *img, size_t npix, const float m[9], const float c[3])
{
    // color.rgb = M * color.rgb + c (shaders/colorspace.c:308,569), row-wise
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *p = img + i * 4;
        const float r = p[0], g = p[1], b = p[2];
        p[0] = (m[0] * r + m[1] * g + m[2] * b) + c[0];
        p[1] = (m[3] * r + m[4] * g + m[5] * b) + c[1];
        p[2] = (m[6] * r + m[7] * g + m[8] * b) + c[2];
    }
}

// pl_shader_dither (src/shaders/dithering.c:109-274); gl_FragCoord = id + 0.5
// (rect-relative in compute passes, dispatch.c:1040). method: 0 = LUT (blue
// noise / bayer: integer index path), 1 = PL_DITHER_ORDERED_FIXED.
static void o_pcg3d(uint32_t s[3], float out[3])
{
    for (int k = 0; k < 3; k++)
        s[k] = 1664525u * s[k] + 1013904223u;
    s[0] += s[1] * s[2]; s[1] += s[2] * s[0]; s[2] += s[0] * s[1];
    for (int k = 0; k < 3; k++)
        s[k] ^= s[k] >> 16;
    s[0] += s[1] * s[2]; s[1] += s[2] * s[0]; s[2] += s[0] * s[1];
    const float k32 = 1.0f / (float) 0xFFFFFFFFu;    // float(0xFFFFFFFF) == 2^32
    for (int k = 0; k < 3; k++)
        out[k] = (float) s[k] * k32;
}

ORC_API void orc_dither(float *img, int w, int h, const float *matrix, int size, int method,
                        int depth, float gamma, int temporal, int frame_index)
{
    const float scale = (float) ((1llu << depth) - 1);
    float rot[4] = {1, 0, 0, 1};
    if (temporal) {
        const int phase = frame_index % 8;
        const float r = phase * (M_PI / 2);
        const float m = phase < 4 ? 1 : -1;
        // column-major mat2 {{cos r, -sin r}, {sin r * m, cos r * m}} (:185-196)
        rot[0] = cos(r); rot[1] = -sin(r); rot[2] = sin(r) * m; rot[3] = cos(r) * m;
    }
    // with an output window set (orc_set_window), `img` is that window of the pass
    const int fx0 = g_win.w ? g_win.x0 : 0, fy0 = g_win.w ? g_win.y0 : 0;
    #pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const float fcx = (float) (x + fx0) + 0.5f, fcy = (float) (y + fy0) + 0.5f;
            float px = fractf(fcx / (float) size), py = fractf(fcy / (float) size); // :183
            if (temporal && method != 2) {
                const float qx = (rot[0] * px + rot[2] * py) + 1.0f;
                const float qy = (rot[1] * px + rot[3] * py) + 1.0f;
                px = fractf(qx); py = fractf(qy);                             // :199
            }
            float bias;
            if (method == 2) {                                                 // :204-207
                uint32_t st[3] = { (uint32_t) fcx, (uint32_t) fcy, temporal ? (uint32_t) frame_index : 0u };
                float rnd[3];
                o_pcg3d(st, rnd);
                bias = rnd[0];
            } else if (method == 0) {
                bias = matrix[(int) (py * (float) size) * size + (int) (px * (float) size)]; // :230
            } else {
                uint32_t ux = (uint32_t) (px * 16.0f) % 16u, uy = (uint32_t) (py * 16.0f) % 16u;
                ux = ux ^ uy;                                                  // :212-224
                ux = (ux | ux << 2) & 0x33333333u; uy = (uy | uy << 2) & 0x33333333u;
                ux = (ux | ux << 1) & 0x55555555u; uy = (uy | uy << 1) & 0x55555555u;
                uint32_t b = ux + (uy << 1);
                b = (b * 0x0802u & 0x22110u) | (b * 0x8020u & 0x88440u);
                b = 0x10101u * b;
                b = (b >> 16) & 0xFFu;
                bias = (float) b * (1.0f / 256.0f);
            }
            float *p = img + ((size_t) y * w + x) * 4;
            for (int c = 0; c < 4; c++) {
                if (gamma != 1.0f && depth <= 4) {                             // :241-266
                    const float v = p[c], lin = powf(v, gamma);
                    float low, high, off;
                    if (depth == 1) {
                        low = 0; high = 1; off = lin;
                    } else {
                        low = floorf(v * scale) / scale;
                        high = ceilf(v * scale) / scale;
                        const float ll = powf(low, gamma), hl = powf(high, gamma);
                        off = (lin - ll) / fmaxf(hl - ll, 1e-6f);
                    }
                    p[c] = off > bias ? high : low;
                } else {
                    p[c] = floorf(scale * p[c] + bias) * (1.0f / scale);       // :269-270
                }
            }
        }
    }
}

/* ======================================================================== */
/* Tier-0 restatement: filter kernels (src/filters.c)                        */

enum { K_BOX = 0, K_TRIANGLE, K_HANN, K_GAUSSIAN, K_SINC, K_JINC, K_CUBIC,
       K_SPLINE16, K_SPLINE36, K_SPLINE64, K_NONE = -1 };

struct orc_filter {
    int kernel, window;         // K_* (window may be K_NONE)
    double kparams[2];          // cubic (b, c) / gaussian
    float kradius, wradius;     // natural radii of kernel / window function
    int resizable;
    float radius;               // config.radius (0 = natural)
    float clamp, blur, taper;
};

static double kweight(int k, const double *prm, double radius, double x)
{
    switch (k) {
    case K_BOX:      return 1.0;                                        // filters.c:254
    case K_TRIANGLE: return 1.0 - x / radius;                           // :273
    case K_HANN:     return 0.5 + 0.5 * cos(M_PI * x);                  // :296
    case K_GAUSSIAN: return exp(-2.0 * x * x / prm[0]);                 // :392
    case K_SINC:     if (x < 1e-8) return 1.0; x *= M_PI; return sin(x) / x;       // :427
    case K_JINC:     if (x < 1e-8) return 1.0; x *= M_PI; return 2.0 * j1(x) / x;  // :442
    case K_CUBIC: {                                                      // :472
        const double b = prm[0], c = prm[1];
        const double p0 = 6.0 - 2.0 * b, p2 = -18.0 + 12.0 * b + 6.0 * c,
                     p3 = 12.0 - 9.0 * b - 6.0 * c, q0 = 8.0 * b + 24.0 * c,
                     q1 = -12.0 * b - 48.0 * c, q2 = 6.0 * b + 30.0 * c, q3 = -b - 6.0 * c;
        if (x < 1.0)
            return (p0 + x * x * (p2 + x * p3)) / p0;
        return (q0 + x * (q1 + x * (q2 + x * q3))) / p0;
    }
    case K_SPLINE16:                                                     // :553
        if (x < 1.0) return ((x - 9.0/5.0 ) * x - 1.0/5.0 ) * x + 1.0;
        return ((-1.0/3.0 * (x-1) + 4.0/5.0) * (x-1) - 7.0/15.0 ) * (x-1);
    case K_SPLINE36:                                                     // :568
        if (x < 1.0) return ((13.0/11.0 * x - 453.0/209.0) * x - 3.0/209.0) * x + 1.0;
        if (x < 2.0) return ((-6.0/11.0 * (x-1) + 270.0/209.0) * (x-1) - 156.0/ 209.0) * (x-1);
        return ((1.0/11.0 * (x-2) - 45.0/209.0) * (x-2) +  26.0/209.0) * (x-2);
    case K_SPLINE64:                                                     // :585
        if (x < 1.0) return ((49.0/41.0 * x - 6387.0/2911.0) * x - 3.0/2911.0) * x + 1.0;
        if (x < 2.0) return ((-24.0/41.0 * (x-1) + 4032.0/2911.0) * (x-1) - 2328.0/2911.0) * (x-1);
        if (x < 3.0) return ((6.0/41.0 * (x-2) - 1008.0/2911.0) * (x-2) + 582.0/2911.0) * (x-2);
        return ((-1.0/41.0 * (x-3) + 168.0/2911.0) * (x-3) - 97.0/2911.0) * (x-3);
    }
    return 0.0;
}

static float radius_bound(const struct orc_filter *f)                   // src/filters.h:22-26
{
    const float r = f->radius && f->resizable ? f->radius : f->kradius;
    return f->blur > 0.0 ? r * f->blur : r;
}

// pl_filter_sample, filters.c:82-124
ORC_API double orc_filter_sample(const struct orc_filter *f, double x)
{
    const float radius = radius_bound(f);
    x = fabs(x);
    if (x > radius)
        return 0.0;
    double kx = x <= f->taper ? 0.0 : (x - f->taper) / (1.0 - f->taper / radius);
    if (f->blur > 0.0)
        kx /= f->blur;
    double k = kweight(f->kernel, f->kparams, radius, kx);
    if (f->window != K_NONE) {
        const double wx = x / radius * f->wradius;
        const double none[2] = {0, 0};
        k *= kweight(f->window, none, f->wradius, wx);
    }
    return k < 0 ? (1 - f->clamp) * k : k;
}

// filter_cutoffs (filters.c:126-151) + polar branch of pl_filter_generate (:215-222)
ORC_API void orc_filter_generate_polar(const struct orc_filter *f, float cutoff, int n,
                                       float *weights, float *out_radius, float *out_radius_zero)
{
    const float bound = radius_bound(f);
    float prev = 0.0, fprev = orc_filter_sample(f, prev);
    int found = 0;
    float radius = bound, radius_zero = bound;
    const float step = 1e-2f;
    for (float x = 0.0; x < bound + step; x += step) {
        const float fx = orc_filter_sample(f, x);
        if ((fprev > cutoff && fx <= cutoff) || (fprev < -cutoff && fx >= -cutoff)) {
            float root = x - fx * (x - prev) / (fx - fprev);
            root = fminf(root, bound);
            radius = root;
            if (!found)
                radius_zero = root;
            found = 1;
        }
        prev = x;
        fprev = fx;
    }
    if (!found)
        radius_zero = radius = bound;
    for (int i = 0; i < n; i++) {
        const double x = radius * i / (n - 1);
        weights[i] = orc_filter_sample(f, x);
    }
    *out_radius = radius;
    *out_radius_zero = radius_zero;
}

// compute_row + separable branch (filters.c:155-177, 224-241); returns row_size
ORC_API int orc_filter_generate_ortho(const struct orc_filter *f, float cutoff, int n,
                                      int stride_align, float *weights, int max_floats,
                                      float *out_radius, float *out_radius_zero)
{
    float dummy[2];
    orc_filter_generate_polar(f, cutoff, 2, dummy, out_radius, out_radius_zero);
    const int row_size = ceilf(*out_radius) * 2;
    const int stride = (row_size + stride_align - 1) / stride_align * stride_align;
    if (n * stride > max_floats)
        return -stride;
    memset(weights, 0, sizeof(float) * n * stride);
    for (int i = 0; i < n; i++) {
        float *row = weights + (size_t) i * stride;
        const double offset = i / (double) (n - 1);
        const double center = (row_size / 2 - 1) + offset;
        double wsum = 0.0;
        for (int k = 0; k < row_size; k++) {
            const double w = orc_filter_sample(f, k - center);
            row[k] = w;
            wsum += w;
        }
        for (int k = 0; k < row_size; k++)
            row[k] /= wsum;
    }
    return row_size;
}

/* ------------------------------------------------------------------------ */
/* CPU baseline driver: BASELINE.json configs[0] — EWA-Lanczos resample of a  */
/* single-channel float image, (a) direct pl_filter_sample per tap,          */
/* (b) 256-entry LUT + lerp (what the GPU path does).                         */

ORC_API double orc_ewa_resample_r32f(const struct orc_filter *f, const float *src, int sw, int sh,
                                     int dw, int dh, int use_lut, float *dst)
{
    float lut[256], radius, radius_zero;
    orc_filter_generate_polar(f, 1e-3f, 256, lut, &radius, &radius_zero);
    const int bound = ceil(radius);
    double taps = 0;
    for (int oy = 0; oy < dh; oy++) {
        for (int ox = 0; ox < dw; ox++) {
            const float px = ((float) ox + 0.5f) / dw, py = ((float) oy + 0.5f) / dh;
            const float tx = px * sw - 0.5f, ty = py * sh - 0.5f;
            const float flx = floorf(tx), fly = floorf(ty);
            const float fcx = tx - flx, fcy = ty - fly;
            float acc = 0, wsum = 0;
            for (int y = 1 - bound; y <= bound; y++) {
                for (int x = 1 - bound; x <= bound; x++) {
                    const float dx = x - fcx, dy = y - fcy;
                    const float d = sqrtf(dx * dx + dy * dy);
                    if (!(d < radius))
                        continue;
                    const float w = use_lut ? lut_lin(lut, 256, d * (1.0f / radius))
                                            : (float) orc_filter_sample(f, d);
                    int xx = (int) flx + x, yy = (int) fly + y;
                    xx = xx < 0 ? 0 : xx >= sw ? sw - 1 : xx;
                    yy = yy < 0 ? 0 : yy >= sh ? sh - 1 : yy;
                    acc = fmaf(w, src[(size_t) yy * sw + xx], acc);
                    wsum += w;
                    taps++;
                }
            }
            dst[(size_t) oy * dw + ox] = acc / wsum;
        }
    }
    return taps / ((double) dw * dh);
}

/* ======================================================================== */
/* K8 / K9: transfer functions and sigmoid (src/shaders/colorspace.c:589-894) */
/*                                                                            */
/* GLSL constants: SH_FLOAT(x) embeds the float exactly, but many constants   */
/* are printed with "%f" (6 decimals) — pf() reproduces that rounding.        */
/* pow/exp/log are libm here; the GPU uses native approximations, so these    */
/* stages are compared within a tolerance (tests/test_gpu_color.py).          */

#include <stdio.h>

static float pf(double v)
{
    char buf[64];
    snprintf(buf, sizeof(buf), "%f", v);
    return strtof(buf, NULL);
}

enum { T_UNKNOWN = 0, T_BT1886, T_SRGB, T_LINEAR, T_G18, T_G20, T_G22, T_G24, T_G26, T_G28,
       T_PROPHOTO, T_ST428, T_PQ, T_HLG, T_VLOG, T_SLOG1, T_SLOG2, T_SCRGB };

static const float O_PQ_M1 = 2610./4096 * 1./4, O_PQ_M2 = 2523./4096 * 128,
                   O_PQ_C1 = 3424./4096, O_PQ_C2 = 2413./4096 * 32, O_PQ_C3 = 2392./4096 * 32;
static const float O_HLG_A = 0.17883277, O_HLG_B = 0.28466892, O_HLG_C = 0.55991073;
static const float O_VLOG_B = 0.00873, O_VLOG_C = 0.241514, O_VLOG_D = 0.598206;
static const float O_SLOG_A = 0.432699, O_SLOG_B = 0.037584, O_SLOG_C = 0.616596 + 0.03,
                   O_SLOG_P = 3.538813, O_SLOG_Q = 0.030001, O_SLOG_K2 = 155.0 / 219.0;

static float trc_gamma(int trc)
{
    switch (trc) {
    case T_G18: return 1.8f; case T_G20: return 2.0f; case T_G24: return 2.4f;
    case T_G26: return 2.6f; case T_G28: return 2.8f; default: return 2.2f;
    }
}

static int black_scaled(int trc)
{
    switch (trc) {
    case T_BT1886: case T_PQ: case T_SCRGB: case T_VLOG: case T_SLOG1: case T_SLOG2: return 0;
    default: return 1;
    }
}

// pl_shader_linearize, colorspace.c:589-720. csp_min/max = nominal luma (NORM).
ORC_API void orc_linearize(float *img, size_t npix, int trc, float csp_min, float csp_max,
                           const float luma[3])
{
    if (trc == T_LINEAR)
        return;
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + i * 4;
        if (trc != T_SCRGB) {
            for (int k = 0; k < 3; k++)
                c[k] = fmaxf(c[k], 0.0f);                                  // :613
        }
        int scale_out = 1;
        switch (trc) {
        case T_SRGB:                                                        // :617-620
            for (int k = 0; k < 3; k++)
                c[k] = 0.04045f < c[k] ? powf((c[k] + 0.055f) / 1.055f, 2.4f)
                                       : c[k] * (1.0f / 12.92f);
            break;
        case T_BT1886: {                                                    // :622-629
            const float lb = powf(csp_min, 1 / 2.4f), lw = powf(csp_max, 1 / 2.4f);
            const float a = powf(lw - lb, 2.4f), b = lb / (lw - lb);
            for (int k = 0; k < 3; k++)
                c[k] = a * powf(c[k] + b, 2.4f);
            scale_out = 0;
            break;
        }
        case T_UNKNOWN: case T_G18: case T_G20: case T_G22: case T_G24: case T_G26: case T_G28:
            for (int k = 0; k < 3; k++)
                c[k] = powf(c[k], trc_gamma(trc));                         // :631-649
            break;
        case T_PROPHOTO:                                                    // :651-653
            for (int k = 0; k < 3; k++)
                c[k] = 0.03125f < c[k] ? powf(c[k], 1.8f) : c[k] * (1.0f / 16.0f);
            break;
        case T_ST428:                                                       // :656
            for (int k = 0; k < 3; k++)
                c[k] = (52.37f / 48.0f) * powf(c[k], 2.6f);
            break;
        case T_PQ: {                                                        // :659-666
            const float im2 = 1.0f / pf(O_PQ_M2), c1 = pf(O_PQ_C1), c2 = pf(O_PQ_C2),
                        c3 = pf(O_PQ_C3), im1 = 1.0f / pf(O_PQ_M1), k10 = pf(10000.0 / 203.0f);
            for (int k = 0; k < 3; k++) {
                float v = powf(c[k], im2);
                v = fmaxf(v - c1, 0.0f) / (c2 - c3 * v);
                v = powf(v, im1);
                c[k] = v * k10;
            }
            scale_out = 0;
            break;
        }
        case T_HLG: {                                                       // :668-683
            const float y = 1.2f * powf(1.111f, log2f(csp_max / (1000.0f / 203.0f)));
            const float b = sqrtf(3 * powf(csp_min / csp_max, 1 / y));
            const float hc = pf(O_HLG_C), ia = 1.0f / pf(O_HLG_A), hb = pf(O_HLG_B);
            for (int k = 0; k < 3; k++) {
                float v = (1 - b) * c[k] + b;
                v = 0.5f < v ? expf((v - hc) * ia) + hb : 4.0f * v * v;
                c[k] = v * (1.0f / 12.0f);
            }
            const float l = luma[0] * c[0] + luma[1] * c[1] + luma[2] * c[2];
            const float g = csp_max * powf(fmaxf(l, 0.0f), y - 1);
            for (int k = 0; k < 3; k++)
                c[k] *= g;
            scale_out = 0;
            break;
        }
        case T_VLOG:                                                        // :686-690
            for (int k = 0; k < 3; k++)
                c[k] = 0.181f <= c[k]
                    ? powf(10.0f, (c[k] - pf(O_VLOG_D)) * (1.0f / pf(O_VLOG_C))) - pf(O_VLOG_B)
                    : (c[k] - 0.125f) * (1.0f / 5.6f);
            scale_out = 0;
            break;
        case T_SLOG1:                                                       // :693-695
            for (int k = 0; k < 3; k++)
                c[k] = powf(10.0f, (c[k] - pf(O_SLOG_C)) * (1.0f / pf(O_SLOG_A))) - pf(O_SLOG_B);
            scale_out = 0;
            break;
        case T_SLOG2:                                                       // :698-702
            for (int k = 0; k < 3; k++)
                c[k] = pf(O_SLOG_Q) <= c[k]
                    ? (powf(10.0f, (c[k] - pf(O_SLOG_C)) * (1.0f / pf(O_SLOG_A))) - pf(O_SLOG_B))
                      * (1.0f / pf(O_SLOG_K2))
                    : (c[k] - pf(O_SLOG_Q)) * (1.0f / pf(O_SLOG_P));
            scale_out = 0;
            break;
        case T_SCRGB:                                                       // :705
            for (int k = 0; k < 3; k++)
                c[k] *= pf(80.0f / 203.0f);
            scale_out = 0;
            break;
        }
        if (scale_out && (csp_max != 1 || csp_min != 0)) {                  // :715-719
            for (int k = 0; k < 3; k++)
                c[k] = (csp_max - csp_min) * c[k] + csp_min;
        }
    }
}

// pl_shader_delinearize, colorspace.c:722-847
ORC_API void orc_delinearize(float *img, size_t npix, int trc, float csp_min, float csp_max,
                             const float luma[3])
{
    if (trc == T_LINEAR)
        return;
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + i * 4;
        if (black_scaled(trc) && trc != T_HLG && (csp_max != 1 || csp_min != 0)) {   // :740-747
            const float m = 1 / (csp_max - csp_min), a = -csp_min / (csp_max - csp_min);
            for (int k = 0; k < 3; k++)
                c[k] = m * c[k] + a;
        }
        if (trc != T_SCRGB) {
            for (int k = 0; k < 3; k++)
                c[k] = fmaxf(c[k], 0.0f);                                  // :750
        }
        switch (trc) {
        case T_SRGB:
            for (int k = 0; k < 3; k++)
                c[k] = 0.0031308f <= c[k] ? 1.055f * powf(c[k], 1.0f / 2.4f) - 0.055f
                                          : c[k] * 12.92f;
            break;
        case T_BT1886: {
            const float lb = powf(csp_min, 1 / 2.4f), lw = powf(csp_max, 1 / 2.4f);
            const float a = powf(lw - lb, 2.4f), b = lb / (lw - lb);
            const float ia = 1.0 / a;
            for (int k = 0; k < 3; k++)
                c[k] = powf(ia * c[k], 1.0f / 2.4f) - b;
            break;
        }
        case T_UNKNOWN: case T_G18: case T_G20: case T_G22: case T_G24: case T_G26: case T_G28:
            for (int k = 0; k < 3; k++)
                c[k] = powf(c[k], 1.0f / trc_gamma(trc));
            break;
        case T_ST428:
            for (int k = 0; k < 3; k++)
                c[k] = powf(c[k] * (48.0f / 52.37f), 1.0f / 2.6f);
            break;
        case T_PROPHOTO:
            for (int k = 0; k < 3; k++)
                c[k] = 0.001953f <= c[k] ? powf(c[k], 1.0f / 1.8f) : c[k] * 16.0f;
            break;
        case T_PQ: {
            const float ik = 1.0f / pf(10000 / 203.0f), m1 = pf(O_PQ_M1), c1 = pf(O_PQ_C1),
                        c2 = pf(O_PQ_C2), c3 = pf(O_PQ_C3), m2 = pf(O_PQ_M2);
            for (int k = 0; k < 3; k++) {
                float v = c[k] * ik;
                v = powf(v, m1);
                v = (c1 + c2 * v) / (1.0f + c3 * v);
                c[k] = powf(v, m2);
            }
            break;
        }
        case T_HLG: {
            const float y = 1.2f * powf(1.111f, log2f(csp_max / (1000.0f / 203.0f)));
            const float b = sqrtf(3 * powf(csp_min / csp_max, 1 / y));
            const float imax = 1.0f / csp_max, ex = (1 - y) / y;
            const float ha = pf(O_HLG_A), hb = pf(O_HLG_B), hc = pf(O_HLG_C);
            const float m = 1 / (1 - b), a = -b / (1 - b);
            for (int k = 0; k < 3; k++)
                c[k] *= imax;
            const float l = luma[0] * c[0] + luma[1] * c[1] + luma[2] * c[2];
            const float g = 12.0f * powf(fmaxf(1e-6f, l), ex);
            for (int k = 0; k < 3; k++) {
                float v = c[k] * g;
                v = 1.0f < v ? ha * logf(v - hb) + hc : 0.5f * sqrtf(v);
                c[k] = m * v + a;
            }
            break;
        }
        case T_VLOG:
            for (int k = 0; k < 3; k++)
                c[k] = 0.01f <= c[k] ? pf(O_VLOG_C / M_LN10) * logf(c[k] + pf(O_VLOG_B)) + pf(O_VLOG_D)
                                     : 5.6f * c[k] + 0.125f;
            break;
        case T_SLOG1:
            for (int k = 0; k < 3; k++)
                c[k] = pf(O_SLOG_A / M_LN10) * logf(c[k] + pf(O_SLOG_B)) + pf(O_SLOG_C);
            break;
        case T_SLOG2:
            for (int k = 0; k < 3; k++)
                c[k] = 0.0f <= c[k]
                    ? pf(O_SLOG_A / M_LN10) * logf(pf(O_SLOG_K2) * c[k] + pf(O_SLOG_B)) + pf(O_SLOG_C)
                    : pf(O_SLOG_P) * c[k] + pf(O_SLOG_Q);
            break;
        case T_SCRGB:
            for (int k = 0; k < 3; k++)
                c[k] *= pf(203.0f / 80.0f);
            break;
        }
    }
}

// pl_shader_sigmoidize / unsigmoidize, colorspace.c:851-894
ORC_API void orc_sigmoid(float *img, size_t npix, float center, float slope, int inverse)
{
    const float offset = 1.0 / (1 + expf(slope * center));
    const float scale = 1.0 / (1 + expf(slope * (center - 1))) - offset;
    const float inv_slope = 1.0 / slope, inv_scale = 1.0 / scale, off_scale = offset / scale;
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + i * 4;
        for (int k = 0; k < 3; k++) {
            const float v = clampf(c[k], 0.0f, 1.0f);
            c[k] = inverse ? inv_scale / (1.0f + expf(slope * (center - v))) - off_scale
                           : center - inv_slope * logf(1.0f / (v * scale + offset) - 1.0f);
        }
    }
}

// pl_shader_set_alpha pieces, colorspace.c:34-47
ORC_API void orc_alpha(float *img, size_t npix, int mode)
{
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + i * 4;
        if (mode == 0) {        // premultiply
            for (int k = 0; k < 3; k++) c[k] *= c[3];
        } else if (mode == 1) { // un-premultiply
            if (c[3] > 1e-6f)
                for (int k = 0; k < 3; k++) c[k] /= c[3];
        } else {                // alpha = 1
            c[3] = 1.0f;
        }
    }
}

/* ======================================================================== */
/* K10: peak detection (src/shaders/colorspace.c:1155-1353)                   */

enum { O_SLICES = 12, O_HIST_BINS = 64, O_PQ_BITS = 14, O_HIST_BITS = 7,
       O_HIST_BIAS = 1 << (O_HIST_BITS - 1) };

struct orc_peak_buf {
    uint32_t frame_wg_count[O_SLICES], frame_wg_active[O_SLICES];
    uint32_t frame_sum_pq[O_SLICES], frame_max_pq[O_SLICES];
    uint32_t frame_hist[O_SLICES][O_HIST_BINS];
};

// `img` holds the colour each compute invocation sees, INCLUDING the padding
// invocations of edge workgroups: it must be ceil(w/16)*16 x ceil(h/16)*16
// (the caller samples those positions like the pass does). `precise` selects
// correctly rounded pow (double) instead of float libm.
ORC_API void orc_detect_peak(const float *img, int pw, int ph, int trc, float csp_min,
                             float csp_max, const float luma[3], float black_cutoff,
                             int use_hist, struct orc_peak_buf *out)
{
    memset(out, 0, sizeof(*out));
    const int nwx = pw / 16, nwy = ph / 16;
    const float cutoff = fmaxf(black_cutoff, 0.0f) * 1e-2f;
    // workgroups are independent; their integer results are accumulated (order-free)
    #pragma omp parallel for schedule(dynamic, 1)
    for (int wy = 0; wy < nwy; wy++) {
        for (int wx = 0; wx < nwx; wx++) {
            const uint32_t wg_idx = wy * nwx + wx, slice = wg_idx % O_SLICES;
            uint32_t wg_sum = 0, wg_max = 0, wg_black = 0, wg_hist[O_HIST_BINS] = {0};
            for (int ly = 0; ly < 16; ly++) {
                for (int lx = 0; lx < 16; lx++) {
                    float c[4];
                    memcpy(c, img + ((size_t) (wy * 16 + ly) * pw + wx * 16 + lx) * 4, 16);
                    orc_linearize(c, 1, trc, csp_min, csp_max, luma);        // :1274-1275
                    float l = luma[0] * c[0] + luma[1] * c[1] + luma[2] * c[2];
                    l *= (float) (203.0f / 10000.0);                         // :1281
                    l = powf(clampf(l, 0.0f, 1.0f), O_PQ_M1);
                    l = (O_PQ_C1 + O_PQ_C2 * l) / (1.0f + O_PQ_C3 * l);
                    l = powf(l, O_PQ_M2);
                    if (cutoff) {                                            // :1286-1287
                        const float t = clampf(l / cutoff, 0.0f, 1.0f);
                        l *= t * t * (3.0f - 2.0f * t);
                    }
                    const uint32_t y_pq = (uint32_t) (16383.0f * l);         // :1288
                    if (use_hist) {                                          // :1292-1294
                        int bin = (int) y_pq >> (O_PQ_BITS - O_HIST_BITS);
                        bin -= O_HIST_BIAS;
                        bin = bin < 0 ? 0 : bin > O_HIST_BINS - 1 ? O_HIST_BINS - 1 : bin;
                        wg_hist[bin]++;
                    }
                    wg_sum += y_pq;
                    wg_max = y_pq > wg_max ? y_pq : wg_max;
                    if (cutoff && y_pq == 0)
                        wg_black++;
                }
            }
            #pragma omp critical(orc_peak)
            {
            if (use_hist) {                                                  // :1329-1337
                if (cutoff)
                    wg_hist[0] -= wg_black;
                for (int i = 0; i < O_HIST_BINS; i++)
                    out->frame_hist[slice][i] += wg_hist[i];
            }
            const uint32_t num = 256 - wg_black;                             // :1340-1347
            out->frame_wg_count[slice] += 1;
            out->frame_wg_active[slice] += num < 1 ? num : 1;
            if (num > 0) {
                out->frame_sum_pq[slice] += wg_sum / num;
                if (wg_max > out->frame_max_pq[slice])
                    out->frame_max_pq[slice] = wg_max;
            }
            }
        }
    }
}

/* ======================================================================== */
/* K11: colour mapping (src/shaders/colorspace.c:1791-1995)                    */

struct orc_color_map {
    float rgb2lms[9], lms2rgb[9];
    int tone_mode;              // -1 none, 0 clip, 1 linear, 2 LUT
    float tone_p[4];            // clip: min,max; linear: a,b,c,d; LUT: scale,offset
    const float *tone_lut;
    int tone_lut_size;
    const uint16_t *gamut_lut;  // rgba16, NULL = none
    int gamut_size[3];
    float gamut_scale, gamut_offset;
    // contrast recovery (:1879-1921): per-pixel low-frequency luma (orc_feature_luma), NULL = off
    const float *lowres;
    float cr_strength, cr_out_min, cr_out_max;
    int gamut_tricubic;         // pl_color_map_params.lut3d_tricubic (shaders/lut.c:718-760)
};

// linear lookup in the rgba16 3-D LUT at normalised coordinates, clamp to edge (what the
// reference's linear LUT sampler does, shaders/lut.c:628-650)
static void o_gamut_trilinear(const struct orc_color_map *m, const float idx[3], float o[3])
{
    int i0[3], i1[3];
    float fr[3];
    for (int k = 0; k < 3; k++) {
        const float pos = clampf(idx[k], 0.0f, 1.0f) * (float) (m->gamut_size[k] - 1);
        const float fl = floorf(pos);
        i0[k] = (int) fl;
        i1[k] = i0[k] + 1 < m->gamut_size[k] ? i0[k] + 1 : m->gamut_size[k] - 1;
        fr[k] = pos - fl;
    }
    const int sx = m->gamut_size[0], sy = m->gamut_size[1];
#define GT(x, y, z, ch) (m->gamut_lut[(((size_t) (z) * sy + (y)) * sx + (x)) * 4 + (ch)] / 65535.0f)
    for (int ch = 0; ch < 3; ch++) {
        const float c00 = mixf(GT(i0[0], i0[1], i0[2], ch), GT(i1[0], i0[1], i0[2], ch), fr[0]);
        const float c10 = mixf(GT(i0[0], i1[1], i0[2], ch), GT(i1[0], i1[1], i0[2], ch), fr[0]);
        const float c01 = mixf(GT(i0[0], i0[1], i1[2], ch), GT(i1[0], i0[1], i1[2], ch), fr[0]);
        const float c11 = mixf(GT(i0[0], i1[1], i1[2], ch), GT(i1[0], i1[1], i1[2], ch), fr[0]);
        o[ch] = mixf(mixf(c00, c10, fr[1]), mixf(c01, c11, fr[1]), fr[2]);
    }
#undef GT
}

// the `lut_tricubic` GLSL function (shaders/lut.c:721-757), statement by statement
static void o_gamut_tricubic(const struct orc_color_map *m, const float idx[3], float o[3])
{
    float g0[3], h0[3], h1[3];
    for (int k = 0; k < 3; k++) {
        const float scale = (float) (m->gamut_size[k] - 1), scale_inv = 1.0f / scale;
        const float pos = idx[k] * scale;
        const float fpos = pos - floorf(pos);
        const float base = pos - fpos;
        const float fpos2 = fpos * fpos, inv = 1.0f - fpos, inv2 = inv * inv;
        const float w0 = 1.0f / 6.0f * inv2 * inv;
        const float w1 = 2.0f / 3.0f - 0.5f * fpos2 * (2.0f - fpos);
        const float w2 = 2.0f / 3.0f - 0.5f * inv2 * (2.0f - inv);
        const float w3 = 1.0f / 6.0f * fpos2 * fpos;
        g0[k] = w0 + w1;
        const float g1 = w2 + w3;
        h0[k] = scale_inv * ((w1 / g0[k]) - 1.0f + base);
        h1[k] = scale_inv * ((w3 / g1) + 1.0f + base);
    }
    float c000[3], c001[3], c010[3], c011[3], c100[3], c101[3], c110[3], c111[3];
#define LUT(out, X, Y, Z) do { const float p_[3] = { X[0], Y[1], Z[2] }; o_gamut_trilinear(m, p_, out); } while (0)
    LUT(c000, h0, h0, h0); LUT(c100, h1, h0, h0);
    LUT(c010, h0, h1, h0); LUT(c110, h1, h1, h0);
    LUT(c001, h0, h0, h1); LUT(c101, h1, h0, h1);
    LUT(c011, h0, h1, h1); LUT(c111, h1, h1, h1);
#undef LUT
    for (int ch = 0; ch < 3; ch++) {
        const float a00 = mixf(c100[ch], c000[ch], g0[0]);
        const float a10 = mixf(c110[ch], c010[ch], g0[0]);
        const float a0 = mixf(a10, a00, g0[1]);
        const float a01 = mixf(c101[ch], c001[ch], g0[0]);
        const float a11 = mixf(c111[ch], c011[ch], g0[0]);
        const float a1 = mixf(a11, a01, g0[1]);
        o[ch] = mixf(a1, a0, g0[2]);
    }
}

static float o_lut1d(const float *lut, int n, float x)
{
    const float fpos = clampf(x, 0.0f, 1.0f) * (float) (n - 1);
    const float fb = floorf(fpos), fc = ceilf(fpos);
    return mixf(lut[(int) fb], lut[(int) fc], fpos - fb);
}

// pl_shader_custom_lut (shaders/lut.c:212-280; 1D linear :731-745, 3D tetrahedral :762-809).
// `lut`: RGB triples as in pl_custom_lut.data; size[1] = size[2] = 0 for 1D.
ORC_API void orc_custom_lut(float *img, size_t npix, const float *lut, const int size[3])
{
    const int sx = size[0], sy = size[1], sz = size[2];
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + i * 4;
        if (!sy) {
            for (int k = 0; k < 3; k++) {
                const float fpos = clampf(c[k], 0.0f, 1.0f) * (float) (sx - 1);
                const float fb = floorf(fpos), fc = ceilf(fpos);
                c[k] = mixf(lut[3 * (int) fb + k], lut[3 * (int) fc + k], fpos - fb);
            }
            continue;
        }
        const float pos[3] = { clampf(c[0], 0.0f, 1.0f) * (float) (sx - 1),
                               clampf(c[1], 0.0f, 1.0f) * (float) (sy - 1),
                               clampf(c[2], 0.0f, 1.0f) * (float) (sz - 1) };
        float fpart[3], s[3];
        int v0[3], v1[3], v2[3], v3[3];
        for (int k = 0; k < 3; k++) {
            const float base = floorf(pos[k]);
            fpart[k] = s[k] = pos[k] - base;
            v0[k] = v1[k] = (int) base;
            v3[k] = v2[k] = (int) ceilf(pos[k]);
        }
        const int cge[3] = { fpart[0] >= fpart[1], fpart[1] >= fpart[2], fpart[2] >= fpart[0] };
        // c_xy = cge[0], c_yx = !cge[0], c_yz = cge[1], c_zy = !cge[1], c_zx = cge[2], c_xz = !cge[2]
        static const int order[6][3] = { {0,1,2}, {0,2,1}, {2,0,1}, {2,1,0}, {1,2,0}, {1,0,2} };
        for (int t = 0; t < 6; t++) {
            const int X = order[t][0], Y = order[t][1], Z = order[t][2];
            // c_AB for A, B in {x, y, z}: A >= B along the cyclic pairs, else the negation
#define CAB(A, B) ((B) == ((A) + 1) % 3 ? cge[A] : !cge[B])
            if (CAB(X, Y) && CAB(Y, Z)) {
                s[0] = fpart[X]; s[1] = fpart[Y]; s[2] = fpart[Z];
                v1[X] = v3[X];
                v2[Z] = v0[Z];
            }
#undef CAB
        }
#define L3(v, k) lut[3 * (((size_t) (v)[2] * sy + (v)[1]) * sx + (v)[0]) + (k)]
        const float w0 = 1.0f - s[0], w1 = s[0] - s[1], w2 = s[1] - s[2], w3 = s[2];
        float o[3];
        for (int k = 0; k < 3; k++)
            o[k] = w0 * L3(v0, k) + w1 * L3(v1, k) + w2 * L3(v2, k) + w3 * L3(v3, k);
#undef L3
        c[0] = o[0]; c[1] = o[1]; c[2] = o[2];
    }
}

static float o_tone(const struct orc_color_map *m, float I)
{
    switch (m->tone_mode) {
    case 0: return clampf(I, m->tone_p[0], m->tone_p[1]);                          // :1826
    case 1:                                                                        // :1836-1842
        I = m->tone_p[0] * I + m->tone_p[1];
        I = clampf(I, 0.0f, 1.0f);
        return m->tone_p[2] * I + m->tone_p[3];
    default:                                                                       // :1873
        return o_lut1d(m->tone_lut, m->tone_lut_size, m->tone_p[0] * I + m->tone_p[1]);
    }
}

// pl_shader_extract_features after the linearization (colorspace.c:1383-1404): img.r = I of
// IPT, img.gba = (0, 0, 1). `klms` = (203/10000 as "%f") * rgb2lms, row-major, in fp32.
ORC_API void orc_extract_features(float *img, size_t npix, const float klms[9])
{
    const float m1 = pf(O_PQ_M1), m2 = pf(O_PQ_M2), c1 = pf(O_PQ_C1), c2 = pf(O_PQ_C2),
                c3 = pf(O_PQ_C3);
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + i * 4;
        float lms[3];
        for (int r = 0; r < 3; r++) {
            float v = klms[3*r] * c[0] + klms[3*r+1] * c[1] + klms[3*r+2] * c[2];
            v = powf(fmaxf(v, 0.0f), m1);
            v = (c1 + c2 * v) / (1.0f + c3 * v);
            lms[r] = powf(v, m2);
        }
        c[0] = 0.4000f * lms[0] + 0.4000f * lms[1] + 0.2000f * lms[2];
        c[1] = c[2] = 0.0f;
        c[3] = 1.0f;
    }
}

// The low-frequency luma the contrast recovery reads (colorspace.c:1886-1906): bicubic
// interpolation of the w x h single-channel feature map `fm`, evaluated as four bilinear taps,
// at the centre of every pixel of an out_w x out_h pass.
static float fm_linear(const float *fm, int w, int h, float px, float py)
{
    const float u = px * (float) w - 0.5f, v = py * (float) h - 0.5f;
    const float fu = floorf(u), fv = floorf(v);
    const float ax = u - fu, ay = v - fv;
    const int x0 = wrap((int) fu, w, ADDR_CLAMP), x1 = wrap((int) fu + 1, w, ADDR_CLAMP);
    const int y0 = wrap((int) fv, h, ADDR_CLAMP), y1 = wrap((int) fv + 1, h, ADDR_CLAMP);
    const float t0 = mixf(fm[(size_t) y0 * w + x0], fm[(size_t) y0 * w + x1], ax);
    const float t1 = mixf(fm[(size_t) y1 * w + x0], fm[(size_t) y1 * w + x1], ax);
    return mixf(t0, t1, ay);
}

ORC_API void orc_feature_luma(const float *fm, int w, int h, int out_w, int out_h, float *luma)
{
    const float osx = 1.0 / out_w, osy = 1.0 / out_h;
    const float lsize[2] = { (float) w, (float) h };
    const float lpt[2] = { 1.0f / (float) w, 1.0f / (float) h };
    for (int y = 0; y < out_h; y++) {
        for (int x = 0; x < out_w; x++) {
            const float pos[2] = { osx * ((float) x + 0.5f), osy * ((float) y + 0.5f) };
            float g[2], hh[2][2];
            for (int a = 0; a < 2; a++) {
                const float t = pos[a] * lsize[a] + 0.5f;
                const float fr = t - floorf(t);
                const float fr2 = fr * fr, inv = 1.0f - fr, inv2 = inv * inv;
                const float w0 = 1.0f / 6.0f * inv2 * inv;
                const float w1 = 2.0f / 3.0f - 0.5f * fr2 * (2.0f - fr);
                const float w2 = 2.0f / 3.0f - 0.5f * inv2 * (2.0f - inv);
                const float w3 = 1.0f / 6.0f * fr2 * fr;
                const float g0 = w0 + w1, g1 = w2 + w3;
                g[a] = g0;
                hh[a][0] = w1 / g0 + inv - 2.0f;
                hh[a][1] = w3 / g1 + inv;
            }
            const float px0 = pos[0] + lpt[0] * hh[0][0], py0 = pos[1] + lpt[1] * hh[1][0];
            const float px1 = pos[0] + lpt[0] * hh[0][1], py1 = pos[1] + lpt[1] * hh[1][1];
            const float l00 = fm_linear(fm, w, h, px0, py0), l01 = fm_linear(fm, w, h, px0, py1);
            const float l0 = mixf(l01, l00, g[1]);
            const float l10 = fm_linear(fm, w, h, px1, py0), l11 = fm_linear(fm, w, h, px1, py1);
            const float l1 = mixf(l11, l10, g[1]);
            luma[(size_t) y * out_w + x] = mixf(l1, l0, g[0]);
        }
    }
}

ORC_API void orc_color_map(float *img, size_t npix, const struct orc_color_map *m)
{
    const float k203 = pf(203.0f / 10000), m1 = pf(O_PQ_M1), m2 = pf(O_PQ_M2),
                c1 = pf(O_PQ_C1), c2 = pf(O_PQ_C2), c3 = pf(O_PQ_C3),
                im1 = 1.0f / pf(O_PQ_M1), im2 = 1.0f / pf(O_PQ_M2), k10 = pf(10000 / 203.0f),
                hpi = pf(0.5f / M_PI);
    #pragma omp parallel for schedule(static) if (npix > 16384)
    for (size_t i = 0; i < npix; i++) {
        float *c = img + i * 4;
        // lms = rgb2lms * rgb; lmspq = PQ(k * lms); ipt = lms2ipt * lmspq         :1792-1799
        float lms[3];
        for (int r = 0; r < 3; r++)
            lms[r] = m->rgb2lms[3*r] * c[0] + m->rgb2lms[3*r+1] * c[1] + m->rgb2lms[3*r+2] * c[2];
        for (int r = 0; r < 3; r++) {
            float v = powf(fmaxf(k203 * lms[r], 0.0f), m1);
            v = (c1 + c2 * v) / (1.0f + c3 * v);
            lms[r] = powf(v, m2);
        }
        float I = 0.4000f * lms[0] + 0.4000f * lms[1] + 0.2000f * lms[2];
        float P = 4.4550f * lms[0] + -4.8510f * lms[1] + 0.3960f * lms[2];
        float T = 0.8056f * lms[0] + 0.3572f * lms[1] + -1.1628f * lms[2];
        const float i_orig = I;

        if (m->tone_mode >= 0) {
            if (m->lowres) {                                                       // :1908-1916
                const float highres = clampf(I, 0.0f, 1.0f);
                const float lowres = clampf(m->lowres[i], 0.0f, 1.0f);
                const float detail = highres - lowres;
                const float base = o_tone(m, highres);
                const float sharp = o_tone(m, lowres) + detail;
                I = clampf(mixf(base, sharp, m->cr_strength), m->cr_out_min, m->cr_out_max);
            } else {
                I = o_tone(m, I);
            }
            const float hx = ((i_orig - 6.0f) * i_orig + 9.0f) * i_orig;         // :1930-1932
            const float hy = ((I - 6.0f) * I + 9.0f) * I;
            const float k = fminf(i_orig / I, hy / hx);
            P *= k;
            T *= k;
        }

        if (m->gamut_lut) {                                                       // :1962-1967
            const float idx[3] = { m->gamut_scale * I + m->gamut_offset,
                                   2.0f * sqrtf(P * P + T * T),
                                   hpi * atan2f(T, P) + 0.5f };
            float o[3];
            if (m->gamut_tricubic) {
                o_gamut_tricubic(m, idx, o);
            } else {
                o_gamut_trilinear(m, idx, o);
            }
            I = o[0];
            P = o[1] - 32768.0f / 65535.0f;
            T = o[2] - 32768.0f / 65535.0f;
        }

        // back to linear RGB                                                      :1985-1995
        float l3[3] = { 1.0f * I + 0.0975689f * P + 0.205226f * T,
                        1.0f * I + -0.1138760f * P + 0.133217f * T,
                        1.0f * I + 0.0326151f * P + -0.676887f * T };
        for (int r = 0; r < 3; r++) {
            float v = powf(fmaxf(l3[r], 0.0f), im2);
            v = fmaxf(v - c1, 0.0f) / (c2 - c3 * v);
            l3[r] = powf(v, im1) * k10;
        }
        for (int r = 0; r < 3; r++)
            c[r] = m->lms2rgb[3*r] * l3[0] + m->lms2rgb[3*r+1] * l3[1] + m->lms2rgb[3*r+2] * l3[2];
    }
}

/* ======================================================================== */
/* K4: separable filters (src/shaders/sampling.c:950-1104, fill_ortho_lut :914-942) */

// `rows`: 256 x row_stride LUT rows as uploaded (i.e. after fill_ortho_lut: raw weights,
// or {w0+w1, w1/(w0+w1)} pairs when use_linear). dir: 0 = horizontal, 1 = vertical.
// Texture-unit rule: taps sit on texel centres along `dir` -> the texel itself;
// across `dir`: nearest when the rect starts on the texel grid there, exact bilinear otherwise.
ORC_API void orc_sample_ortho(const struct orc_src *s, const float *rows, int row_size,
                              int row_stride, int dir, int use_linear, int use_ar,
                              float antiring, float scale, unsigned mask,
                              int out_w, int out_h, float *out)
{
    float p[4][2], pt[2];
    corners(s, p, pt);
    const float osx = 1.0 / out_w, osy = 1.0 / out_h;
    const int N = row_size;
    const float r0v = dir ? s->rect[0] : s->rect[1];
    const int across_linear = r0v != truncf(r0v);
    const int na = dir ? s->h : s->w, no = dir ? s->w : s->h;

    WIN_SETUP(out_w, out_h);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int y = wy0; y < wy1; y++) {
        for (int x = wx0; x < wx1; x++) {
            const float fx = osx * ((float) x + 0.5f), fy = osy * ((float) y + 0.5f);
            const float px = attr(p, 0, fx, fy), py = attr(p, 1, fx, fy);
            const float pa = dir ? py : px, po = dir ? px : py;
            const float ta = pa * (float) na - 0.5f, fla = floorf(ta);
            const float fcoord = ta - fla;                                    // :1060-1061
            const int first = (int) fla - (N / 2 - 1);                        // :1062

            int o0, o1 = 0;
            float ofrac = 0.0f;
            if (!across_linear) {
                o0 = (int) floorf(po * (float) no);
            } else {
                const float to = po * (float) no - 0.5f, flo = floorf(to);
                ofrac = to - flo;
                o0 = (int) flo;
                o1 = o0 + 1;
            }

            const float fpos = clampf(fcoord, 0.0f, 1.0f) * 255.0f;
            const float fbase = floorf(fpos), fr = fpos - fbase;
            const float *ra = rows + (size_t) (int) fbase * row_stride;
            const float *rb = rows + (size_t) ((int) fbase + 1 > 255 ? 255 : (int) fbase + 1) * row_stride;

            float ca[4] = {0, 0, 0, 0}, lo[4] = {1e9f, 1e9f, 1e9f, 1e9f}, hi[4] = {0, 0, 0, 0};
            for (int n = 0; n < N; n += use_linear ? 2 : 1) {
                const float w = mixf(ra[n], rb[n], fr);
                float c[4];
                for (int half = 0; half < (use_linear ? 2 : 1); half++) {
                    const int i = first + n + half;
                    float t[4];
                    const float *a = dir ? texel(s, o0, i) : texel(s, i, o0);
                    if (across_linear) {
                        const float *b = dir ? texel(s, o1, i) : texel(s, i, o1);
                        for (int k = 0; k < 4; k++)
                            t[k] = mixf(a[k], b[k], ofrac);
                    } else {
                        memcpy(t, a, 16);
                    }
                    if (!half) {
                        memcpy(c, t, 16);
                    } else {
                        const float f = mixf(ra[n + 1], rb[n + 1], fr);      // :1072-1073
                        for (int k = 0; k < 4; k++)
                            c[k] = mixf(c[k], t[k], f);
                    }
                }
                if (use_ar && (n == N / 2 - 1 || n == N / 2)) {               // :1076-1079
                    for (int k = 0; k < 4; k++) {
                        lo[k] = fminf(lo[k], c[k]);
                        hi[k] = fmaxf(hi[k], c[k]);
                    }
                }
                for (int k = 0; k < 4; k++)
                    ca[k] = fmaf(w, c[k], ca[k]);                             // :1082
            }
            float *o = WIN_OUT(out, x, y);
            const float def[4] = {0, 0, 0, 1};
            for (int k = 0; k < 4; k++) {
                if (use_ar)
                    ca[k] = mixf(ca[k], clampf(ca[k], lo[k], hi[k]), antiring); // :1085
                o[k] = (mask & (1u << k)) ? scale * ca[k] : def[k];            // :1086
            }
        }
    }
}

/* ======================================================================== */
/* K6: debanding (src/shaders/sampling.c:183-275), PRNG src/shaders.c:965-998  */


ORC_API void orc_deband(const struct orc_src *s, int iterations, float threshold, float radius,
                        float grain, const float grain_neutral[3], float scale, unsigned mask,
                        unsigned frame_index, int out_w, int out_h, float *out)
{
    float p[4][2], pt[2];
    corners(s, p, pt);
    const float osx = 1.0 / out_w, osy = 1.0 / out_h;
    mask &= 7u;                                                               // :201
    const float thr = threshold / (1000 * scale);                             // :225
    const float two_pi = pf(M_PI * 2);                                        // "%f"
    WIN_SETUP(out_w, out_h);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int y = wy0; y < wy1; y++) {
        for (int x = wx0; x < wx1; x++) {
            const float fx = osx * ((float) x + 0.5f), fy = osy * ((float) y + 0.5f);
            const float px = attr(p, 0, fx, fy), py = attr(p, 1, fx, fy);
            float color[4], res[3];
            tex_nearest(s, px, py, color);
            memcpy(res, color, 12);
            uint32_t st[3] = { (uint32_t) ((float) x + 0.5f), (uint32_t) ((float) y + 0.5f),
                               frame_index };
            float rnd[3];
            for (int i = 1; mask && i <= iterations; i++) {
                o_pcg3d(st, rnd);
                float dx = rnd[0] * ((float) i * radius);                     // :232
                const float ang = rnd[1] * two_pi;
                const float dy = dx * sinf(ang);
                dx = dx * cosf(ang);                                          // :233
                const float ox[4] = { dx, -dx, -dx, dx }, oy[4] = { dy, dy, -dy, -dy };
                float avg[3] = {0, 0, 0};
                for (int k = 0; k < 4; k++) {                                 // :236-239
                    float t[4];
                    tex_nearest(s, px + pt[0] * ox[k], py + pt[1] * oy[k], t);
                    for (int c = 0; c < 3; c++)
                        avg[c] += t[c];
                }
                const float bound = thr / (float) i;                          // :244
                for (int c = 0; c < 3; c++) {
                    const float a = avg[c] * 0.25f;
                    if ((mask & (1u << c)) && !(fabsf(res[c] - a) > bound))   // :247-250
                        res[c] = a;
                }
            }
            if (mask && grain > 0) {                                          // :255-268
                o_pcg3d(st, rnd);
                const float g = grain / (1000.0 * scale);
                int k = 0;
                for (int c = 0; c < 3; c++) {
                    if (!(mask & (1u << c)))
                        continue;
                    const float neutral = grain_neutral[k] / scale;
                    const float strength = fminf(fabsf(res[c] - neutral), g);
                    res[c] += strength * (rnd[k] - 0.5f);
                    k++;
                }
            }
            float *o = WIN_OUT(out, x, y);
            for (int c = 0; c < 3; c++)
                o[c] = ((mask & (1u << c)) ? res[c] : color[c]) * scale;      // :270-271
            o[3] = color[3] * scale;
        }
    }
}

/* ======================================================================== */
/* K14: error diffusion (src/shaders/dithering.c:326-527)                       */

// Sequential walk of the same sheared-column order with the same packed ring buffer
// (ids of one workgroup step never touch each other's slots, see k_errdiff.hip, so the
// serial order reproduces the parallel result exactly). pattern[dy][dx + 2].
ORC_API void orc_error_diffusion(const float *img, int w, int h, int depth, int shift,
                                 int divisor, const int pattern[3][5], float *out)
{
    int ring_cols = 0;
    for (int dy = 0; dy <= 2; dy++) {
        for (int dx = -2; dx <= 2; dx++) {
            if (pattern[dy][dx + 2] && dx + dy * shift > ring_cols)
                ring_cols = dx + dy * shift;
        }
    }
    ring_cols += 1;
    const int ring_rows = h + 2;
    const uint32_t ring_size = (uint32_t) ring_rows * ring_cols;
    uint32_t *ring = calloc(ring_size, sizeof(uint32_t));
    const int shifted_width = w + (h - 1) * shift;
    const float quant = (float) ((1 << depth) - 1);

    for (int64_t id = 0; id < (int64_t) h * shifted_width; id++) {
        const int y = (int) (id % h), xs = (int) (id / h);
        const int x = xs - y * shift;
        if (x < 0 || x >= w)
            continue;
        const uint32_t idx = (uint32_t) (xs * ring_rows + y) % ring_size;
        const float *po = img + ((size_t) y * w + x) * 4;
        const uint32_t e32 = ring[idx] + ((128u << 24) | (128u << 12) | 128u);
        ring[idx] = 0;
        const int e[3] = { (int) ((e32 >> 24) & 0xFF) - 128, (int) ((e32 >> 12) & 0xFF) - 128,
                           (int) (e32 & 0xFF) - 128 };
        float pix[3], dith[3], ediv[3];
        float *o = out + ((size_t) y * w + x) * 4;
        for (int c = 0; c < 3; c++) {
            pix[c] = po[c] * quant + (float) e[c] / 254.0f;                  // :462-465
            dith[c] = rintf(pix[c]);                                         // round(): half-even
            o[c] = dith[c] / quant;
            ediv[c] = (pix[c] - dith[c]) * 254.0f / (float) divisor;         // :470
        }
        o[3] = po[3];
        for (int dividend = 1; dividend <= divisor; dividend++) {
            int assigned = 0;
            uint32_t packed = 0;
            for (int dy = 0; dy <= 2; dy++) {
                for (int dx = -2; dx <= 2; dx++) {
                    if (pattern[dy][dx + 2] != dividend)
                        continue;
                    if (!assigned) {
                        assigned = 1;
                        const int t[3] = { (int) rintf(ediv[0] * (float) dividend),
                                           (int) rintf(ediv[1] * (float) dividend),
                                           (int) rintf(ediv[2] * (float) dividend) };
                        packed = ((uint32_t) (t[0] & 0xFF) << 24) | ((uint32_t) (t[1] & 0xFF) << 12) |
                                 (uint32_t) (t[2] & 0xFF);
                    }
                    if (dx < 0 && x < -dx)                                   // :508-509
                        continue;
                    const uint32_t delta = (uint32_t) ((dx + dy * shift) * ring_rows + dy);
                    ring[(idx + delta) % ring_size] += packed;
                }
            }
        }
    }
    free(ring);
}

/* ======================================================================== */
/* Self-checks of arithmetic identities the product relies on (tests/test_host.py) */

// plh_un8 / plh_un16 (csrc/hip/devmath.hiph): q = v*(1/d); q += fma(-q, d, v)*(1/d) must be
// the correctly rounded v/d for every code value. Returns the number of mismatches.
/* ======================================================================== */
/* Dolby Vision (src/shaders/colorspace.c:51-271, :285-292, :392-420)             */

// struct pl_reshape_data (include/libplacebo/colorspace.h:139-148), as the caller fills it
struct orc_dovi_comp {
    int num_pivots;
    float pivots[9];
    int method[8];              // 0 = polynomial, 1 = MMR
    float poly_coeffs[8][3];
    int mmr_order[8];
    float mmr_constant[8];
    float mmr_coeffs[8][3][7];
};

static float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static float dot4(const float a[4], const float b[4])
{
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
}

// pl_shader_dovi_reshape: `sig` is the clamped input colour for all three components (:119);
// the piece is the one whose pivots bracket s (the shader's tree of mix(., ., s >= pivot[i]) over
// the inner pivots and 1e9 sentinels selects exactly the number of inner pivots <= s, :186-217)
ORC_API void orc_dovi_reshape(float *img, size_t npix, const struct orc_dovi_comp comp[3])
{
    for (size_t i = 0; i < npix; i++) {
        float *c = img + 4 * i;
        const float sig[3] = { clampf(c[0], 0.0f, 1.0f), clampf(c[1], 0.0f, 1.0f),
                               clampf(c[2], 0.0f, 1.0f) };
        for (int ch = 0; ch < 3; ch++) {
            const struct orc_dovi_comp *k = &comp[ch];
            if (!k->num_pivots)
                continue;
            float s = sig[ch];
            int piece = 0;
            for (int p = 1; p < k->num_pivots - 1; p++)
                piece += s >= k->pivots[p];
            if (k->method[piece] == 0) {
                const float *co = k->poly_coeffs[piece];
                s = (co[2] * s + co[1]) * s + co[0];                        // reshape_poly (:101)
            } else {                                                        // reshape_mmr (:52-97)
                const int order = k->mmr_order[piece];
                const float (*w)[7] = k->mmr_coeffs[piece];
                float sigX[4] = { sig[0] * sig[1], sig[0] * sig[2], sig[1] * sig[2], 0.0f };
                sigX[3] = sigX[0] * sig[2];
                s = k->mmr_constant[piece];
                s += dot3(w[0], sig);
                s += dot4(w[0] + 3, sigX);
                if (order >= 2) {
                    const float sig2[3] = { sig[0] * sig[0], sig[1] * sig[1], sig[2] * sig[2] };
                    const float sigX2[4] = { sigX[0] * sigX[0], sigX[1] * sigX[1], sigX[2] * sigX[2],
                                             sigX[3] * sigX[3] };
                    s += dot3(w[1], sig2);
                    s += dot4(w[1] + 3, sigX2);
                    if (order >= 3) {
                        const float sig3[3] = { sig2[0] * sig[0], sig2[1] * sig[1], sig2[2] * sig[2] };
                        const float sigX3[4] = { sigX2[0] * sigX[0], sigX2[1] * sigX[1],
                                                 sigX2[2] * sigX[2], sigX2[3] * sigX[3] };
                        s += dot3(w[2], sig3);
                        s += dot4(w[2] + 3, sigX3);
                    }
                }
            }
            c[ch] = clampf(s, k->pivots[0], k->pivots[k->num_pivots - 1]);
        }
    }
}

// The non-linear tail of Dolby Vision decoding (:392-420): PQ EOTF, LMS -> RGB (row-major 3 x 3,
// the hard-coded BT.2020 HPE inverse times the stream's matrix: the caller multiplies), PQ OETF,
// with the constants as the shader text prints them
ORC_API void orc_dovi_lms(float *img, size_t npix, const float m[9])
{
    const float im2 = 1.0f / pf(O_PQ_M2), c1 = pf(O_PQ_C1), c2 = pf(O_PQ_C2), c3 = pf(O_PQ_C3),
                im1 = 1.0f / pf(O_PQ_M1), m1 = pf(O_PQ_M1), m2 = pf(O_PQ_M2);
    for (size_t i = 0; i < npix; i++) {
        float *c = img + 4 * i, v[3];
        for (int k = 0; k < 3; k++) {
            float x = powf(fmaxf(c[k], 0.0f), im2);
            x = fmaxf(x - c1, 0.0f) / (c2 - c3 * x);
            v[k] = powf(x, im1);
        }
        for (int r = 0; r < 3; r++) {
            float x = m[3 * r] * v[0] + m[3 * r + 1] * v[1] + m[3 * r + 2] * v[2];
            x = powf(fmaxf(x, 0.0f), m1);
            x = (c1 + c2 * x) / (1.0f + c3 * x);
            c[r] = powf(x, m2);
        }
    }
}

/* ======================================================================== */
/* pl_shader_distort (src/shaders/sampling.c:1108-1217)                          */

// The canvas [-1, 1]^2 (y up: the attribute runs from +1 at the top row to -1 at the bottom,
// :1156-1160) through `tf` (canvas -> texture coordinates, row-major 2 x 2 + offset: what the
// host inverts and binds, :1166-1176), a bilinear fetch or the bicubic of sampling.c:335-361,
// and with an alpha mode the fade of everything outside the texture over one texel (:1204-1211).
ORC_API void orc_distort(const struct orc_src *s, const float tf[6], int bicubic, int alpha_mode,
                         int out_w, int out_h, float *out)
{
    const float canvas[4][2] = { { -1.0f, 1.0f }, { 1.0f, 1.0f }, { -1.0f, -1.0f }, { 1.0f, -1.0f } };
    const float pt[2] = { (float) (1.0 / s->w), (float) (1.0 / s->h) };
    const float size[2] = { (float) s->w, (float) s->h };
    const float osx = 1.0 / out_w, osy = 1.0 / out_h;
    for (int y = 0; y < out_h; y++) {
        for (int x = 0; x < out_w; x++) {
            const float fx = osx * ((float) x + 0.5f), fy = osy * ((float) y + 0.5f);
            const float cx = attr(canvas, 0, fx, fy), cy = attr(canvas, 1, fx, fy);
            const float pos[2] = { (tf[0] * cx + tf[1] * cy) + tf[4], (tf[2] * cx + tf[3] * cy) + tf[5] };
            float c[4];
            if (bicubic) {
                float g[4], h[4];
                const float off[2] = {0, 0};
                for (int k = 0; k < 2; k++) {
                    const float fr = fractf(pos[k] * size[k] + 0.5f);
                    const float fr2 = fr * fr, inv = 1.0f - fr, inv2 = inv * inv;
                    const float w0 = 1.0f / 6.0f * inv2 * inv;
                    const float w1 = 2.0f / 3.0f - 0.5f * fr2 * (2.0f - fr);
                    const float w2 = 2.0f / 3.0f - 0.5f * inv2 * (2.0f - inv);
                    const float w3 = 1.0f / 6.0f * fr2 * fr;
                    g[k] = w0 + w1;
                    g[k + 2] = w2 + w3;
                    h[k] = w1 / g[k] + inv - 2.0f;
                    h[k + 2] = w3 / g[k + 2] + inv;
                }
                fast4(s, pt, pos[0], pos[1], g, h, off, 1.0f, c);
            } else {
                tex_linear(s, pos[0], pos[1], c);
            }
            if (alpha_mode) {
                const float bx = smoothstep01(fminf(pos[0], 1.0f - pos[0]) / pt[0]);
                const float by = smoothstep01(fminf(pos[1], 1.0f - pos[1]) / pt[1]);
                const float border = bx * by;
                if (alpha_mode == 2) {      // PL_ALPHA_PREMULTIPLIED
                    for (int k = 0; k < 3; k++)
                        c[k] *= border;
                }
                c[3] *= border;
            }
            memcpy(out + ((size_t) y * out_w + x) * 4, c, 16);
        }
    }
}

/* ======================================================================== */
/* overlays (src/renderer.c:811-1020) and the blend unit (gpu.h pl_blend_params)  */

// One overlay part as it lands on a plane. The reference emits two triangles whose vertices carry
// `pos` (the part's dst corners through the overlay transform) and `coord` (its src corners over
// the texture size) and lets the rasteriser interpolate (:896-931); on a parallelogram that
// interpolation is the affine function restated here, and a pixel belongs to the part when its
// centre lies inside (top-left rule: the low edges count, the high ones do not).
//     coord.x = u0 + ((p.x - ox) * ux + (p.y - oy) * uy),  coord.y likewise with v
struct orc_overlay_part {
    float x0, y0, x1, y1;
    float ox, oy;
    float ux, uy, u0;
    float vx, vy, v0;
    float color[4];
};

enum { ORC_OVERLAY_NORMAL = 0, ORC_OVERLAY_MONOCHROME };

// The fragments of one part over a w x h plane: mask = covered, color = what the overlay shader
// starts from (:952-961: the texture, or the part's colour), coverage = texture.r (the factor of
// :987-991; 1 for NORMAL)
ORC_API void orc_overlay_fragments(const float *tex, int tw, int th, int linear, int mode,
                                   const struct orc_overlay_part *q, int w, int h,
                                   float *color, float *coverage, uint8_t *mask)
{
    const struct orc_src s = { .tex = tex, .w = tw, .h = th, .address_mode = ADDR_CLAMP };
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            const size_t i = (size_t) y * w + x;
            const float px = (float) x + 0.5f, py = (float) y + 0.5f;
            mask[i] = px >= q->x0 && px < q->x1 && py >= q->y0 && py < q->y1;
            coverage[i] = 1.0f;
            if (!mask[i])
                continue;
            const float u = q->u0 + ((px - q->ox) * q->ux + (py - q->oy) * q->uy);
            const float v = q->v0 + ((px - q->ox) * q->vx + (py - q->oy) * q->vy);
            float t[4];
            if (linear)
                tex_linear(&s, u, v, t);
            else
                tex_nearest(&s, u, v, t);
            if (mode == ORC_OVERLAY_MONOCHROME) {
                memcpy(color + 4 * i, q->color, 16);
                coverage[i] = t[0];
            } else {
                memcpy(color + 4 * i, t, 16);
            }
        }
    }
}

// pl_blend_mode (gpu.h): ZERO, ONE, SRC_ALPHA, ONE_MINUS_SRC_ALPHA
static float blend_factor(int mode, float src_alpha)
{
    return mode == 1 ? 1.0f : mode == 2 ? src_alpha : mode == 3 ? 1.0f - src_alpha : 0.0f;
}

// The blend unit over the masked pixels: a fixed-point target clamps the fragment to [0, 1]
// first; result = src * Sf + dst * Df per channel, rgb and alpha with their own factors
// (factors = src_rgb, dst_rgb, src_alpha, dst_alpha; enable = 0: the fragment replaces the
// target). The caller rounds `dst` through the target format afterwards.
ORC_API void orc_blend(float *dst, const float *src, const uint8_t *mask, size_t npix,
                       const int factors[4], int enable, int fixed_point)
{
    for (size_t i = 0; i < npix; i++) {
        if (mask && !mask[i])
            continue;
        float c[4];
        memcpy(c, src + 4 * i, 16);
        if (fixed_point) {
            for (int k = 0; k < 4; k++)
                c[k] = clampf(c[k], 0.0f, 1.0f);
        }
        float *d = dst + 4 * i;
        if (enable) {
            const float fs = blend_factor(factors[0], c[3]), fd = blend_factor(factors[1], c[3]);
            const float as = blend_factor(factors[2], c[3]), ad = blend_factor(factors[3], c[3]);
            for (int k = 0; k < 3; k++)
                c[k] = c[k] * fs + d[k] * fd;
            c[3] = c[3] * as + d[3] * ad;
        }
        memcpy(d, c, 16);
    }
}

/* ======================================================================== */
/* deinterlacing (src/shaders/deinterlacing.c:26-370)                            */

enum { ORC_DEINT_WEAVE = 0, ORC_DEINT_BOB, ORC_DEINT_YADIF, ORC_DEINT_BWDIF };

// GET(TEX, X, Y) = textureLod(TEX, pos + pt * vec2(X, Y)) with NEAREST filtering and the MIRROR
// address mode the shader binds its textures with (:51-53): texel (x + X, y + Y), mirrored
static const float *deint_get(const float *img, int w, int h, int x, int y)
{
    return img + ((size_t) wrap(y, h, ADDR_MIRROR) * w + wrap(x, w, ADDR_MIRROR)) * 4;
}

// spatial_predictor (:131-157). a..g: row above, x - 3 .. x + 3; h..n: row below
static float yadif_spatial(const float up[7], const float dn[7], float bias)
{
    const float a = up[0], b = up[1], c = up[2], d = up[3], e = up[4], f = up[5], g = up[6];
    const float h = dn[0], i = dn[1], j = dn[2], k = dn[3], l = dn[4], m = dn[5], n = dn[6];
    float pred = (d + k) / 2.0f;
    float best = fabsf(c - j) + fabsf(d - k) + fabsf(e - l) - bias;
    float score = fabsf(b - k) + fabsf(c - l) + fabsf(d - m);
    if (score < best) {
        pred = (c + l) / 2.0f;
        best = score;
        score = fabsf(a - l) + fabsf(b - m) + fabsf(c - n);
        if (score < best) {
            pred = (b + m) / 2.0f;
            best = score;
        }
    }
    score = fabsf(d - i) + fabsf(e - j) + fabsf(f - k);
    if (score < best) {
        pred = (e + j) / 2.0f;
        best = score;
        score = fabsf(e - h) + fabsf(f - i) + fabsf(g - j);
        if (score < best) {
            pred = (f + i) / 2.0f;
            best = score;
        }
    }
    return pred;
}

// temporal_predictor (:188-216)
static float yadif_temporal(float A, float B, float C, float D, float E, float F, float G, float H,
                            float I, float J, float K, float L, float pred, int skip_spatial_check)
{
    const float p0 = (C + H) / 2.0f, p1 = F, p2 = (D + I) / 2.0f, p3 = G, p4 = (E + J) / 2.0f;
    const float tdiff0 = fabsf(D - I) / 2.0f;
    const float tdiff1 = (fabsf(A - F) + fabsf(B - G)) / 2.0f;
    const float tdiff2 = (fabsf(K - F) + fabsf(G - L)) / 2.0f;
    float diff = fmaxf(tdiff0, fmaxf(tdiff1, tdiff2));
    if (!skip_spatial_check) {
        const float maxi = fmaxf(p2 - fminf(p3, p1), fminf(p0 - p1, p4 - p3));
        const float mini = fminf(p2 - fmaxf(p3, p1), fmaxf(p0 - p1, p4 - p3));
        diff = fmaxf(diff, fmaxf(mini, -maxi));
    }
    if (pred > p2 + diff)
        pred = p2 + diff;
    if (pred < p2 - diff)
        pred = p2 - diff;
    return pred;
}

// process_bwdif / process_intra_bwdif (:270-322). cur = rows -3 -1 +1 +3; prev, next = rows -1 +1;
// prev2, next2 = rows -4 -2 0 +2 +4
static float bwdif_intra(const float cur[4])
{
    const float sp0 = 5077.0f / 8192.0f, sp1 = 981.0f / 8192.0f;
    return sp0 * (cur[1] + cur[2]) - sp1 * (cur[0] + cur[3]);
}

static float bwdif_process(const float cur[4], const float prev[2], const float next[2],
                           const float prev2[5], const float next2[5])
{
    const float lf0 = 4309.0f / 8192.0f, lf1 = 213.0f / 8192.0f;
    const float hf0 = 5570.0f / 8192.0f, hf1 = 3801.0f / 8192.0f, hf2 = 1016.0f / 8192.0f;
    const float sp0 = 5077.0f / 8192.0f, sp1 = 981.0f / 8192.0f;

    const float s = prev2[2] + next2[2];
    const float d = s / 2.0f;
    const float c = cur[1], e = cur[2];

    const float tdiff0 = fabsf(prev2[2] - next2[2]);
    const float tdiff1 = fabsf(prev[0] - c) + fabsf(prev[1] - e);
    const float tdiff2 = fabsf(next[0] - c) + fabsf(next[1] - e);
    float diff = fmaxf(tdiff0, fmaxf(tdiff1, tdiff2)) / 2.0f;
    const int still = diff == 0.0f;

    const float bs = prev2[1] + next2[1], fs = prev2[3] + next2[3];
    const float b = (bs / 2.0f) - c, f = (fs / 2.0f) - c;
    const float dc = d - c, de = d - e;
    const float mmax = fmaxf(de, fmaxf(dc, fminf(b, f)));
    const float mmin = fminf(de, fminf(dc, fmaxf(b, f)));
    diff = fmaxf(diff, fmaxf(mmin, -mmax));

    const float single = sp0 * (c + e) - sp1 * (cur[0] + cur[3]);
    float all = (hf0 * s - hf1 * (bs + fs) +
                 hf2 * (prev2[0] + next2[0] + prev2[4] + next2[4])) / 4.0f;
    all += lf0 * (c + e) - lf1 * (cur[0] + cur[3]);

    float interpol = fabsf(c - e) > tdiff0 ? all : single;
    interpol = fminf(fmaxf(interpol, d - diff), d + diff);     // clamp(x, lo, hi)
    return still ? d : interpol;
}

// pl_shader_deinterlace over whole w x h frames of decoded texels; prev / next may be NULL.
// field: 1 = top / even rows are output as they are, 2 = bottom (enum pl_field); 0 = no-op.
// The components outside comp_mask keep the shader's initial colour (0, 0, 0, 1).
ORC_API void orc_deinterlace(const float *cur, const float *prev, const float *next, int w, int h,
                             int field, int first_field, int algo, int skip_spatial_check,
                             unsigned comp_mask, float *out)
{
    if (!first_field)
        first_field = 1;
    int intra_only = algo != ORC_DEINT_YADIF;
    if (algo == ORC_DEINT_BWDIF)
        intra_only = (!prev && field == first_field) || (!next && field != first_field);
    const float *pv = !intra_only && prev ? prev : cur;
    const float *nx = !intra_only && next ? next : cur;
    const float *pv2 = field == first_field ? pv : cur;
    const float *nx2 = field == first_field ? cur : nx;
    // "%f" of 1 / 255 as the shader text carries it (:129, :134)
    const float bias = 0.003922f;

    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            float res[4];
            memcpy(res, deint_get(cur, w, h, x, y), 16);
            const int kept = field == 0 || (y % 2) == (field == 1 ? 0 : 1);
            for (int ch = 0; ch < 4 && !kept; ch++) {
#define G(IMG, X, Y) (deint_get(IMG, w, h, x + (X), y + (Y))[ch])
                switch (algo) {
                case ORC_DEINT_WEAVE:
                    break;
                case ORC_DEINT_BOB:
                    res[ch] = G(cur, 0, field == 1 ? -1 : 1);
                    break;
                case ORC_DEINT_YADIF: {
                    float up[7], dn[7];
                    for (int t = 0; t < 7; t++) {
                        up[t] = G(cur, t - 3, -1);
                        dn[t] = G(cur, t - 3, +1);
                    }
                    const float sp = yadif_spatial(up, dn, bias);
                    res[ch] = yadif_temporal(G(pv, 0, -1), G(pv, 0, 1),
                                             G(pv2, 0, -2), G(pv2, 0, 0), G(pv2, 0, 2),
                                             G(cur, 0, -1), G(cur, 0, 1),
                                             G(nx2, 0, -2), G(nx2, 0, 0), G(nx2, 0, 2),
                                             G(nx, 0, -1), G(nx, 0, 1), sp, skip_spatial_check);
                    break;
                }
                case ORC_DEINT_BWDIF: {
                    const float c4[4] = { G(cur, 0, -3), G(cur, 0, -1), G(cur, 0, 1), G(cur, 0, 3) };
                    if (intra_only) {
                        res[ch] = bwdif_intra(c4);
                        break;
                    }
                    const float p2[2] = { G(pv, 0, -1), G(pv, 0, 1) };
                    const float n2[2] = { G(nx, 0, -1), G(nx, 0, 1) };
                    float p5[5], n5[5];
                    for (int t = 0; t < 5; t++) {
                        p5[t] = G(pv2, 0, 2 * t - 4);
                        n5[t] = G(nx2, 0, 2 * t - 4);
                    }
                    res[ch] = bwdif_process(c4, p2, n2, p5, n5);
                    break;
                }
                }
#undef G
            }
            float *o = out + ((size_t) y * w + x) * 4;
            o[0] = o[1] = o[2] = 0.0f;
            o[3] = 1.0f;
            for (int ch = 0; ch < 4; ch++) {
                if (comp_mask & (1u << ch))
                    o[ch] = res[ch];
            }
        }
    }
}

ORC_API int orc_check_unorm_decode(int bits)
{
    const float d = bits == 8 ? 255.0f : 65535.0f;
    const int n = bits == 8 ? 255 : 65535;
    const float r = 1.0f / d;
    int bad = 0;
    for (int i = 0; i <= n; i++) {
        const float x = (float) i, q = x * r;
        const float q2 = fmaf(fmaf(-q, d, x), r, q);
        bad += q2 != x / d;
    }
    return bad;
}

// (int) fract((n + 0.5) / size) * size == n mod size for power-of-two `size`: the integer
// shortcut of the dither matrix index (colorops.hiph, dither_bias). Returns mismatches.
ORC_API int orc_check_dither_index(int size, int limit)
{
    int bad = 0;
    const float inv = (float) (1.0 / size);
    for (int n = 0; n < limit; n++) {
        const float f = ((float) n + 0.5f) * inv;
        const float p = f - floorf(f);
        bad += (int) (p * (float) size) != (n & (size - 1));
    }
    return bad;
}
