// Which work shape suits a 2x bilinear upscale of 8-byte texels on gfx950? The arithmetic of
// k_bilinear_fast (decode u16 -> f32, separable lerps with the 0.25 / 0.75 weights of an exact 2x,
// encode to unorm16) in several lane-to-pixel mappings, against the bare data movement
// (hbm_rate.hip: 12.7 us). 1080p -> 4K, rgba16, alpha forced to 1.
// Build: hipcc --offload-arch=gfx950 -O2 -o bilerp_shapes.bin bilerp_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

#define SW 1920
#define SH 1080

struct rgb { float r, g, b; };

__device__ __forceinline__ rgb decode(u32x2 t)
{
    const float k = 1.0f / 65535.0f;
    return { (float) (t.x & 0xffff) * k, (float) (t.x >> 16) * k, (float) (t.y & 0xffff) * k };
}
__device__ __forceinline__ rgb mix(rgb a, rgb b, float w)
{
    return { __builtin_fmaf(b.r - a.r, w, a.r), __builtin_fmaf(b.g - a.g, w, a.g), __builtin_fmaf(b.b - a.b, w, a.b) };
}
__device__ __forceinline__ u32x2 encode(rgb c)
{
    const u16x2 lo = __builtin_amdgcn_cvt_pknorm_u16(c.r, c.g), hi = __builtin_amdgcn_cvt_pknorm_u16(c.b, 1.0f);
    return { (unsigned) lo.x | ((unsigned) lo.y << 16), (unsigned) hi.x | ((unsigned) hi.y << 16) };
}
__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// the weights come from memory so that the compiler cannot fold them
struct params { float wx0, wx1, wy0, wy1; };

// A: one 2x2 output cell per lane, four 8-byte loads (the shape of k_bilinear_fast<.., ITERS = 1>)
template <int OFF>    // OFF = 1: every row pair starts 8 bytes off a 16-byte boundary (the cell phase of a 2x upscale)
__global__ __launch_bounds__(256) void k_cell4(u32x4 *dst, const u32x2 *src, params p)
{
    const int cx = blockIdx.x * 64 + (threadIdx.x & 63), cy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cx >= SW || cy >= SH) return;
    const int x0 = clampi(cx - 1, SW - 1), x1 = cx, y0 = clampi(cy - 1, SH - 1), y1 = cy;
    const rgb a = decode(src[(size_t) y0 * SW + x0]), b = decode(src[(size_t) y0 * SW + x1]);
    const rgb c = decode(src[(size_t) y1 * SW + x0]), d = decode(src[(size_t) y1 * SW + x1]);
    const rgb t0 = mix(a, b, p.wx0), t1 = mix(a, b, p.wx1), u0 = mix(c, d, p.wx0), u1 = mix(c, d, p.wx1);
    const u32x2 o00 = encode(mix(t0, u0, p.wy0)), o01 = encode(mix(t1, u1, p.wy0));
    const u32x2 o10 = encode(mix(t0, u0, p.wy1)), o11 = encode(mix(t1, u1, p.wy1));
    u32x4 *r0 = (u32x4 *) ((u32x2 *) (dst + (size_t) (2 * cy) * SW + cx) + OFF), *r1 = r0 + SW;
    __builtin_nontemporal_store((u32x4) { o00.x, o00.y, o01.x, o01.y }, r0);
    __builtin_nontemporal_store((u32x4) { o10.x, o10.y, o11.x, o11.y }, r1);
}

// B: the same cell, the two texels of a row in one (8-byte aligned) 16-byte load
__global__ __launch_bounds__(256) void k_cell2(u32x4 *dst, const u32x2 *src, params p)
{
    const int cx = blockIdx.x * 64 + (threadIdx.x & 63), cy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (cx >= SW || cy >= SH) return;
    const int x0 = max(cx - 1, 0), y0 = clampi(cy - 1, SH - 1), y1 = cy;   // (left edge: a real kernel patches lane 0)
    const u32x4 q0 = *(const u32x4 *) (src + (size_t) y0 * SW + x0), q1 = *(const u32x4 *) (src + (size_t) y1 * SW + x0);
    const rgb a = decode((u32x2) { q0.x, q0.y }), b = decode((u32x2) { q0.z, q0.w });
    const rgb c = decode((u32x2) { q1.x, q1.y }), d = decode((u32x2) { q1.z, q1.w });
    const rgb t0 = mix(a, b, p.wx0), t1 = mix(a, b, p.wx1), u0 = mix(c, d, p.wx0), u1 = mix(c, d, p.wx1);
    const u32x2 o00 = encode(mix(t0, u0, p.wy0)), o01 = encode(mix(t1, u1, p.wy0));
    const u32x2 o10 = encode(mix(t0, u0, p.wy1)), o11 = encode(mix(t1, u1, p.wy1));
    u32x4 *r0 = dst + (size_t) (2 * cy) * SW + cx, *r1 = r0 + SW;
    __builtin_nontemporal_store((u32x4) { o00.x, o00.y, o01.x, o01.y }, r0);
    __builtin_nontemporal_store((u32x4) { o10.x, o10.y, o11.x, o11.y }, r1);
}

// C: a lane owns a column of ROWS cells: texel rows are loaded once and carried down (each source
// row feeds two cell rows), ROWS + 1 row loads for ROWS cells
template <int ROWS>
__global__ __launch_bounds__(256) void k_column(u32x4 *dst, const u32x2 *src, params p)
{
    const int cx = blockIdx.x * 64 + (threadIdx.x & 63), cyb = (blockIdx.y * 4 + (threadIdx.x >> 6)) * ROWS;
    if (cx >= SW || cyb >= SH) return;
    const int x0 = max(cx - 1, 0);
    u32x4 q[ROWS + 1];
#pragma unroll
    for (int j = 0; j <= ROWS; j++)
        q[j] = *(const u32x4 *) (src + (size_t) clampi(cyb - 1 + j, SH - 1) * SW + x0);
    rgb t0, t1;
    {
        const rgb a = decode((u32x2) { q[0].x, q[0].y }), b = decode((u32x2) { q[0].z, q[0].w });
        t0 = mix(a, b, p.wx0); t1 = mix(a, b, p.wx1);
    }
#pragma unroll
    for (int j = 0; j < ROWS; j++) {
        const rgb c = decode((u32x2) { q[j + 1].x, q[j + 1].y }), d = decode((u32x2) { q[j + 1].z, q[j + 1].w });
        const rgb u0 = mix(c, d, p.wx0), u1 = mix(c, d, p.wx1);
        const u32x2 o00 = encode(mix(t0, u0, p.wy0)), o01 = encode(mix(t1, u1, p.wy0));
        const u32x2 o10 = encode(mix(t0, u0, p.wy1)), o11 = encode(mix(t1, u1, p.wy1));
        if (cyb + j < SH) {
            u32x4 *r0 = dst + (size_t) (2 * (cyb + j)) * SW + cx, *r1 = r0 + SW;
            __builtin_nontemporal_store((u32x4) { o00.x, o00.y, o01.x, o01.y }, r0);
            __builtin_nontemporal_store((u32x4) { o10.x, o10.y, o11.x, o11.y }, r1);
        }
        t0 = u0; t1 = u1;
    }
}

static float time_us(void (*launch)(int), int reps)
{
    hipEvent_t a, b;
    (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    for (int i = 0; i < 4; i++) launch(i);
    (void) hipDeviceSynchronize();
    (void) hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) launch(i);
    (void) hipEventRecord(b, 0);
    (void) hipEventSynchronize(b);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

#define POOL 8
static u32x2 *g_src[POOL];
static u32x4 *g_dst[POOL];
static const params g_p = { 0.75f, 0.25f, 0.75f, 0.25f };

int main()
{
    for (int i = 0; i < POOL; i++) {
        (void) hipMalloc(&g_src[i], (size_t) SW * SH * 8 + 64);
        (void) hipMalloc(&g_dst[i], (size_t) SW * SH * 32 + 64);
        (void) hipMemset(g_src[i], 0x3c + i, (size_t) SW * SH * 8 + 64);
    }
    struct { const char *name; void (*fn)(int); } ks[] = {
        {"A  cell, 4 x 8-byte loads", [](int i) { k_cell4<0><<<dim3(30, 270), 256>>>(g_dst[i % POOL], g_src[i % POOL], g_p); }},
        {"A' same, stores 8 B off 16", [](int i) { k_cell4<1><<<dim3(30, 270), 256>>>(g_dst[i % POOL], g_src[i % POOL], g_p); }},
        {"B  cell, 2 x 16-byte loads", [](int i) { k_cell2<<<dim3(30, 270), 256>>>(g_dst[i % POOL], g_src[i % POOL], g_p); }},
        {"C2 column of 2 cells", [](int i) { k_column<2><<<dim3(30, 135), 256>>>(g_dst[i % POOL], g_src[i % POOL], g_p); }},
        {"C4 column of 4 cells", [](int i) { k_column<4><<<dim3(30, 68), 256>>>(g_dst[i % POOL], g_src[i % POOL], g_p); }},
        {"C8 column of 8 cells", [](int i) { k_column<8><<<dim3(30, 34), 256>>>(g_dst[i % POOL], g_src[i % POOL], g_p); }},
    };
    const double bytes = (double) SW * SH * 8 * 5;
    for (int rep = 0; rep < 2; rep++)
        for (auto &k : ks) {
            const float us = time_us(k.fn, 80);
            printf("%-30s %7.2f us  %7.1f GB/s  %4.1f %% of 8 TB/s\n", k.name, us, bytes / us / 1e3, bytes / us / 1e3 / 80);
        }
    return 0;
}
