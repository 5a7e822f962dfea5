// What the memory system of an MI355X delivers to plain streaming kernels of the shapes the
// renderer's passes have: fill (write only), sum (read only), copy, and the 1:4 "expand" of a 2x
// upscale (read N texels, write 4N) -- the practical ceilings the roofline fractions in DESIGN.md
// are to be read against (the 8 TB/s figure is the HBM3E interface, not what a kernel sees).
// Build: hipcc --offload-arch=gfx950 -O2 -o hbm_rate.bin hbm_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <bool NT>
__global__ __launch_bounds__(256) void k_fill(u32x4 *dst, size_t n, unsigned v)
{
    const u32x4 val = { v, v + 1, v + 2, v + 3 };
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        if (NT)
            __builtin_nontemporal_store(val, dst + i);
        else
            dst[i] = val;
    }
}

__global__ __launch_bounds__(256) void k_sum(const u32x4 *src, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        const u32x4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        out[0] = acc;
}

template <bool NT>
__global__ __launch_bounds__(256) void k_copy(u32x4 *dst, const u32x4 *src, size_t n)
{
    for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) {
        const u32x4 v = src[i];
        if (NT)
            __builtin_nontemporal_store(v, dst + i);
        else
            dst[i] = v;
    }
}

// 2x upscale shape: a lane reads one 8-byte texel and writes a 2x2 block (two 16-byte row pieces)
template <bool NT>
__global__ __launch_bounds__(256) void k_expand(u32x4 *dst, const u32x2 *src, int sw, int sh)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= sw || y >= sh)
        return;
    const u32x2 t = src[(size_t) y * sw + x];
    const u32x4 o = { t.x, t.y, t.x + 1, t.y + 1 };
    u32x4 *r0 = dst + (size_t) (2 * y) * sw + x, *r1 = r0 + sw;
    if (NT) {
        __builtin_nontemporal_store(o, r0);
        __builtin_nontemporal_store(o, r1);
    } else {
        *r0 = o;
        *r1 = o;
    }
}

static float time_us(void (*launch)(void *), void *ctx, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++)
        launch(ctx);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++)
        launch(ctx);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

struct ctx { u32x4 *a, *b; unsigned *out; size_t n; int blocks; int sw, sh; };

int main()
{
    // sizes of the passes: a 4K rgba16 plane = 66.4 MB, a 1080p one = 16.6 MB; the pool rotates
    // through > 256 MB so that no launch finds its data in the Infinity Cache
    const size_t plane4k = (size_t) 3840 * 2160 * 8, plane1080 = (size_t) 1920 * 1080 * 8;
    const int pool = 6;
    ctx c[pool];
    for (int i = 0; i < pool; i++) {
        hipMalloc(&c[i].a, plane4k); hipMalloc(&c[i].b, plane4k); hipMalloc(&c[i].out, 64);
        hipMemset(c[i].a, 1, plane4k); hipMemset(c[i].b, 2, plane4k);
        c[i].sw = 1920; c[i].sh = 1080;
    }
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d; bytes per launch: fill/sum 66.4 MB, copy 132.7 MB, expand 16.6 + 66.4 MB\n", cus);
    static int turn;
    static ctx *cc;
    cc = c;
    for (int bpc = 2; bpc <= 16; bpc *= 2) {
        for (int i = 0; i < pool; i++) { c[i].blocks = cus * bpc; c[i].n = plane4k / 16; }
        struct { const char *name; void (*fn)(void *); double bytes; } ks[] = {
            {"fill, plain stores", [](void *) { ctx &x = cc[turn++ % pool]; k_fill<false><<<x.blocks, 256>>>(x.a, x.n, 7); }, (double) plane4k},
            {"fill, nontemporal", [](void *) { ctx &x = cc[turn++ % pool]; k_fill<true><<<x.blocks, 256>>>(x.a, x.n, 7); }, (double) plane4k},
            {"sum (read only)", [](void *) { ctx &x = cc[turn++ % pool]; k_sum<<<x.blocks, 256>>>(x.a, x.n, x.out); }, (double) plane4k},
            {"copy, plain stores", [](void *) { ctx &x = cc[turn++ % pool]; k_copy<false><<<x.blocks, 256>>>(x.b, x.a, x.n); }, 2.0 * plane4k},
            {"copy, nontemporal", [](void *) { ctx &x = cc[turn++ % pool]; k_copy<true><<<x.blocks, 256>>>(x.b, x.a, x.n); }, 2.0 * plane4k},
        };
        printf("---- grid-stride, %d blocks per CU\n", bpc);
        for (auto &k : ks) {
            const float us = time_us(k.fn, nullptr, 60);
            printf("%-24s %8.2f us  %7.1f GB/s\n", k.name, us, k.bytes / us / 1e3);
        }
    }
    printf("---- expand 1080p -> 4K (one 2x2 block per lane, 64x4 lanes per block)\n");
    struct { const char *name; void (*fn)(void *); } es[] = {
        {"expand, plain stores", [](void *) { ctx &x = cc[turn++ % pool]; k_expand<false><<<dim3(30, 270), 256>>>(x.b, (const u32x2 *) x.a, 1920, 1080); }},
        {"expand, nontemporal", [](void *) { ctx &x = cc[turn++ % pool]; k_expand<true><<<dim3(30, 270), 256>>>(x.b, (const u32x2 *) x.a, 1920, 1080); }},
    };
    for (auto &k : es) {
        const float us = time_us(k.fn, nullptr, 60);
        printf("%-24s %8.2f us  %7.1f GB/s\n", k.name, us, (double) (plane1080 + plane4k) / us / 1e3);
    }
    return 0;
}
