// What does a launch cost the GPU timeline between two dependent kernels of one stream? The same
// 12 us streaming kernel (1080p -> 4K "expand") launched back to back 200 times: on the null stream
// or a non-blocking one, with 24 bytes of arguments or a 2.5 KB struct by value (the size of plh_pass),
// with and without an event recorded after every launch. Time per launch = kernel + gap.
// Build: hipcc --offload-arch=gfx950 -O2 -o launch_gap.bin launch_gap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstring>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define SW 1920
#define SH 1080

struct small_args { u32x4 *dst; const u32x2 *src; int pad[2]; };
struct big_args { u32x4 *dst; const u32x2 *src; int pad[636]; };     // 2560 bytes

template <typename A>
__global__ __launch_bounds__(256) void k_expand(const A a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= SW || y >= SH)
        return;
    const u32x2 t = a.src[(size_t) y * SW + x];
    const u32x4 o = { t.x, t.y, t.x + (unsigned) a.pad[1], t.y + 1 };
    u32x4 *r0 = a.dst + (size_t) (2 * y) * SW + x;
    __builtin_nontemporal_store(o, r0);
    __builtin_nontemporal_store(o, r0 + SW);
}

#define POOL 8
static u32x2 *g_src[POOL];
static u32x4 *g_dst[POOL];

// mode 0: plain launches; 1: hipEventRecord behind each; 2: the event attached to the launch itself
// (hipExtLaunchKernelGGL's stop event: the dispatch packet's own completion signal)
template <typename A>
static double run(hipStream_t s, int reps, int mode)
{
    hipEvent_t a, b, e[8];
    (void) hipEventCreate(&a); (void) hipEventCreate(&b);
    for (auto &x : e) (void) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    A args;
    memset(&args, 0, sizeof(args));
    for (int i = 0; i < reps + 8; i++) {
        if (i == 8) {
            (void) hipStreamSynchronize(s);
            (void) hipEventRecord(a, s);
        }
        args.dst = g_dst[i % POOL];
        args.src = g_src[i % POOL];
        if (mode == 2)
            hipExtLaunchKernelGGL(k_expand<A>, dim3(30, 270), dim3(256), 0, s, nullptr, e[i % 8], 0, args);
        else
            hipLaunchKernelGGL(k_expand<A>, dim3(30, 270), dim3(256), 0, s, args);
        if (mode == 1)
            (void) hipEventRecord(e[i % 8], s);
    }
    (void) hipEventRecord(b, s);
    (void) hipEventSynchronize(b);
    float ms = 0;
    (void) hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}

int main()
{
    for (int i = 0; i < POOL; i++) {
        (void) hipMalloc(&g_src[i], (size_t) SW * SH * 8);
        (void) hipMalloc(&g_dst[i], (size_t) SW * SH * 32);
        (void) hipMemset(g_src[i], 0x3c + i, (size_t) SW * SH * 8);
    }
    hipStream_t nb;
    (void) hipStreamCreateWithFlags(&nb, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; rep++) {
        printf("null stream,   24 B args            %7.2f us per launch\n", run<small_args>(0, 200, 0));
        printf("null stream,   2560 B args          %7.2f us per launch\n", run<big_args>(0, 200, 0));
        printf("own stream,    24 B args            %7.2f us per launch\n", run<small_args>(nb, 200, 0));
        printf("own stream,    2560 B args          %7.2f us per launch\n", run<big_args>(nb, 200, 0));
        printf("own stream,    2560 B args + event  %7.2f us per launch\n", run<big_args>(nb, 200, 1));
        printf("own stream,    2560 B args, stop event on the launch %7.2f us per launch\n", run<big_args>(nb, 200, 2));
    }
    return 0;
}
