// Issue rate of the f16 MFMA forms on gfx950: cycles per wave-instruction per SIMD, one wave per SIMD
// (independent accumulators) and back-to-back on ONE accumulator. Decides whether the 12 live columns
// of k_polar_mxd's second K block are worth a 16-column MFMA (v_mfma_f32_16x16x16_f16) instead of
// the 32-column one. Build: hipcc --offload-arch=gfx950 -O2 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define ITERS 4096

template <int MODE>
__global__ void k(float *out, float seed)
{
    h8 a8, b8;
    h4 a4, b4;
    for (int i = 0; i < 8; i++) { a8[i] = (_Float16) (seed + i); b8[i] = (_Float16) (seed - i); }
    for (int i = 0; i < 4; i++) { a4[i] = a8[i]; b4[i] = b8[i]; }
    f4 acc[4] = { (f4) (0.0f), (f4) (0.0f), (f4) (0.0f), (f4) (0.0f) };
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int d = MODE & 1 ? 0 : u;     // odd modes: one accumulator, dependent chain
            if (MODE < 2)
                acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[d], 0, 0, 0);
            else
                acc[d] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[d], 0, 0, 0);
        }
    }
    float s = 0;
    for (int u = 0; u < 4; u++)
        s += acc[u][0] + acc[u][3];
    if (s == 12345.678f)
        out[0] = s;
}

template <int MODE> static void run(const char *name, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // 4 waves per CU = one per SIMD, every CU busy
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double cyc = ms * 1e-3 * khz * 1e3 / (ITERS * 4.0);
    printf("%-34s %7.2f cycles per MFMA per SIMD (at the nominal %d MHz; %.3f ms)\n", name, cyc, khz / 1000, ms);
}

int main()
{
    float *out;
    hipMalloc(&out, 4);
    run<0>("16x16x32_f16, 4 accumulators", out);
    run<1>("16x16x32_f16, 1 accumulator", out);
    run<2>("16x16x16_f16, 4 accumulators", out);
    run<3>("16x16x16_f16, 1 accumulator", out);
    return 0;
}
