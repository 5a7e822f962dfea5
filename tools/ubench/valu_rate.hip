// Throughput of single VALU instructions on gfx950: cycles per wave64 instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip ; run on the GPU box.
// (Measurement aid for DESIGN.md: decides whether packed fp32 / transcendental-light
// formulations pay on CDNA4.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define ITERS 2048
#define UNROLL 16

#define KERNEL(name, ASM)                                                          \
    __global__ void name(float *out, float seed)                                   \
    {                                                                              \
        float v[UNROLL];                                                           \
        float2 p[UNROLL];                                                          \
        for (int i = 0; i < UNROLL; i++) {                                         \
            v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;                        \
            p[i] = make_float2(v[i], v[i] + 0.5f);                                 \
        }                                                                          \
        for (int it = 0; it < ITERS; it++) {                                       \
            _Pragma("unroll") for (int i = 0; i < UNROLL; i++) { ASM; }            \
        }                                                                          \
        float s = 0;                                                               \
        for (int i = 0; i < UNROLL; i++)                                           \
            s += v[i] + p[i].x + p[i].y;                                           \
        if (s == 12345.678f)                                                       \
            out[0] = s;                                                            \
    }

KERNEL(k_fma, asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i])))
KERNEL(k_mul, asm volatile("v_mul_f32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_pkfma, asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i])))
KERNEL(k_pkmul, asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(p[i])))
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_log, asm volatile("v_log_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_rcp, asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_sqrt, asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_rsq, asm volatile("v_rsq_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_sin, asm volatile("v_sin_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_cvt, asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[i])))
KERNEL(k_max, asm volatile("v_max_f32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(v[i])))
KERNEL(k_divfix, asm volatile("v_div_fixup_f32 %0, %0, %0, %0" : "+v"(v[i])))
KERNEL(k_floor, asm volatile("v_floor_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_mix, asm volatile("v_fma_mix_f32 %0, %0, %0, %0" : "+v"(v[i])))
KERNEL(k_exp_fma, asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(v[i]), "+v"(p[i].x)))

typedef void (*kern)(float *, float);

int main()
{
    float *out;
    hipMalloc(&out, 64);
    int cus = 0, khz = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("CUs %d, nominal clock %d kHz\n", cus, khz);
    struct { const char *name; kern k; int per; } ks[] = {
        {"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_pk_fma_f32", k_pkfma, 1},
        {"v_pk_mul_f32", k_pkmul, 1}, {"v_exp_f32", k_exp, 1}, {"v_log_f32", k_log, 1},
        {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1}, {"v_rsq_f32", k_rsq, 1},
        {"v_sin_f32", k_sin, 1}, {"v_cvt_f32_u32", k_cvt, 1}, {"v_max_f32", k_max, 1},
        {"v_cndmask_b32", k_cndmask, 1}, {"v_div_fixup_f32", k_divfix, 1},
        {"v_floor_f32", k_floor, 1}, {"v_fma_mix_f32", k_mix, 1},
        {"v_exp_f32 + v_fma_f32 (pair)", k_exp_fma, 1},
    };
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
        const int blocks = cus * waves_per_simd;    // 256 threads = 4 waves = one per SIMD
        printf("---- %d wave(s) per SIMD\n", waves_per_simd);
        for (auto &e : ks) {
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            e.k<<<blocks, 256>>>(out, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(a);
            e.k<<<blocks, 256>>>(out, 1.0f);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            const double instr_per_simd = (double) waves_per_simd * ITERS * UNROLL;
            const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
            printf("%-32s %8.3f ms  %6.2f cycles/instr/SIMD @2.4GHz\n", e.name, ms, cyc);
        }
    }
    return 0;
}
