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

// integer / select / conversion instructions the colour and debanding kernels are full of
KERNEL(k_cnd_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %0, s[2:3]" : "+v"(v[i])))
KERNEL(k_mullo, asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_mulhi, asm volatile("v_mul_hi_u32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_mul24, asm volatile("v_mul_u32_u24 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_mad24, asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(v[i])))
KERNEL(k_addu, asm volatile("v_add_u32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_add3, asm volatile("v_add3_u32 %0, %0, %0, %0" : "+v"(v[i])))
KERNEL(k_lshladd, asm volatile("v_lshl_add_u32 %0, %0, 3, %0" : "+v"(v[i])))
KERNEL(k_and, asm volatile("v_and_b32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_xor, asm volatile("v_xor_b32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_lshr, asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(v[i])))
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(v[i])))
KERNEL(k_mini, asm volatile("v_min_i32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_med3, asm volatile("v_med3_f32 %0, %0, %0, %0" : "+v"(v[i])))
KERNEL(k_addf, asm volatile("v_add_f32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_fmac, asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_fmaak, asm volatile("v_fmaak_f32 %0, %0, %0, 0x3f000000" : "+v"(v[i])))
KERNEL(k_cmp, asm volatile("v_cmp_gt_f32 vcc, %0, %0" : "+v"(v[i]) : : "vcc"))
KERNEL(k_cvt_sdwa, asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v[i])))
KERNEL(k_or_sdwa, asm volatile("v_or_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(v[i])))
KERNEL(k_cvti, asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_cvtf16, asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_cvtf32h, asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[i])))
KERNEL(k_pkrtz, asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_pknorm, asm volatile("v_cvt_pknorm_u16_f32 %0, %0, %0" : "+v"(v[i])))
KERNEL(k_rndne, asm volatile("v_rndne_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_fract, asm volatile("v_fract_f32 %0, %0" : "+v"(v[i])))
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %0" : "+v"(v[i])))
KERNEL(k_ldexp, asm volatile("v_ldexp_f32 %0, %0, 1" : "+v"(v[i])))
KERNEL(k_mad64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(p[i]), "+v"(v[i]) : : "vcc"))
KERNEL(k_pkfma16, asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(v[i])))
KERNEL(k_dot2, asm volatile("v_dot2c_f32_f16 %0, %0, %0" : "+v"(v[i])))

// compare + select as the compiler emits it (VOP2 select reading vcc) vs through an SGPR pair
KERNEL(k_cmp_cnd_vcc, asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]), "+v"(p[i].x) : : "vcc"))
KERNEL(k_cmp_cnd_sgpr, asm volatile("v_cmp_gt_f32_e64 s[4:5], %0, %1\n v_cndmask_b32_e64 %0, %0, %1, s[4:5]" : "+v"(v[i]), "+v"(p[i].x) : : "s4", "s5"))
KERNEL(k_cnd_vcc_2src, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]), "+v"(p[i].x)))

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
        {"v_cndmask_b32_e64 (sgpr mask)", k_cnd_e64, 1}, {"v_mul_lo_u32", k_mullo, 1},
        {"v_mul_hi_u32", k_mulhi, 1}, {"v_mul_u32_u24", k_mul24, 1}, {"v_mad_u32_u24", k_mad24, 1},
        {"v_add_u32", k_addu, 1}, {"v_add3_u32", k_add3, 1}, {"v_lshl_add_u32", k_lshladd, 1},
        {"v_and_b32", k_and, 1}, {"v_xor_b32", k_xor, 1}, {"v_lshrrev_b32", k_lshr, 1},
        {"v_bfe_u32", k_bfe, 1}, {"v_min_i32", k_mini, 1}, {"v_med3_f32", k_med3, 1},
        {"v_add_f32", k_addf, 1}, {"v_fmac_f32", k_fmac, 1}, {"v_fmaak_f32", k_fmaak, 1},
        {"v_cmp_gt_f32 (vcc)", k_cmp, 1}, {"v_cvt_f32_u32_sdwa", k_cvt_sdwa, 1},
        {"v_or_b32_sdwa", k_or_sdwa, 1}, {"v_cvt_i32_f32", k_cvti, 1},
        {"v_cvt_f16_f32", k_cvtf16, 1}, {"v_cvt_f32_f16", k_cvtf32h, 1},
        {"v_cvt_pkrtz_f16_f32", k_pkrtz, 1}, {"v_cvt_pknorm_u16_f32", k_pknorm, 1},
        {"v_rndne_f32", k_rndne, 1}, {"v_fract_f32", k_fract, 1}, {"v_mov_b32", k_mov, 1},
        {"v_ldexp_f32", k_ldexp, 1}, {"v_mad_u64_u32", k_mad64, 1},
        {"v_pk_fma_f16", k_pkfma16, 1}, {"v_dot2c_f32_f16", k_dot2, 1},
        {"v_cmp + v_cndmask (vcc) pair", k_cmp_cnd_vcc, 1},
        {"v_cmp + v_cndmask (sgpr) pair", k_cmp_cnd_sgpr, 1},
        {"v_cndmask_b32 vcc, 2 sources", k_cnd_vcc_2src, 1},
    };
    for (int waves_per_simd = 2; waves_per_simd <= 4; waves_per_simd *= 2) {
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
