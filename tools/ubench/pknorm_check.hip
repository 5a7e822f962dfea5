// Is v_cvt_pknorm_u16_f32 bit-identical to rint(clamp(x, 0, 1) * 65535) (float multiply,
// round-half-even) for EVERY float? Exhaustive over all 2^32 bit patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

template <int MODE> __global__ void check(unsigned long long *mismatch, uint32_t *first)
{
    const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 4096u;
    unsigned long long bad = 0;
    for (uint32_t i = 0; i < 4096u; i++) {
        const uint32_t bits = base + i;
        const float x = __uint_as_float(bits);
        const uint32_t ref_f = (uint32_t) __builtin_rintf(fminf(fmaxf(x, 0.0f), 1.0f) * 65535.0f);
        // exact product (fits a double), one rounding to nearest-even
        const uint32_t ref = MODE ? (uint32_t) __builtin_rint((double) fminf(fmaxf(x, 0.0f), 1.0f) * 65535.0)
                                  : ref_f;
        const ushort2_t pk = __builtin_amdgcn_cvt_pknorm_u16(x, x);
        if (pk.x != ref || pk.y != ref) {
            if (!bad && atomicCAS(first, 0u, bits ? bits : 1u) == 0u) {}
            bad++;
        }
    }
    if (bad)
        atomicAdd(mismatch, bad);
}

int main()
{
    unsigned long long *d, h = 0;
    uint32_t *f, hf = 0;
    hipMalloc(&d, 8); hipMalloc(&f, 4);
    hipMemset(d, 0, 8); hipMemset(f, 0, 4);
    for (int mode = 0; mode < 2; mode++) {
        hipMemset(d, 0, 8); hipMemset(f, 0, 4);
        if (mode) check<1><<<4096, 256>>>(d, f); else check<0><<<4096, 256>>>(d, f);   // 2^32 values
        hipDeviceSynchronize();
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hf, f, 4, hipMemcpyDeviceToHost);
        float x; memcpy(&x, &hf, 4);
        printf("v_cvt_pknorm_u16_f32 vs %s: %llu mismatches of 2^32",
               mode ? "rint((double) clamp(x) * 65535.0) [exact product, RNE]"
                    : "rintf(clamp(x) * 65535.0f) [float product, RNE]", h);
        if (h) printf(", e.g. bits 0x%08x (%.9g)", hf, x);
        printf("\n");
    }
    return 0;
}
