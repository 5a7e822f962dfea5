// Instruction budgets of the colour-map building blocks: each kernel wraps one device function
// of csrc/hip (one pixel per lane, operands from memory); tools/isa_count.py disassembles the
// object and prints VALU instructions per kernel by issue class (profiles/r02_valu_rate.txt).
#include <hip/hip_runtime.h>
#include "plh_device.h"
#include "colorops.hiph"

#define K(name) extern "C" __global__ void name(const float4_t *in, float4_t *out, const plh_op *ops)
K(k_base) { float4_t c = in[threadIdx.x]; out[threadIdx.x] = c; }
K(k_eotf3) {
    float4_t c = in[threadIdx.x];
    const pq_consts k = { ops[0].f[10], ops[0].f[13], ops[0].f[14], ops[0].f[9], ops[0].f[13] };
    c.x = pq_eotf_acc1(c.x, k); c.y = pq_eotf_acc1(c.y, k); c.z = pq_eotf_acc1(c.z, k);
    out[threadIdx.x] = c;
}
K(k_oetf3) {
    float4_t c = in[threadIdx.x];
    const pq_consts k = { ops[0].f[10], ops[0].f[13], ops[0].f[14], ops[0].f[9], ops[0].f[13] };
    c.x = pq_oetf_acc1(c.x, k); c.y = pq_oetf_acc1(c.y, k); c.z = pq_oetf_acc1(c.z, k);
    out[threadIdx.x] = c;
}
K(k_atan2) { float4_t c = in[threadIdx.x]; c.x = atan2_poly(c.x, c.y); out[threadIdx.x] = c; }
K(k_gamut) {
    float4_t c = in[threadIdx.x];
    const float idx[3] = { c.x, c.y, c.z };
    float o[3];
    gamut_lookup(ops[0].ptr, ops[0].i0, ops[0].i1, ops[0].i2, idx, o);
    c.x = o[0]; c.y = o[1]; c.z = o[2];
    out[threadIdx.x] = c;
}
K(k_tone) { float4_t c = in[threadIdx.x]; c.x = tone_curve1(ops[0], c.x); out[threadIdx.x] = c; }
K(k_cm1) { float4_t c[1] = { in[threadIdx.x] }; cm_fused<1>(c, ops[0], &ops[1], &ops[2], ops[3]); out[threadIdx.x] = c[0]; }
K(k_cm2) {
    float4_t c[2] = { in[threadIdx.x], in[threadIdx.x + 64] };
    cm_fused<2>(c, ops[0], &ops[1], &ops[2], ops[3]);
    out[threadIdx.x] = c[0]; out[threadIdx.x + 64] = c[1];
}
K(k_lin) { float4_t c = in[threadIdx.x]; op_linearize(c, ops[0]); out[threadIdx.x] = c; }
K(k_delin) { float4_t c = in[threadIdx.x]; op_delinearize(c, ops[0]); out[threadIdx.x] = c; }
K(k_dither) { float4_t c = in[threadIdx.x]; frag_t fc = { c.w, c.w + 1.0f }; op_dither<true>(c, ops[0], fc); out[threadIdx.x] = c; }
K(k_affine) { float4_t c = in[threadIdx.x]; op_affine(c, ops[0].f); out[threadIdx.x] = c; }
