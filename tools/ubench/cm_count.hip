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
    gamut_lookup(ops[0], idx, o);
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
// the whole map pass of configs[3] as straight-line code (no op interpreter): two pixels per lane
// from an rgba16hf source, colour map, BT.1886 delinearize, uniform scale, rgba16 store
K(k_map_line) {
    const uint4 v = ((const uint4 *) in)[threadIdx.x];
    const uint32_t q[4] = { v.x, v.y, v.z, v.w };
    float4_t c[2];
    for (int i = 0; i < 2; i++)
        c[i] = { plh_h2f(q[2 * i] & 0xffff), plh_h2f(q[2 * i] >> 16), plh_h2f(q[2 * i + 1] & 0xffff), plh_h2f(q[2 * i + 1] >> 16) };
    cm_fused<2>(c, ops[0], &ops[1], &ops[2], ops[3]);
    uint32_t o[4];
    for (int i = 0; i < 2; i++) {
        const float *f = ops[4].f;   // BT.1886 with both flags, as configs[3] records it
        c[i].x = delin1(fmaxf(f[0] * c[i].x + f[1], 0.0f), TRC_BT_1886, f);
        c[i].y = delin1(fmaxf(f[0] * c[i].y + f[1], 0.0f), TRC_BT_1886, f);
        c[i].z = delin1(fmaxf(f[0] * c[i].z + f[1], 0.0f), TRC_BT_1886, f);
        c[i].x *= ops[5].f[0]; c[i].y *= ops[5].f[0]; c[i].z *= ops[5].f[0]; c[i].w *= ops[5].f[0];
        o[2 * i] = plh_unorm16x2(c[i].x, c[i].y);
        o[2 * i + 1] = plh_unorm16x2(c[i].z, c[i].w);
    }
    ((uint4 *) out)[threadIdx.x] = make_uint4(o[0], o[1], o[2], o[3]);
}
// ... with the PQ linearize in front (the recorded chain of configs[3]'s map pass: LINEARIZE
// RGB2IPT TONE_MAP GAMUT_LUT IPT2RGB DELINEARIZE SCALE), NP pixels per lane
template <int NP> DEV void map_chain(const uint32_t *q, uint32_t *o, const plh_op *ops)
{
    float4_t c[NP];
    for (int i = 0; i < NP; i++)
        c[i] = { plh_h2f(q[2 * i] & 0xffff), plh_h2f(q[2 * i] >> 16), plh_h2f(q[2 * i + 1] & 0xffff), plh_h2f(q[2 * i + 1] >> 16) };
    for (int i = 0; i < NP; i++) {
        const float *f = ops[6].f;
        c[i].x = f[0] * lin1(fmaxf(c[i].x, 0.0f), TRC_PQ, f) + f[1];
        c[i].y = f[0] * lin1(fmaxf(c[i].y, 0.0f), TRC_PQ, f) + f[1];
        c[i].z = f[0] * lin1(fmaxf(c[i].z, 0.0f), TRC_PQ, f) + f[1];
    }
    cm_fused<NP>(c, ops[0], &ops[1], &ops[2], ops[3]);
    for (int i = 0; i < NP; i++) {
        const float *f = ops[4].f;
        c[i].x = delin1(fmaxf(f[0] * c[i].x + f[1], 0.0f), TRC_BT_1886, f);
        c[i].y = delin1(fmaxf(f[0] * c[i].y + f[1], 0.0f), TRC_BT_1886, f);
        c[i].z = delin1(fmaxf(f[0] * c[i].z + f[1], 0.0f), TRC_BT_1886, f);
        c[i].x *= ops[5].f[0]; c[i].y *= ops[5].f[0]; c[i].z *= ops[5].f[0]; c[i].w *= ops[5].f[0];
        o[2 * i] = plh_unorm16x2(c[i].x, c[i].y);
        o[2 * i + 1] = plh_unorm16x2(c[i].z, c[i].w);
    }
}
K(k_chain1) {
    const uint2 v = ((const uint2 *) in)[threadIdx.x];
    const uint32_t q[2] = { v.x, v.y };
    uint32_t o[2];
    map_chain<1>(q, o, ops);
    ((uint2 *) out)[threadIdx.x] = make_uint2(o[0], o[1]);
}
K(k_chain2) {
    const uint4 v = ((const uint4 *) in)[threadIdx.x];
    const uint32_t q[4] = { v.x, v.y, v.z, v.w };
    uint32_t o[4];
    map_chain<2>(q, o, ops);
    ((uint4 *) out)[threadIdx.x] = make_uint4(o[0], o[1], o[2], o[3]);
}
