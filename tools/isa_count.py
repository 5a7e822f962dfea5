#!/usr/bin/env python3
"""VALU instructions per kernel of a HIP source, by issue class on gfx950 (cycles per wave64
instruction per SIMD measured by tools/ubench/valu_rate.hip, profiles/r02_valu_rate.txt):
A 2.5 (fma/mul/add/and/or/shift/mov), B 4.4 (cvt, min/max, floor, cndmask, cmp, mul_lo, fma_mix,
med3, bfe, add3, lshl_add ...), C 8.3 (exp/log/rcp/sqrt/rsq/sin, mad_u64_u32), P 9.5 (v_pk_*_f32).
usage: tools/isa_count.py file.hip [kernel-substring ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = ("v_fma_f32 v_fmac_f32 v_fmaak_f32 v_fmamk_f32 v_mul_f32 v_add_f32 v_sub_f32 v_subrev_f32 v_add_u32 "
     "v_sub_u32 v_subrev_u32 v_and_b32 v_or_b32 v_xor_b32 v_lshrrev_b32 v_lshlrev_b32 v_ashrrev_i32 "
     "v_mov_b32 v_and_or_b32 v_or3_b32 v_not_b32 v_add_co_u32 v_addc_co_u32 v_lshl_or_b32 v_xad_u32 v_accvgpr_write_b32 v_accvgpr_read_b32").split()
C = "v_exp_f32 v_log_f32 v_rcp_f32 v_sqrt_f32 v_rsq_f32 v_sin_f32 v_cos_f32 v_mad_u64_u32 v_rcp_iflag_f32".split()


def classify(op):
    op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "M"
    if op.startswith("v_pk_") and op.endswith("_f32"):
        return "P"
    if op in C:
        return "C"
    if op in A:
        return "A"
    return "B"


def main():
    src = sys.argv[1]
    pats = sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, "x.s")
        inc = os.path.join(ROOT, "libplacebo_amd", "csrc", "hip")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-slp-vectorize", "-I" + inc, "-I" + os.path.join(ROOT, "include"),
                               "--cuda-device-only", "-S", src, "-o", obj])
        dis = open(obj).read()
    cur = None
    counts = collections.OrderedDict()
    for line in dis.splitlines():
        m = re.match(r"^([A-Za-z_][A-Za-z_0-9$.]*):", line)
        if m and not m.group(1).startswith(".L"):
            cur = m.group(1)
            counts[cur] = collections.Counter()
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if cur and m:
            op = m.group(1)
            if op.startswith("v_"):
                counts[cur][classify(op)] += 1
                counts[cur]["op:" + re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)] += 1
            elif op.startswith("ds_"):
                counts[cur]["LDS"] += 1
            elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
                counts[cur]["VMEM"] += 1
            elif op.startswith("s_"):
                counts[cur]["S"] += 1
    base = None
    for k, c in counts.items():
        if pats and not any(p in k for p in pats):
            continue
        valu = c["A"] + c["B"] + c["C"] + c["P"]
        cyc = 2.5 * c["A"] + 4.4 * c["B"] + 8.3 * c["C"] + 9.5 * c["P"]
        print("%-40s VALU %4d (A %4d B %4d C %3d P %2d) ~%6.0f cyc  MFMA %3d LDS %3d VMEM %3d SALU %4d" %
              (k[:40], valu, c["A"], c["B"], c["C"], c["P"], cyc, c["M"], c["LDS"], c["VMEM"], c["S"]))
        if os.environ.get("TOP"):
            top = sorted(((n, k2[3:]) for k2, n in c.items() if k2.startswith("op:")), reverse=True)[:int(os.environ["TOP"])]
            print("     " + " ".join("%s:%d" % (o, n) for n, o in top))


if __name__ == "__main__":
    main()
