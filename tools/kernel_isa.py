#!/usr/bin/env python3
"""Static VALU instruction counts of one kernel of a HIP source, per basic block (between .LBB
labels), by issue class (tools/isa_count.py's classes), compiled with the library's flags.
usage: tools/kernel_isa.py <file.hip> <mangled-substring> [extra flags...]
Loop bodies show up as blocks; the caller multiplies by trip counts."""
import collections
import os
import re
import subprocess
import sys
import tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_count import classify

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "x.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S",
                               src, "-o", out] + extra, stderr=subprocess.DEVNULL)
        dis = open(out).read()
    cur, blocks, blk = None, [], None
    for line in dis.splitlines():
        m = re.match(r"^([.A-Za-z_][A-Za-z_0-9$.]*):", line)
        if m:
            lab = m.group(1)
            if not lab.startswith(".L"):
                cur = lab if pat in lab else None
                if cur:
                    print(cur)
            if cur:
                blk = [lab, collections.Counter()]
                blocks.append(blk)
            continue
        if not cur or blk is None:
            continue
        if re.match(r"^\s+\.(end_amdhsa_kernel|section|amdhsa_)", line) and "end_amdhsa" in line:
            cur = None
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", line)
        if m:
            op = m.group(1)
            c = blk[1]
            if op.startswith("v_"):
                c[classify(op)] += 1
            elif op.startswith("ds_"):
                c["LDS"] += 1
            elif op.startswith(("global_", "flat_", "buffer_")):
                c["VMEM"] += 1
            elif op.startswith("s_"):
                c["S"] += 1
                if op.startswith("s_cbranch") or op == "s_branch":
                    c["br:" + line.split()[-1]] += 1
    tot = collections.Counter()
    for lab, c in blocks:
        valu = c["A"] + c["B"] + c["C"] + c["P"]
        for k in "ABCPM":
            tot[k] += c[k]
        if valu + c["M"] >= int(os.environ.get("MINB", "40")):
            br = " ".join(k[3:] for k in c if k.startswith("br:"))[:60]
            print("  %-12s VALU %4d (A %4d B %4d C %3d) ~%5.0f cyc MFMA %3d LDS %3d VMEM %3d S %4d  -> %s" %
                  (lab[:12], valu, c["A"], c["B"], c["C"], 2.5 * c["A"] + 4.4 * c["B"] + 8.3 * c["C"],
                   c["M"], c["LDS"], c["VMEM"], c["S"], br))
    print("  total VALU %d (A %d B %d C %d) MFMA %d" % (tot["A"] + tot["B"] + tot["C"] + tot["P"], tot["A"], tot["B"], tot["C"], tot["M"]))


main()
