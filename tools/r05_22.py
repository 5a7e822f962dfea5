#!/usr/bin/env python3
"""The driver's short run, again and again inside one process: is the 20-step figure low because of
what precedes the first timed region (then only region 1 is slow) or because of the sync that opens
every region (then all are)?  usage: tools/r05_22.py [prime seconds]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

torch.cuda.set_device(0)
torch.cuda.synchronize()
(sw, sh), (dw, dh), _, _ = bench.WORKLOADS["ewa_1080p_to_4k_hdr_tonemap"]
st = bench.Stream(0, "ewa_1080p_to_4k_hdr_tonemap", 10)
bench.prime(st, float(sys.argv[1]) if len(sys.argv) > 1 else None)
import ctypes
import libplacebo_amd as pl
L = pl.lib()
L.plh_test_peak_wait_ns.restype = ctypes.c_long
out = []
for rep in range(8):
    L.plh_test_peak_wait_ns()
    dt = bench.run_timed(st, 20, 5, sync=torch.cuda.synchronize)
    out.append((round(dt / 20 * 1e3, 4), round(L.plh_test_peak_wait_ns() / 25 / 1e3, 1)))
    if rep == 3:
        time.sleep(0.05)    # an idle period in the middle
L.plh_test_peak_wait_ns()
dt = bench.run_timed(st, 300, 5, sync=torch.cuda.synchronize)
print("20-step regions (ms per frame, us waited for the measurement per frame):", out,
      "| 300 steps:", round(dt / 300 * 1e3, 4), round(L.plh_test_peak_wait_ns() / 305 / 1e3, 1))
st.close()
