#!/usr/bin/env python3
"""How long do N frames of the metric's workload take from an EMPTY pipeline (finish before, finish
after)? t(N) = a + b N: b = the steady frame time, a = what the driver's short command
(--steps 20 --warmup 5) pays once: first-frame latency + the closing syncs. Also: per-call host
time of pl_render_image (is the queue fed fast enough?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import bench

def main():
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
    st = bench.Stream(0, "ewa_1080p_to_4k_hdr_tonemap", 10)
    for _ in range(30):
        st.step()
    st.g.finish()
    rows = []
    for n in (1, 2, 3, 5, 10, 20, 20, 40, 80, 160):
        best = None
        for rep in range(5):
            st.g.finish(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            host = []
            for _ in range(n):
                h0 = time.perf_counter(); st.step(); host.append(time.perf_counter() - h0)
            t1 = time.perf_counter()
            st.g.finish()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            cand = (t3 - t0, t1 - t0, t2 - t1, t3 - t2, float(np.median(host)))
            best = cand if best is None or cand[0] < best[0] else best
        rows.append((n,) + best)
        print("N=%3d total %.3f ms = %.4f ms/frame | submit loop %.3f ms, finish %.3f ms, device sync %.3f ms, host per call %.1f us"
              % (n, best[0] * 1e3, best[0] * 1e3 / n, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3, best[4] * 1e6))
    ns = np.array([r[0] for r in rows], float); ts = np.array([r[1] for r in rows]) * 1e3
    b, a = np.polyfit(ns, ts, 1)
    print("fit: t(N) = %.3f ms + %.4f ms * N" % (a, b))
    # after an idle gap (clocks): sleep 50 ms, then 20 frames
    for gap in (0.0, 0.005, 0.05, 0.5):
        st.g.finish(); time.sleep(gap)
        t0 = time.perf_counter()
        for _ in range(20):
            st.step()
        st.g.finish()
        print("after %.0f ms idle: 20 frames %.4f ms/frame" % (gap * 1e3, (time.perf_counter() - t0) * 1e3 / 20))
    st.close()

main()
