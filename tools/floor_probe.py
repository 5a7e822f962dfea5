"""Fixed cost of one pass: tiny and growing copies through k_pass_generic."""
import sys, numpy as np
sys.path[:0] = ["tests", "."]
import libplacebo_amd as pl
with pl.HipGpu() as g:
    for w, h in ((64, 64), (512, 512), (1920, 1080), (3840, 2160), (7680, 4320)):
        src = g.tex_create(w, h, "rgba16", np.zeros((h, w, 4), np.uint16))
        dst = g.tex_create(w, h, "rgba16hf")
        t = g.timer()
        for it in range(20):
            a = g.begin(); a.sample("direct", src); assert a.finish(dst, timer=t)
        g.finish()
        v = []
        while True:
            ns = g.timer_query(t)
            if not ns: break
            v.append(ns)
        mb = w * h * 16 / 1e6
        print("%5dx%-5d %8.1f us  %7.1f MB  %6.2f TB/s" % (w, h, np.median(v) / 1e3, mb, mb / np.median(v) * 1e3 / 1e6))
        src.destroy(); dst.destroy()
