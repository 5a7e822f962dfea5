"""Timing probe for the polar pass variants (1080p rgba16hf -> 4K)."""
import sys, os, numpy as np
sys.path[:0] = ["tests", "."]
import libplacebo_amd as pl
import util
W, H = 1920, 1080
with pl.HipGpu() as g:
    frame = util.chirp_rgba16(W, H)
    src = g.tex_create(W, H, "rgba16", frame)
    fbo = g.tex_create(W, H, "rgba16hf")
    a = g.begin(); a.sample("direct", src); assert a.finish(fbo)
    cfg = pl.filter_config("ewa_lanczos")
    for name, fmt, dither, comps in (("dither10->rgba16", "rgba16", 10, 3), ("plain->rgba16", "rgba16", 0, 3),
                              ("plain->rgba16hf", "rgba16hf", 0, 3), ("plain->rgba32f", "rgba32f", 0, 3),
                              ("rgba plain->rgba16", "rgba16", 0, 4)):
        dst = g.tex_create(2 * W, 2 * H, fmt)
        lut, ds = pl.ShaderObj(), pl.ShaderObj()
        t = g.timer()
        for it in range(20):
            g.reset_frame()
            b = g.begin()
            assert b.sample_polar(fbo, cfg, lut, new_w=2 * W, new_h=2 * H, components=comps)
            if dither:
                b.dither(dither, ds)
            assert b.finish(dst, timer=t)
        g.finish()
        v = []
        while True:
            ns = g.timer_query(t)
            if not ns: break
            v.append(ns)
        print("%-22s %.1f us" % (name, np.median(v) / 1e3))
        dst.destroy(); lut.destroy(); ds.destroy()
