"""Timing probe: separable lanczos 1080p -> 4K (vertical pass, then horizontal), deband 4K."""
import sys, numpy as np
sys.path[:0] = ["tests", "."]
import libplacebo_amd as pl
import util
W, H = 1920, 1080
with pl.HipGpu() as g:
    src = g.tex_create(W, H, "rgba16", util.chirp_rgba16(W, H))
    fbo_a = g.tex_create(W, H, "rgba16hf")
    a = g.begin(); a.sample("direct", src); assert a.finish(fbo_a)
    fbo_v = g.tex_create(W, 2 * H, "rgba16hf")
    dst = g.tex_create(2 * W, 2 * H, "rgba16")
    lut = pl.ShaderObj()
    cfg = pl.filter_config(sys.argv[1] if len(sys.argv) > 1 else "lanczos")
    tv, th, td = g.timer(), g.timer(), g.timer()
    for it in range(15):
        g.reset_frame()
        v = g.begin(); assert v.sample_ortho(fbo_a, cfg, lut, new_w=W, new_h=2 * H, components=3)
        assert v.finish(fbo_v, timer=tv)
        h = g.begin(); assert h.sample_ortho(fbo_v, cfg, lut, new_w=2 * W, new_h=2 * H, components=3)
        assert h.finish(dst, timer=th)
        d = g.begin(); assert d.deband(dst)
        big = fbo_big if it else None
        if it == 0:
            fbo_big = g.tex_create(2 * W, 2 * H, "rgba16hf")
        assert d.finish(fbo_big, timer=td)
    g.finish()
    for name, t in (("ortho vertical 1080->2160", tv), ("ortho horizontal 1920->3840", th), ("deband 4K", td)):
        v = []
        while True:
            ns = g.timer_query(t)
            if not ns: break
            v.append(ns)
        print("%-28s %.1f us" % (name, np.median(v) / 1e3))
