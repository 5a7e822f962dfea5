"""Scratch timing of the first kernels (not the contract bench)."""
import time, sys
import numpy as np
import libplacebo_amd as pl
sys.path.insert(0, "tests")
import util

with pl.HipGpu(0) as g:
    sw, sh, dw, dh = 1920, 1080, 3840, 2160
    src = util.chirp_rgba16(sw, sh)
    N = 8
    srcs = [g.tex_create(sw, sh, "rgba16", src) for _ in range(N)]
    fbos = [g.tex_create(sw, sh, "rgba16hf") for _ in range(N)]
    dsts = [g.tex_create(dw, dh, "rgba16") for _ in range(N)]
    dst8 = [g.tex_create(dw, dh, "rgba8") for _ in range(N)]
    lut, ds = pl.ShaderObj(), pl.ShaderObj()
    cfg = pl.filter_config("ewa_lanczos")

    def bilinear(i):
        s = g.begin(); s.sample("bilinear", srcs[i % N], new_w=dw, new_h=dh); s.finish(dsts[i % N])

    def passA(i):
        s = g.begin(); s.sample("direct", srcs[i % N]); s.finish(fbos[i % N])

    def polar(i):
        s = g.begin(); s.sample_polar(fbos[i % N], cfg, lut, new_w=dw, new_h=dh, components=3)
        s.dither(8, ds); s.finish(dst8[i % N])

    for name, fn in (("bilinear", bilinear), ("passA", passA), ("polar+dither", polar)):
        for i in range(3): fn(i)
        g.finish()
        K = 20
        t0 = time.perf_counter()
        for i in range(K): fn(i)
        g.finish()
        dt = (time.perf_counter() - t0) / K
        print(f"{name}: {dt*1e6:.1f} us/frame  {dw*dh/dt/1e6:.0f} Mpx/s")
