"""cfg 4 timing probe: 4K PQ/BT.2020 rgba16 -> BT.709 SDR rgba16, detect + map."""
import sys, numpy as np
sys.path[:0] = ["tests", "."]
import libplacebo_amd as pl
from test_gpu_color import hdr_test_frame
W, H = 3840, 2160
with pl.HipGpu() as g:
    base = (hdr_test_frame(64, 48)[..., :] * 65535).astype(np.uint16)
    frame = np.tile(base, (H // 48, W // 64, 1))
    src = g.tex_create(W, H, "rgba16", frame)
    fbo = g.tex_create(W, H, "rgba16hf")
    dst = g.tex_create(W, H, "rgba16")
    csp = pl.color_space("bt2020", "pq", max_luma=1000.0)
    out = pl.color_space("bt709", "bt1886")
    state = pl.ShaderObj()
    t1, t2 = g.timer(), g.timer()
    for it in range(30):
        g.reset_frame()
        a = g.begin(); a.sample("direct", src)
        assert a.detect_peak(csp, state)
        assert a.finish(fbo, timer=t1)
        b = g.begin(); b.sample("direct", fbo)
        b.color_map(csp, out, state, None)
        assert b.finish(dst, timer=t2)
    g.finish()
    for name, t in (("detect", t1), ("map", t2)):
        v = []
        while True:
            ns = g.timer_query(t)
            if not ns: break
            v.append(ns)
        print(name, "us:", np.median(v) / 1e3, len(v))
