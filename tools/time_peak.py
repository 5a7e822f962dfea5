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
    csp = pl.color_space("bt2020", "pq", max_luma=1000.0)
    for name, kw in (("hist+cutoff", dict(percentile=99.995)), ("nohist", dict(percentile=100.0)),
                     ("nohist nocutoff", dict(percentile=100.0, black_cutoff=0.0)),
                     ("copy only", None)):
        state = pl.ShaderObj()
        t1 = g.timer()
        for it in range(12):
            g.reset_frame()
            a = g.begin(); a.sample("direct", src)
            if kw is not None:
                assert a.detect_peak(csp, state, **kw)
            assert a.finish(fbo, timer=t1)
        g.finish()
        v = []
        while True:
            ns = g.timer_query(t1)
            if not ns: break
            v.append(ns)
        print("%-18s %.1f us" % (name, np.median(v) / 1e3))
        state.destroy()
