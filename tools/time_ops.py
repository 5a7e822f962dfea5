"""Per-op cost probe at 4K: time single-op and short-chain passes (rgba16 -> rgba16hf / rgba16)
and print each one's cost above the plain copy pass. PL_HIP_PASS_TRACE=1 shows the op lists."""
import sys, numpy as np
sys.path[:0] = ["tests", "."]
import libplacebo_amd as pl
from test_gpu_color import hdr_test_frame
W, H = 3840, 2160
with pl.HipGpu() as g:
    if len(sys.argv) > 1 and sys.argv[1] == "chirp":     # bench.py's synthetic HDR content
        import util
        f = util.chirp_rgba16(W, H).astype(np.float32) * 0.75
        f[..., 3] = 65535
        frame = f.astype(np.uint16)
    else:
        base = (hdr_test_frame(64, 48)[..., :] * 65535).astype(np.uint16)
        frame = np.tile(base, (H // 48, W // 64, 1))
    src = g.tex_create(W, H, "rgba16", frame)
    fbo = g.tex_create(W, H, "rgba16hf")
    dst = g.tex_create(W, H, "rgba16")
    pq = pl.color_space("bt2020", "pq", max_luma=1000.0)
    hlg = pl.color_space("bt2020", "hlg", max_luma=1000.0)
    sdr = pl.color_space("bt709", "bt1886")
    srgb = pl.color_space("bt709", "srgb")
    state = pl.ShaderObj()

    def chain(name, fn, target):
        t = g.timer()
        for it in range(12):
            g.reset_frame()
            s = g.begin(); s.sample("direct", src)
            fn(s)
            assert s.finish(target, timer=t), name
        g.finish()
        v = []
        while True:
            ns = g.timer_query(t)
            if not ns: break
            v.append(ns)
        return np.median(v) / 1e3

    cases = [
        ("copy -> f16", lambda s: None, fbo),
        ("copy -> rgba16", lambda s: None, dst),
        ("linearize pq", lambda s: s.linearize(pq), fbo),
        ("delinearize pq", lambda s: s.delinearize(pq), fbo),
        ("linearize hlg", lambda s: s.linearize(hlg), fbo),
        ("linearize bt1886", lambda s: s.linearize(sdr), fbo),
        ("delinearize bt1886", lambda s: s.delinearize(sdr), fbo),
        ("linearize srgb", lambda s: s.linearize(srgb), fbo),
        ("delinearize srgb", lambda s: s.delinearize(srgb), fbo),
        ("sigmoidize", lambda s: s.sigmoidize(), fbo),
        ("map clip/clip", lambda s: s.color_map(pq, sdr, None, pl.color_map_params("clip", "clip")), dst),
        ("map spline/clip", lambda s: s.color_map(pq, sdr, None, pl.color_map_params("spline", "clip")), dst),
        ("map clip/perceptual", lambda s: s.color_map(pq, sdr, None, pl.color_map_params("clip", "perceptual")), dst),
        ("map spline/perceptual", lambda s: s.color_map(pq, sdr, None, pl.color_map_params("spline", "perceptual")), dst),
        ("map bt2390/relative", lambda s: s.color_map(pq, sdr, None, pl.color_map_params("bt.2390", "relative")), dst),
    ]
    res = {}
    for name, fn, target in cases:
        res[name] = chain(name, fn, target)
    b16, bun = res["copy -> f16"], res["copy -> rgba16"]
    for name, fn, target in cases:
        b = b16 if target is fbo else bun
        print(f"{name:24s} {res[name]:8.1f} us   (+{res[name] - b:6.1f})")
