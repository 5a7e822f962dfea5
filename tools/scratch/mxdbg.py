import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["PL_HIP_POLAR_MFMA"] = "1"
import numpy as np
import libplacebo_amd as pl, util
for (sw, sh) in [(96, 64), (1920, 1080)]:
    img = util.random_rgba16(sw, sh, seed=3)
    with pl.HipGpu(0, log_level=5) as g:
        src = g.tex_create(sw, sh, "rgba16", img)
        dst = g.tex_create(2*sw, 2*sh, "rgba32f")
        rr = pl.Renderer(g)
        params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"))
        assert rr.render(pl.frame(src, components=3), pl.frame(dst), params)
        for lv, m in g.messages:
            if "matrix" in m or "polar" in m:
                print(sw, m)
        mx = dst.download()
    os.environ["PL_HIP_POLAR_MFMA"] = "0"
    with pl.HipGpu(0, log_level=5) as g:
        src = g.tex_create(sw, sh, "rgba16", img)
        dst = g.tex_create(2*sw, 2*sh, "rgba32f")
        rr = pl.Renderer(g)
        assert rr.render(pl.frame(src, components=3), pl.frame(dst), params)
        pp = dst.download()
    os.environ["PL_HIP_POLAR_MFMA"] = "1"
    e = np.abs(mx[..., :3].astype(np.float64) - pp[..., :3])
    print(sw, "max", e.max(), "mean", e.mean(), "per-column max (every 240th):", e.max(axis=(0, 2))[::max(1, 2*sw//16)])
