import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] + "/tests")
os.environ["PL_HIP_MXR_DEBUG"] = "1"
import numpy as np
import libplacebo_amd as pl, util
img = util.chirp_rgba16(960, 540)
outs = []
for mfma in ("1", "0"):
    os.environ["PL_HIP_POLAR_MFMA"] = mfma
    with pl.HipGpu(0, log_level=5) as g:
        src = g.tex_create(960, 540, "rgba16", img); dst = g.tex_create(3840, 2160, "rgba16")
        rr = pl.Renderer(g)
        p = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"))
        assert rr.render(pl.frame(src, components=3), pl.frame(dst), p)
        outs.append(dst.download())
        if mfma == "1":
            for _, m in g.messages:
                if "mxr axis" in m or "matrix" in m:
                    print(m)
d = np.abs(outs[0].astype(np.int64) - outs[1].astype(np.int64))[..., :3]
ys, xs = np.nonzero((d > 1).any(axis=-1))
print("4x chirp: > 1 code on", len(ys), "pixels; max", d.max())
if len(ys):
    print("rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "x mod 4", np.bincount(xs % 4, minlength=4), "y mod 4", np.bincount(ys % 4, minlength=4))
    print("x mod 256 histogram (16 bins)", np.histogram(xs % 256, bins=16)[0], "y mod 64 (16 bins)", np.histogram(ys % 64, bins=16)[0])
    i = np.argmax(d.max(axis=-1)); y, x = divmod(i, 3840); print("worst at", x, y, outs[0][y, x], outs[1][y, x], "src around", img[y // 4, max(x // 4 - 2, 0):x // 4 + 3, 0])
