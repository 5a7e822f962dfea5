import sys, numpy as np
sys.path[:0] = ["tests", "."]
import libplacebo_amd as pl, util
with pl.HipGpu(log_level=6) as g:
    rr = pl.Renderer(g)
    sw, sh = 64, 48
    img = util.random_rgba16(sw, sh, seed=5)
    src = g.tex_create(sw, sh, "rgba16", img)
    dst = g.tex_create(100, 80, "rgba16", np.full((80, 100, 4), 1234, np.uint16))
    image = pl.frame(src, crop=(8, 4, 40, 36))
    target = pl.frame(dst, crop=(42, 20, 10, 52))
    print(rr.render(image, target, pl.render_params("fast")))
    got = dst.download()
    print(got[20, 8:14], got[19, 10], img[4, 36:40])
    for lvl, m in g.messages[-12:]:
        print(lvl, m[:300])
    rr.destroy()
