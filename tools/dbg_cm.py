import sys, ctypes as C, numpy as np
sys.path[:0] = ["tests", "."]
import libplacebo_amd as pl, orc, colormap_ref as cr
from test_gpu_color import hdr_test_frame, run_ops
tone, gamut = sys.argv[1], sys.argv[2]
with pl.HipGpu() as g:
    src_img = hdr_test_frame()
    src = pl.color_space("bt2020", "pq", max_luma=1000.0)
    dst = pl.color_space("bt709", "bt1886")
    state = pl.ShaderObj()
    L = []
    def rec(sh):
        sh.color_map(src, dst, state, pl.color_map_params(tone=tone, gamut=gamut))
        L.append(sh.listing())
    got = run_ops(g, src_img, rec)
    print(L[0])
    r = cr.resolve(cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=1000.0),
                   cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]), tone=tone.encode(), gamut=gamut.encode())
    print("tone", r["tone"].input_min, r["tone"].input_max, r["tone"].output_min, r["tone"].output_max, r["need_tone"], r["need_gamut"])
    print("gamut", r["gamut"].min_luma, r["gamut"].max_luma, r["kw"].get("gamut_scale"), r["kw"].get("gamut_offset"), r["kw"].get("tone_p"))
    print("lin", r["lin"], "delin", r["delin"])
    if tone == "clip":
        r["kw"].update(tone_mode=0, tone_p=(r["tone"].input_min, r["tone"].input_max, 0, 0), tone_lut=None)
    ref = cr.apply(src_img.copy(), r)
    d = np.abs(got - ref)
    idx = np.argsort(d.max(axis=2).ravel())[::-1][:8]
    for i in idx:
        y, x = divmod(i, d.shape[1])
        print(y, x, src_img[y, x], got[y, x], ref[y, x])
    print("frac>1.5lsb", (d > 1.5/65535).mean(), "mean", d.mean()*65535)
    for q in (0.5, 0.9, 0.99, 0.999):
        print("quantile", q, np.quantile(d[..., :3], q) * 65535)
