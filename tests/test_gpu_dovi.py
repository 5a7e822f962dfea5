"""Dolby Vision decoding (PL_COLOR_SYSTEM_DOLBYVISION: pl_shader_dovi_reshape and the LMS tail of
pl_shader_decode_color) against the oracle's restatement of src/shaders/colorspace.c:51-271,
:285-292, :392-420. Reshaping is polynomial arithmetic: bit-exact. The tail has two PQ curves:
held to float64 like the other PQ stages. The decoding matrix and offsets come from
pl_color_repr_decode, pinned bit for bit to the reference build (tests/test_tier0_ref.py)."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util
from libplacebo_amd import _capi as capi
from test_gpu_color import run_ops

pytestmark = pytest.mark.gpu


LMS2RGB = np.array([[3.06441879, -2.16597676, 0.10155818],
                    [-0.65612108, 1.78554118, -0.12943749],
                    [0.01736321, -0.04725154, 1.03004253]])


def metadata(seed=0):
    """What a stream's RPU looks like: luma in three nearly-linear quadratic pieces, one chroma
    component as a single third-order MMR close to the identity, the other in three pieces of
    mixed kinds and orders; a BT.2020-like YCC matrix with its offsets; an RGB -> LMS matrix close
    to the inverse of the decoder's fixed LMS -> RGB."""
    rng = np.random.default_rng(seed)
    m = capi.DoviMetadata()
    o = (orc.DoviComp * 3)()

    def poly(c, i, k0, k1, k2):
        m.comp[c].method[i] = o[c].method[i] = 0
        for k, v in enumerate((k0, k1, k2)):
            m.comp[c].poly_coeffs[i][k] = o[c].poly_coeffs[i][k] = v

    def mmr(c, i, order):
        m.comp[c].method[i] = o[c].method[i] = 1
        m.comp[c].mmr_order[i] = o[c].mmr_order[i] = order
        m.comp[c].mmr_constant[i] = o[c].mmr_constant[i] = float(rng.normal() * 0.005)
        for j in range(order):
            for k in range(7):
                v = float(rng.normal() * 0.02 / (j + 1)) + (0.97 if (j, k) == (0, c) else 0.0)
                m.comp[c].mmr_coeffs[i][j][k] = o[c].mmr_coeffs[i][j][k] = v

    def pivots(c, *p):
        m.comp[c].num_pivots = o[c].num_pivots = len(p)
        for k, v in enumerate(p):
            m.comp[c].pivots[k] = o[c].pivots[k] = v

    pivots(0, 0.0, 0.3, 0.7, 1.0)
    poly(0, 0, 0.004, 0.98, 0.05)
    poly(0, 1, -0.01, 1.05, -0.04)
    poly(0, 2, 0.03, 0.93, 0.04)
    pivots(1, 0.0, 1.0)
    mmr(1, 0, 3)
    pivots(2, 0.0, 0.47, 0.53, 1.0)
    mmr(2, 0, 1)
    poly(2, 1, 0.002, 0.995, 0.003)
    mmr(2, 2, 2)
    for i in range(3):
        m.nonlinear_offset[i] = (0.0, 0.5, 0.5)[i]
    ycc = ((1.0, 0.0, 1.4746), (1.0, -0.16455, -0.57135), (1.0, 1.8814, 0.0))
    lms = np.linalg.inv(LMS2RGB) * (1 + 0.01 * rng.normal(size=(3, 3)))
    for i in range(3):
        for j in range(3):
            m.nonlinear[i][j] = ycc[i][j]
            m.linear[i][j] = lms[i][j]
    return m, o


def source(seed=1):
    """(Y, Cb, Cr, a) samples of ordinary pictures -- mid-range luma, modest chroma -- plus a few
    rows ON pivots and outside [0, 1] (clamped for the curves)"""
    rng = np.random.default_rng(seed)
    src = rng.random((48, 64, 4)).astype(np.float32)
    src[..., 0] = 0.2 + 0.6 * src[..., 0]
    src[..., 1:3] = 0.5 + 0.16 * (src[..., 1:3] - 0.5)
    src[0, :8, :3] = [0.0, 0.3, 0.47]
    src[1, :8, :3] = [1.0, 0.7, 0.53]
    src[2, :8, :3] = [-0.5, 1.5, 0.3]
    return src


def in_gamut(rgb_pq, lms2rgb):
    """Pixels the PQ pair is defined for: PQ-coded values inside (0, 1) and a positive result of
    the LMS -> RGB matrix. Outside, the shader text divides by (c2 - c3 x) near or beyond its pole
    and takes pow() of negatives: undefined in GLSL, not comparable."""
    v = rgb_pq[..., :3].astype(np.float64)
    ok = np.all((v > 1e-3) & (v < 0.999), axis=-1)
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 4096 * 128, 3424 / 4096, 2413 / 128, 2392 / 128
    x = np.clip(v, 0, 1) ** (1 / m2)
    lin = (np.maximum(x - c1, 0) / (c2 - c3 * x)) ** (1 / m1)
    out = lin @ np.asarray(lms2rgb, dtype=np.float64).T
    return ok & np.all(out > 1e-3, axis=-1)


def test_reshape_bit_exact(gpu):
    meta, ocomp = metadata()
    src = source()

    def rec(sh):
        pl.lib().pl_shader_dovi_reshape(sh.sh, C.byref(meta))
    got = run_ops(gpu, src, rec)
    ref = orc.dovi_reshape(src.copy(), ocomp)
    assert np.array_equal(got, ref), util.diff_stats(got, ref)
    assert not np.array_equal(got[..., :3], src[..., :3])
    assert np.array_equal(got[..., 3], src[..., 3])
    # a component without pivots passes through; only polynomials; only MMR of mixed orders
    for keep in ((0,), (1,), (2,)):
        m2, o2 = metadata(seed=3)
        for c in range(3):
            if c not in keep:
                m2.comp[c].num_pivots = o2[c].num_pivots = 0
        got = run_ops(gpu, src, lambda sh: pl.lib().pl_shader_dovi_reshape(sh.sh, C.byref(m2)))
        ref = orc.dovi_reshape(src.copy(), o2)
        assert np.array_equal(got, ref), (keep, util.diff_stats(got, ref))


def test_decode_color_dolby_vision(gpu):
    """the whole decode: integer scale, reshape, the stream's YCC -> RGB' matrix and offsets, PQ
    EOTF, LMS -> RGB, PQ OETF"""
    meta, ocomp = metadata()
    src = source(2)
    src[..., :3] *= 4095.0 / 65535.0 * 16       # 12 bits in the upper bits of 16: scale = 1 / (that)
    bits = dict(sample_depth=16, color_depth=12, bit_shift=4)
    r1 = pl.color_repr("dolbyvision", "unknown", **bits)
    r1.dovi = C.addressof(meta)
    got = run_ops(gpu, src, lambda sh: sh.decode_color(r1))
    assert r1.sys == pl.SYS["rgb"]

    r2 = pl.color_repr("dolbyvision", "unknown", **bits)
    r2.dovi = C.addressof(meta)
    L = pl.lib()
    L.pl_color_repr_normalize.restype = C.c_float
    scale = L.pl_color_repr_normalize(C.byref(r2))
    ref = src.copy()
    ref[..., :3] *= np.float32(scale)
    orc.dovi_reshape(ref, ocomp)
    tr = L.pl_color_repr_decode(C.byref(r2), None)
    orc.op_affine(ref, [tr.mat.m[i][j] for i in range(3) for j in range(3)], list(tr.c))
    lms2rgb = LMS2RGB.astype(np.float32)
    lin = np.array([[meta.linear[i][j] for j in range(3)] for i in range(3)], dtype=np.float32)
    # pl_matrix3x3_mul (a := a * b) in float, row by row
    m = np.zeros((3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            acc = np.float32(0)
            for k in range(3):
                acc = np.float32(acc + np.float32(lms2rgb[i, k] * lin[k, j]))
            m[i, j] = acc
    valid = in_gamut(ref, m)
    orc.dovi_lms(ref, [float(v) for v in m.reshape(-1)])
    d = np.abs(got[..., :3].astype(np.float64) - ref[..., :3].astype(np.float64))[valid]
    print("dolby vision decode vs oracle: max %.2e, mean %.2e over %.0f %% of the pixels"
          % (d.max(), d.mean(), 100 * valid.mean()))
    assert valid.mean() > 0.6
    # two pow() pairs around a matrix with entries of 3 and -2 on nearly equal L, M, S: the EOTF's
    # 7.5e-6 relative error (test_gpu_color.py) comes out ~ 5 times larger: one code of 16 bits
    assert d.max() <= 1.5e-5 and d.mean() <= 4e-6, (float(d.max()), float(d.mean()))
    assert np.array_equal(got[..., 3], src[..., 3])
    assert 0.05 < got[..., :3].mean() < 0.95


def test_renderer_decodes_a_dolby_vision_frame(gpu):
    """pl_render_image on a packed Dolby Vision frame (PQ / BT.2020 with the stream's metadata) to
    an HDR10 target of the same size: the frame is the decode of the texels (the colour spaces are
    equal, nothing else happens). Through the measuring path as well (peak detection on)."""
    w, h = 64, 48
    meta, ocomp = metadata()
    img = orc.tex_encode(source(6), "rgba16")
    src = gpu.tex_create(w, h, "rgba16", img)
    dst = gpu.tex_create(w, h, "rgba16")
    hdr = pl.color_space("bt2020", "pq", max_luma=1000.0)
    repr_ = pl.color_repr("dolbyvision", "unknown", sample_depth=16, color_depth=16)
    repr_.dovi = C.addressof(meta)
    image = pl.frame(src, components=3, repr_=repr_, color=hdr)
    rr = pl.Renderer(gpu)
    assert rr.render(image, pl.frame(dst, components=3, color=hdr),
                     pl.render_params("fast", dither_params=None)), gpu.messages[-4:]
    assert rr.errors() == 0, gpu.messages[-4:]
    got = dst.download()

    ref = orc.tex_decode(img, "rgba16")
    ref[..., 3] = 1.0
    orc.dovi_reshape(ref, ocomp)
    r2 = pl.color_repr("dolbyvision", "unknown", sample_depth=16, color_depth=16)
    r2.dovi = C.addressof(meta)
    tr = pl.lib().pl_color_repr_decode(C.byref(r2), None)
    orc.op_affine(ref, [tr.mat.m[i][j] for i in range(3) for j in range(3)], list(tr.c))
    lin = np.array([[meta.linear[i][j] for j in range(3)] for i in range(3)], dtype=np.float64)
    m = (LMS2RGB.astype(np.float32).astype(np.float64) @ lin).astype(np.float32)
    valid = in_gamut(ref, m)
    orc.dovi_lms(ref, [float(v) for v in m.reshape(-1)])
    want = orc.tex_encode(ref, "rgba16")
    d = np.abs(got[..., :3].astype(np.int64) - want[..., :3].astype(np.int64))[valid]
    assert valid.mean() > 0.5 and d.max() <= 1, (float(valid.mean()), int(d.max()))

    # tone-mapped to SDR with peak detection: the measuring pass needs the decoded frame as a
    # texture (the Dolby Vision ops live in one kernel variant); it must render and not be black
    sdr = gpu.tex_create(w, h, "rgba16")
    params = pl.render_params("default", dither_params=None,
                              peak_detect_params=pl.peak_detect_params())
    assert rr.render(image, pl.frame(sdr, components=3, color=pl.color_space("bt709", "bt1886")),
                     params), gpu.messages[-4:]
    assert rr.errors() == 0, gpu.messages[-4:]
    out = sdr.download()
    assert 2000 < out[..., :3].mean() < 60000
    rr.destroy()
    for t in (src, dst, sdr):
        t.destroy()


def test_reshaping_curves_never_shared_between_recorded_shaders(gpu):
    """The reshaping curves of a shader live in a ring of 32 device slots (pl_shader_decode_color has
    no state object to keep them in). A slot belongs to its shader until that shader is dispatched,
    reset or freed: recording a 33rd such shader while 32 are pending FAILS (round 5 handed it slot 0
    again, whose first owner then read the newcomer's curves -- ADVICE r05), and after the pending ones
    have run or been dropped the ring is whole again."""
    meta, ocomp = metadata()
    src = source()
    h, w = src.shape[:2]
    t = gpu.tex_create(w, h, "rgba32f", src)
    pending = []
    for _ in range(32):
        sh = gpu.begin()
        assert sh.sample("nearest", t)
        pl.lib().pl_shader_dovi_reshape(sh.sh, C.byref(meta))
        assert not sh.failed()
        pending.append(sh)
    extra = gpu.begin()
    assert extra.sample("nearest", t)
    pl.lib().pl_shader_dovi_reshape(extra.sh, C.byref(meta))
    assert extra.failed()
    pl.lib().pl_dispatch_abort(gpu.dp, C.byref(extra.sh))
    # the first 32 still render what their own curves say; dispatching / dropping them frees the slots
    d = gpu.tex_create(w, h, "rgba32f")
    assert pending[0].finish(d)
    ref = orc.dovi_reshape(src.copy(), ocomp)
    assert np.array_equal(d.download(), ref)
    for sh in pending[1:]:
        pl.lib().pl_dispatch_abort(gpu.dp, C.byref(sh.sh))
    for _ in range(3):
        got = run_ops(gpu, src, lambda sh: pl.lib().pl_shader_dovi_reshape(sh.sh, C.byref(meta)))
        assert np.array_equal(got, ref)
    t.destroy(); d.destroy()
