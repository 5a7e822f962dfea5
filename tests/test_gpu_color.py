"""Colour stages on the GPU: the reference's own GPU known-answer tests
(src/tests/gpu_tests.c:560-703, 753-785) restated against this backend, plus
parity with the CPU oracle.

Tolerances: these stages use pow/exp/log — native v_exp_f32/v_log_f32 on the
GPU, libm in the oracle — so the bar is the reference's own epsilons
(1e-6 SDR / 1e-4 HDR round trips, 1e-4 golden vectors) and <= 1 LSB at 16 bit
against the oracle, not bit-exactness.
"""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu

W = H = 16


def ramp():
    """The reference test pattern: ((x+.5)/W, (y+.5)/H, 0, 1) as rgba32f."""
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 4), np.float32)
    img[..., 0] = (x + 0.5) / W
    img[..., 1] = (y + 0.5) / H
    img[..., 3] = 1.0
    return img


def run_ops(g, src, record, out_fmt="rgba32f"):
    h, w = src.shape[:2]
    t = g.tex_create(w, h, "rgba32f", src)
    d = g.tex_create(w, h, out_fmt)
    sh = g.begin()
    assert sh.sample("nearest", t)
    record(sh)
    assert not sh.failed(), g.messages[-3:]
    assert sh.finish(d), g.messages[-3:]
    out = d.download()
    t.destroy(); d.destroy()
    return out


def nominal(csp):
    """(csp_min, csp_max) in PL_HDR_NORM, from the (reference-pinned) product Tier-0."""
    mn, mx = C.c_float(), C.c_float()

    class NLP(C.Structure):
        _fields_ = [("color", C.POINTER(capi.ColorSpace)), ("metadata", C.c_int),
                    ("scaling", C.c_int), ("out_min", C.POINTER(C.c_float)),
                    ("out_max", C.POINTER(C.c_float)), ("out_avg", C.POINTER(C.c_float))]
    p = NLP(color=C.pointer(csp), metadata=2, scaling=0, out_min=C.pointer(mn),
            out_max=C.pointer(mx))
    pl.lib().pl_color_space_nominal_luma_ex(C.byref(p))
    return mn.value, mx.value


def luma_coeffs(primaries):
    m = pl.lib().pl_get_rgb2xyz_matrix(pl.lib().pl_raw_primaries_get(primaries))
    return [m.m[1][0], m.m[1][1], m.m[1][2]]


TRCS = [k for k in pl.TRC if k != "linear"]
HDR_TRCS = {"pq", "hlg", "vlog", "slog1", "slog2", "scrgb"}


@pytest.mark.parametrize("trc", list(pl.TRC))
def test_kat_transfer_roundtrip(gpu, trc):
    # gpu_tests.c:560-577: delinearize then linearize must be the identity
    csp = pl.color_space("unknown", trc, min_luma=1e-6)
    out = run_ops(gpu, ramp(), lambda sh: (sh.delinearize(csp), sh.linearize(csp)))
    eps = 1e-4 if trc in HDR_TRCS else 1e-6
    assert np.abs(out - ramp()).max() <= eps * 1.5, np.abs(out - ramp()).max()


@pytest.mark.parametrize("trc", TRCS)
@pytest.mark.parametrize("direction", ["linearize", "delinearize"])
def test_transfer_vs_oracle(gpu, trc, direction):
    csp = pl.color_space("bt2020" if trc in ("pq", "hlg") else "bt709", trc)
    pl.lib().pl_color_space_infer(C.byref(csp))
    mn, mx = nominal(csp)
    luma = luma_coeffs(csp.primaries)
    rng = np.random.default_rng(3)
    src = rng.random((32, 32, 4)).astype(np.float32)
    if direction == "delinearize":
        src[..., :3] *= mx
    got = run_ops(gpu, src, lambda sh: getattr(sh, direction)(csp))
    ref = getattr(orc, direction)(src.copy(), int(csp.transfer), mn, mx, luma)
    # relative to the signal range: <= 1 LSB of 16 bit. The PQ EOTF is ill-conditioned near
    # its peak (1 ulp of the inner pow is amplified ~700x by (c2 - c3*v) and the 6.28 power),
    # so for it the bar is the reference's own HDR epsilon (1e-4 of range, gpu_tests.c:571).
    scale = max(1.0, float(np.abs(ref[..., :3]).max()))
    tol = 1e-4 * scale if (trc == "pq" and direction == "linearize") else scale / 65535.0
    assert np.abs(got - ref).max() <= tol, np.abs(got - ref).max()


def test_pq_pair_error_bound_over_every_code(gpu):
    """pqmath.hiph's pq_eotf_acc1 / pq_oetf_acc1 (every PQ conversion of the image path) against a
    float64 evaluation of the reference's formulas (colorspace.c:645-651, :749-755, with the
    six-decimal constants it prints), for ALL 65536 sixteen-bit PQ codes. The bound that matters is
    the distance to one 16-bit code: 1.5e-5 relative at the steepest point of the curve; the
    EOTF is evaluated as exp2(log2(inner) * 6.277) with native log / exp and plain reciprocals
    (ADVICE r03), so its error is stated and held here: <= 7.5e-6 relative over the whole range
    (measured 5.4e-6), near black included, and the OETF puts every code's linear value back within
    0.1 code."""
    import colormap_f64 as c64
    codes = np.arange(65536, dtype=np.float64)
    v = (codes / 65535.0).reshape(256, 256)
    src = np.zeros((256, 256, 4), np.float32)
    src[..., 0] = v
    src[..., 1] = v[::-1, ::-1]
    src[..., 2] = v.T
    src[..., 3] = 1.0
    csp = pl.color_space("bt2020", "pq")
    pl.lib().pl_color_space_infer(C.byref(csp))
    got = run_ops(gpu, src, lambda sh: sh.linearize(csp)).astype(np.float64)
    ref = c64.pq_eotf(src[..., :3].astype(np.float64)) * c64.K10
    nz = src[..., :3] > 0
    rel = np.abs(got[..., :3] - ref)[nz] / ref[nz]
    dark = (src[..., :3] < 0.1)[nz]
    worst = src[..., :3][nz][np.argmax(rel)]
    print("PQ EOTF vs float64: max relative error %.2e at PQ %.4f (codes below 0.1: %.2e, mean %.2e)" %
          (rel.max(), worst, rel[dark].max(), rel.mean()))
    assert rel.max() <= 7.5e-6      # (half of what one 16-bit code is at the curve's steepest point)
    assert np.all(got[..., :3][~nz] == 0.0)
    # ... and back: the OETF of the float64 linear value is the code it came from
    lin = np.zeros_like(src)
    lin[..., :3] = ref.astype(np.float32)
    lin[..., 3] = 1.0
    back = run_ops(gpu, lin, lambda sh: sh.delinearize(csp)).astype(np.float64)
    err = np.abs(back[..., :3] - src[..., :3].astype(np.float64)) * 65535.0
    # (the fp32 rounding of the INPUT alone moves the result by up to 6e-8 * slope: near black,
    # where a code is 1e-7 of the linear range, that is the dominant term)
    truth = c64.pq_oetf(lin[..., :3].astype(np.float64) / c64.K10)
    own = np.abs(back[..., :3] - truth) * 65535.0
    print("PQ OETF vs float64: max %.3f codes of 16 bits (%.3f against the float64 OETF of the fp32 input)"
          % (err.max(), own.max()))
    assert own.max() <= 0.1 and err.max() <= 0.6


@pytest.mark.parametrize("inverse", [False, True])
def test_sigmoid_vs_oracle(gpu, inverse):
    rng = np.random.default_rng(4)
    src = rng.random((32, 32, 4)).astype(np.float32)
    got = run_ops(gpu, src, lambda sh: sh.sigmoidize(inverse=inverse))
    ref = orc.sigmoid(src.copy(), inverse=inverse)
    assert np.abs(got - ref).max() <= 2e-6


def test_sigmoid_roundtrip(gpu):
    out = run_ops(gpu, ramp(), lambda sh: (sh.sigmoidize(), sh.sigmoidize(inverse=True)))
    assert np.abs(out - ramp())[..., :3].max() <= 2e-6


SYSTEMS = [k for k in pl.SYS if k not in ("dolbyvision",)]


@pytest.mark.parametrize("sysname", SYSTEMS)
def test_kat_color_system_roundtrip(gpu, sysname):
    # gpu_tests.c:673-703: encode then decode is the identity
    if sysname in ("bt2100pq", "bt2100hlg"):
        pytest.skip("skipped by the reference too ('horrifically noisy')")
    def rec(sh):
        sh.encode_color(pl.color_repr(sysname, "unknown"))
        sh.decode_color(pl.color_repr(sysname, "unknown"))
    out = run_ops(gpu, ramp(), rec)
    eps = 1e-5 if sysname in ("bt2020c", "xyz") else 1e-6
    assert np.abs(out - ramp()).max() <= eps * 1.5, np.abs(out - ramp()).max()


@pytest.mark.parametrize("sysname,levels,bits", [("bt709", "limited", (16, 10, 6)),
                                                  ("bt2020nc", "limited", (10, 10, 0)),
                                                  ("bt601", "full", (8, 8, 0)),
                                                  ("rgb", "limited", (16, 16, 0))])
def test_decode_color_bit_exact_vs_oracle(gpu, sysname, levels, bits):
    # affine-only decode: no transcendental -> bit-exact
    rng = np.random.default_rng(5)
    src = rng.random((32, 32, 4)).astype(np.float32)
    r1 = pl.color_repr(sysname, levels, sample_depth=bits[0], color_depth=bits[1],
                       bit_shift=bits[2])
    got = run_ops(gpu, src, lambda sh: sh.decode_color(r1))
    r2 = pl.color_repr(sysname, levels, sample_depth=bits[0], color_depth=bits[1],
                       bit_shift=bits[2])
    tr = pl.lib().pl_color_repr_decode(C.byref(r2), None)
    m = [tr.mat.m[i][j] for i in range(3) for j in range(3)]
    ref = orc.op_affine(src.copy(), m, list(tr.c))
    assert np.array_equal(got, ref)


CLIP_REF = [(2.5375, 2.5375, 2.5375), (4.2135, -0.3160, -0.0461),
            (-1.4911, 2.8747, -0.2552), (-0.1849, -0.0212, 2.8388)]
SAT_REF = [(2.5375, 2.5375, 2.5375), (3.1083, -0.1067, -0.0464),
           (-0.5708, 2.6442, -0.1995), (0.0000, 0.0000, 2.7834)]


@pytest.mark.parametrize("gamut,ref", [("clip", CLIP_REF), ("saturation", SAT_REF)])
def test_kat_bt2020_to_scrgb_golden(gpu, gamut, ref):
    # gpu_tests.c:582-667: BT.2020 linear -> scRGB with Display-P3 display primaries
    src = np.array([[(1, 1, 1, 1), (1, 0, 0, 1), (0, 1, 0, 1), (0, 0, 1, 1)]], np.float32)
    state = pl.ShaderObj()
    params = pl.color_map_params(tone="clip", gamut=gamut)
    params.tone_mapping_function = None  # the reference passes only .gamut_mapping
    params.lut_size = 0
    params.lut3d_size = (C.c_int * 3)(0, 0, 0)
    params.contrast_smoothness = 0
    params.gamut_constants = capi.GamutMapConstants()
    params.tone_constants = capi.ToneMapConstants()
    dst = pl.color_space("bt709", "scrgb")
    dst.hdr.prim = pl.lib().pl_raw_primaries_get(pl.PRIM["display_p3"]).contents
    out = run_ops(gpu, src, lambda sh: sh.color_map(pl.color_space("bt2020", "linear"), dst,
                                                     state, params))
    state.destroy()
    assert np.abs(out[0, :, :3] - np.array(ref, np.float32)).max() <= 1e-4, out[0, :, :3]


def test_kat_peak_detection(gpu):
    # gpu_tests.c:753-785: avg/max of the ramp vs the CPU formula
    src = ramp()
    t = gpu.tex_create(W, H, "rgba32f", src)
    state = pl.ShaderObj()
    sh = gpu.begin()
    assert sh.sample("nearest", t)
    pp = capi.PeakDetectParams(minimum_peak=0.01)  # everything else 0, like the reference test
    assert pl.lib().pl_shader_detect_peak(sh.sh, pl.color_space("unknown", "gamma22"),
                                          C.byref(state.slot), C.byref(pp)), gpu.messages[-3:]
    assert sh.compute(W, H), gpu.messages[-3:]
    hdr = capi.HdrMetadata()
    assert pl.lib().pl_get_detected_hdr_metadata(state.slot, C.byref(hdr))
    luma = (0.212639 * src[..., 0] ** 2.2 + 0.715169 * src[..., 1] ** 2.2 +
            0.072192 * src[..., 2] ** 2.2)
    pq = np.array([pl.lib().pl_hdr_rescale(pl.HDR_NORM, pl.HDR_PQ, float(v))
                   for v in luma.ravel()])
    assert abs(hdr.max_pq_y - pq.max()) <= 1e-4
    assert abs(hdr.avg_pq_y - pq.mean()) <= 1e-3
    state.destroy(); t.destroy()


# ---- cfg 4: BT.2020 PQ HDR10 -> BT.709 SDR through tone-map LUT + gamut 3DLUT ----------
def hdr_test_frame(w=64, h=48, seed=7):
    """PQ-encoded BT.2020 content: a luminance ramp with saturated colour patches."""
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 4)).astype(np.float32)
    img[..., :3] *= np.linspace(0.05, 0.75, w, dtype=np.float32)[None, :, None] / 0.75
    img[..., :3] *= 0.75       # PQ 0.75 ~ 1000 nits
    img[: h // 4, :, 1:3] *= 0.1     # saturated reds
    img[h // 4: h // 2, :, 0] *= 0.1  # cyans
    img[..., 3] = 1.0
    return img


@pytest.mark.parametrize("tone,gamut,tricubic", [
    ("spline", "perceptual", False), ("bt2390", "softclip", False),
    ("st2094-40", "relative", False), ("hable", "darken", False),
    ("mobius", "desaturate", False), ("clip", "clip", False),
    # pl_color_map_params.lut3d_tricubic (shaders/lut.c:718-760): B-spline LUT lookup
    ("spline", "perceptual", True), ("bt2390", "softclip", True)])
def test_color_map_hdr10_to_sdr_vs_oracle(gpu, tone, gamut, tricubic):
    import colormap_ref as cr
    import ref_structs as R
    src_img = hdr_test_frame()
    src = pl.color_space("bt2020", "pq", max_luma=1000.0)
    dst = pl.color_space("bt709", "bt1886")
    state = pl.ShaderObj()
    params = pl.color_map_params(tone=tone, gamut=gamut, lut3d_tricubic=tricubic)
    got = run_ops(gpu, src_img, lambda sh: sh.color_map(src, dst, state, params))
    state.destroy()

    r = cr.resolve(cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=1000.0),
                   cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]),
                   tone=tone.encode(), gamut=gamut.encode())
    if tone == "clip":
        # pl_tone_map_clip without force_lut takes the closed-form path (colorspace.c:1824)
        r["kw"].update(tone_mode=0, tone_p=(r["tone"].input_min, r["tone"].input_max, 0, 0),
                       tone_lut=None)
    r["kw"]["gamut_tricubic"] = tricubic
    ref = cr.apply(src_img.copy(), r)
    if tricubic:
        # the option must change the picture (else this parametrisation proves nothing)
        r2 = dict(r, kw=dict(r["kw"], gamut_tricubic=False))
        assert np.abs(cr.apply(src_img.copy(), r2) - ref).max() * 65535 > 4
    # --- tolerance: see util.assert_colormap_parity ---------------------------------------
    import colormap_f64 as c64
    import util
    truth, _ = c64.hdr10_to_sdr(src_img, r, 0.0)
    util.assert_colormap_parity(got, ref, truth)
    assert np.array_equal(got[..., 3], ref[..., 3])
    # sanity: the output must be a plausible SDR image, not zeros
    assert 0.05 < ref[..., :3].mean() < 0.9


def test_color_map_sdr_to_hdr_matrix_fast_path(gpu):
    # no tone/gamut work needed -> single 3x3 (colorspace.c:1782-1789)
    import colormap_ref as cr
    src_img = ramp()
    src = pl.color_space("bt709", "linear")
    dst = pl.color_space("bt2020", "linear")
    sh_list = []
    def rec(sh):
        sh.color_map(src, dst, None, pl.color_map_params(tone="clip", gamut="clip"))
        sh_list.append(sh.listing())
    got = run_ops(gpu, src_img, rec)
    assert "rgb2ipt" not in sh_list[0] and "affine" in sh_list[0], sh_list[0]
    lib = cr._cpu()
    import ref_structs as R
    a = R.m3(lib.pl_ipt_rgb2lms(lib.pl_raw_primaries_get(pl.PRIM["bt709"])))
    b = R.m3(lib.pl_ipt_lms2rgb(lib.pl_raw_primaries_get(pl.PRIM["bt2020"])))
    m = (np.array(b, np.float64).reshape(3, 3) @ np.array(a, np.float64).reshape(3, 3))
    ref = src_img.copy()
    ref[..., :3] = (src_img[..., :3].astype(np.float64) @ m.T).astype(np.float32)
    assert np.abs(got - ref).max() <= 2e-6


def _read_device(ptr, nbytes):
    import util
    hip = util.hip_runtime()
    hip.hipDeviceSynchronize()
    out = np.zeros(nbytes // 4, np.uint32)
    rc = hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(nbytes), 2)
    assert rc == 0
    return out


@pytest.mark.parametrize("use_hist", [False, True])
def test_peak_detect_buffer_vs_oracle(gpu, use_hist):
    """Raw 12-slice measurement buffer (colorspace.c:936-942) against the oracle's
    restatement of the detection shader (:1155-1353)."""
    w, h = 96, 64
    src_img = hdr_test_frame(w, h, seed=11)
    csp = pl.color_space("bt2020", "pq", max_luma=1000.0)
    t = gpu.tex_create(w, h, "rgba32f", src_img)
    d = gpu.tex_create(w, h, "rgba32f")
    state = pl.ShaderObj()
    sh = gpu.begin()
    assert sh.sample("nearest", t)
    pp = pl.peak_detect_params(percentile=99.995 if use_hist else 100.0)
    assert pl.lib().pl_shader_detect_peak(sh.sh, csp, C.byref(state.slot), C.byref(pp))
    assert sh.finish(d), gpu.messages[-3:]
    # the colour itself must pass through unchanged
    assert np.array_equal(d.download(), src_img)

    size = C.c_size_t()
    pl.lib().pl_hip_peak_buffer.restype = C.c_void_p
    ptr = pl.lib().pl_hip_peak_buffer(state.slot, C.byref(size))
    assert ptr and size.value == 816 * 4
    got = _read_device(ptr, size.value)

    csp2 = pl.color_space("bt2020", "pq", max_luma=1000.0)
    pl.lib().pl_color_space_infer(C.byref(csp2))
    mn, mx = nominal(csp2)
    ref = orc.detect_peak(src_img, pl.TRC["pq"], mn, mx, luma_coeffs(csp2.primaries),
                          black_cutoff=1.0, use_hist=use_hist)
    nwg = (w // 16) * (h // 16)
    g_cnt, r_cnt = got[0:12], ref[0:12]
    assert np.array_equal(g_cnt, r_cnt) and g_cnt.sum() == nwg
    assert np.array_equal(got[12:24], ref[12:24])           # active WGs
    # per-WG averages are integer divisions of sums of floor(16383 * PQ): native pow vs libm
    # may move single pixels across an integer boundary -> <= 1 code per WG
    assert np.abs(got[24:36].astype(np.int64) - ref[24:36]).max() <= r_cnt.max()
    assert np.abs(got[36:48].astype(np.int64) - ref[36:48]).max() <= 1
    if use_hist:
        gh, rh = got[48:].reshape(12, 64).astype(np.int64), ref[48:].reshape(12, 64)
        assert gh.sum() == rh.sum()
        assert np.abs(gh - rh).sum() <= 4     # boundary pixels may switch bin
    # and the host-side reduction must report the same metadata
    hdr = capi.HdrMetadata()
    assert pl.lib().pl_get_detected_hdr_metadata(state.slot, C.byref(hdr))
    assert abs(hdr.max_pq_y - ref[36:48].max() / 16383.0) <= 2 / 16383.0
    state.destroy(); t.destroy(); d.destroy()


def test_detected_peak_feeds_color_map(gpu):
    """Frame N: detect; frame N: colour map picks the measured peak up through the shared
    state object (renderer.c:1183-1250 pattern) -> different tone curve than static metadata."""
    w, h = 64, 48
    src_img = hdr_test_frame(w, h) * np.float32(0.8)
    src_img[..., 3] = 1
    csp = pl.color_space("bt2020", "pq", max_luma=4000.0)
    dst = pl.color_space("bt709", "bt1886")
    state = pl.ShaderObj()
    static = run_ops(gpu, src_img, lambda sh: sh.color_map(csp, dst, state, None))
    t = gpu.tex_create(w, h, "rgba32f", src_img)
    sh = gpu.begin()
    assert sh.sample("nearest", t)
    assert sh.detect_peak(csp, state, smoothing_period=0.0)
    assert sh.compute(w, h)
    listing = []
    def rec(sh):
        sh.color_map(csp, dst, state, None)
        listing.append(sh.listing())
    dynamic = run_ops(gpu, src_img, rec)
    assert not np.array_equal(static, dynamic)
    # brighter, since the measured peak (~600 nits) is far below the mastering peak
    assert dynamic[..., :3].mean() > static[..., :3].mean()
    state.destroy(); t.destroy()


@pytest.mark.parametrize("vision", [v for v in pl.VISION if v != "normal"])
@pytest.mark.parametrize("trc", ["srgb", "pq"])
def test_cone_distort_vs_oracle(gpu, vision, trc):
    # pl_shader_cone_distort (shaders/colorspace.c:2040-2064) = linearize -> cone matrix ->
    # delinearize; the matrix is pinned bit-exact against the reference in test_tier0_ref.py
    csp = pl.color_space("bt2020" if trc == "pq" else "bt709", trc)
    pl.lib().pl_color_space_infer(C.byref(csp))
    mn, mx = nominal(csp)
    luma = luma_coeffs(csp.primaries)
    cp = pl.cone_params(vision)
    m = pl.lib().pl_get_cone_matrix(C.byref(cp), pl.lib().pl_raw_primaries_get(csp.primaries))
    mat = np.array([[m.m[i][j] for j in range(3)] for i in range(3)], np.float32)
    rng = np.random.default_rng(11)
    src = rng.random((32, 32, 4)).astype(np.float32)
    got = run_ops(gpu, src, lambda sh: sh.cone_distort(csp, cp))
    ref = orc.linearize(src.copy(), int(csp.transfer), mn, mx, luma)
    lin = ref[..., :3].astype(np.float64)
    ref[..., :3] = (lin @ mat.astype(np.float64).T).astype(np.float32)
    ref = orc.delinearize(ref, int(csp.transfer), mn, mx, luma)
    assert np.abs(got - src)[..., :3].max() > 0.01   # it did something
    if trc == "pq":
        # the PQ OETF has an unbounded slope at black, and the cone matrix cancels to near-black
        # values: compare in linear light. The PQ EOTF alone is good to the reference's HDR
        # epsilon (1e-4 of the range, test_transfer_vs_oracle); the matrix rows sum |coeffs| up
        # to ~2.5 on top of it, hence 3e-4.
        got, ref = (orc.linearize(x.copy(), int(csp.transfer), mn, mx, luma) for x in (got, ref))
        assert np.abs(got - ref).max() <= 3e-4 * mx, np.abs(got - ref).max()
    else:
        # <= 2 LSB of 16 bit on the encoded signal (the matrix runs in f32 FMAs on the GPU)
        assert np.abs(got - ref).max() <= 2 / 65535.0, np.abs(got - ref).max()


def test_cone_distort_normal_is_noop(gpu):
    csp = pl.color_space("bt709", "srgb")
    src = ramp()
    got = run_ops(gpu, src, lambda sh: sh.cone_distort(csp, pl.cone_params("normal")))
    assert np.array_equal(got, src)


def test_peak_result_timing_and_state_lifecycle(gpu):
    """pl_peak_detect_params.allow_delayed and the life cycle of the measurement
    (src/shaders/colorspace.c:1041-1150):
      * asking for the result inside the shader that measures: a usage error unless
        allow_delayed (then: silently the previous result);
      * a recorded-but-never-dispatched measurement is abandoned by the next one;
      * consecutive frames are low-passed (smoothing_period), a scene cut snaps;
      * pl_reset_detected_peak forgets the result, the next frame starts over."""
    w, h = 64, 48
    dim = hdr_test_frame(w, h) * np.float32(0.5)
    bright = np.clip(hdr_test_frame(w, h) * np.float32(1.15), 0, 1)
    for im in (dim, bright):
        im[..., 3] = 1
    csp = pl.color_space("bt2020", "pq")
    t_dim, t_bright = gpu.tex_create(w, h, "rgba32f", dim), gpu.tex_create(w, h, "rgba32f", bright)
    state = pl.ShaderObj()
    hdr = capi.HdrMetadata()
    get = lambda: bool(pl.lib().pl_get_detected_hdr_metadata(state.slot, C.byref(hdr)))

    def measure(tex, dispatch=True, **kw):
        sh = gpu.begin()
        assert sh.sample("nearest", tex)
        assert sh.detect_peak(csp, state, **kw)
        return sh if not dispatch else sh.compute(w, h)

    # inside the measuring shader, without allow_delayed: warning, no result
    n_warn = sum("usage error" in m for _, m in gpu.messages)
    sh = measure(t_dim, dispatch=False, smoothing_period=0.0)
    assert not get()
    assert sum("usage error" in m for _, m in gpu.messages) == n_warn + 1
    assert sh.compute(w, h)
    assert get()
    first = (hdr.max_pq_y, hdr.avg_pq_y)
    assert 0.3 < first[0] < 0.5 and 0.0 < first[1] < first[0]

    # same, with allow_delayed: no warning, the previous result stays in force
    sh = measure(t_bright, dispatch=False, smoothing_period=0.0, allow_delayed=True)
    assert get() and (hdr.max_pq_y, hdr.avg_pq_y) == first
    assert sum("usage error" in m for _, m in gpu.messages) == n_warn + 1
    sh.abort()
    # ... and the abandoned measurement does not poison the next one
    assert measure(t_bright, smoothing_period=0.0, allow_delayed=True)
    gpu.finish()
    assert get() and hdr.max_pq_y > first[0] + 0.2
    bright_peak = hdr.max_pq_y

    # low-pass: period 20 moves 1 - exp(-1/20) of the way per frame (below the scene-cut
    # thresholds when they are disabled)
    assert measure(t_dim, smoothing_period=0.0, scene_threshold_low=0.0, scene_threshold_high=0.0)
    assert get()
    lo = hdr.max_pq_y
    assert abs(lo - first[0]) < 1e-6
    kw = dict(smoothing_period=20.0, scene_threshold_low=0.0, scene_threshold_high=0.0)
    pl.lib().pl_reset_detected_peak(state.slot)
    assert not get()
    assert measure(t_dim, **kw) and get() and abs(hdr.max_pq_y - lo) < 1e-6     # first sample: adopted
    assert measure(t_bright, **kw) and get()
    step = 1.0 - np.exp(-1.0 / 20.0)
    assert abs(hdr.max_pq_y - (lo + step * (bright_peak - lo))) < 1e-5
    # with the default thresholds the same jump is a scene cut: snaps (almost) all the way
    pl.lib().pl_reset_detected_peak(state.slot)
    kw = dict(smoothing_period=20.0)
    assert measure(t_dim, **kw) and get()
    assert measure(t_bright, **kw) and get()
    assert abs(hdr.max_pq_y - bright_peak) < 1e-3
    state.destroy(); t_dim.destroy(); t_bright.destroy()
