"""k_polar_mx -- the EWA 2x upscale as a tile contraction on the f16 matrix pipe (the library's
default for that geometry) -- against k_polar_pp, the sequential-fma kernel that is bit-exact
with the oracle (tests/test_gpu_fullsize.py, PL_HIP_POLAR_MFMA=0), and against the oracle itself.

Parity statement (north star: +-1 code of 16 bits per channel, dither index bit-exact):
  * before any quantisation (rgba32f target): |mx - pp| <= 4e-6 of full scale (a quarter of a
    16-bit code): f16 hi + lo weight halves, exact products, fp32 sums in another order;
  * rgba16 target: never more than one code apart, identical on the bulk;
  * 10-bit dithered target: never more than one 10-bit step apart, and only where the value
    lies within that quarter code of a dither threshold (a fraction of a percent).
Same geometries as the bit-exact tests: the BASELINE frame (chirp and white noise), odd sizes
whose edge tiles are clipped, sampled alpha, an rgba16hf source (no fused pass), the HDR colour
map behind it (full interpreter in the epilogue), flipped and offset targets.
"""
import os

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu

TEN_BIT = dict(sample_depth=16, color_depth=10, bit_shift=6)


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def dither10():
    return dict(dither_params=capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0),
                disable_dither_gamma_correction=True)


def render(img, dw, dh, params, mfma, src_fmt="rgba16", dst_fmt="rgba16", ten_bit=False,
           components=3, image_kw=None, target_kw=None, expect_mx=None):
    """one frame through pl_render_image on a fresh backend whose log is kept (the kernel choice
    is logged at debug level)"""
    with env(PL_HIP_POLAR_MFMA="1" if mfma else "0"):
        with pl.HipGpu(0, log_level=5) as g:
            sh, sw = img.shape[:2]
            src = g.tex_create(sw, sh, src_fmt, img)
            dst = g.tex_create(dw, dh, dst_fmt)
            rr = pl.Renderer(g)
            util.srand(1)
            image = pl.frame(src, components=components, **(image_kw or {}))
            target = pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT) if ten_bit else None,
                              **(target_kw or {}))
            assert rr.render(image, target, params), g.messages[-4:]
            assert rr.errors() == 0
            out = dst.download()
            used = any("polar on the matrix pipe" in m for _, m in g.messages)
            rr.destroy(); src.destroy(); dst.destroy()
    if expect_mx is not None:
        assert used == expect_mx, "kernel choice: matrix pipe %s" % used
    return out


def ewa(**kw):
    return pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"), **kw)


def assert_codes(mx, pp, step=1, max_frac=0.15):
    d = np.abs(mx.astype(np.int64) - pp.astype(np.int64))
    if d.max() > step:
        ys, xs = np.nonzero((d > step).any(axis=-1))
        print("samples more than %d apart: %d, rows %d..%d, columns %d..%d; distinct rows %d, distinct columns %d; "
              "x mod 12: %s, y mod 12: %s" % (step, len(ys), ys.min(), ys.max(), xs.min(), xs.max(),
                                             len(np.unique(ys)), len(np.unique(xs)),
                                             np.bincount(xs % 12, minlength=12), np.bincount(ys % 12, minlength=12)))
    assert d.max() <= step, (int(d.max()), int((d > step).sum()))
    assert (d > 0).mean() <= max_frac, float((d > 0).mean())
    return float((d > 0).mean())


@pytest.mark.parametrize("size", [(96, 64), (130, 77), (1920, 1080)])
@pytest.mark.parametrize("content", ["chirp", "noise"])
def test_mx_vs_reference_kernel_and_oracle(size, content):
    sw, sh = size
    img = util.chirp_rgba16(sw, sh) if content == "chirp" else util.random_rgba16(sw, sh, seed=3)
    dw, dh = 2 * sw, 2 * sh
    # -- unquantised
    f_mx = render(img, dw, dh, ewa(), True, dst_fmt="rgba32f", expect_mx=True)
    f_pp = render(img, dw, dh, ewa(), False, dst_fmt="rgba32f", expect_mx=False)
    err = np.abs(f_mx[..., :3].astype(np.float64) - f_pp[..., :3])
    print("k_polar_mx vs k_polar_pp %dx%d %s: max |diff| %.2e, mean %.2e (one 16-bit code = 1.5e-5)"
          % (sw, sh, content, err.max(), err.mean()))
    assert err.max() <= 4e-6, err.max()
    assert np.array_equal(f_mx[..., 3], f_pp[..., 3])
    # -- 16-bit target, against the reference kernel and against the oracle
    q_mx = render(img, dw, dh, ewa(), True, expect_mx=True)
    q_pp = render(img, dw, dh, ewa(), False)
    assert_codes(q_mx, q_pp)
    a = orc.tex_decode(img, "rgba16")
    a[..., 3] = 1.0
    a = orc.op_quant_f16(a)
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
    ref16 = orc.tex_encode(orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7), "rgba16")
    assert np.array_equal(q_pp, ref16)      # (the reference variant IS the oracle, bit for bit)
    assert_codes(q_mx, ref16)
    # -- 10-bit blue-noise dither: a step only where a threshold lies within the difference
    d_mx = render(img, dw, dh, ewa(**dither10()), True, ten_bit=True, expect_mx=True)
    d_pp = render(img, dw, dh, ewa(**dither10()), False, ten_bit=True)
    assert np.all((d_mx & 63 == 0) | (d_mx == 65535))   # (samples dithered beyond 1: clamped by the store)
    frac = assert_codes(d_mx >> 6, d_pp >> 6, step=1, max_frac=0.004)
    print("  10-bit dithered frames differ on %.5f of the samples, by one step" % frac)


def test_mx_sampled_alpha_and_f16_source():
    """4-channel tile; an rgba16hf source is copied into the planes bit for bit (no fused pass)"""
    sw, sh = 120, 70
    rng = np.random.default_rng(9)
    img = rng.random((sh, sw, 4), dtype=np.float32).astype(np.float16)
    for comps in (3, 4):
        mx = render(img, 2 * sw, 2 * sh, ewa(), True, src_fmt="rgba16hf", dst_fmt="rgba32f",
                    components=comps, expect_mx=True)
        pp = render(img, 2 * sw, 2 * sh, ewa(), False, src_fmt="rgba16hf", dst_fmt="rgba32f",
                    components=comps)
        err = np.abs(mx.astype(np.float64) - pp)
        assert err.max() <= 4e-6, (comps, err.max())
    img16 = util.random_rgba16(sw, sh, seed=4)
    mx = render(img16, 2 * sw, 2 * sh, ewa(**dither10()), True, ten_bit=True, components=4,
                image_kw=dict(repr_=pl.color_repr("rgb", "full", alpha="independent")), expect_mx=True)
    pp = render(img16, 2 * sw, 2 * sh, ewa(**dither10()), False, ten_bit=True, components=4,
                image_kw=dict(repr_=pl.color_repr("rgb", "full", alpha="independent")))
    assert_codes(mx >> 6, pp >> 6, max_frac=0.01)


def test_mx_offset_and_flipped_targets():
    """target crop (offset store, gl_FragCoord offset of the dither) and a flipped target"""
    sw, sh = 100, 60
    img = util.chirp_rgba16(sw, sh)
    for crop in [(16.0, 8.0, 216.0, 128.0), (200.0, 120.0, 0.0, 0.0)]:
        kw = dict(target_kw=dict(crop=crop), ten_bit=True)
        mx = render(img, 232, 140, ewa(**dither10()), True, **kw)
        pp = render(img, 232, 140, ewa(**dither10()), False, **kw)
        assert_codes(mx >> 6, pp >> 6, max_frac=0.01)


def test_mx_not_used_where_it_does_not_apply():
    """non-integer and anisotropic ratios keep the phase-class kernel; both switches give the same
    frame there"""
    sw, sh = 96, 64
    img = util.chirp_rgba16(sw, sh)
    for dw, dh in ((168, 112), (192, 96), (240, 160)):
        a = render(img, dw, dh, ewa(), True, expect_mx=False)
        b = render(img, dw, dh, ewa(), False, expect_mx=False)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("ratio", [3, 4, 1.5])
@pytest.mark.parametrize("size", [(80, 48), (131, 77), (1280, 720)])
@pytest.mark.parametrize("content", ["chirp", "noise"])
def test_mxr_integer_upscale_vs_reference_kernel_and_oracle(ratio, size, content):
    """k_polar_mxr -- the EWA upscale by exactly 3, 4 or 3 : 2 on the matrix pipe (720p -> 4K is 3x,
    1440p -> 4K is 3 : 2) -- against k_polar_pp, which IS the oracle bit for bit, at sizes whose edge tiles are clipped and
    at the real one. Same statement as the 2x kernel: <= 1 code of 16 bits, identical on the bulk;
    a 10-bit dithered frame differs by one step on a fraction of a percent, and every sample of
    it is the dither of the frame's own pre-dither value (index path exact)."""
    from test_gpu_metric import dither_consistency
    sw, sh = size
    if ratio == 4 and sw > 1000:
        sw, sh = 960, 540
    if ratio == 1.5:
        sw, sh = {80: (80, 48), 131: (130, 78), 1280: (2560, 1440)}[sw]
    img = util.chirp_rgba16(sw, sh) if content == "chirp" else util.random_rgba16(sw, sh, seed=3)
    dw, dh = int(ratio * sw), int(ratio * sh)
    q_mx = render(img, dw, dh, ewa(), True, expect_mx=True)
    q_pp = render(img, dw, dh, ewa(), False, expect_mx=False)
    frac = assert_codes(q_mx, q_pp)
    print("k_polar_mxr %gx %dx%d %s: 16-bit frames differ on %.4f of the samples, by one code" %
          (ratio, sw, sh, content, frac))
    if sw < 1000:
        a = orc.tex_decode(img, "rgba16")
        a[..., 3] = 1.0
        a = orc.op_quant_f16(a)
        w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
        ref16 = orc.tex_encode(orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7), "rgba16")
        assert np.array_equal(q_pp, ref16)
        assert_codes(q_mx, ref16)
    d_mx = render(img, dw, dh, ewa(**dither10()), True, ten_bit=True, expect_mx=True)
    d_pp = render(img, dw, dh, ewa(**dither10()), False, ten_bit=True)
    assert_codes(d_mx >> 6, d_pp >> 6, step=1, max_frac=0.004)
    assert dither_consistency(d_mx, q_mx, util.blue_noise(pl)) == 0.0


@pytest.mark.parametrize("ratio", [3, 4, 1.5])
def test_mxr_hdr_colour_map_epilogue_and_f16_source(ratio):
    """the HDR map chain behind the 3x / 4x upscale (720p HDR10 -> 4K SDR: the metric's frame from a
    720p source) and an rgba16hf source (no fused decode): within the conditioning of the colour
    map of the sequential-fma kernel, as for 2x (test_mx_hdr_colour_map_epilogue)"""
    from test_gpu_fullsize import hdr_frame16
    sw, sh = 150, 84
    img = hdr_frame16(sw, sh)
    hdr = pl.color_space("bt2020", "pq", max_luma=1000.0)
    sdr = pl.color_space("bt709", "bt1886")
    params = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=None,
                              peak_detect_params=pl.peak_detect_params(percentile=99.995))
    kw = dict(image_kw=dict(color=hdr), target_kw=dict(color=sdr))
    dw, dh = int(ratio * sw), int(ratio * sh)
    mx = render(img, dw, dh, params, True, expect_mx=True, **kw)
    pp = render(img, dw, dh, params, False, **kw)
    d = np.abs(mx[..., :3].astype(np.int64) - pp[..., :3])
    print("HDR epilogue %gx: |mxr - pp| codes: median %.1f p99 %.1f max %d" %
          (ratio, np.median(d), np.quantile(d, 0.99), d.max()))
    assert np.quantile(d, 0.5) <= 1 and np.quantile(d, 0.99) <= 4 and d.max() <= 64
    assert np.array_equal(mx[..., 3], pp[..., 3])
    rng = np.random.default_rng(9)
    f16 = rng.random((sh, sw, 4), dtype=np.float32).astype(np.float16)
    a = render(f16, dw, dh, ewa(), True, src_fmt="rgba16hf", expect_mx=True)
    b = render(f16, dw, dh, ewa(), False, src_fmt="rgba16hf")
    assert_codes(a, b)


def test_mx_hdr_colour_map_epilogue():
    """the metric's second pass at small size: f16 intermediate -> polar on the matrix pipe ->
    PQ linearize, tone + gamut map, BT.1886, dither in the epilogue (full interpreter).
    Pre-dither the two kernels must agree to the conditioning of the colour map."""
    from test_gpu_fullsize import hdr_frame16
    sw, sh = 160, 90
    img = hdr_frame16(sw, sh)
    hdr = pl.color_space("bt2020", "pq", max_luma=1000.0)
    sdr = pl.color_space("bt709", "bt1886")
    params = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=None,
                              peak_detect_params=pl.peak_detect_params(percentile=99.995))
    kw = dict(image_kw=dict(color=hdr), target_kw=dict(color=sdr))
    mx = render(img, 2 * sw, 2 * sh, params, True, expect_mx=True, **kw)
    pp = render(img, 2 * sw, 2 * sh, params, False, **kw)
    d = np.abs(mx[..., :3].astype(np.int64) - pp[..., :3])
    print("HDR epilogue: |mx - pp| codes: median %.1f p99 %.1f max %d" %
          (np.median(d), np.quantile(d, 0.99), d.max()))
    # a quarter code into the PQ EOTF and the inverse display gamma: a few codes at most,
    # identical or adjacent on the bulk
    assert np.quantile(d, 0.5) <= 1 and np.quantile(d, 0.99) <= 4 and d.max() <= 64
    assert np.array_equal(mx[..., 3], pp[..., 3])


def ewa_down(**kw):
    return pl.render_params("fast", downscaler=pl.filter_config("ewa_lanczos"), disable_linear_scaling=True, **kw)


@pytest.mark.parametrize("size", [(128, 64), (300, 170), (2048, 1024), (3840, 2160)])
def test_mxd_2to1_downscale_vs_reference_kernel(size):
    """k_polar_mxd -- the widened EWA 2 : 1 downscale (148 taps) on the matrix pipe, rgba16hf source
    to rgba16hf target, the pass of BASELINE configs[4] -- against k_polar_pp (bit-exact with the
    oracle, test_gpu_fullsize.py): the results are fp32 sums in another order rounded to half
    precision, so they are identical on the bulk and never more than one f16 ulp apart."""
    dw, dh = size
    rng = np.random.default_rng(11)
    img = (0.05 + 0.9 * rng.random((2 * dh, 2 * dw, 4), dtype=np.float32)).astype(np.float16)
    img[..., 3] = 1.0
    mx = render(img, dw, dh, ewa_down(), True, src_fmt="rgba16hf", dst_fmt="rgba16hf", expect_mx=True)
    pp = render(img, dw, dh, ewa_down(), False, src_fmt="rgba16hf", dst_fmt="rgba16hf", expect_mx=False)
    assert np.array_equal(mx[..., 3], pp[..., 3])
    a, b = mx[..., :3].view(np.uint16).astype(np.int64), pp[..., :3].view(np.uint16).astype(np.int64)
    d = np.abs(a - b)       # (positive halves: the code distance is the ulp distance)
    print("k_polar_mxd vs k_polar_pp %dx%d: %.4f of the samples differ, max %d f16 ulp"
          % (dw, dh, (d > 0).mean(), d.max()))
    assert d.max() <= 1 and (d > 0).mean() < 0.02, (int(d.max()), float((d > 0).mean()))
    assert pp[..., :3].astype(np.float32).std() > 0.01


def test_mxd_cropped_source_and_clipped_tiles():
    """a source rectangle that starts inside the plane (even and odd offsets: the tile's texel
    pairs are then unaligned) and an output whose size is no multiple of the 64 x 32 tile; a
    fractional offset has other phases than 1/2 and must stay on k_polar_pp"""
    dw, dh = 200, 90
    rng = np.random.default_rng(12)
    img = (0.05 + 0.9 * rng.random((2 * dh + 40, 2 * dw + 40, 4), dtype=np.float32)).astype(np.float16)
    img[..., 3] = 1.0
    for x0, y0, expect in ((2, 4, True), (7, 13, True), (2.5, 4.0, False)):
        kw = dict(image_kw=dict(crop=(x0, y0, x0 + 2 * dw, y0 + 2 * dh)))
        mx = render(img, dw, dh, ewa_down(), True, src_fmt="rgba16hf", dst_fmt="rgba16hf", **kw)
        pp = render(img, dw, dh, ewa_down(), False, src_fmt="rgba16hf", dst_fmt="rgba16hf", **kw)
        d = np.abs(mx[..., :3].view(np.uint16).astype(np.int64) - pp[..., :3].view(np.uint16).astype(np.int64))
        if expect:
            assert 0 < (d > 0).mean() < 0.02 and d.max() <= 1, (x0, y0, int(d.max()), float((d > 0).mean()))
        else:
            assert d.max() == 0, (x0, y0)


@pytest.mark.parametrize("size", [(200, 90), (1920, 1080)])
def test_mxd_unorm_source_and_rgba16_target(size):
    """The other variants of k_polar_mxd: an rgba16 source (decoded and rounded to f16 while it is
    staged = the fused PASS A) and an rgba16 target behind the fused epilogue, 16-bit and 10-bit
    dithered -- the plain SDR 2 : 1 EWA downscale (4K -> 1080p) -- against k_polar_pp: never more
    than one code / one 10-bit step apart, identical on the bulk."""
    dw, dh = size
    img = util.random_rgba16(2 * dw, 2 * dh, seed=5)
    q_mx = render(img, dw, dh, ewa_down(), True, expect_mx=True)
    q_pp = render(img, dw, dh, ewa_down(), False, expect_mx=False)
    frac = assert_codes(q_mx, q_pp)
    f_mx = render(img, dw, dh, ewa_down(), True, dst_fmt="rgba16hf", expect_mx=True)
    f_pp = render(img, dw, dh, ewa_down(), False, dst_fmt="rgba16hf")
    d = np.abs(f_mx[..., :3].view(np.uint16).astype(np.int64) - f_pp[..., :3].view(np.uint16).astype(np.int64))
    assert d.max() <= 1 and (d > 0).mean() < 0.02
    d_mx = render(img, dw, dh, ewa_down(**dither10()), True, ten_bit=True, expect_mx=True)
    d_pp = render(img, dw, dh, ewa_down(**dither10()), False, ten_bit=True)
    assert np.all((d_mx & 63 == 0) | (d_mx == 65535))
    frac10 = assert_codes(d_mx >> 6, d_pp >> 6, step=1, max_frac=0.004)
    print("k_polar_mxd rgba16 -> rgba16 %dx%d: %.4f of the 16-bit samples one code apart; 10-bit dithered: %.5f"
          % (dw, dh, frac, frac10))


# ---- k_polar_mxp: the same contraction on persistent workgroups (round 6) ------------------------
@pytest.mark.parametrize("size", [(96, 64), (130, 77), (1000, 562), (1920, 1080)])
@pytest.mark.parametrize("case", ["dither10", "rgba16", "f16_source", "flipped", "offset", "bayer16"])
def test_mxp_persistent_kernel_against_k_polar_mx(size, case):
    """k_polar_mxp (the library's default for RGB tiles behind the fused epilogue: 2 x CUs resident
    workgroups walking the tiles of their XCD's band, the next tile's texels prefetched during a
    tile's contraction, B fragments and dither matrix in LDS once per workgroup, clipped lanes
    storing to a sink) against k_polar_mx, one workgroup per tile (PL_HIP_MX_PERSIST=0): the same
    arithmetic, bit for bit. Sizes: fewer tiles than workgroups, clipped edge tiles, several tiles
    per workgroup (1080p -> 4K: 1020 tiles on 512 groups) with bands that do not divide; targets:
    10-bit dithered (blue noise 64 x 64, Bayer 16 x 16), plain rgba16, flipped, offset by a
    multiple of 8 rows (the dither column stays contiguous) and by an odd amount (k_polar_mx
    keeps the pass: logged kernel choice is still the matrix pipe)."""
    sw, sh = size
    rng = np.random.default_rng(sw)
    kw, params, ten, fmt = {}, ewa(), False, "rgba16"
    dw, dh = 2 * sw, 2 * sh
    img = rng.integers(0, 65536, (sh, sw, 4), dtype=np.uint16)
    if case == "dither10":
        params, ten = ewa(**dither10()), True
    elif case == "bayer16":
        params = ewa(dither_params=capi.DitherParams(method=pl.DITHER_ORDERED_LUT, lut_size=4, transfer=0),
                     disable_dither_gamma_correction=True)
        ten = True
    elif case == "f16_source":
        img = (img.astype(np.float32) / 65535.0).astype(np.float16)
        fmt = "rgba16hf"
    elif case == "flipped":
        kw = dict(target_kw=dict(crop=(2.0 * sw, 2.0 * sh, 0.0, 0.0)))
    elif case == "offset":
        # target 24 columns / 16 + 5 rows larger: one crop on the 8-row grid, one off it
        params, ten = ewa(**dither10()), True
        dw, dh = 2 * sw + 24, 2 * sh + 21
    crops = [None]
    if case == "offset":
        crops = [(8.0, 16.0, 8.0 + 2 * sw, 16.0 + 2 * sh), (3.0, 5.0, 3.0 + 2 * sw, 5.0 + 2 * sh)]
    for crop in crops:
        k2 = dict(kw)
        if crop:
            k2 = dict(target_kw=dict(crop=crop))
        outs = []
        for persist in ("1", "0"):
            with env(PL_HIP_MX_PERSIST=persist):
                outs.append(render(img, dw, dh, params, True, src_fmt=fmt, ten_bit=ten, expect_mx=True, **k2))
        a, b = outs
        assert a[..., :3].std() > 1000
        assert np.array_equal(a, b), util.diff_stats(a, b)
