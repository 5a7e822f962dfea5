"""pl_render_image at the BASELINE.json geometries (configs[1..4]), end to end against the
oracle composed stage by stage the way the renderer composes the passes (reference
src/renderer.c:1553-1962, 1964-2087, 2157-2279, 2586-2964; pattern of the reference's own
pl_render_tests, src/tests/gpu_tests.c:1155-1216).

The GPU launches are the full-size ones (1080p->4K phase-class tables and tile map, 4K colour
map, 8K deband / widened polar), the oracle runs the whole frame too (it is OpenMP-parallel
over independent output rows). Bit-exact wherever no transcendental is involved (cfg 2, cfg 3,
the 8K->4K polar pass); stated tolerances where pow/exp/log are (cfg 4, cfg 5), and there the
GPU is additionally held against a float64 evaluation of the same formulas.

Every end-to-end case has a small twin (same code path, ~1e4 pixels) so that a failure at full
size can be localised quickly.
"""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu

P1080, P4K, P8K = (1920, 1080), (3840, 2160), (7680, 4320)


@pytest.fixture()
def rr(gpu):
    r = pl.Renderer(gpu)
    yield r
    r.destroy()


def normalize(repr_):
    fn = pl.lib().pl_color_repr_normalize
    fn.restype = C.c_float
    return fn(C.byref(repr_))


def chirp16(w, h):
    """util.chirp_rgba16 in float32 (fast enough for 8K)"""
    yc, xc = h / 2.0, w / 2.0
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    r2 = (x - np.float32(xc)) ** 2 + (y - np.float32(yc)) ** 2
    phi = (1 + 5 ** 0.5) / 2
    f = 0.1 * np.pi * 0.5 / np.sqrt(xc * xc + yc * yc)
    out = np.empty((h, w, 4), np.uint16)
    for k in range(3):
        out[..., k] = np.rint((0.5 * np.sin(np.float32(f / phi ** k) * r2) + 0.5) * 65535)
    out[..., 3] = 65535
    return out


def hdr_frame16(w, h, seed=1):
    """the chirp read as PQ code values (scaled so that it spans ~0..1000 nits),
    plus a seeded 0.1 % sprinkle of code 0.9 highlights so that max != percentile."""
    img = chirp16(w, h).astype(np.float32)
    img[..., :3] *= np.float32(0.75)
    rng = np.random.default_rng(seed)
    n = max(w * h // 1000, 1)
    ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
    img[ys, xs, :3] = 0.9 * 65535
    return np.rint(img).astype(np.uint16)


# ---- cfg 2 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [(96, 54), P1080])
def test_cfg2_bilinear_1080p_to_4k(gpu, rr, size):
    sw, sh = size
    img = chirp16(sw, sh)
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(2 * sw, 2 * sh, "rgba16")
    assert rr.render(pl.frame(src, components=3), pl.frame(dst, components=4, mapping=[0, 1, 2, 3]),
                     pl.render_params("fast")), gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    ref = orc.sample_simple(orc.tex_decode(img, "rgba16"), orc.S_BILINEAR, 2 * sw, 2 * sh)
    ref[..., 3] = 1.0
    ref16 = orc.tex_encode(ref, "rgba16")
    util.assert_polar_equal(got, ref16)
    src.destroy(); dst.destroy()


# ---- cfg 3 -------------------------------------------------------------------------------------
def cfg3_oracle(img, dw, dh, depth, shift, matrix):
    sh_, sw = img.shape[:2]
    a = orc.tex_decode(img, "rgba16")
    a[..., 3] = 1.0
    a = orc.op_quant_f16(a)                                 # PASS A: rgba16hf FBO
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
    out = orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7)
    orc.dither(out, matrix, depth)
    scale = np.float32(normalize(pl.color_repr("rgb", "full", sample_depth=16, color_depth=depth,
                                               bit_shift=shift)))
    out[...] = out * (np.float32(1.0) / scale)
    return orc.tex_encode(out, "rgba16")


@pytest.mark.parametrize("size", [(96, 64), P1080])
def test_cfg3_ewa_lanczos_1080p_to_4k_dither10(gpu, rr, size):
    """The headline launch: 23 x 21 phase classes, host-built XCD tile map, 3 rows per lane."""
    sw, sh = size
    img = chirp16(sw, sh)
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(2 * sw, 2 * sh, "rgba16")
    target = pl.frame(dst, repr_=pl.color_repr("rgb", "full", sample_depth=16, color_depth=10,
                                                bit_shift=6))
    params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"),
                              dither_params=capi.DitherParams(method=pl.DITHER_BLUE_NOISE,
                                                              lut_size=6, transfer=0),
                              disable_dither_gamma_correction=True)
    util.srand(1)
    assert rr.render(pl.frame(src, components=3), target, params), gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    ref16 = cfg3_oracle(img, 2 * sw, 2 * sh, 10, 6, util.blue_noise(pl))
    # (alpha = 1.0 goes through `color *= 1/scale` too: 1023 << 6)
    util.assert_polar_equal(got, ref16, step=64)       # (one 10-bit step)
    assert np.all(got[..., 3] == 1023 << 6)
    # a second frame through the same renderer (cached tables / LUTs) must not change anything
    assert rr.render(pl.frame(src, components=3), target, params)
    assert np.array_equal(dst.download(), got)
    src.destroy(); dst.destroy()


def test_cfg3_random_content_full_size(gpu, rr):
    """Same launch on white noise (every tap matters, no smooth-signal cancellation)."""
    sw, sh = P1080
    img = util.random_rgba16(sw, sh, seed=3)
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(2 * sw, 2 * sh, "rgba16")
    params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"))
    assert rr.render(pl.frame(src, components=3), pl.frame(dst), params), gpu.messages[-4:]
    got = dst.download()
    a = orc.tex_decode(img, "rgba16")
    a[..., 3] = 1.0
    a = orc.op_quant_f16(a)
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
    ref16 = orc.tex_encode(orc.sample_polar(a, w, r, rz, 2 * sw, 2 * sh, mask=0x7), "rgba16")
    util.assert_polar_equal(got, ref16)
    src.destroy(); dst.destroy()


# ---- cfg 4 -------------------------------------------------------------------------------------
def inferred(csp_src, csp_dst):
    a, b = capi.ColorSpace(), capi.ColorSpace()
    C.memmove(C.byref(a), C.byref(csp_src), C.sizeof(a))
    C.memmove(C.byref(b), C.byref(csp_dst), C.sizeof(b))
    pl.lib().pl_color_space_infer_map(C.byref(a), C.byref(b))
    return a, b


def resolve_with_peak(meta, tone=b"spline", gamut=b"perceptual"):
    """colormap_ref.resolve for BT.2020 PQ -> BT.709 BT.1886 with the measured scene peak / average
    in the source metadata (what pl_shader_color_map_ex sees after hdr_update_peak)."""
    import colormap_ref as cr
    src = cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"])
    if meta is not None:
        src.hdr.max_pq_y, src.hdr.avg_pq_y = meta.max_pq_y, meta.avg_pq_y
    dst = cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"])
    return cr.resolve(src, dst, tone=tone, gamut=gamut)


def colormap_tolerance(got16, ref16, truth=None, sel=None, **kw):
    """util.assert_colormap_parity on rgba16 images; `truth` (float, [0, 1]) covers the flat pixel
    indices `sel`."""
    if truth is None:
        util.assert_colormap_parity(got16, ref16, None, scale=1.0)
        return
    g = got16[..., :3].reshape(-1, 3)[sel]
    o = ref16[..., :3].reshape(-1, 3)[sel]
    util.assert_colormap_parity(g, o, truth.reshape(-1, truth.shape[-1])[:, :3], scale=1.0, **kw)
    util.assert_colormap_parity(got16, ref16, None, scale=1.0)


@pytest.mark.parametrize("size", [(128, 80), P4K])
def test_cfg4_hdr10_4k_peak_detect_tone_map(gpu, rr, size):
    """4K BT.2020 PQ -> BT.709 SDR: A: plane -> peak measurement + rgba16hf FBO; B: colour map
    (PQ -> IPT -> tone LUT -> gamut 3D-LUT -> BT.1886) -> target."""
    import colormap_f64 as c64
    import colormap_ref as cr
    from test_gpu_color import luma_coeffs, nominal
    w, h = size
    img = hdr_frame16(w, h)
    src = gpu.tex_create(w, h, "rgba16", img)
    dst = gpu.tex_create(w, h, "rgba16")
    hdr = pl.color_space("bt2020", "pq")
    sdr = pl.color_space("bt709", "bt1886")

    # -- the measurement pass by hand (the renderer consumes its own buffer): integer buffer
    tex = orc.tex_decode(img, "rgba16")
    tex[..., 3] = 1.0
    hdr_i, sdr_i = inferred(hdr, sdr)
    fbo = gpu.tex_create(w, h, "rgba16hf")
    state = pl.ShaderObj()
    a = gpu.begin()
    assert a.sample("direct", src, components=3)
    pp = capi.PeakDetectParams.in_dll(pl.lib(), "pl_peak_detect_default_params")
    assert pl.lib().pl_shader_detect_peak(a.sh, hdr_i, C.byref(state.slot), C.byref(pp))
    assert a.finish(fbo), gpu.messages[-3:]
    size_ = C.c_size_t()
    pl.lib().pl_hip_peak_buffer.restype = C.c_void_p
    ptr = pl.lib().pl_hip_peak_buffer(state.slot, C.byref(size_))
    assert ptr and size_.value == 816 * 4
    from test_gpu_color import _read_device
    buf = _read_device(ptr, size_.value)
    mn, mx = nominal(hdr_i)
    pad_w, pad_h = -(-w // 16) * 16, -(-h // 16) * 16
    padded = np.zeros((pad_h, pad_w, 4), np.float32)
    padded[:h, :w] = tex
    # invocations beyond the image sample the clamped edge (the pass samples like any other)
    padded[:h, w:] = tex[:, -1:, :]
    padded[h:, :] = padded[h - 1:h, :]
    refbuf = orc.detect_peak(padded, pl.TRC["pq"], mn, mx, luma_coeffs(hdr_i.primaries),
                             black_cutoff=pp.black_cutoff, use_hist=pp.percentile < 100)
    nwg = (pad_w // 16) * (pad_h // 16)
    assert np.array_equal(buf[0:12], refbuf[0:12]) and buf[0:12].sum() == nwg
    assert np.array_equal(buf[12:24], refbuf[12:24])
    # per-workgroup means / maxima of floor(16383 * PQ(Y)): native pow vs libm moves single
    # pixels across an integer boundary -> at most one code per workgroup
    assert np.abs(buf[24:36].astype(np.int64) - refbuf[24:36]).max() <= refbuf[0:12].max()
    assert np.abs(buf[36:48].astype(np.int64) - refbuf[36:48]).max() <= 1
    gh, rh = buf[48:].reshape(12, 64).astype(np.int64), refbuf[48:].reshape(12, 64).astype(np.int64)
    assert gh.sum() == rh.sum()
    assert np.abs(gh - rh).sum() <= max(4, w * h // 100000)     # bin-boundary pixels
    state.destroy(); fbo.destroy()

    # -- end to end
    params = pl.render_params("default", dither_params=None)
    assert rr.render(pl.frame(src, components=3, color=hdr), pl.frame(dst, color=sdr), params), \
        gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    meta = capi.HdrMetadata()
    assert pl.lib().pl_renderer_get_hdr_metadata(rr.rr, C.byref(meta))
    # the reported scene maximum is the histogram percentile, below the sprinkled highlights
    assert 0.5 < meta.max_pq_y <= 0.9 + 1e-3 and 0.0 < meta.avg_pq_y < meta.max_pq_y
    # ... and consistent with the oracle's buffer: avg = sum / (active * 16383)
    avg = refbuf[24:36].sum() / (refbuf[12:24].sum() * 16383.0)
    assert abs(meta.avg_pq_y - avg) <= 2e-4, (meta.avg_pq_y, avg)

    r = resolve_with_peak(meta)
    assert r["need_tone"] and r["need_gamut"]
    ref = cr.apply(orc.op_quant_f16(tex.copy()), r)          # FBO rounding, then the colour map
    ref16 = orc.tex_encode(ref, "rgba16")
    # float64 evaluation on every 13th pixel
    sel = np.arange(0, w * h, 13)
    sub = orc.op_quant_f16(tex.copy()).reshape(-1, 1, 4)[sel]
    truth, _ = c64.hdr10_to_sdr(sub, r, 0.0)
    colormap_tolerance(got, ref16, truth.reshape(-1, 4), sel)
    assert np.all(got[..., 3] == 65535)
    assert 0.05 < orc.tex_decode(got, "rgba16")[..., :3].mean() < 0.9
    src.destroy(); dst.destroy()


# ---- cfg 5 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("size", [((256, 144), (128, 72)), (P8K, P4K)])
def test_cfg5_widened_polar_8k_to_4k_bit_exact(gpu, rr, size):
    """The 8K -> 4K EWA pass on its own (SDR, no linear-light scaling): plane -> rgba16hf FBO,
    widened (blur 2, radius 6.5, gather tap order) polar pass, store. No transcendental ->
    bit-exact at full size."""
    (sw, sh), (dw, dh) = size
    img = chirp16(sw, sh)
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(dw, dh, "rgba16")
    params = pl.render_params("fast", downscaler=pl.filter_config("ewa_lanczos", 2),
                              disable_linear_scaling=True)
    assert rr.render(pl.frame(src, components=3), pl.frame(dst), params), gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    a = orc.tex_decode(img, "rgba16")
    a[..., 3] = 1.0
    a = orc.op_quant_f16(a)
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos(blur=float(sw) / dw))
    ref = orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7, gather_order=not r < 6.0)
    ref16 = orc.tex_encode(ref, "rgba16")
    util.assert_polar_equal(got, ref16)
    src.destroy(); dst.destroy()


@pytest.mark.parametrize("size", [((256, 144), (128, 72)), (P8K, P4K)])
def test_cfg5_8k_to_4k_deband_ewa_tone_map(gpu, size):
    """One stream of BASELINE configs[4]: deband (native resolution) + PQ linearize -> 8K
    rgba16hf FBO; widened EWA downscale in linear light -> 4K FBO; peak measurement of that FBO;
    colour map (prelinearized) + BT.1886 -> target."""
    import colormap_ref as cr
    from test_gpu_color import luma_coeffs, nominal
    (sw, sh), (dw, dh) = size
    img = hdr_frame16(sw, sh)
    rr = pl.Renderer(gpu)       # fresh: the deband PRNG is seeded by the frame counter
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(dw, dh, "rgba16")
    hdr = pl.color_space("bt2020", "pq")
    sdr = pl.color_space("bt709", "bt1886")
    deband = capi.DebandParams.in_dll(pl.lib(), "pl_deband_default_params")
    params = pl.render_params("default", dither_params=None, deband_params=deband,
                              downscaler=pl.filter_config("ewa_lanczos", 2))
    assert rr.render(pl.frame(src, components=3, color=hdr), pl.frame(dst, color=sdr), params), \
        gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    meta = capi.HdrMetadata()
    assert pl.lib().pl_renderer_get_hdr_metadata(rr.rr, C.byref(meta))

    hdr_i, sdr_i = inferred(hdr, sdr)
    mn, mx = nominal(hdr_i)
    luma = luma_coeffs(hdr_i.primaries)
    tex = orc.tex_decode(img, "rgba16")
    # A: deband (grain pre-divided by max_luma / 203, renderer.c:1341), alpha = 1, PQ linearize
    grain = deband.grain / (hdr_i.hdr.max_luma / 203.0)
    a = orc.deband(tex, sw, sh, iterations=deband.iterations, threshold=deband.threshold,
                   radius=deband.radius, grain=float(np.float32(grain)), frame_index=1)
    del tex
    a[..., 3] = 1.0
    orc.linearize(a, pl.TRC["pq"], mn, mx, luma)
    orc.op_quant_f16(a)
    # B: widened polar in linear light -> 4K rgba16hf FBO
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos(blur=float(sw) / dw))
    b = orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7, gather_order=not r < 6.0)
    del a
    orc.op_quant_f16(b)
    # C: the measurement the renderer reported must be the one of this image
    lin = capi.ColorSpace()
    C.memmove(C.byref(lin), C.byref(hdr_i), C.sizeof(lin))
    lin.transfer = pl.TRC["linear"]
    pad_w, pad_h = -(-dw // 16) * 16, -(-dh // 16) * 16
    padded = np.zeros((pad_h, pad_w, 4), np.float32)
    padded[:dh, :dw] = b
    padded[:dh, dw:] = b[:, -1:, :]
    padded[dh:, :] = padded[dh - 1:dh, :]
    pp = capi.PeakDetectParams.in_dll(pl.lib(), "pl_peak_detect_default_params")
    refbuf = orc.detect_peak(padded, pl.TRC["linear"], mn, mx, luma,
                             black_cutoff=pp.black_cutoff, use_hist=pp.percentile < 100)
    avg = refbuf[24:36].sum() / (refbuf[12:24].sum() * 16383.0)
    assert abs(meta.avg_pq_y - avg) <= 3e-4, (meta.avg_pq_y, avg)
    # D: colour map on linear light, BT.1886, store
    import colormap_f64 as c64
    r_ = resolve_with_peak(meta)
    sel = np.arange(0, dw * dh, 13)
    truth, _ = c64.hdr10_to_sdr(b.reshape(-1, 1, 4)[sel], r_, 0.0, prelinearized=True)
    ref = cr.apply(b, r_, prelinearized=True)
    ref16 = orc.tex_encode(ref, "rgba16")
    # `truth` is float64 on the ORACLE's intermediate image. The renderer's own differs from it
    # in a few f16 codes of the 8K FBO (a PQ EOTF ulp that crosses an f16 rounding boundary,
    # ~1e-4 of the texels); where that texel is a 3900-nit highlight on a black surround, one
    # f16 ulp of it is thousands of output codes on the negative lobes around it. Those ~1e-5 of
    # the pixels make the maximum meaningless end to end: the quantiles up to 99.9 % are stated
    # here, the maximum stage by stage in test_cfg5_stage_by_stage (each stage fed the oracle's
    # output of the previous one).
    colormap_tolerance(got, ref16, truth.reshape(-1, 4), sel, quantiles=(0.5, 0.9, 0.99, 0.999), per_sample=False)
    far = np.abs(got[..., :3].astype(np.int64) - ref16[..., :3]).max(axis=2) > 300
    assert far.mean() <= 2e-5, far.sum()
    assert np.all(got[..., 3] == 65535)
    src.destroy(); dst.destroy(); rr.destroy()


@pytest.mark.parametrize("size", [((256, 144), (128, 72)), (P8K, P4K)])
def test_cfg5_stage_by_stage(gpu, size):
    """configs[4] with every stage run on its own through the shader API and fed the ORACLE's
    output of the previous stage, so that each comparison is about that stage alone:
    A deband + PQ linearize -> f16 (identical codes but for EOTF-ulp rounding flips and the
    debanding decisions that follow a sin/cos ulp); B widened EWA in linear light (bit-exact);
    C colour map + BT.1886 (never further from float64 than the oracle, maximum included)."""
    import types
    import colormap_ref as cr
    import colormap_f64 as c64
    from test_gpu_color import luma_coeffs, nominal
    (sw, sh), (dw, dh) = size
    img = hdr_frame16(sw, sh)
    hdr = pl.color_space("bt2020", "pq")
    sdr = pl.color_space("bt709", "bt1886")
    hdr_i, sdr_i = inferred(hdr, sdr)
    mn, mx = nominal(hdr_i)
    luma = luma_coeffs(hdr_i.primaries)
    deband = capi.DebandParams.in_dll(pl.lib(), "pl_deband_default_params")
    grain = float(np.float32(deband.grain / (hdr_i.hdr.max_luma / 203.0)))

    # -- A
    src = gpu.tex_create(sw, sh, "rgba16", img)
    ta = gpu.tex_create(sw, sh, "rgba16hf")
    s = gpu.begin()
    assert s.deband(src, iterations=deband.iterations, threshold=deband.threshold,
                    radius=deband.radius, grain=grain, components=3), gpu.messages[-3:]
    s.linearize(hdr_i)
    seed = int(s.listing().split("seed=")[1].split(")")[0])
    assert s.finish(ta), gpu.messages[-3:]
    got_a = ta.download()[..., :3]
    src.destroy()
    a = orc.deband(orc.tex_decode(img, "rgba16"), sw, sh, iterations=deband.iterations,
                   threshold=deband.threshold, radius=deband.radius, grain=grain, frame_index=seed)
    a[..., 3] = 1.0
    # float64 EOTF of the oracle's debanded image, rounded to f16 once
    t16 = (c64.pq_eotf(a[..., :3].astype(np.float64)) * c64.K10).astype(np.float16)
    orc.linearize(a, pl.TRC["pq"], mn, mx, luma)
    orc.op_quant_f16(a)
    ref_a = a[..., :3].astype(np.float16)
    u16 = lambda x: x.view(np.uint16)
    ulps = np.abs(u16(got_a).astype(np.int32) - u16(ref_a))
    # an EOTF error that straddles an f16 rounding boundary flips the stored code by one. The
    # oracle's float-libm EOTF does that on ~1.5 % of the texels, the kernel's (pqmath.hiph) on
    # fewer: it must match the float64 codes at least as often as the oracle does
    miss_gpu, miss_orc = (u16(got_a) != u16(t16)).mean(), (u16(ref_a) != u16(t16)).mean()
    assert miss_gpu <= miss_orc + 1e-4, (miss_gpu, miss_orc)
    assert (ulps == 0).mean() >= 1.0 - (miss_gpu + miss_orc) - 1e-4, (ulps == 0).mean()
    assert (ulps <= 1).mean() >= 0.9999, (ulps <= 1).mean()
    # the rest: a debanding decision that went the other way (a sin/cos ulp moved a sample point
    # across a texel edge) -- bounded by what debanding may change at all (threshold + grain in
    # PQ code space: < 10 % of the linear value + a floor)
    far = ulps > 1
    if far.any():
        g, r = got_a[far].astype(np.float64), ref_a[far].astype(np.float64)
        assert np.all(np.abs(g - r) <= 0.1 * np.maximum(np.abs(g), np.abs(r)) + 1e-4), \
            np.abs(g - r).max()
    print("cfg5 stage A: f16 codes != float64: GPU %.5f oracle %.5f; GPU != oracle %.5f, > 1 ulp %.2e"
          % (miss_gpu, miss_orc, (ulps != 0).mean(), far.mean()))
    del t16
    del got_a, ref_a, ulps, far

    # -- B: the oracle's A, uploaded
    ta.upload(a.astype(np.float16))
    tb = gpu.tex_create(dw, dh, "rgba16hf")
    lut = pl.ShaderObj()
    s = gpu.begin()
    assert s.sample_polar(ta, pl.filter_config("ewa_lanczos", 2), lut, new_w=dw, new_h=dh,
                          components=3), gpu.messages[-3:]
    assert s.finish(tb), gpu.messages[-3:]
    got_b = tb.download()
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos(blur=float(sw) / dw))
    b = orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7, gather_order=not r < 6.0)
    del a
    orc.op_quant_f16(b)
    util.assert_polar_equal(got_b[..., :3], b[..., :3].astype(np.float16))
    ta.destroy(); lut.destroy()

    # -- C: the oracle's B, uploaded; a fixed scene measurement in the source metadata
    b[..., 3] = 1.0
    tb.upload(b.astype(np.float16))
    meta = types.SimpleNamespace(max_pq_y=0.70, avg_pq_y=0.35)
    src_csp = capi.ColorSpace()
    C.memmove(C.byref(src_csp), C.byref(hdr_i), C.sizeof(src_csp))
    src_csp.hdr.max_pq_y, src_csp.hdr.avg_pq_y = meta.max_pq_y, meta.avg_pq_y
    dst = gpu.tex_create(dw, dh, "rgba16")
    state = pl.ShaderObj()
    s = gpu.begin()
    assert s.sample("direct", tb)
    s.color_map(src_csp, sdr_i, state, prelinearized=True)
    assert s.finish(dst), gpu.messages[-3:]
    got = dst.download()
    r_ = resolve_with_peak(meta)
    sel = np.arange(0, dw * dh, 7)
    truth, _ = c64.hdr10_to_sdr(b.reshape(-1, 1, 4)[sel], r_, 0.0, prelinearized=True)
    ref16 = orc.tex_encode(cr.apply(b, r_, prelinearized=True), "rgba16")
    colormap_tolerance(got, ref16, truth.reshape(-1, 4), sel)
    tb.destroy(); dst.destroy(); state.destroy()
