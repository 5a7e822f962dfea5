"""Contrast recovery of the tone mapper (first item; reference
src/shaders/colorspace.c:1383-1404 `pl_shader_extract_features`, :1879-1921, and
src/renderer.c:2089-2154 `get_feature_map`), which completes pl_render_high_quality_params.

Parity: each stage against the oracle's restatement -- the feature extraction, the bicubic
(four bilinear taps) lookup of the low-resolution map + detail re-injection inside the colour
map -- with the colour-map bar of test_gpu_color.py (the chain is ill-conditioned in fp32)."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
from libplacebo_amd import _capi as capi
from test_gpu_color import hdr_test_frame, run_ops

pytestmark = pytest.mark.gpu


def klms(primaries):
    m = pl.lib().pl_ipt_rgb2lms(pl.lib().pl_raw_primaries_get(pl.PRIM[primaries]))
    k = np.float32(float("%f" % (203.0 / 10000)))
    return [np.float32(k * np.float32(m.m[i][j])) for i in range(3) for j in range(3)]


def test_extract_features_vs_oracle(gpu):
    import colormap_ref as cr
    src_img = hdr_test_frame()
    csp = pl.color_space("bt2020", "pq", max_luma=1000.0)
    got = run_ops(gpu, src_img, lambda sh: sh.extract_features(csp))
    r = cr.resolve(cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=1000.0),
                   cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]))
    ref = orc.linearize(src_img.copy(), *r["lin"])
    ref = orc.extract_features(ref, klms("bt2020"))
    assert np.array_equal(got[..., 1:], ref[..., 1:])       # (0, 0, 1)
    # I is PQ-coded luminance: round-trips the PQ-coded input up to the matrices
    d = np.abs(got[..., 0] - ref[..., 0])
    assert d.max() <= 1e-4 and np.median(d) <= 2e-6, (d.max(), np.median(d))
    assert 0.05 < got[..., 0].mean() < 0.75


@pytest.mark.parametrize("tone", ["spline", "bt2390", "clip"])
def test_color_map_with_contrast_recovery_vs_oracle(gpu, tone):
    import colormap_ref as cr
    src_img = hdr_test_frame()
    h, w = src_img.shape[:2]
    # a smooth low-resolution map, values representable in the r16hf texture
    rng = np.random.default_rng(5)
    fm = (0.15 + 0.5 * rng.random((12, 16))).astype(np.float16)
    fm_tex = gpu.tex_create(16, 12, "r16hf", fm)
    src = pl.color_space("bt2020", "pq", max_luma=1000.0)
    dst = pl.color_space("bt709", "bt1886")
    res = {}
    for strength in (0.0, 0.3):
        state = pl.ShaderObj()
        params = pl.color_map_params(tone=tone, gamut="perceptual", contrast_recovery=strength)
        res[strength] = run_ops(gpu, src_img, lambda sh: sh.color_map(src, dst, state, params,
                                                                      feature_map=fm_tex))
        state.destroy()
    fm_tex.destroy()

    r = cr.resolve(cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=1000.0),
                   cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]),
                   tone=tone.encode(), gamut=b"perceptual")
    if tone == "clip":
        r["kw"].update(tone_mode=0, tone_p=(r["tone"].input_min, r["tone"].input_max, 0, 0),
                       tone_lut=None)
    lowres = orc.feature_luma(fm.astype(np.float32), w, h)
    ref = cr.apply(src_img.copy(), r, lowres=lowres, strength=0.3)
    ref0 = cr.apply(src_img.copy(), r)

    import colormap_f64 as c64
    import util
    if tone == "clip":
        truths = (None, None)      # (the float64 model covers the LUT curves)
    else:
        cr_out = (r["tone"].output_min, r["tone"].output_max)
        truths = (c64.hdr10_to_sdr(src_img, r, 0.0, lowres=lowres, strength=0.3, cr_out=cr_out)[0],
                  c64.hdr10_to_sdr(src_img, r, 0.0)[0])
    for got, want, truth in ((res[0.3], ref, truths[0]), (res[0.0], ref0, truths[1])):
        util.assert_colormap_parity(got, want, truth)
    # the recovery really changes the picture, the same way in both implementations
    delta_gpu = (res[0.3] - res[0.0])[..., :3]
    delta_ref = (ref - ref0)[..., :3]
    assert np.abs(delta_gpu).max() * 65535 > 500
    assert np.quantile(np.abs(delta_gpu - delta_ref) * 65535, 0.9) <= 3.0


def hdr_frames(gpu, sw, sh, dw, dh):
    import util
    f = util.chirp_rgba16(sw, sh).astype(np.float32) * 0.75
    f[..., 3] = 65535
    src = gpu.tex_create(sw, sh, "rgba16", f.astype(np.uint16))
    dst = gpu.tex_create(dw, dh, "rgba16")
    image = pl.frame(src, components=3, color=pl.color_space("bt2020", "pq", max_luma=1000.0),
                     repr_=pl.color_repr("rgb", "full"))
    target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"))
    return src, dst, image, target


def test_high_quality_preset_runs_contrast_recovery(gpu):
    """pl_render_high_quality_params on HDR -> SDR: feature map passes + recovery in the colour
    map; switching the strength / smoothness off gives the plain tone mapping."""
    src, dst, image, target = hdr_frames(gpu, 96, 64, 96, 64)
    outs = {}
    for name, kw in (("hq", {}), ("off", dict(contrast_recovery=0.0)),
                     ("smooth1", dict(contrast_smoothness=1.0))):
        cm = pl.color_map_params("spline", "perceptual",
                                 **{**dict(contrast_recovery=0.3, contrast_smoothness=3.5), **kw})
        params = pl.render_params("high_quality", color_map_params=cm, peak_detect_params=None)
        rr = pl.Renderer(gpu)
        assert rr.render(image, target, params), gpu.messages[-4:]
        assert rr.errors() == 0
        outs[name] = dst.download()
        rr.destroy()
    assert np.array_equal(outs["off"], outs["smooth1"])
    d = np.abs(outs["hq"].astype(np.int64) - outs["off"].astype(np.int64))[..., :3]
    assert d.max() > 200 and d.mean() > 5, (int(d.max()), float(d.mean()))
    # recovery adds local contrast: the high-pass energy of the luma goes up
    def hp_energy(img):
        y = img[..., :3].astype(np.float64).mean(axis=2)
        return np.abs(y[1:-1, 1:-1] * 4 - y[:-2, 1:-1] - y[2:, 1:-1] - y[1:-1, :-2] - y[1:-1, 2:]).mean()
    assert hp_energy(outs["hq"]) > hp_energy(outs["off"])
    src.destroy(); dst.destroy()


@pytest.mark.parametrize("size", [((256, 144), (128, 72)), ((300, 170), (150, 85)), ((192, 108), (192, 108))])
def test_measuring_pass_extracts_the_features_too(gpu, size, monkeypatch):
    """high_quality on HDR10 with peak detection: the pass that reads the scaled intermediate for
    the measurement also writes the contrast-recovery feature plane ([PEAK_DETECT] [FEATURES] into
    an r16hf target: k_peak_tiles<.., 2, ..>) instead of a second pass reading the same image
    (PL_HIP_FUSED_FEATURES=0 keeps the two passes; PL_HIP_PEAK_TILES=0: k_peak_fast<.., 2> and the
    fold kernel; PL_HIP_PEAK_FAST=0: the generic measuring kernel with the fused op list). Same
    measurement, same feature plane, so the same frame bit for bit
    and the same scene metadata -- behind an EWA 2:1 downscale (the measurement of an existing
    intermediate: configs[4]) and at 1:1 (where the measurement rides on the decoding pass and
    nothing is merged)."""
    import ctypes as C
    import util
    from libplacebo_amd import _capi as capi
    (sw, sh), (dw, dh) = size
    outs = []
    for env in ({"PL_HIP_FUSED_FEATURES": "1"}, {"PL_HIP_FUSED_FEATURES": "0"},
                {"PL_HIP_FUSED_FEATURES": "1", "PL_HIP_PEAK_TILES": "0"},
                {"PL_HIP_FUSED_FEATURES": "1", "PL_HIP_PEAK_FAST": "0"}):
        for k in ("PL_HIP_FUSED_FEATURES", "PL_HIP_PEAK_FAST", "PL_HIP_PEAK_TILES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        src, dst, image, target = hdr_frames(gpu, sw, sh, dw, dh)
        params = pl.render_params("high_quality", downscaler=pl.filter_config("ewa_lanczos", 2),
                                  peak_detect_params=pl.peak_detect_params(percentile=99.995))
        rr = pl.Renderer(gpu)
        util.srand(1)
        assert rr.render(image, target, params), gpu.messages[-4:]
        assert rr.errors() == 0
        meta = capi.HdrMetadata()
        assert pl.lib().pl_renderer_get_hdr_metadata(rr.rr, C.byref(meta))
        outs.append((dst.download(), (meta.max_pq_y, meta.avg_pq_y)))
        rr.destroy(); src.destroy(); dst.destroy()
    for out, meta in outs[1:]:
        assert meta == outs[0][1]
        assert np.array_equal(out, outs[0][0]), util.diff_stats(out, outs[0][0])
    assert outs[0][0][..., :3].std() > 1000


@pytest.mark.parametrize("size", [((96, 64), (96, 64)), ((257, 131), (257, 131)), ((640, 360), (1280, 720)),
                                  ((1031, 577), (517, 301)), ((3840, 2160), (3840, 2160))])
def test_fused_lowpass_equals_the_two_passes(gpu, size):
    """The low-pass behind the contrast-recovery feature map (reference src/renderer.c:2089-2154: a
    separable bicubic downscale of the full-size r16hf luminance plane by contrast_smoothness, two
    pl_shader_sample_ortho2 passes through an r16hf intermediate) as ONE launch that keeps the
    intermediate in LDS (k_lowpass2, csrc/hip/k_ortho.hip) against the two passes
    (PL_HIP_LOWPASS_FUSED=0): the same geometry per pixel, the same blended weight rows, the same fma
    order, the same f16 rounding of the intermediate -- the feature map is the same plane bit for
    bit, and with it the tone-mapped frame. Sizes: tiles that are partial on both axes, planes smaller
    than a tile's footprint (mirrored on both sides at once), up- and downscaled targets, the bench's
    4K frame."""
    import os
    import util
    from test_gpu_fullsize import hdr_frame16
    (sw, sh), (dw, dh) = size
    hdr = hdr_frame16(sw, sh)
    outs = []
    for fused in ("1", "0"):
        old = os.environ.get("PL_HIP_LOWPASS_FUSED")
        os.environ["PL_HIP_LOWPASS_FUSED"] = fused
        try:
            src = gpu.tex_create(sw, sh, "rgba16", hdr)
            dst = gpu.tex_create(dw, dh, "rgba16")
            rr = pl.Renderer(gpu)
            util.srand(1)
            params = pl.render_params("high_quality", peak_detect_params=pl.peak_detect_params(percentile=99.995))
            assert rr.render(pl.frame(src, components=3, color=pl.color_space("bt2020", "pq", max_luma=1000.0)),
                             pl.frame(dst, color=pl.color_space("bt709", "bt1886")), params)
            assert rr.errors() == 0
            outs.append(dst.download())
            rr.destroy(); src.destroy(); dst.destroy()
        finally:
            if old is None:
                os.environ.pop("PL_HIP_LOWPASS_FUSED", None)
            else:
                os.environ["PL_HIP_LOWPASS_FUSED"] = old
    assert outs[0][..., :3].std() > 1000
    assert np.array_equal(outs[0], outs[1]), util.diff_stats(outs[0], outs[1])
