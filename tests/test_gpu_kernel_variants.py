"""The specialised kernels (k_bilinear_fast, k_ortho_fast, k_polar_pp + fused PASS A / fused
epilogue) against the generic ones they replace: same pl_render_image call, kernel selected by
environment switches that the library reads per launch. Bar: bit-identical frames. (Each generic
kernel is pinned against the oracle elsewhere: test_gpu_renderer.py, test_gpu_ortho_deband.py.)"""
import ctypes as C
import os

import numpy as np
import pytest

import libplacebo_amd as pl
import util
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu

TEN_BIT = dict(sample_depth=16, color_depth=10, bit_shift=6)


def render(gpu, img, dw, dh, params, ten_bit, env, crop=None, src_fmt="rgba16"):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        sh, sw = img.shape[:2]
        src = gpu.tex_create(sw, sh, src_fmt, img)
        dst = gpu.tex_create(dw, dh, "rgba16")
        image = pl.frame(src, components=3, crop=crop)
        target = pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT) if ten_bit else None)
        rr = pl.Renderer(gpu)
        util.srand(1)
        assert rr.render(image, target, params), gpu.messages[-4:]
        assert rr.errors() == 0
        out = dst.download()
        rr.destroy(); src.destroy(); dst.destroy()
        return out
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def dither():
    return capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0)


@pytest.mark.parametrize("scale", [(2, 2), (3, 2), (1.5, 1.25)])
@pytest.mark.parametrize("ten_bit", [False, True])
def test_bilinear_fast_equals_generic(gpu, scale, ten_bit):
    sw, sh = 100, 58
    img = util.chirp_rgba16(sw, sh)
    dw, dh = int(sw * scale[0]), int(sh * scale[1])
    kw = dict(dither_params=dither(), disable_dither_gamma_correction=True) if ten_bit else {}
    params = pl.render_params("fast", **kw)
    outs = [render(gpu, img, dw, dh, params, ten_bit, {"PL_HIP_BILIN_ITERS": it})
            for it in ("0", "1", "2", "4")]
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])
    assert outs[0][..., :3].std() > 1000


@pytest.mark.parametrize("size", [((100, 58), (200, 116)), ((101, 59), (202, 118)), ((96, 64), (288, 128)),
                                  ((1920, 1080), (3840, 2160))])
@pytest.mark.parametrize("ten_bit", [False, True])
def test_bilinear_tables_equal_per_pixel_geometry(gpu, size, ten_bit):
    """k_bilinear_strip (round 6, opt-in -- it lost, profiles/r06_12_strip_rows.txt: one 16-byte load per
    2 x 2 cell, a wave sliding down 64 cell columns) and k_bilinear_tab (geometry from per-column /
    per-row tables; PL_HIP_BILIN_TABLES=1) against k_bilinear_fast (PL_HIP_BILIN_STRIP=0: per-pixel
    attribute interpolation, floor, fract) and the generic kernel: the same frames bit for bit -- 2x
    at even and odd sizes, 3x2 (where a cell's two columns do not share a texel: the tables decline
    and every run is the per-pixel kernel), the BASELINE frame; rgba16hf source too."""
    (sw, sh), (dw, dh) = size
    img = util.chirp_rgba16(sw, sh)
    kw = dict(dither_params=dither(), disable_dither_gamma_correction=True) if ten_bit else {}
    params = pl.render_params("fast", **kw)
    per_px_env = {"PL_HIP_BILIN_STRIP": "0", "PL_HIP_BILIN_TABLES": "0"}
    strip = render(gpu, img, dw, dh, params, ten_bit, {"PL_HIP_BILIN_STRIP": "1", "PL_HIP_BILIN_TABLES": "0"})
    tab = render(gpu, img, dw, dh, params, ten_bit, {"PL_HIP_BILIN_TABLES": "1"})    # (opt-in: it lost, profiles/r04_12)
    per_px = render(gpu, img, dw, dh, params, ten_bit, per_px_env)
    assert np.array_equal(tab, per_px), util.diff_stats(tab, per_px)
    assert np.array_equal(strip, per_px), util.diff_stats(strip, per_px)
    if sw < 1000:
        generic = render(gpu, img, dw, dh, params, ten_bit, {"PL_HIP_BILIN_ITERS": "0"})
        assert np.array_equal(tab, generic)
        f16 = (img.astype(np.float32) / 65535.0).astype(np.float16)
        a = render(gpu, f16, dw, dh, params, ten_bit, {"PL_HIP_BILIN_TABLES": "1"}, src_fmt="rgba16hf")
        b = render(gpu, f16, dw, dh, params, ten_bit, per_px_env, src_fmt="rgba16hf")
        c = render(gpu, f16, dw, dh, params, ten_bit, {"PL_HIP_BILIN_STRIP": "1"}, src_fmt="rgba16hf")
        assert np.array_equal(a, b) and np.array_equal(c, b)
    assert tab[..., :3].std() > 1000


@pytest.mark.parametrize("size", [(7, 5), (64, 9), (65, 8), (129, 17), (300, 170), (2, 2)])
def test_bilinear_strip_odd_sizes_flips_and_alpha(gpu, size):
    """k_bilinear_strip at sizes whose last wave / last strip are partial (a strip is 8 cell rows, a
    wave 64 cell columns; the first cell of either axis is the padded one), a flipped source rect
    (the tables decline: both runs are the per-pixel kernel) and a 10-bit dithered target, against
    k_bilinear_fast: bit for bit."""
    sw, sh = size
    rng = np.random.default_rng(sw * 31 + sh)
    img = rng.integers(0, 65536, (sh, sw, 4), dtype=np.uint16)
    params = pl.render_params("fast")
    for crop in (None, (float(sw), float(sh), 0.0, 0.0)):      # (the second: a flipped source rect)
        a = render(gpu, img, 2 * sw, 2 * sh, params, False, {"PL_HIP_BILIN_STRIP": "1"}, crop=crop)
        b = render(gpu, img, 2 * sw, 2 * sh, params, False, {"PL_HIP_BILIN_STRIP": "0"}, crop=crop)
        assert np.array_equal(a, b), util.diff_stats(a, b)
    params10 = pl.render_params("fast", dither_params=dither(), disable_dither_gamma_correction=True)
    a = render(gpu, img, 2 * sw, 2 * sh, params10, True, {"PL_HIP_BILIN_STRIP": "1"})
    b = render(gpu, img, 2 * sw, 2 * sh, params10, True, {"PL_HIP_BILIN_STRIP": "0"})
    assert np.array_equal(a, b), util.diff_stats(a, b)


def test_bilinear_fast_crop_and_flip(gpu):
    sw, sh = 96, 64
    img = util.chirp_rgba16(sw, sh)
    params = pl.render_params("fast")
    for crop in [(10.5, 7.25, 80.0, 50.5), (90.0, 60.0, 6.0, 4.0)]:     # second one is flipped
        a = render(gpu, img, 160, 120, params, False, {"PL_HIP_BILIN_ITERS": "0"}, crop=crop)
        b = render(gpu, img, 160, 120, params, False, {"PL_HIP_BILIN_ITERS": "1"}, crop=crop)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["lanczos", "mitchell", "catmull_rom", "spline36", "ginseng"])
@pytest.mark.parametrize("scale", [(2, 2), (1.6, 2.3)])
def test_ortho_fast_equals_generic(gpu, name, scale):
    sw, sh = 90, 62
    img = util.chirp_rgba16(sw, sh)
    dw, dh = int(sw * scale[0]), int(sh * scale[1])
    for ar in (0.0, 0.6):
        params = pl.render_params("fast", upscaler=pl.filter_config(name), dither_params=dither(),
                                  disable_dither_gamma_correction=True, antiringing_strength=ar)
        a = render(gpu, img, dw, dh, params, True, {"PL_HIP_ORTHO_FAST": "0"})
        b = render(gpu, img, dw, dh, params, True, {"PL_HIP_ORTHO_FAST": "1"})
        assert np.array_equal(a, b), (name, scale, ar, util.diff_stats(a, b))
    assert a[..., :3].std() > 1000


def test_ortho_fast_without_epilogue_and_with_lite_ops(gpu):
    """16-bit target (no dither: plain store) and a colour-adjusted target (LITE op chain)."""
    sw, sh = 90, 62
    img = util.chirp_rgba16(sw, sh)
    params = pl.render_params("fast", upscaler=pl.filter_config("lanczos"))
    a = render(gpu, img, 180, 124, params, False, {"PL_HIP_ORTHO_FAST": "0"})
    b = render(gpu, img, 180, 124, params, False, {"PL_HIP_ORTHO_FAST": "1"})
    assert np.array_equal(a, b)
    adj = capi.ColorAdjustment(brightness=0.05, contrast=1.1, saturation=0.9, hue=0.0, gamma=1.0,
                               temperature=0.0)
    params = pl.render_params("fast", upscaler=pl.filter_config("lanczos"), color_adjustment=adj)
    a = render(gpu, img, 180, 124, params, False, {"PL_HIP_ORTHO_FAST": "0"})
    b = render(gpu, img, 180, 124, params, False, {"PL_HIP_ORTHO_FAST": "1"})
    assert np.array_equal(a, b)


def test_ortho_fast_default_preset(gpu):
    """pl_render_default_params: lanczos in linear + sigmoidized light; the horizontal pass
    carries unsigmoidize + delinearize + dither (the full op interpreter, EPI 3)."""
    sw, sh = 90, 62
    img = util.chirp_rgba16(sw, sh)
    params = pl.render_params("default")
    a = render(gpu, img, 180, 124, params, True, {"PL_HIP_ORTHO_FAST": "0"})
    b = render(gpu, img, 180, 124, params, True, {"PL_HIP_ORTHO_FAST": "1"})
    assert np.array_equal(a, b)
    assert a[..., :3].std() > 1000


@pytest.mark.parametrize("scale", [(2, 2), (3, 3), (1.5, 1.5)])
def test_polar_fused_equals_unfused_and_per_pixel(gpu, scale):
    sw, sh = 96, 64
    img = util.chirp_rgba16(sw, sh)
    dw, dh = int(sw * scale[0]), int(sh * scale[1])
    params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"),
                              dither_params=dither(), disable_dither_gamma_correction=True)
    base = render(gpu, img, dw, dh, params, True, {})
    for env in ({"PL_HIP_NO_FUSION": "1"}, {"PL_HIP_POLAR_PER_PIXEL": "1"},
                {"PL_HIP_NT_STORE": "0"}):
        o = render(gpu, img, dw, dh, params, True, env)
        util.assert_polar_equal(o, base, step=64, what=env)      # (10-bit dithered frames)


def test_fuzz_sizes_crops_specialised_vs_generic(gpu):
    """Random source / target sizes and crops (fractional, flipped, partly outside): the
    specialised kernels must agree with the generic ones on every frame."""
    rng = np.random.default_rng(2024)
    scalers = [None, "lanczos", "mitchell", "ewa_lanczos"]
    for case in range(36):
        sw, sh = int(rng.integers(17, 140)), int(rng.integers(9, 90))
        dw, dh = int(rng.integers(16, 260)), int(rng.integers(8, 180))
        img = util.chirp_rgba16(sw, sh)
        x0, x1 = sorted(rng.uniform(-3, sw + 3, 2))
        y0, y1 = sorted(rng.uniform(-3, sh + 3, 2))
        if x1 - x0 < 4 or y1 - y0 < 4:
            x0, y0, x1, y1 = 0, 0, sw, sh
        crop = [x0, y0, x1, y1]
        if case % 5 == 0:
            crop = [x1, y0, x0, y1]     # flipped in x
        if case % 7 == 0:
            crop = None
        name = scalers[case % len(scalers)]
        ten_bit = bool(case % 2)
        kw = dict(dither_params=dither(), disable_dither_gamma_correction=True) if ten_bit else {}
        if name:
            kw.update(upscaler=pl.filter_config(name), downscaler=pl.filter_config(name, 2))
        params = pl.render_params("fast", **kw)
        generic = {"PL_HIP_BILIN_ITERS": "0", "PL_HIP_ORTHO_FAST": "0", "PL_HIP_NO_FUSION": "1",
                   "PL_HIP_NT_STORE": "0"}
        a = render(gpu, img, dw, dh, params, ten_bit, generic, crop=crop)
        b = render(gpu, img, dw, dh, params, ten_bit, {}, crop=crop)
        assert np.array_equal(a, b), (case, name, (sw, sh), (dw, dh), crop, util.diff_stats(a, b))


@pytest.mark.parametrize("fmt", ["8", "16"])
def test_planar_chroma_polar_phase_classes_equal_per_pixel(gpu, fmt):
    """NV12 / P016: the chroma plane (2 components) and, with a luma-only image, the 1-component
    plane go through the phase-class polar kernel (k_polar_pp_*_c12.hip); per-pixel weights
    (PL_HIP_POLAR_PER_PIXEL=1) must give the same frame."""
    rng = np.random.default_rng(4)
    w, h = 96, 64
    dt, mx = (np.uint8, 255) if fmt == "8" else (np.uint16, 65535)
    yy, xx = np.mgrid[0:h, 0:w]
    y = ((0.5 + 0.4 * np.sin(xx * 0.21) * np.cos(yy * 0.17)) * mx).astype(dt)[..., None]
    uv = (rng.integers(mx // 4, 3 * mx // 4, (h // 2, w // 2, 2))).astype(dt)
    outs = []
    for env in ({"PL_HIP_POLAR_PER_PIXEL": "1"}, {}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ty = gpu.tex_create(w, h, "r8" if fmt == "8" else "r16", y)
            tuv = gpu.tex_create(w // 2, h // 2, "rg8" if fmt == "8" else "rg16", uv)
            dst = gpu.tex_create(2 * w, 2 * h, "rgba16")
            f = capi.Frame(num_planes=2)
            f.planes[0] = capi.Plane(texture=ty.ptr, components=1)
            f.planes[1] = capi.Plane(texture=tuv.ptr, components=2)
            for c in range(4):
                f.planes[0].component_mapping[c] = [0, -1, -1, -1][c]
                f.planes[1].component_mapping[c] = [1, 2, -1, -1][c]
            bits = 8 if fmt == "8" else 16
            f.repr = pl.color_repr("bt709", "limited", sample_depth=bits, color_depth=bits)
            f.color = pl.color_space("bt709", "bt1886")
            pl.lib().pl_frame_set_chroma_location(C.byref(f), 1)
            target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"))
            params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"))
            rr = pl.Renderer(gpu)
            assert rr.render(f, target, params), gpu.messages[-4:]
            assert rr.errors() == 0
            outs.append(dst.download())
            rr.destroy(); ty.destroy(); tuv.destroy(); dst.destroy()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    util.assert_polar_equal(outs[0], outs[1])
    assert outs[0][..., :3].std() > 1000


def test_one_renderer_across_changing_geometries(gpu):
    """Cached state (phase-class tables, scaler LUTs, FBO pool, dither matrix) must follow the
    frame geometry and the params: one long-lived renderer == a fresh renderer per frame."""
    rng = np.random.default_rng(77)
    rr = pl.Renderer(gpu)
    names = ["ewa_lanczos", "lanczos", None, "ewa_lanczos", "mitchell", "ewa_lanczos"]
    for case in range(12):
        sw, sh = int(rng.integers(24, 120)), int(rng.integers(16, 80))
        dw, dh = int(rng.integers(24, 240)), int(rng.integers(16, 160))
        if case in (3, 4):      # same geometry twice in a row (cache hit), then a new one
            sw, sh, dw, dh = 64, 40, 128, 80
        img = util.chirp_rgba16(sw, sh)
        name = names[case % len(names)]
        kw = dict(dither_params=dither(), disable_dither_gamma_correction=True)
        if name:
            kw.update(upscaler=pl.filter_config(name), downscaler=pl.filter_config(name, 2))
        params = pl.render_params("fast", **kw)
        src = gpu.tex_create(sw, sh, "rgba16", img)
        dst = gpu.tex_create(dw, dh, "rgba16")
        image = pl.frame(src, components=3)
        target = pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT))
        util.srand(1)
        assert rr.render(image, target, params), gpu.messages[-4:]
        got = dst.download()
        fresh = pl.Renderer(gpu)
        util.srand(1)
        assert fresh.render(image, target, params)
        want = dst.download()
        fresh.destroy()
        assert np.array_equal(got, want), (case, name, (sw, sh), (dw, dh))
        src.destroy(); dst.destroy()
    assert rr.errors() == 0
    rr.destroy()


@pytest.mark.parametrize("fmt", ["8", "16"])
@pytest.mark.parametrize("name", ["lanczos", "mitchell", "spline36"])
def test_planar_ortho_fast_equals_generic(gpu, fmt, name):
    """NV12 / P016 with a separable scaler: luma / chroma planes (r8, rg8, r16, rg16 and their
    f16 FBOs) go through k_ortho_fast's plane instantiations."""
    rng = np.random.default_rng(6)
    w, h = 96, 64
    dt, mx = (np.uint8, 255) if fmt == "8" else (np.uint16, 65535)
    yy, xx = np.mgrid[0:h, 0:w]
    y = ((0.5 + 0.4 * np.sin(xx * 0.21) * np.cos(yy * 0.17)) * mx).astype(dt)[..., None]
    uv = (rng.integers(mx // 4, 3 * mx // 4, (h // 2, w // 2, 2))).astype(dt)
    outs = []
    for env in ({"PL_HIP_ORTHO_FAST": "0"}, {}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            ty = gpu.tex_create(w, h, "r8" if fmt == "8" else "r16", y)
            tuv = gpu.tex_create(w // 2, h // 2, "rg8" if fmt == "8" else "rg16", uv)
            dst = gpu.tex_create(2 * w, 2 * h, "rgba16")
            f = capi.Frame(num_planes=2)
            f.planes[0] = capi.Plane(texture=ty.ptr, components=1)
            f.planes[1] = capi.Plane(texture=tuv.ptr, components=2)
            for c in range(4):
                f.planes[0].component_mapping[c] = [0, -1, -1, -1][c]
                f.planes[1].component_mapping[c] = [1, 2, -1, -1][c]
            bits = 8 if fmt == "8" else 16
            f.repr = pl.color_repr("bt709", "limited", sample_depth=bits, color_depth=bits)
            f.color = pl.color_space("bt709", "bt1886")
            pl.lib().pl_frame_set_chroma_location(C.byref(f), 1)
            target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"))
            params = pl.render_params("fast", upscaler=pl.filter_config(name))
            rr = pl.Renderer(gpu)
            assert rr.render(f, target, params), gpu.messages[-4:]
            assert rr.errors() == 0
            outs.append(dst.download())
            rr.destroy(); ty.destroy(); tuv.destroy(); dst.destroy()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    util.assert_polar_equal(outs[0], outs[1])
    assert outs[0][..., :3].std() > 1000


@pytest.mark.parametrize("name,ratio", [("lanczos", 2.0), ("lanczos", 2.5), ("mitchell", 3.5),
                                        ("catmull_rom", 4.0), ("hermite", 5.0), ("spline36", 1.7),
                                        # all-positive kernels: "linear trick" LUTs
                                        ("bicubic", 3.5), ("bicubic", 1.5), ("gaussian", 2.0),
                                        ("bilinear", 6.0), ("hermite", 2.2)])
def test_ortho_fast_downscales_equal_generic(gpu, name, ratio):
    """Widened (anti-aliased) kernels: 10-16 taps take the run-time tap count variant."""
    sw, sh = 200, 140
    img = util.chirp_rgba16(sw, sh)
    dw, dh = int(sw / ratio), int(sh / ratio)
    params = pl.render_params("fast", downscaler=pl.filter_config(name, 2),
                              dither_params=dither(), disable_dither_gamma_correction=True)
    a = render(gpu, img, dw, dh, params, True, {"PL_HIP_ORTHO_FAST": "0"})
    b = render(gpu, img, dw, dh, params, True, {"PL_HIP_ORTHO_FAST": "1"})
    assert np.array_equal(a, b), (name, ratio, util.diff_stats(a, b))
    assert a[..., :3].std() > 500


def test_tiny_frames_do_not_break_the_fast_paths(gpu):
    """1..8 pixel sources and targets through every scaler family: the specialised kernels must
    either decline or agree with the generic ones (and nothing may fault)."""
    rng = np.random.default_rng(5)
    scalers = [None, "lanczos", "bicubic", "ewa_lanczos", "mitchell", "gaussian"]
    for case in range(30):
        sw, sh = int(rng.integers(1, 9)), int(rng.integers(1, 9))
        dw, dh = int(rng.integers(1, 17)), int(rng.integers(1, 17))
        img = util.chirp_rgba16(max(sw, 2), max(sh, 2))[:sh, :sw].copy()
        name = scalers[case % len(scalers)]
        kw = {}
        if name:
            kw.update(upscaler=pl.filter_config(name), downscaler=pl.filter_config(name, 2))
        if case % 2:
            kw.update(dither_params=dither(), disable_dither_gamma_correction=True)
        params = pl.render_params("fast", **kw)
        generic = {"PL_HIP_BILIN_ITERS": "0", "PL_HIP_ORTHO_FAST": "0", "PL_HIP_NO_FUSION": "1",
                   "PL_HIP_POLAR_PER_PIXEL": "1"}
        a = render(gpu, img, dw, dh, params, bool(case % 2), generic)
        b = render(gpu, img, dw, dh, params, bool(case % 2), {})
        assert np.array_equal(a, b), (case, name, (sw, sh), (dw, dh))


@pytest.mark.parametrize("ten_bit", [False, True])
@pytest.mark.parametrize("src_fmt", ["rgba16", "rgba16hf"])
def test_nearest_fast_equals_generic(gpu, ten_bit, src_fmt):
    """k_nearest_fast (1:1 and integer-ratio nearest fetch + fused epilogue: the output pass of a
    single cached frame in pl_render_image_mix, plain conversions) against k_pass_generic"""
    sw, sh = 101, 57
    img = util.chirp_rgba16(sw, sh)
    if src_fmt == "rgba16hf":
        img = (img.astype(np.float32) / 65535.0).astype(np.float16)
    kw = dict(dither_params=dither(), disable_dither_gamma_correction=True) if ten_bit else {}
    nearest = pl.filter_config("nearest")
    for (dw, dh), crop in [((sw, sh), None), ((2 * sw, 3 * sh), None), ((sw, sh), (sw, 0, 0, sh)),
                           ((77, 41), (3.0, 2.0, 80.0, 43.0))]:
        params = pl.render_params("fast", upscaler=nearest, downscaler=nearest, **kw)
        a = render(gpu, img, dw, dh, params, ten_bit, {"PL_HIP_BILIN_ITERS": "0"}, crop=crop,
                   src_fmt=src_fmt)
        b = render(gpu, img, dw, dh, params, ten_bit, {}, crop=crop, src_fmt=src_fmt)
        assert np.array_equal(a, b), ((dw, dh), crop, util.diff_stats(a, b))
        assert a[..., :3].std() > 1000


@pytest.mark.parametrize("size", [(16, 16), (70, 45), (128, 64), (1920, 1080), (3840, 2160)])
@pytest.mark.parametrize("src_fmt,trc", [("rgba16", "pq"), ("rgba16hf", "linear")])
@pytest.mark.parametrize("store", [True, False])
def test_peak_fast_equals_generic(gpu, size, src_fmt, trc, store):
    """The renderer's measuring pass (plane texel for texel -> f16 intermediate + measurement, or
    the target-less measurement of an FBO) on its three kernels -- k_peak_tiles (the default: the
    measurement kept on chip, slice by slice, folded by the last workgroup), k_peak_fast
    (PL_HIP_PEAK_TILES=0: one wave per tile, scratch copies, k_peak_fold) and k_pass_peak
    (PL_HIP_PEAK_FAST=0: the interpreter): the 816-word buffer word for word and the intermediate
    bit for bit, for sizes that are not multiples of the 16 x 16 tiling (padding invocations
    measure the clamped edge texel). Every kernel measures twice through the same state object:
    the second result must equal the first (the scratch words are left zeroed)."""
    from test_gpu_color import _read_device
    from test_gpu_fullsize import hdr_frame16, inferred
    w, h = size
    img = hdr_frame16(w, h)
    if src_fmt == "rgba16hf":
        img = (img.astype(np.float32) / 65535.0 * 4.0).astype(np.float16)
    csp, _ = inferred(pl.color_space("bt2020", trc, max_luma=1000.0), pl.color_space("bt709", "bt1886"))
    res = []
    modes = [{}, {"PL_HIP_PEAK_TILES": "0"}, {"PL_HIP_PEAK_FAST": "0"}]
    for env in modes:
        old = {k: os.environ.get(k) for k in ("PL_HIP_PEAK_FAST", "PL_HIP_PEAK_TILES")}
        for k in old:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            src = gpu.tex_create(w, h, src_fmt, img)
            fbo = gpu.tex_create(w, h, "rgba16hf")
            state = pl.ShaderObj()
            bufs = []
            for rep in range(2):
                a = gpu.begin()
                assert a.sample("direct", src, components=3)
                pp = pl.peak_detect_params(percentile=99.995)
                assert pl.lib().pl_shader_detect_peak(a.sh, csp, C.byref(state.slot), C.byref(pp))
                if store:
                    assert a.finish(fbo), gpu.messages[-3:]
                else:
                    assert a.compute(w, h), gpu.messages[-3:]
                size_ = C.c_size_t()
                pl.lib().pl_hip_peak_buffer.restype = C.c_void_p
                ptr = pl.lib().pl_hip_peak_buffer(state.slot, C.byref(size_))
                assert ptr and size_.value == 816 * 4
                bufs.append(_read_device(ptr, size_.value).copy())
            assert np.array_equal(bufs[0], bufs[1]), env
            res.append((bufs[0], fbo.download().view(np.uint16).copy()))
            state.destroy(); fbo.destroy(); src.destroy()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    (buf_t, fbo_t), (buf_f, fbo_f), (buf_g, fbo_g) = res
    assert buf_g[0:12].sum() == (-(-w // 16)) * (-(-h // 16)) and buf_g[24:36].sum() > 0
    assert np.array_equal(buf_f, buf_g)
    assert np.array_equal(buf_t, buf_g), np.nonzero(buf_t != buf_g)
    if store:
        assert np.array_equal(fbo_f, fbo_g) and fbo_g.any()
        assert np.array_equal(fbo_t, fbo_g)


@pytest.mark.parametrize("size", [(97, 61), (256, 130)])
def test_pass_native_equals_generic(gpu, size):
    """k_pass_native (a pass that reads its source texel for texel: the colour-map pass behind an
    intermediate, the plane decode in front of a separable scaler) against k_pass_generic: the
    same frames, bit for bit -- HDR10 -> SDR tone + gamut map with dither (full interpreter) and
    the default preset's decode / linearize / sigmoid pass (odd widths: the last column is a
    single pixel)."""
    from test_gpu_fullsize import hdr_frame16
    w, h = size
    hdr = hdr_frame16(w, h)
    sdr = util.chirp_rgba16(w, h)
    cases = [
        (hdr, w, h, pl.render_params("default", peak_detect_params=pl.peak_detect_params(percentile=99.995)),
         dict(color=pl.color_space("bt2020", "pq", max_luma=1000.0)), dict(color=pl.color_space("bt709", "bt1886"))),
        (sdr, 2 * w, 2 * h, pl.render_params("default"), {}, {}),
    ]
    for img, dw, dh, params, ikw, tkw in cases:
        outs = []
        for native in ("1", "0"):
            old = os.environ.get("PL_HIP_PASS_NATIVE")
            os.environ["PL_HIP_PASS_NATIVE"] = native
            # (both on the interpreter: the map chain, which k_pass_native's pass would otherwise
            # run as, is compared with it in test_map_chain_equals_interpreter)
            old_chain = os.environ.get("PL_HIP_MAP_CHAIN")
            os.environ["PL_HIP_MAP_CHAIN"] = "0"
            try:
                src = gpu.tex_create(w, h, "rgba16", img)
                dst = gpu.tex_create(dw, dh, "rgba16")
                rr = pl.Renderer(gpu)
                util.srand(1)
                assert rr.render(pl.frame(src, components=3, **ikw),
                                 pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT), **tkw), params)
                assert rr.errors() == 0
                outs.append(dst.download())
                rr.destroy(); src.destroy(); dst.destroy()
            finally:
                for key, val in (("PL_HIP_PASS_NATIVE", old), ("PL_HIP_MAP_CHAIN", old_chain)):
                    if val is None:
                        os.environ.pop(key, None)
                    else:
                        os.environ[key] = val
        assert np.array_equal(outs[0], outs[1]) and outs[0][..., :3].std() > 1000


def _env(name, value):
    class _E:
        def __enter__(self):
            self.old = os.environ.get(name)
            os.environ[name] = value
        def __exit__(self, *a):
            if self.old is None:
                os.environ.pop(name, None)
            else:
                os.environ[name] = self.old
    return _E()


@pytest.mark.parametrize("case", ["map_pass", "map_pass_contrast_recovery", "ewa_2x_map", "ewa_2x_map_no_peak"])
@pytest.mark.parametrize("size", [(97, 61), (256, 130)])
def test_map_chain_equals_interpreter(gpu, case, size):
    """The op list of an HDR map pass as straight-line code (struct plh_map_chain: k_pass_chain, the
    CHAIN epilogue of k_polar_mx) against the same pass through the op interpreter
    (PL_HIP_MAP_CHAIN=0): the same device functions in the same order -- up to where the uniform
    scale factors are multiplied in (assert_same_up_to_a_dither_step) -- HDR10 -> BT.1886 with a 10-bit dither behind an intermediate (with and without
    contrast recovery reading the feature map) and behind the EWA 2x upscale on the matrix pipe
    (the metric's launch), odd sizes included (single-pixel last column, partial tiles)."""
    from test_gpu_fullsize import hdr_frame16
    w, h = size
    hdr = hdr_frame16(w, h)
    peak = pl.peak_detect_params(percentile=99.995)
    if case == "map_pass":
        dw, dh, params = w, h, pl.render_params("default", peak_detect_params=peak)
    elif case == "map_pass_contrast_recovery":
        dw, dh, params = w, h, pl.render_params("high_quality", peak_detect_params=peak)
    elif case == "ewa_2x_map":
        dw, dh, params = 2 * w, 2 * h, pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), peak_detect_params=peak)
    else:
        dw, dh, params = 2 * w, 2 * h, pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), peak_detect_params=None)
    outs = []
    for chain in ("1", "0"):
        # (the closed forms of the PQ pair on both sides -- the chain's piecewise cubics are held to them
        # by test_pq_segments_against_closed_forms)
        with _env("PL_HIP_MAP_CHAIN", chain), _env("PL_HIP_POLAR_MFMA", "1"), _env("PL_HIP_PQ_SEGMENTS", "0"):
            src = gpu.tex_create(w, h, "rgba16", hdr)
            dst = gpu.tex_create(dw, dh, "rgba16")
            rr = pl.Renderer(gpu)
            util.srand(1)
            assert rr.render(pl.frame(src, components=3, color=pl.color_space("bt2020", "pq", max_luma=1000.0)),
                             pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT),
                                      color=pl.color_space("bt709", "bt1886")), params)
            assert rr.errors() == 0
            outs.append(dst.download())
            rr.destroy(); src.destroy(); dst.destroy()
    assert outs[0][..., :3].std() > 1000
    if case == "map_pass_contrast_recovery":
        # The interpreter keeps contrast recovery on its per-op path (op_tone_map between
        # op_rgb2ipt and op_ipt2rgb, IEEE divisions); the chain has it inside the fused map
        # (cm_fused<NP, true>: reciprocal + Newton step, the well-conditioned PQ pair). Same
        # formulas, both within the oracle's tolerance (test_gpu_metric.py): behind a 10-bit
        # dither they may differ by one 10-bit step on a few samples.
        d = np.abs(outs[0].astype(np.int64) - outs[1].astype(np.int64))
        assert d.max() <= 65 and (d > 0).mean() < 0.01, (d.max(), (d > 0).mean())
    else:
        assert_same_up_to_a_dither_step(outs[0], outs[1])


@pytest.mark.parametrize("case", ["ewa_2x_map", "ewa_2x_map_no_peak", "ewa_3x_map"])
@pytest.mark.parametrize("size", [(97, 61), (256, 130), (480, 270)])
def test_pq_segments_against_closed_forms(gpu, case, size):
    """The PQ pair of the map chain as piecewise cubics in LDS (csrc/hip/pqseg.hiph: the default of the
    chain kernels that have the variant) against the closed forms of pqmath.hiph (PL_HIP_PQ_SEGMENTS=0)
    on the metric's launch (k_polar_mx's chain epilogue; "ewa_3x_map": k_polar_mxr's, the 3x upscale of
    720p -> 4K), 16-bit target without a dither so that every difference shows: both are
    approximations of the same curves (tests/test_pqseg.py: the pieces are the closer one). On the bulk
    the frames agree (97 % of the samples identical, 99.8 % within a code); the colour map amplifies
    what is left on saturated colours -- a few codes on a few samples, the same kind and size of
    difference either has against the fp32 oracle, and the statement against float64 is
    tests/util.py::assert_colormap_parity's (test_gpu_metric.py runs it on the default, the pieces).
    The number of copies per piece in LDS (a bank-conflict matter) changes
    nothing at all; a frame with values beyond the tables' range (PQ codes above 1.25 out of an
    rgba16hf source) takes the closed forms wave by wave -- no clamp, no NaN, the same frame to the
    same tolerance. (That frame is also what found the closed form's own clamp of the EOTF's quotient
    to [0, 1], which the reference does not have: pqmath.hiph.)"""
    from test_gpu_fullsize import hdr_frame16
    w, h = size
    hdr = hdr_frame16(w, h)
    peak = pl.peak_detect_params(percentile=99.995) if case != "ewa_2x_map_no_peak" else None
    up = 3 if case == "ewa_3x_map" else 2
    params = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), peak_detect_params=peak,
                              dither_params=None)

    def run(env, img=hdr, fmt="rgba16"):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            src = gpu.tex_create(w, h, fmt, img)
            dst = gpu.tex_create(up * w, up * h, "rgba16")
            rr = pl.Renderer(gpu)
            util.srand(1)
            assert rr.render(pl.frame(src, components=3, color=pl.color_space("bt2020", "pq", max_luma=1000.0)),
                             pl.frame(dst, color=pl.color_space("bt709", "bt1886")), params)
            assert rr.errors() == 0
            out = dst.download()
            rr.destroy(); src.destroy(); dst.destroy()
            return out
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

    closed = run({"PL_HIP_PQ_SEGMENTS": "0", "PL_HIP_POLAR_MFMA": "1"})
    seg = run({"PL_HIP_PQ_SEGMENTS": "1", "PL_HIP_POLAR_MFMA": "1"})
    assert closed[..., :3].std() > 1000
    d = np.abs(seg.astype(np.int64) - closed.astype(np.int64))
    print("pq segments vs closed forms: max", d.max(), "differing", (d > 0).mean(), "> 1 code", (d > 1).mean())
    assert d.max() <= 16 and (d > 1).mean() < 0.005 and (d > 0).mean() < 0.06, (d.max(), (d > 0).mean(), (d > 1).mean())
    for copies in ("2", "4", "16"):
        again = run({"PL_HIP_PQ_SEGMENTS": "1", "PL_HIP_POLAR_MFMA": "1", "PL_HIP_PQ_SEG_COPIES": copies})
        assert np.array_equal(again, seg), copies
    # beyond the tables: an rgba16hf frame whose left half carries PQ "codes" up to 1.6
    f16 = (hdr.astype(np.float32) / 65535.0)
    f16[:, : w // 2, :3] *= 1.6 / max(float(f16[..., :3].max()), 1e-3)
    f16 = f16.astype(np.float16)
    a = run({"PL_HIP_PQ_SEGMENTS": "0", "PL_HIP_POLAR_MFMA": "1"}, f16, "rgba16hf")
    b = run({"PL_HIP_PQ_SEGMENTS": "1", "PL_HIP_POLAR_MFMA": "1"}, f16, "rgba16hf")
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    # (up to 160 000 cd/m^2 through the colour map: what either form rounds differently is amplified a
    # little further than on the frame above)
    assert d.max() <= 24 and (d > 1).mean() < 0.005, (d.max(), (d > 1).mean())


def assert_same_up_to_a_dither_step(chain, interp):
    """HDR chains: the chain kernels take the uniform scale factors between the colour map's stages
    (203 / 10000 and back, the black-scaling slope) folded into its two matrices by the launcher,
    the interpreter multiplies them into every pixel (struct plh_map_chain: in_mat / out_mat).
    Same formulas, products rounded at another place: an fp32 ulp before the dither, which moves a
    10-bit code only where it lands on a threshold -- one step, on a vanishing share of the samples."""
    d = np.abs(chain.astype(np.int64) - interp.astype(np.int64))
    assert d.max() <= 65 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())


@pytest.mark.parametrize("size", [(97, 61), (256, 130)])
def test_features_pass_equals_generic(gpu, size):
    """k_pass_features (pl_shader_extract_features as its own pass, the feature map of contrast
    recovery) against k_pass_generic: the high_quality preset on an HDR10 frame, everything else
    held on the interpreter (PL_HIP_MAP_CHAIN=0), must give the same frame bit for bit."""
    from test_gpu_fullsize import hdr_frame16
    w, h = size
    hdr = hdr_frame16(w, h)
    params = pl.render_params("high_quality", peak_detect_params=pl.peak_detect_params(percentile=99.995))
    outs = []
    for native in ("1", "0"):
        with _env("PL_HIP_MAP_CHAIN", "0"), _env("PL_HIP_PASS_NATIVE", native):
            src = gpu.tex_create(w, h, "rgba16", hdr)
            dst = gpu.tex_create(w, h, "rgba16")
            rr = pl.Renderer(gpu)
            util.srand(1)
            assert rr.render(pl.frame(src, components=3, color=pl.color_space("bt2020", "pq", max_luma=1000.0)),
                             pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT),
                                      color=pl.color_space("bt709", "bt1886")), params)
            assert rr.errors() == 0
            outs.append(dst.download())
            rr.destroy(); src.destroy(); dst.destroy()
    assert np.array_equal(outs[0], outs[1]) and outs[0][..., :3].std() > 1000


@pytest.mark.parametrize("scaler", ["lanczos", "ewa_lanczos", "mitchell"])
@pytest.mark.parametrize("size", [(90, 62), (256, 130)])
def test_sdr_chain_equals_interpreter(gpu, scaler, size):
    """pl_render_default_params on an SDR frame: the last scaler pass carries UNSIGMOIDIZE +
    DELINEARIZE + dither + scale -- the chain without a colour map (k_ortho_fast EPI 4, the CHAIN
    epilogues of k_polar_pp and k_polar_mx) against the op interpreter (PL_HIP_MAP_CHAIN=0):
    bit-identical -- for the separable kernels and k_polar_pp by construction (the same device
    functions in the same order on the same sums). The two variants of k_polar_mx apply the row-phase
    term of the contraction at different places (the CHAIN variant to the finished sums of a row
    phase, the full-interpreter variant inside the contraction: k_polar_mx.hiph, YPHASE), an fp32
    ulp apart wherever that term is not zero: identical on the bulk, one dither step on a sample in
    10^4 at most (round 5's 81-MFMA pairing happened to flip none on these two frames; round 6's
    54-MFMA pairing flips one of 67 000)."""
    sw, sh = size
    img = util.chirp_rgba16(sw, sh)
    params = pl.render_params("default", upscaler=pl.filter_config(scaler))
    for mfma in ("1", "0"):
        outs = []
        for chain in ("1", "0"):
            with _env("PL_HIP_MAP_CHAIN", chain), _env("PL_HIP_POLAR_MFMA", mfma):
                outs.append(render(gpu, img, 2 * sw, 2 * sh, params, True, {}))
        assert outs[0][..., :3].std() > 1000
        if scaler == "ewa_lanczos" and mfma == "1":
            d = np.abs(outs[0].astype(np.int64) - outs[1].astype(np.int64))
            assert d.max() <= 64 and (d > 0).mean() <= 1e-4, (int(d.max()), float((d > 0).mean()))
        else:
            assert np.array_equal(outs[0], outs[1]), util.diff_stats(outs[0], outs[1])


@pytest.mark.parametrize("case", ["hdr_2x", "hdr_1.5x", "sdr_default_1.5x", "hdr_rgba_2x"])
def test_polar_pp_chain_equals_interpreter(gpu, case):
    """The CHAIN variant of the phase-class polar kernel (k_polar_pp: every EWA geometry that is not
    on the matrix pipe, and the bit-exact variant of those that are) against its full-interpreter
    variant (PL_HIP_MAP_CHAIN=0): the HDR colour map behind a 2x and a 1.5x upscale, the default
    preset's unsigmoidize + delinearize behind a 1.5x one, sampled alpha. Bit-identical."""
    from test_gpu_fullsize import hdr_frame16
    sw, sh = 120, 70
    hdr = case.startswith("hdr")
    img = hdr_frame16(sw, sh) if hdr else util.chirp_rgba16(sw, sh)
    scale = 2.0 if "2x" in case else 1.5
    dw, dh = int(sw * scale), int(sh * scale)
    kw = dict(upscaler=pl.filter_config("ewa_lanczos"))
    if hdr:
        kw["peak_detect_params"] = pl.peak_detect_params(percentile=99.995)
    params = pl.render_params("default", **kw)
    outs = []
    for chain in ("1", "0"):
        with _env("PL_HIP_MAP_CHAIN", chain), _env("PL_HIP_POLAR_MFMA", "0"):
            src = gpu.tex_create(sw, sh, "rgba16", img)
            dst = gpu.tex_create(dw, dh, "rgba16")
            rr = pl.Renderer(gpu)
            util.srand(1)
            ikw = dict(color=pl.color_space("bt2020", "pq", max_luma=1000.0)) if hdr else {}
            tkw = dict(color=pl.color_space("bt709", "bt1886")) if hdr else {}
            assert rr.render(pl.frame(src, components=4 if "rgba" in case else 3, **ikw),
                             pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT), **tkw), params)
            assert rr.errors() == 0
            outs.append(dst.download())
            rr.destroy(); src.destroy(); dst.destroy()
    assert outs[0][..., :3].std() > 1000
    if hdr:
        assert_same_up_to_a_dither_step(outs[0], outs[1])
    else:
        assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("preset", ["fast", "default", "high_quality"])
@pytest.mark.parametrize("semi,bits,sub", [(True, 8, (2, 2)), (False, 8, (2, 2)), (True, 16, (2, 1))])
def test_merge_pass_equals_generic(gpu, preset, semi, bits, sub):
    """k_pass_merge (the pass that assembles a planar frame: reference plane + fetched planes + YCbCr
    matrix + the preset's linearize / sigmoidize, odd widths included) against k_pass_generic
    (PL_HIP_PASS_NATIVE=0), NV12 / I420 / P216-style input through three presets: bit-identical."""
    from test_gpu_renderer import planar_frame
    w, h = 66, 46
    outs = []
    for native in ("1", "0"):
        with _env("PL_HIP_PASS_NATIVE", native):
            f, texs, _ = planar_frame(gpu, w, h, seed=5, sub=sub, bits=bits, semi=semi)
            f.repr = pl.color_repr("bt709", "limited", sample_depth=bits, color_depth=bits)
            f.color = pl.color_space("bt709", "bt1886")
            pl.lib().pl_frame_set_chroma_location.argtypes = [C.POINTER(capi.Frame), C.c_int]
            pl.lib().pl_frame_set_chroma_location(C.byref(f), 1)
            dst = gpu.tex_create(2 * w, 2 * h, "rgba16")
            rr = pl.Renderer(gpu)
            util.srand(1)
            assert rr.render(f, pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT),
                                         color=pl.color_space("bt709", "bt1886")),
                             pl.render_params(preset, deband_params=None)), gpu.messages[-4:]
            assert rr.errors() == 0
            outs.append(dst.download())
            rr.destroy(); dst.destroy()
            for t in texs:
                t.destroy()
    assert np.array_equal(outs[0], outs[1]) and outs[0][..., :3].std() > 1000
