"""Degenerate and ragged frame sizes through pl_render_image (the edge cases the
reference's own tests walk -- gpu_tests.c renders 1x1 ... odd sizes through every preset): one
pixel, one row, one column, primes, sizes just below / above the tile shapes of the specialised
kernels (16x16 measuring tiles, 64x32 debanding windows, 128x64 polar tiles). For each, the frame
the library renders with its DEFAULT kernels must be the frame the generic kernels render (every
specialised kernel switched off): bit for bit where no matrix-pipe scaler is involved, within one
16-bit code where one is (tests/util.py: assert_polar_equal). The generic kernels are the ones
pinned to the oracle at ordinary sizes; what this file adds is that no kernel reads or writes
outside a tiny image, mis-handles a tile that is mostly padding, or divides by a zero extent."""
import ctypes as C
import os

import numpy as np
import pytest

import libplacebo_amd as pl
import util
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu

GENERIC = {"PL_HIP_POLAR_MFMA": "0", "PL_HIP_PEAK_FAST": "0", "PL_HIP_PASS_NATIVE": "0",
           "PL_HIP_DEBAND_FAST": "0", "PL_HIP_ORTHO_FAST": "0", "PL_HIP_BILIN_ITERS": "0",
           "PL_HIP_MAP_CHAIN": "0", "PL_HIP_FUSED_FEATURES": "0"}
TEN_BIT = dict(sample_depth=16, color_depth=10, bit_shift=6)


def frame16(w, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 65536, (h, w, 4)).astype(np.uint16)
    img[..., 3] = 65535
    return img


def render(gpu, img, dw, dh, preset, hdr, env, **kw):
    old = {k: os.environ.get(k) for k in GENERIC}
    for k in GENERIC:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        sh, sw = img.shape[:2]
        src = gpu.tex_create(sw, sh, "rgba16", img)
        dst = gpu.tex_create(dw, dh, "rgba16")
        if hdr:
            image = pl.frame(src, components=3, color=pl.color_space("bt2020", "pq", max_luma=1000.0))
            target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"),
                              repr_=pl.color_repr("rgb", "full", **TEN_BIT))
            kw.setdefault("peak_detect_params", pl.peak_detect_params(percentile=99.995))
        else:
            image = pl.frame(src, components=3)
            target = pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT))
        kw.setdefault("dither_params", capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0))
        params = pl.render_params(preset, **kw)
        rr = pl.Renderer(gpu)
        util.srand(1)
        assert rr.render(image, target, params), gpu.messages[-4:]
        assert rr.errors() == 0, (rr.errors(), gpu.messages[-4:])
        out = dst.download()
        rr.destroy(); src.destroy(); dst.destroy()
        return out
    finally:
        for k in GENERIC:
            os.environ.pop(k, None)
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


SIZES = [((1, 1), (1, 1)), ((1, 1), (7, 5)), ((2, 1), (4, 2)), ((1, 2), (2, 4)), ((3, 2), (1, 1)),
         ((5, 3), (10, 6)), ((13, 11), (26, 22)), ((17, 15), (34, 30)), ((31, 17), (62, 34)),
         ((63, 33), (126, 66)), ((65, 33), (130, 66)), ((129, 65), (258, 130)),
         ((16, 16), (8, 8)), ((34, 18), (17, 9)), ((33, 31), (11, 10)), ((127, 3), (254, 6)),
         ((3, 127), (6, 254))]


@pytest.mark.parametrize("preset,hdr,kw", [
    ("fast", False, {}),
    ("default", False, {}),
    ("default", False, {"upscaler": "ewa_lanczos", "downscaler": "ewa_lanczos"}),
    ("high_quality", False, {}),
    ("default", True, {}),
    ("default", True, {"upscaler": "ewa_lanczos", "downscaler": "ewa_lanczos"}),
    ("high_quality", True, {}),
], ids=["fast", "default", "default-ewa", "hq", "hdr-default", "hdr-ewa", "hdr-hq"])
def test_tiny_and_ragged_frames_default_kernels_equal_the_generic_ones(gpu, preset, hdr, kw):
    worst = 0
    for n, ((sw, sh), (dw, dh)) in enumerate(SIZES):
        img = frame16(sw, sh, 100 + n)
        k = {key: (pl.filter_config(v) if key in ("upscaler", "downscaler") else v) for key, v in kw.items()}
        fast = render(gpu, img, dw, dh, preset, hdr, {}, **k)
        k = {key: (pl.filter_config(v) if key in ("upscaler", "downscaler") else v) for key, v in kw.items()}
        slow = render(gpu, img, dw, dh, preset, hdr, GENERIC, **k)
        assert fast.shape == (dh, dw, 4)
        d = np.abs(fast.astype(np.int64) - slow.astype(np.int64))
        # a 10-bit dithered frame: one code of 16 bits before the dither is at most one 10-bit step
        # (64) behind it, on the samples whose dither decision it flips
        polar = preset == "high_quality" or "upscaler" in kw
        if hdr:
            # (the HDR map is ill-conditioned on saturated noise: where two routes round the f16
            # intermediate differently a sample can move by several codes -- tests/util.py,
            # DESIGN.md section 6 -- and with it by a second dither step, on a handful of samples)
            worst = max(worst, d.max())
            assert d.max() <= 192 and (d > 64).mean() <= 5e-3, ((sw, sh), (dw, dh), d.max(), (d > 64).mean())
            assert (d > 0).mean() <= 0.05 or fast.size <= 256, ((sw, sh), (dw, dh), (d > 0).mean())
        elif polar:
            assert d.max() <= 64, ((sw, sh), (dw, dh), d.max())
            assert (d > 0).mean() <= 0.02 or fast.size <= 64, ((sw, sh), (dw, dh), (d > 0).mean())
        else:
            assert d.max() == 0, ((sw, sh), (dw, dh), d.max(), np.argwhere(d > 0)[:3])


def render_frames(gpu, image, target, dst, preset, env, **kw):
    old = {k: os.environ.get(k) for k in GENERIC}
    for k in GENERIC:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        kw.setdefault("dither_params", capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0))
        rr = pl.Renderer(gpu)
        util.srand(1)
        assert rr.render(image, target, pl.render_params(preset, **kw)), gpu.messages[-4:]
        assert rr.errors() == 0, (rr.errors(), gpu.messages[-4:])
        out = dst.download()
        rr.destroy()
        return out
    finally:
        for k in GENERIC:
            os.environ.pop(k, None)
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v


@pytest.mark.parametrize("preset", ["fast", "default", "high_quality"])
def test_ragged_crops_flips_and_rotations(gpu, preset):
    """Fractional and flipped source crops, a target crop that leaves a border, the frame rotated by
    90 / 180 / 270 degrees, at sizes that are no multiple of anything: default kernels = generic
    kernels (bit for bit where no matrix-pipe scaler can be involved, one 10-bit step otherwise)."""
    sw, sh = 53, 37
    img = frame16(sw, sh, 7)
    cases = [dict(crop=(3.25, 2.5, 47.75, 33.0), tcrop=None, rot=0),
             dict(crop=(50.0, 35.0, 2.0, 1.0), tcrop=None, rot=0),            # flipped on both axes
             dict(crop=None, tcrop=(5.0, 3.0, 71.0, 49.0), rot=0),           # border around the image
             dict(crop=None, tcrop=None, rot=1), dict(crop=None, tcrop=None, rot=2),
             dict(crop=(1.5, 0.0, 40.0, 37.0), tcrop=None, rot=3)]
    for case in cases:
        outs = []
        for env in ({}, GENERIC):
            src = gpu.tex_create(sw, sh, "rgba16", img)
            dst = gpu.tex_create(79, 61, "rgba16")
            image = pl.frame(src, components=3, crop=case["crop"])
            image.rotation = case["rot"]
            target = pl.frame(dst, repr_=pl.color_repr("rgb", "full", **TEN_BIT), crop=case["tcrop"])
            outs.append(render_frames(gpu, image, target, dst, preset, env))
            src.destroy(); dst.destroy()
        d = np.abs(outs[0].astype(np.int64) - outs[1].astype(np.int64))
        if preset == "high_quality":
            assert d.max() <= 64 and (d > 0).mean() <= 0.02, (case, d.max(), (d > 0).mean())
        else:
            assert d.max() == 0, (case, d.max(), np.argwhere(d > 0)[:3])


@pytest.mark.parametrize("size", [(2, 2), (6, 4), (18, 10), (34, 18), (66, 34), (130, 66)])
@pytest.mark.parametrize("preset", ["fast", "default", "high_quality"])
def test_tiny_planar_video_frames(gpu, preset, size):
    """NV12 (8-bit, 4:2:0) frames down to one chroma texel, upscaled 2x into a 10-bit target: the
    plane merge, chroma scaling and main scaler on their default kernels against the generic ones."""
    w, h = size
    rng = np.random.default_rng(w * 31 + h)
    y = rng.integers(16, 236, (h, w, 1)).astype(np.uint8)
    uv = rng.integers(16, 241, (h // 2, w // 2, 2)).astype(np.uint8)
    outs = []
    for env in ({}, GENERIC):
        ty, tuv = gpu.tex_create(w, h, "r8", y), gpu.tex_create(w // 2, h // 2, "rg8", uv)
        f = capi.Frame(num_planes=2)
        for i, (t, comps, mapping) in enumerate([(ty, 1, [0]), (tuv, 2, [1, 2])]):
            f.planes[i].texture = t.ptr
            f.planes[i].components = comps
            for c in range(4):
                f.planes[i].component_mapping[c] = mapping[c] if c < comps else -1
        f.repr = pl.color_repr("bt709", "limited", sample_depth=8, color_depth=8)
        f.color = pl.color_space("bt709", "bt1886")
        pl.lib().pl_frame_set_chroma_location.argtypes = [C.POINTER(capi.Frame), C.c_int]
        pl.lib().pl_frame_set_chroma_location(C.byref(f), 1)     # PL_CHROMA_LEFT
        dst = gpu.tex_create(2 * w, 2 * h, "rgba16")
        target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"),
                          repr_=pl.color_repr("rgb", "full", **TEN_BIT))
        outs.append(render_frames(gpu, f, target, dst, preset, env))
        ty.destroy(); tuv.destroy(); dst.destroy()
    d = np.abs(outs[0].astype(np.int64) - outs[1].astype(np.int64))
    if preset == "high_quality":
        assert d.max() <= 64 and ((d > 0).mean() <= 0.02 or d.size <= 256), (size, d.max(), (d > 0).mean())
    else:
        assert d.max() == 0, (size, d.max(), np.argwhere(d > 0)[:3])


@pytest.mark.parametrize("size", [((16384, 24), (16384, 24)), ((24, 16384), (24, 16384)),
                                  ((8192, 18), (16384, 36)), ((18, 8192), (36, 16384))])
@pytest.mark.parametrize("preset,hdr", [("fast", False), ("default", False), ("default", True)])
def test_very_wide_and_very_tall_frames(gpu, preset, hdr, size):
    """16384 texels along one axis, a couple of tiles along the other: index arithmetic, pitches and
    the measuring pass's tile walk at the far end of a row / column (default = generic kernels)."""
    (sw, sh), (dw, dh) = size
    img = frame16(sw, sh, sw + sh)
    fast = render(gpu, img, dw, dh, preset, hdr, {})
    slow = render(gpu, img, dw, dh, preset, hdr, GENERIC)
    d = np.abs(fast.astype(np.int64) - slow.astype(np.int64))
    if hdr:
        assert d.max() <= 192 and (d > 64).mean() <= 5e-3 and (d > 0).mean() <= 0.05, (d.max(), (d > 0).mean())
    else:
        assert d.max() == 0, (d.max(), np.argwhere(d > 0)[:3])
