"""Second opinions on the oracle's shader half (CPU only).

The reference holds no numeric vectors for its samplers (src/tests/gpu_tests.c:1216 "TODO"), so
oracle/pl_oracle.c is a restatement of GLSL. To keep a transcription slip from being "parity",
each stage below is evaluated a second time, independently of that C code and in another
arithmetic: written in numpy directly from the reference's shader text / definitions, float64
where the oracle is float32, the filter evaluated directly (pl_filter_sample per tap) where the
oracle goes through the 256-row LUT. The polar sampler's counterpart is in tests/test_host.py."""
import ctypes as C

import numpy as np
import pytest

import orc
import util


def filter_value(f, x):
    orc.lib().orc_filter_sample.restype = C.c_double
    return orc.lib().orc_filter_sample(C.byref(f), C.c_double(float(abs(x))))


def separable_direct(img, f, taps, axis, out_n):
    """src/shaders/sampling.c:1040-1090 (pl_shader_sample_ortho2), one pass: position of the
    output sample on the source axis, `taps` neighbours starting at floor(pos) - (taps/2 - 1),
    weight = the filter at the tap's distance, weights normalised, clamped addressing. float64."""
    img = np.moveaxis(img.astype(np.float64), axis, 0)
    n_in = img.shape[0]
    out = np.zeros((out_n,) + img.shape[1:])
    for o in range(out_n):
        pos = (o + 0.5) / out_n * n_in - 0.5
        base = int(np.floor(pos))
        frac = pos - base
        w = np.array([filter_value(f, k - (taps // 2 - 1) - frac) for k in range(taps)])
        w /= w.sum()
        for k in range(taps):
            out[o] += w[k] * img[min(max(base - (taps // 2 - 1) + k, 0), n_in - 1)]
    return np.moveaxis(out, 0, axis)


@pytest.mark.parametrize("mk,name", [(orc.lanczos, "lanczos"), (orc.mitchell, "mitchell")])
@pytest.mark.parametrize("sizes", [((40, 28), (100, 28)), ((40, 28), (40, 70)), ((40, 28), (57, 28))])
def test_separable_sampler_against_direct_float64_evaluation(mk, name, sizes):
    (sw, sh), (dw, dh) = sizes
    img = orc.tex_decode(util.random_rgba16(sw, sh, seed=9), "rgba16")
    f = mk()
    rows, taps, radius, _ = orc.filter_generate_ortho(f)
    direction = 0 if dw != sw else 1
    got = orc.sample_ortho(img, rows, taps, direction, dw, dh)
    want = separable_direct(img, f, taps, 1 if direction == 0 else 0, dw if direction == 0 else dh)
    # the LUT's 256 phases, linearly interpolated: ~1e-5 of weight error per tap
    assert np.abs(got - want).max() < 2e-4, np.abs(got - want).max()
    # and the LUT rows themselves are the normalised direct weights
    for phase in (0, 17, 128, 255):
        frac = phase / 255.0
        w = np.array([filter_value(f, k - (taps // 2 - 1) - frac) for k in range(taps)])
        assert np.abs(rows[phase, :taps] - w / w.sum()).max() < 1e-6


def pcg3d(s):
    """src/shaders.c:970-983, on uint32 arrays (numpy wraps modulo 2^32)"""
    x, y, z = (np.uint32(1664525) * c + np.uint32(1013904223) for c in s)
    x = x + y * z; y = y + z * x; z = z + x * y
    x = x ^ (x >> np.uint32(16)); y = y ^ (y >> np.uint32(16)); z = z ^ (z >> np.uint32(16))
    x = x + y * z; y = y + z * x; z = z + x * y
    k = 1.0 / 4294967296.0      # 1.0 / float(0xFFFFFFFFu): the float conversion rounds to 2^32
    return (x, y, z), (x * k, y * k, z * k)


def deband_numpy(tex, iterations, threshold, radius, grain, neutral, seed):
    """src/shaders/sampling.c:183-275, written from the GLSL: float64 positions and averages"""
    h, w = tex.shape[:2]
    ys, xs = np.mgrid[0:h, 0:w]
    px, py = (xs + 0.5) / w, (ys + 0.5) / h
    with np.errstate(over="ignore"):
        state = (xs.astype(np.uint32), ys.astype(np.uint32), np.full((h, w), seed, np.uint32))
        color = tex.astype(np.float64).copy()
        res = color[..., :3].copy()

        def get(dx, dy):
            ix = np.clip(np.floor((px + dx / w) * w).astype(int), 0, w - 1)
            iy = np.clip(np.floor((py + dy / h) * h).astype(int), 0, h - 1)
            return tex[iy, ix, :3].astype(np.float64)

        for i in range(1, iterations + 1):
            state, r = pcg3d(state)
            dist, ang = r[0] * (i * radius), r[1] * 6.283185
            dx, dy = dist * np.cos(ang), dist * np.sin(ang)
            avg = 0.25 * (get(dx, dy) + get(-dx, dy) + get(-dx, -dy) + get(dx, -dy))
            keep = np.abs(res - avg) > (threshold / 1000.0) / i
            res = np.where(keep, res, avg)
        if grain > 0:
            state, r = pcg3d(state)
            strength = np.minimum(np.abs(res - np.asarray(neutral)), grain / 1000.0)
            res = res + strength * (np.stack(r, axis=-1) - 0.5)
    color[..., :3] = res
    return color


@pytest.mark.parametrize("kw", [dict(), dict(iterations=3, grain=0.0),
                                dict(iterations=2, threshold=6.0, radius=8.0, grain=8.0,
                                     neutral=(0.1, 0.2, 0.0))])
def test_deband_against_numpy_restatement_of_the_glsl(kw):
    w, h = 96, 64
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 4), np.float32)
    img[..., 0] = np.floor((x / w) * 63) / 63
    img[..., 1] = np.floor(((x + y) / (w + h)) * 63) / 63
    img[..., 2] = np.floor((y / h) * 63) / 63
    img[..., 3] = 1.0
    p = dict(iterations=1, threshold=3.0, radius=16.0, grain=4.0, neutral=(0.0, 0.0, 0.0))
    p.update(kw)
    got = orc.deband(img, w, h, iterations=p["iterations"], threshold=p["threshold"],
                     radius=p["radius"], grain=p["grain"], grain_neutral=p["neutral"], frame_index=5)
    want = deband_numpy(img, p["iterations"], p["threshold"], p["radius"], p["grain"], p["neutral"], 5)
    d = np.abs(got - want)
    # float32 vs float64: a sample point within ~1e-5 px of a texel edge, or a difference within
    # an ulp of the threshold, may go the other way on a few pixels; everywhere else the two agree
    # to float32 rounding
    assert (d.max(axis=2) < 1e-6).mean() > 0.995, (d.max(axis=2) < 1e-6).mean()
    assert d.max() <= (p["threshold"] + p["grain"]) / 1000.0 + 1e-6
    if p["grain"]:
        assert len(np.unique(got[..., 0])) > 2 * 64     # it did something
