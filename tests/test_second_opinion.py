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


# ---- deinterlacing (src/shaders/deinterlacing.c), a whole frame at a time in float64 ----------

def _mirror(i, n):
    m = np.mod(i, 2 * n)
    return np.where(m < n, m, 2 * n - 1 - m)


def _get(img, dx, dy):
    """GET(TEX, X, Y) for every pixel at once: the frame shifted, MIRROR addressing"""
    h, w = img.shape[:2]
    ys = _mirror(np.arange(h) + dy, h)
    xs = _mirror(np.arange(w) + dx, w)
    return img[ys][:, xs]


def yadif_numpy(cur, prev, nxt, field, first, skip_spatial_check=False):
    cur, prev, nxt = (a.astype(np.float64) for a in (cur, prev, nxt))
    a_, b, c, d, e, f, g = (_get(cur, k, -1) for k in range(-3, 4))
    h, i, j, k, l, m, n = (_get(cur, q, +1) for q in range(-3, 4))
    pred = (d + k) / 2
    best = np.abs(c - j) + np.abs(d - k) + np.abs(e - l) - 0.003922
    s1 = np.abs(b - k) + np.abs(c - l) + np.abs(d - m)
    s2 = np.abs(a_ - l) + np.abs(b - m) + np.abs(c - n)
    left1 = s1 < best
    left2 = left1 & (s2 < s1)
    pred = np.where(left1, (c + l) / 2, pred)
    best = np.where(left1, s1, best)
    pred = np.where(left2, (b + m) / 2, pred)
    best = np.where(left2, s2, best)
    s3 = np.abs(d - i) + np.abs(e - j) + np.abs(f - k)
    s4 = np.abs(e - h) + np.abs(f - i) + np.abs(g - j)
    right1 = s3 < best
    right2 = right1 & (s4 < s3)
    pred = np.where(right1, (e + j) / 2, pred)
    pred = np.where(right2, (f + i) / 2, pred)

    prev2, next2 = (prev, cur) if field == first else (cur, nxt)
    A, B = _get(prev, 0, -1), _get(prev, 0, 1)
    C_, D, E = _get(prev2, 0, -2), _get(prev2, 0, 0), _get(prev2, 0, 2)
    F, G = _get(cur, 0, -1), _get(cur, 0, 1)
    H, I, J = _get(next2, 0, -2), _get(next2, 0, 0), _get(next2, 0, 2)
    K, L = _get(nxt, 0, -1), _get(nxt, 0, 1)
    p0, p1, p2, p3, p4 = (C_ + H) / 2, F, (D + I) / 2, G, (E + J) / 2
    diff = np.maximum(np.abs(D - I) / 2,
                      np.maximum((np.abs(A - F) + np.abs(B - G)) / 2, (np.abs(K - F) + np.abs(G - L)) / 2))
    if not skip_spatial_check:
        maxi = np.maximum(p2 - np.minimum(p3, p1), np.minimum(p0 - p1, p4 - p3))
        mini = np.minimum(p2 - np.maximum(p3, p1), np.maximum(p0 - p1, p4 - p3))
        diff = np.maximum(diff, np.maximum(mini, -maxi))
    pred = np.minimum(pred, p2 + diff)
    pred = np.where(pred < p2 - diff, p2 - diff, pred)
    return pred


def bwdif_numpy(cur, prev, nxt, field, first):
    cur, prev, nxt = (a.astype(np.float64) for a in (cur, prev, nxt))
    prev2, next2 = (prev, cur) if field == first else (cur, nxt)
    c3m, c, e, c3p = _get(cur, 0, -3), _get(cur, 0, -1), _get(cur, 0, 1), _get(cur, 0, 3)
    p2 = [_get(prev2, 0, k) for k in (-4, -2, 0, 2, 4)]
    n2 = [_get(next2, 0, k) for k in (-4, -2, 0, 2, 4)]
    s = p2[2] + n2[2]
    d = s / 2
    tdiff0 = np.abs(p2[2] - n2[2])
    tdiff1 = np.abs(_get(prev, 0, -1) - c) + np.abs(_get(prev, 0, 1) - e)
    tdiff2 = np.abs(_get(nxt, 0, -1) - c) + np.abs(_get(nxt, 0, 1) - e)
    diff = np.maximum(tdiff0, np.maximum(tdiff1, tdiff2)) / 2
    still = diff == 0
    bs, fs = p2[1] + n2[1], p2[3] + n2[3]
    b, f = bs / 2 - c, fs / 2 - c
    dc, de = d - c, d - e
    mmax = np.maximum(de, np.maximum(dc, np.minimum(b, f)))
    mmin = np.minimum(de, np.minimum(dc, np.maximum(b, f)))
    diff = np.maximum(diff, np.maximum(mmin, -mmax))
    single = 5077 / 8192 * (c + e) - 981 / 8192 * (c3m + c3p)
    allf = (5570 / 8192 * s - 3801 / 8192 * (bs + fs) + 1016 / 8192 * (p2[0] + n2[0] + p2[4] + n2[4])) / 4
    allf = allf + 4309 / 8192 * (c + e) - 213 / 8192 * (c3m + c3p)
    interpol = np.where(np.abs(c - e) > tdiff0, allf, single)
    interpol = np.clip(interpol, d - diff, d + diff)
    return np.where(still, d, interpol)


@pytest.mark.parametrize("algo", ["yadif", "bwdif"])
def test_deinterlacers_against_numpy_restatements_of_the_glsl(algo):
    """pl_shader_deinterlace written a third time: whole frames at once in numpy, float64, from
    the shader text (the oracle walks pixels in C, float32; the kernels own row pairs). The
    decisions (yadif's direction, bwdif's filter choice and its clamp) sit on comparisons, so the
    inputs are 8-bit frames, where float32 and float64 agree on every comparison that is not an
    exact tie -- and ties resolve the same way in both (strict <, >)."""
    rng = np.random.default_rng(12)
    h, w = 38, 52
    frames = [np.zeros((h, w, 4), np.float32) for _ in range(3)]
    yy, xx = np.mgrid[0:h, 0:w]
    for t, f in enumerate(frames):
        for ch in range(4):
            v = 128 + 100 * np.sin((xx + 4 * t) * (0.2 + 0.05 * ch) + yy * 0.3) + rng.integers(-20, 20, (h, w))
            f[..., ch] = np.clip(np.rint(v), 0, 255) / 255
    for field in (1, 2):
        for first in (1, 2):
            want = orc.deinterlace(frames[1], frames[0], frames[2], field, first,
                                   orc.DEINT_YADIF if algo == "yadif" else orc.DEINT_BWDIF)
            if algo == "yadif":
                full = yadif_numpy(frames[1], frames[0], frames[2], field, first)
            else:
                full = bwdif_numpy(frames[1], frames[0], frames[2], field, first)
            rebuilt = slice(1, None, 2) if field == 1 else slice(0, None, 2)
            kept = slice(0, None, 2) if field == 1 else slice(1, None, 2)
            assert np.array_equal(want[kept], frames[1][kept])
            d = np.abs(want[rebuilt].astype(np.float64) - full[rebuilt])
            # a handful of pixels may sit on a float32 / float64 tie of a comparison: they then
            # take the other branch, whose value differs by much more than rounding
            assert (d > 1e-6).mean() < 0.002, (field, first, float((d > 1e-6).mean()), float(d.max()))


def test_dovi_reshape_against_a_numpy_restatement():
    """pl_shader_dovi_reshape per pixel in numpy float64 from the shader text (colorspace.c:106-271):
    the binary tree of pivots written out, the MMR dot products term by term"""
    rng = np.random.default_rng(4)
    comps = (orc.DoviComp * 3)()
    for c in range(3):
        k = comps[c]
        k.num_pivots = 5
        for i, v in enumerate((0.0, 0.2, 0.5, 0.8, 1.0)):
            k.pivots[i] = v
        for i in range(4):
            if (i + c) % 2:
                k.method[i] = 0
                for j in range(3):
                    k.poly_coeffs[i][j] = float(rng.normal() * 0.3 + (1 if j == 1 else 0))
            else:
                k.method[i] = 1
                k.mmr_order[i] = 1 + (i + c) % 3
                k.mmr_constant[i] = float(rng.normal() * 0.05)
                for o in range(3):
                    for j in range(7):
                        k.mmr_coeffs[i][o][j] = float(rng.normal() * 0.2)
    img = rng.random((16, 16, 4)).astype(np.float32)
    got = orc.dovi_reshape(img.copy(), comps)
    for y in range(16):
        for x in range(16):
            sig = np.clip(img[y, x, :3].astype(np.float64), 0, 1)
            for c in range(3):
                k = comps[c]
                s = sig[c]
                piv = [k.pivots[i] for i in range(1, 4)] + [1e9] * 4
                t = [s >= np.float32(p) for p in piv]
                # mix(mix(mix(c0, c1, t0), mix(c2, c3, t2), t1), mix(mix(c4, c5, t4), mix(c6, c7, t6), t5), t3)
                left = (3 if t[2] else 2) if t[1] else (1 if t[0] else 0)
                right = (7 if t[6] else 6) if t[5] else (5 if t[4] else 4)
                piece = right if t[3] else left
                if k.method[piece] == 0:
                    co = [np.float64(k.poly_coeffs[piece][j]) for j in range(3)]
                    v = (co[2] * s + co[1]) * s + co[0]
                else:
                    x3 = np.array([sig[0] * sig[1], sig[0] * sig[2], sig[1] * sig[2], sig[0] * sig[1] * sig[2]])
                    v = np.float64(k.mmr_constant[piece])
                    for o in range(k.mmr_order[piece]):
                        wts = np.array([k.mmr_coeffs[piece][o][j] for j in range(7)], dtype=np.float64)
                        v += wts[:3] @ sig ** (o + 1) + wts[3:] @ x3 ** (o + 1)
                v = min(max(v, 0.0), 1.0)
                assert abs(got[y, x, c] - v) < 2e-6, (x, y, c, float(got[y, x, c]), float(v))
    assert np.array_equal(got[..., 3], img[..., 3])


def test_distort_against_a_numpy_bilinear_resampling():
    """pl_shader_distort's bilinear path evaluated directly in float64: canvas position of every
    output pixel, the affine map, a bilinear fetch with CLAMP addressing, the one-texel alpha fade"""
    rng = np.random.default_rng(8)
    sh, sw, oh, ow = 12, 20, 30, 44
    img = rng.random((sh, sw, 4)).astype(np.float32)
    img[..., 3] = 1.0
    tf = [0.41, -0.13, 0.09, -0.37, 0.52, 0.47]        # canvas -> texture coordinates
    got = orc.distort(img, tf, ow, oh, alpha_mode=1)
    ys, xs = np.mgrid[0:oh, 0:ow]
    cx = -1 + 2 * (xs + 0.5) / ow
    cy = 1 - 2 * (ys + 0.5) / oh
    u = tf[0] * cx + tf[1] * cy + tf[4]
    v = tf[2] * cx + tf[3] * cy + tf[5]
    px, py = u * sw - 0.5, v * sh - 0.5
    x0, y0 = np.floor(px).astype(int), np.floor(py).astype(int)
    ax, ay = (px - x0)[..., None], (py - y0)[..., None]
    cl = lambda i, n: np.clip(i, 0, n - 1)
    f = img.astype(np.float64)
    t00, t10 = f[cl(y0, sh), cl(x0, sw)], f[cl(y0, sh), cl(x0 + 1, sw)]
    t01, t11 = f[cl(y0 + 1, sh), cl(x0, sw)], f[cl(y0 + 1, sh), cl(x0 + 1, sw)]
    want = (t00 * (1 - ax) + t10 * ax) * (1 - ay) + (t01 * (1 - ax) + t11 * ax) * ay

    def fade(p, pt):
        t = np.clip(np.minimum(p, 1 - p) / pt, 0, 1)
        return t * t * (3 - 2 * t)
    want[..., 3] *= fade(u, 1 / sw) * fade(v, 1 / sh)
    assert np.abs(got - want).max() < 2e-5, float(np.abs(got - want).max())
    assert (got[..., 3] == 0).any() and (got[..., 3] > 0.99).any()
