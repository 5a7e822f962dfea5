"""float64 evaluation of the colour-mapping formulas (reference
src/shaders/colorspace.c:1791-1995 + PQ linearize :645-651 + BT.1886 delinearize
:749-755), used to derive a *conditioning-aware* tolerance for fp32 implementations.

Why: the IPT/PQ round trip is ill-conditioned in fp32. PQ's outer exponent (m2 = 78.84)
and the (c2 - c3 v) cancellation of its inverse amplify one ulp of an intermediate by
~10^2..10^3, lms2rgb has cancelling coefficients of magnitude ~5, and the inverse display
gamma has unbounded slope towards 0. Any two correct fp32 evaluations (float libm vs native
v_exp_f32/v_log_f32, or two Vulkan drivers) therefore differ by far more than one 16-bit
code on saturated / bright samples, while agreeing to ~0.01 LSB on most. The tests bound the
GPU-vs-oracle difference by first-order propagation of a stated relative accuracy EPS of the
PQ-decoded LMS values through the last matrix and the output transfer function.

Test infrastructure only.
"""
import numpy as np

M1, M2 = 0.159302, 78.843750          # "%f"-printed constants, as the shader embeds them
C1, C2, C3 = 0.835938, 18.851562, 18.687500
K203 = 0.020300
K10 = 49.261084

LMS2IPT = np.array([[0.4000, 0.4000, 0.2000], [4.4550, -4.8510, 0.3960],
                    [0.8056, 0.3572, -1.1628]])
IPT2LMS = np.array([[1.0, 0.0975689, 0.205226], [1.0, -0.1138760, 0.133217],
                    [1.0, 0.0326151, -0.676887]])


def pq_oetf(x):
    y = np.maximum(x, 0.0) ** M1
    return ((C1 + C2 * y) / (1.0 + C3 * y)) ** M2


def pq_eotf(v):
    p = np.maximum(v, 0.0) ** (1.0 / M2)
    return (np.maximum(p - C1, 0.0) / (C2 - C3 * p)) ** (1.0 / M1)


def bt1886_inverse(L, csp_min, csp_max):
    lb, lw = csp_min ** (1 / 2.4), csp_max ** (1 / 2.4)
    a, b = (lw - lb) ** 2.4, lb / (lw - lb)
    return np.maximum(L, 0.0) ** (1 / 2.4) * (1.0 / a) ** (1 / 2.4) - b


def lerp_lut1d(lut, x):
    n = len(lut)
    pos = np.clip(x, 0.0, 1.0) * (n - 1)
    i0 = np.floor(pos).astype(int)
    i1 = np.minimum(i0 + 1, n - 1)
    f = pos - i0
    lut = lut.astype(np.float64)
    return lut[i0] * (1 - f) + lut[i1] * f


def cubic_lut3d(lut_u16, size, idx):
    """shaders/lut.c:721-757 in float64: B-spline weights, eight linear fetches"""
    g0, h = [], []
    for k, n in enumerate(size):
        pos = idx[k] * (n - 1)
        fpos = pos - np.floor(pos)
        base = pos - fpos
        inv = 1.0 - fpos
        w0, w3 = inv ** 3 / 6.0, fpos ** 3 / 6.0
        w1 = 2.0 / 3.0 - 0.5 * fpos ** 2 * (2.0 - fpos)
        w2 = 2.0 / 3.0 - 0.5 * inv ** 2 * (2.0 - inv)
        g0.append(w0 + w1)
        h.append(((w1 / (w0 + w1) - 1.0 + base) / (n - 1), (w3 / (w2 + w3) + 1.0 + base) / (n - 1)))
    out = 0.0
    for t in range(8):
        bits = [(t >> k) & 1 for k in range(3)]
        w = 1.0
        for k in range(3):
            w = w * ((1.0 - g0[k]) if bits[k] else g0[k])
        out = out + w[..., None] * lerp_lut3d(lut_u16, size, [h[k][bits[k]] for k in range(3)])
    return out


def lerp_lut3d(lut_u16, size, idx):
    sx, sy, sz = size
    lut = lut_u16.reshape(sz, sy, sx, 4).astype(np.float64) / 65535.0
    pos = [np.clip(idx[k], 0.0, 1.0) * (s - 1) for k, s in enumerate(size)]
    i0 = [np.floor(p).astype(int) for p in pos]
    i1 = [np.minimum(i + 1, s - 1) for i, s in zip(i0, size)]
    f = [p - i for p, i in zip(pos, i0)]
    out = 0.0
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                w = ((f[0] if dx else 1 - f[0]) * (f[1] if dy else 1 - f[1]) *
                     (f[2] if dz else 1 - f[2]))
                t = lut[(i1[2] if dz else i0[2]), (i1[1] if dy else i0[1]),
                        (i1[0] if dx else i0[0])]
                out = out + w[..., None] * t[..., :3]
    return out


def hdr10_to_sdr(img, r, eps, prelinearized=False, lowres=None, strength=0.0, cr_out=(0.0, 1.0)):
    """PQ/HDR10 -> BT.1886 colour map in float64.

    `r` = colormap_ref.resolve(...). Returns (out, allow): the float64 result and the per-sample
    first-order bound on |fp32 result - out| given that the PQ-decoded LMS values (the inputs of
    the last matrix) carry a relative error <= eps. prelinearized: `img` already is linear light
    (1.0 = 203 cd/m^2). lowres / strength / cr_out: contrast recovery (colorspace.c:1880-1921),
    `lowres` = the low-pass luma per pixel."""
    kw = r["kw"]
    if prelinearized:
        rgb = img[..., :3].astype(np.float64)
    else:
        rgb = pq_eotf(img[..., :3].astype(np.float64)) * K10          # linearize (PQ)
    lms = rgb @ np.array(kw["rgb2lms"], np.float64).reshape(3, 3).T
    ipt = pq_oetf(K203 * lms) @ LMS2IPT.T
    I, P, T = ipt[..., 0], ipt[..., 1], ipt[..., 2]
    i_orig = I
    mode = kw.get("tone_mode", -1)
    if mode >= 0:
        tp = kw["tone_p"]
        if mode == 0:
            I = np.clip(I, tp[0], tp[1])
        elif mode == 2:
            curve = lambda v: lerp_lut1d(kw["tone_lut"], tp[0] * v + tp[1])  # noqa: E731
            if lowres is not None:
                hi = np.clip(I, 0.0, 1.0)
                lo = np.clip(np.asarray(lowres, np.float64).reshape(I.shape), 0.0, 1.0)
                base, sharp = curve(hi), curve(lo) + (hi - lo)
                I = np.clip(base * (1 - strength) + sharp * strength, cr_out[0], cr_out[1])
            else:
                I = curve(I)
        hull = lambda v: ((v - 6.0) * v + 9.0) * v   # noqa: E731
        k = np.minimum(i_orig / I, hull(I) / hull(i_orig))
        P, T = P * k, T * k
    if kw.get("gamut_lut") is not None:
        idx = [kw["gamut_scale"] * I + kw["gamut_offset"], 2.0 * np.hypot(P, T),
               0.159155 * np.arctan2(T, P) + 0.5]
        if kw.get("gamut_tricubic"):
            o = cubic_lut3d(kw["gamut_lut"], kw["gamut_size"], idx)
        else:
            o = lerp_lut3d(kw["gamut_lut"], kw["gamut_size"], idx)
        I, P, T = o[..., 0], o[..., 1] - 32768.0 / 65535.0, o[..., 2] - 32768.0 / 65535.0
    lmspq = np.stack([I, P, T], -1) @ IPT2LMS.T
    lms_out = pq_eotf(lmspq) * K10
    M = np.array(kw["lms2rgb"], np.float64).reshape(3, 3)
    lin = lms_out @ M.T
    _, dmin, dmax, _ = r["delin"]
    # delinearize: clamp, BT.1886 inverse (BT.1886 is not black-scaled: it carries its own lift)
    g = lambda L: bt1886_inverse(L, dmin, dmax)  # noqa: E731
    out = g(lin)
    dlin = eps * (lms_out @ np.abs(M).T)
    allow = np.maximum(np.abs(g(lin + dlin) - out), np.abs(g(lin - dlin) - out))
    res = img.astype(np.float64).copy()
    res[..., :3] = out
    al = np.zeros_like(res)
    al[..., :3] = allow
    return res, al
