"""The PQ transfer pair as piecewise cubics (csrc/hip/pqseg.hiph, round 6): the tables the host
builds (plh_pqseg_build, backend.hip, through its test hook), evaluated here exactly as the device evaluates them (fp32
position, truncation, fract, three fp32 FMAs -- numpy emulates an FMA as a float64 product-sum
rounded once, which is what it is for fp32 operands), against the closed forms in float64
(SMPTE ST 2084 with the constants the reference prints into its shaders:
/root/reference/src/shaders/colorspace.c:643-668, 745-775). Host only -- no GPU.

Bounds held here (the closed-form device functions for comparison,
tests/test_gpu_color.py::test_pq_pair_error_bound_over_every_code: EOTF 5.4e-6 max relative, OETF
2.7e-7):
  EOTF, every 16-bit code and 2.5e6 random values below 1.25: relative error <= 3.5e-7 from v = 1/16
        (0.1 cd/m^2) up, <= 4.5e-6 on [1/64, 1/16), absolute error <= 3e-12 (3e-8 cd/m^2) below;
  OETF, 2e6 values log-uniform over [2^-64, 16): absolute error <= 2e-7 of the PQ range;
  every 16-bit code through the EOTF pieces and back through the OETF pieces: within 0.01 code.
"""
import ctypes as C

import numpy as np

import libplacebo_amd as pl

O_T0, O_N, E_LO, E_HI, SPLIT = -64, 68, 128, 158, 2
A = float(np.float32(0.164062))


def fmt(x):
    return float(np.float32(float("%f" % x)))


M1, M2, C3 = fmt(2610.0 / 4096 / 4), fmt(2523.0 / 4096 * 128), fmt(2392.0 / 4096 * 32)
CONSTS = np.array([M1, C3, M2, np.float32(1.0) / np.float32(M2), np.float32(1.0) / np.float32(M1)], np.float32)


def tables():
    out = np.zeros((O_N + E_LO + E_HI, 4), np.float32)
    fn = pl.lib().plh_test_pqseg_build
    fn.restype = None
    fn(CONSTS.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out[:O_N], out[O_N:]


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def cubic(tab, piece, pos):
    u = (pos - np.floor(pos)).astype(np.float32)
    c = tab[piece]
    return fma32(fma32(fma32(c[:, 3], u, c[:, 2]), u, c[:, 1]), u, c[:, 0]).astype(np.float64)


def eotf_seg(tab, v):
    p = (v.astype(np.float32) * np.float32(128.0)).astype(np.float32)
    fine = p < SPLIT
    pos = np.maximum(np.where(fine, p * np.float32(E_LO // SPLIT), p), np.float32(0.0)).astype(np.float32)
    return cubic(tab, pos.astype(np.int64) + np.where(fine, 0, E_LO - SPLIT), pos)


def oetf_seg(tab, x):
    t = np.log2(x.astype(np.float64)).astype(np.float32)       # (v_log_f32: one ulp)
    pos = np.clip(t, np.float32(O_T0), np.float32(O_T0 + O_N - 1.0 / 262144)).astype(np.float32)
    return cubic(tab, np.floor(pos).astype(np.int64) - O_T0, pos)


def eotf64(v):
    m2, m1, c3 = float(CONSTS[2]), float(CONSTS[0]), float(CONSTS[1])
    p = np.power(np.maximum(v, 0.0), 1.0 / m2)
    return np.power(np.maximum(p - (1 - A), 0.0) / ((c3 + A) - c3 * p), 1.0 / m1)


def oetf64(x):
    m2, m1, c3 = float(CONSTS[2]), float(CONSTS[0]), float(CONSTS[1])
    y = np.power(np.maximum(x, 0.0), m1)
    return np.power(((1 - A) + (c3 + A) * y) / (1 + c3 * y), m2)


def test_eotf_pieces_against_float64():
    _, te = tables()
    rng = np.random.default_rng(3)
    v = np.concatenate([np.arange(65536) / 65535.0, rng.random(1000000) * 1.2499, rng.random(1000000) / 16,
                        rng.random(500000) / 64]).astype(np.float32).astype(np.float64)
    got, want = eotf_seg(te, v), eotf64(v)
    hi, mid = v >= 1.0 / 16, (v >= 1.0 / 64) & (v < 1.0 / 16)
    assert (np.abs(got - want)[hi] / want[hi]).max() <= 3.5e-7
    assert (np.abs(got - want)[mid] / want[mid]).max() <= 4.5e-6
    assert np.abs(got - want)[v < 1.0 / 64].max() <= 3e-12
    # black is black, and the curve is monotone across every seam
    assert eotf_seg(te, np.zeros(1))[0] == 0.0 and eotf_seg(te, -np.ones(1))[0] == 0.0
    seams = np.concatenate([np.arange(1, E_LO) / 8192.0, np.arange(SPLIT, 160) / 128.0])
    below = eotf_seg(te, np.nextafter(seams.astype(np.float32), np.float32(0)).astype(np.float64))
    at = eotf_seg(te, seams)
    assert np.all(at >= below * (1 - 5e-6))


def test_oetf_pieces_against_float64():
    to, _ = tables()
    rng = np.random.default_rng(4)
    x = np.exp2(rng.random(2000000) * 67.999 - 64).astype(np.float32).astype(np.float64)
    # (the device's v_log_f32 is good to one ulp of log2 x; np.log2 in float64 rounded to fp32 is half)
    got, want = oetf_seg(to, x), oetf64(x)
    assert np.abs(got - want).max() <= 2e-7, np.abs(got - want).max()
    # zero and negative values: the first piece's left end, OETF(2^-64) for OETF(0) = c1^m2
    with np.errstate(divide="ignore", invalid="ignore"):
        z = oetf_seg(to, np.array([0.0]))[0]
    assert abs(z - oetf64(np.array([0.0]))[0]) <= 3e-7


def test_round_trip_of_every_code():
    """code -> EOTF pieces -> OETF pieces -> the code it came from, to a hundredth of a 16-bit code"""
    to, te = tables()
    v = np.arange(1, 65536) / 65535.0
    lin = eotf_seg(te, v.astype(np.float32).astype(np.float64))
    pos = lin > 0
    back = oetf_seg(to, lin[pos].astype(np.float32).astype(np.float64))
    assert np.abs(back - v[pos]).max() * 65535 <= 0.02


def test_tables_against_the_committed_ones():
    """The 354 pieces as committed (tests/golden/pqseg_tables.npy, written by this file's tables() on the
    build container): another libm may round a coefficient's last bit differently -- nothing more."""
    import os
    to, te = tables()
    got = np.concatenate([to, te])
    want = np.load(os.path.join(os.path.dirname(__file__), "golden", "pqseg_tables.npy"))
    assert got.shape == want.shape == (O_N + E_LO + E_HI, 4)
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-30)
    assert (np.abs(got - want) / scale).max() <= 4e-7
