"""mxr_axis (csrc/host/shader_sampling.c): the geometry check behind k_polar_mxr -- is an axis of a
polar pass an exact R : G upscale (3x, 4x, 3 : 2), with which shift, origin and per-phase texel
offsets, and how is the phase at fcoord = 0 of an odd ratio canonicalised. Host logic only (no
GPU): the per-output base texels and fcoords are computed here the way the kernels compute them
(fp32: pos * size - 0.5, floor, fract), the function is called through a test hook.
tests/test_gpu_polar_mfma.py renders these geometries and holds the kernel to k_polar_pp."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl


def axis(src, dst, offset=0.0):
    """per-output (fcoord, base) of a full-frame scale src -> dst texels, in fp32 like plh_attr +
    the sampler: pos = (i + 0.5) / dst normalised, then pos * src - 0.5"""
    i = np.arange(dst, dtype=np.float32)
    pos = (i + np.float32(0.5)) * np.float32(1.0 / dst) + np.float32(offset / src)
    t = pos * np.float32(src) - np.float32(0.5)
    fl = np.floor(t)
    return (t - fl).astype(np.float32), fl.astype(np.int32)


def mxr_axis(fc, base, R, G):
    fn = pl.lib().plh_test_mxr_axis
    fn.restype = C.c_int
    n = len(fc)
    out = (C.c_int * 10)()
    canon = (C.c_float * n)()
    ok = fn(fc.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p), n, R, G, out, canon)
    if not ok:
        return None
    return dict(shift=out[0], origin=out[1], off=list(out[2:2 + R]), rep=list(out[6:6 + R]),
                canon=np.array(canon[:], np.float32))


@pytest.mark.parametrize("src,R,G", [(1280, 3, 1), (720, 3, 1), (131, 3, 1), (960, 4, 1), (540, 4, 1),
                                     (2560, 3, 2), (1440, 3, 2), (130, 3, 2)])
def test_exact_ratios_are_recognised(src, R, G):
    dst = src * R // G
    fc, base = axis(src, dst)
    m = mxr_axis(fc, base, R, G)
    assert m is not None
    # every output's canonical base is origin + G * ((i + shift) / R) + off[phase]
    i = np.arange(dst)
    wrapped = fc > 0.98
    cb = base + wrapped
    q = (i + m["shift"]) % R
    want = m["origin"] + G * ((i + m["shift"]) // R) + np.array(m["off"])[q]
    assert np.array_equal(cb, want)
    assert all(0 <= o <= (0 if G == 1 else 2) for o in m["off"])
    # the canonical fcoord of an output is its phase's up to fp32 noise; wrapped ones are negative
    rep = np.array(m["rep"])
    assert np.all(~wrapped[rep])
    dev = m["canon"] - fc[rep][q]
    assert np.abs(dev).max() < 1e-5 + 1.5 * np.finfo(np.float32).eps * dst
    assert np.all(m["canon"][wrapped] <= 0) and np.all(m["canon"][wrapped] > -1e-3)


def test_odd_ratio_has_a_wrapping_phase_even_ratio_has_none():
    fc3, base3 = axis(720, 2160)
    assert (fc3 > 0.98).any()       # the phase at fcoord = 0: rounding puts some outputs just below
    m = mxr_axis(fc3, base3, 3, 1)
    assert m is not None and min(np.abs(fc3[m["rep"]])) < 1e-4
    # ... and whichever way an implementation's rounding falls, the structure is found: the same
    # axis with the phase-0 outputs pushed below / above the texel centre at random
    rng = np.random.default_rng(2)
    i = np.arange(2160)
    t = (i + 0.5) / 3.0 - 0.5 + np.where(i % 3 == 1, rng.uniform(-4e-5, 4e-5, 2160), 0.0)
    fl = np.floor(t)
    fcn, basen = (t - fl).astype(np.float32), fl.astype(np.int32)
    assert (fcn > 0.98).sum() > 100 and ((fcn < 0.02) & (i % 3 == 1)).sum() > 100
    mn = mxr_axis(fcn, basen, 3, 1)
    assert mn is not None and mn["shift"] == m["shift"] and mn["origin"] == m["origin"]
    fc4, base4 = axis(960, 3840)
    assert not (fc4 > 0.98).any()
    assert mxr_axis(fc4, base4, 4, 1) is not None


@pytest.mark.parametrize("src,dst", [(1920, 3840), (1280, 2240), (1000, 2500), (96, 168), (1920, 2560)])
def test_other_ratios_are_refused(src, dst):
    fc, base = axis(src, dst)
    for R, G in ((3, 1), (4, 1), (3, 2)):
        assert mxr_axis(fc, base, R, G) is None, (R, G)


def test_a_constant_fractional_offset_keeps_the_geometry():
    """a cropped source whose rect starts off the texel grid: the phases are other constants, the
    structure is the same"""
    fc, base = axis(1280, 3840, offset=0.3)
    m = mxr_axis(fc, base, 3, 1)
    assert m is not None
    assert len(set(np.round(fc[m["rep"]], 3))) == 3


def test_short_axes_are_refused():
    fc, base = axis(2, 6)
    assert mxr_axis(fc, base, 3, 1) is None
