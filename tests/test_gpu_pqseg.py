"""The PQ pieces as the DEVICE evaluates them (csrc/hip/pqseg.hiph: pq_eotf_seg / pq_oetf_seg, the
functions the chain epilogues of k_polar_mx / k_polar_mxr call, off a copy of the tables staged in
LDS) against the host's emulation of the same lookup (tests/test_pqseg.py), which is what is pinned
against float64 there. The EOTF lookup has no transcendental in it -- position, floor, fract, three
FMAs -- so the device must return the emulation's value BIT FOR BIT for every 16-bit code and for
negative inputs; values at and beyond the tables' end (v >= 1.25) take the closed form. The OETF
lookup goes through v_log_f32 (one ulp), so it is held to the emulation within 2.5e-7 of the PQ range
and to float64 within the bound of test_pqseg.py."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import test_pqseg as host

pytestmark = pytest.mark.gpu


def device_eval(gpu, values, which):
    v = np.ascontiguousarray(values, np.float32)
    out = np.empty_like(v)
    fn = pl.lib().plh_test_pqseg_eval
    fn.restype = C.c_int
    rc = fn(host.CONSTS.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
            C.c_int(v.size), C.c_int(which))
    assert rc == 0, rc
    return out


def test_eotf_pieces_on_the_device_are_the_emulation_bit_for_bit(gpu):
    _, te = host.tables()
    rng = np.random.default_rng(11)
    v = np.concatenate([np.arange(65536) / 65535.0, rng.random(200000) * 1.2499, -rng.random(1000),
                        rng.random(100000) / 64]).astype(np.float32)
    got = device_eval(gpu, v, 0)
    want = host.eotf_seg(te, v.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, want), (np.abs(got.astype(np.float64) - want).max(), int((got != want).sum()))
    # beyond the tables: the closed form (the reference's formula, no clamp), to fp32 accuracy
    far = np.array([1.25, 1.3, 1.5, 1.75], np.float32)
    got = device_eval(gpu, np.concatenate([far, np.full(252, 0.5, np.float32)]), 0)[:4].astype(np.float64)
    want = host.eotf64(far.astype(np.float64))
    assert np.all(np.isfinite(got)) and (np.abs(got - want) / want).max() < 2e-5, (got, want)
    assert np.all(got > 10.0)       # (above 100 000 cd/m^2: nothing was held at 10 000)


def test_oetf_pieces_on_the_device(gpu):
    to, _ = host.tables()
    rng = np.random.default_rng(12)
    x = np.concatenate([np.exp2(rng.random(300000) * 67.99 - 64), [0.0, 1.0, 15.9], rng.random(50000)]).astype(np.float32)
    got = device_eval(gpu, x, 1).astype(np.float64)
    with np.errstate(divide="ignore"):
        emu = host.oetf_seg(to, x.astype(np.float64))
    # (one ulp of v_log_f32 is at most two fp32 ulps of a result near 1: 2.4e-7)
    assert np.abs(got - emu).max() <= 2.5e-7, np.abs(got - emu).max()
    assert np.abs(got - host.oetf64(x.astype(np.float64))).max() <= 3e-7
    # beyond the tables (x >= 16): the closed form
    far = np.concatenate([np.array([16.0, 40.0], np.float32), np.full(254, 0.01, np.float32)])
    g2 = device_eval(gpu, far, 1)[:2].astype(np.float64)
    assert np.abs(g2 - host.oetf64(far[:2].astype(np.float64))).max() <= 3e-6
