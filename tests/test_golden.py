"""Tier-0 parity without the reference: the product's host-side maths (filters, dither
matrices, tone / gamut mapping, colour matrices, CPU transfer functions) and the oracle's
filter restatement against golden vectors generated from the REAL reference
(tests/golden/make_golden.py). Bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

import golden_cases as gc
import orc
from libplacebo_amd import _capi as capi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tier0.npz")


@pytest.fixture(scope="module")
def golden():
    assert os.path.exists(GOLDEN), "tests/golden/tier0.npz is missing (make_golden.py)"
    return dict(np.load(GOLDEN))


@pytest.fixture(scope="module")
def product(built):
    return gc.evaluate(gc.Lib(C.CDLL(capi.LIB_PATH)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_every_case_is_present(golden, product):
    assert set(golden) == set(product)
    assert len(golden) > 80


@pytest.mark.parametrize("group", ["filter", "dither", "tone", "gamut", "matrices", "trc"])
def test_product_host_maths_bit_exact(golden, product, group):
    keys = [k for k in golden if k.startswith(group + "/")]
    assert keys
    for k in keys:
        assert np.array_equal(bits(product[k]), bits(golden[k])), k


def test_oracle_filter_restatement_bit_exact(golden):
    """oracle/pl_oracle.c restates pl_filter_generate (filters.c:186-252); the GPU parity tests
    rely on it for the weights of the scalers."""
    cases = {"ewa_lanczos": (orc.ewa_lanczos, True), "lanczos": (orc.lanczos, False),
             "mitchell": (orc.mitchell, False), "bilinear": (orc.triangle, False)}
    for name, (mk, polar) in cases.items():
        for blur in gc.BLURS:
            meta = golden[f"filter/{name}/{blur}/meta"]
            ref = golden[f"filter/{name}/{blur}/weights"]
            if polar:
                w, r, rz = orc.filter_generate_polar(mk(blur=blur), cutoff=1e-3)
                assert bits(np.float32(r)) == bits(meta[0]) and bits(np.float32(rz)) == bits(meta[1])
                assert np.array_equal(bits(w), bits(ref)), (name, blur)
            else:
                rows, n, r, rz = orc.filter_generate_ortho(mk(blur=blur))
                assert n == int(meta[2]) and rows.shape[1] == int(meta[3])
                assert np.array_equal(bits(rows.ravel()), bits(ref)), (name, blur)


def test_reference_known_answers(golden):
    """Known answers the reference's own tests assert (src/tests/filters.c:16-75,
    src/tests/tone_mapping.c:46-86), evaluated on the golden vectors."""
    # separable rows are normalised to sum 1 (filters.c test: fabs(sum - 1) < 1e-6)
    for name in ("lanczos", "mitchell", "spline36"):
        meta = golden[f"filter/{name}/0.0/meta"]
        rows = golden[f"filter/{name}/0.0/weights"].reshape(256, int(meta[3]))
        assert np.abs(rows[:, :int(meta[2])].sum(axis=1) - 1.0).max() < 1e-6
    # polar LUTs start at the kernel's peak (1.0) and end inside the cutoff
    for name in ("ewa_lanczos", "ewa_lanczossharp"):
        w = golden[f"filter/{name}/0.0/weights"]
        assert abs(w[0] - 1.0) < 1e-6 and abs(w[-1]) < 1e-3
    # tone curves are monotonic and land inside the output range (tone_mapping.c:60-86)
    lo, hi = 0.0, 1.0
    for k, lut in golden.items():
        if k.startswith("tone/") and not k.startswith("tone/clip"):
            assert np.all(np.diff(lut) >= -1e-6), k
            assert lut.min() >= lo - 1e-6 and lut.max() <= hi + 1e-6, k
    # blue noise / bayer matrices are permutations of k / size^2
    for k in ("dither/blue/64", "dither/bayer/16"):
        m = np.sort(golden[k])
        assert np.allclose(m, np.arange(m.size) / m.size, atol=1e-6), k
