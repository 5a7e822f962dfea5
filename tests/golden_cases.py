"""The Tier-0 golden cases: one evaluation routine run against (a) the real reference
(tests/golden/make_golden.py -> tests/golden/tier0.npz), (b) the product's host code and
(c) where applicable the oracle's restatement (tests/test_golden.py)."""
import ctypes as C

import numpy as np

import util
from libplacebo_amd import _capi as capi
from ref_structs import *  # noqa: F401,F403

FILTERS = ["ewa_lanczos", "ewa_lanczossharp", "ewa_hann", "lanczos", "mitchell", "bilinear",
           "hermite", "gaussian", "spline36", "catmull_rom", "bicubic"]
BLURS = [0.0, 2.0, 1.7]


class RefFilterParams(C.Structure):  # the reference keeps a deprecated trailing field
    _fields_ = capi.FilterParams._fields_ + [("filter_scale", C.c_float)]


class RefFilter(C.Structure):
    _fields_ = [("params", RefFilterParams), ("radius", C.c_float), ("radius_zero", C.c_float),
                ("weights", C.POINTER(C.c_float)), ("row_size", C.c_int),
                ("insufficient", C.c_bool), ("row_stride", C.c_int)]


class Lib:
    """A library exposing the reference's Tier-0 API."""
    params_t, filter_t = capi.FilterParams, capi.Filter

    def __init__(self, lib):
        self.lib = declare(lib)
        self.lib.pl_filter_generate.restype = C.POINTER(self.filter_t)
        self.lib.pl_find_filter_config.restype = C.POINTER(capi.FilterConfig)
        self.lib.pl_find_filter_config.argtypes = [C.c_char_p, C.c_int]


class RefLib(Lib):
    params_t, filter_t = RefFilterParams, RefFilter


def evaluate(L):
    lib, out = L.lib, {}

    # ---- filters ----
    for name in FILTERS:
        cfg = lib.pl_find_filter_config(name.encode(), 1)  # PL_FILTER_UPSCALING
        assert cfg, name
        for blur in BLURS:
            p = L.params_t()
            C.memmove(C.byref(p.config), cfg, C.sizeof(capi.FilterConfig))
            if blur:
                p.config.blur = (cfg.contents.blur or 1.0) * blur
            p.lut_entries, p.cutoff, p.row_stride_align = 256, 1e-3 if cfg.contents.polar else 0.0, 4
            f = lib.pl_filter_generate(None, C.byref(p)).contents
            cnt = 256 if cfg.contents.polar else 256 * f.row_stride
            key = f"filter/{name}/{blur}"
            out[key + "/meta"] = np.array([f.radius, f.radius_zero, f.row_size, f.row_stride],
                                          np.float32)
            out[key + "/weights"] = np.ctypeslib.as_array(f.weights, (cnt,)).copy()

    # ---- dither matrices ----
    for size in (4, 16):
        m = np.zeros(size * size, np.float32)
        lib.pl_generate_bayer_matrix(m.ctypes.data_as(C.c_void_p), size)
        out[f"dither/bayer/{size}"] = m
    for size in (16, 64):
        m = np.zeros(size * size, np.float32)
        util.srand(1)
        lib.pl_generate_blue_noise(m.ctypes.data_as(C.c_void_p), size)
        out[f"dither/blue/{size}"] = m

    # ---- tone mapping ----
    for name in TONE_NAMES:
        for imin, imax, iavg, omin, omax in ((0.005, 1000, 0, 0.203, 203), (0.005, 4000, 90, 0.05, 600)):
            r = (lambda x: lib.pl_hdr_rescale(HDR_NITS, HDR_PQ, x))
            p = TMP(function=lib.pl_find_tone_map_function(name), constants=TMC(*TMC_DEFAULT),
                    input_scaling=HDR_PQ, output_scaling=HDR_PQ, lut_size=128,
                    input_min=r(imin), input_max=r(imax), input_avg=r(iavg),
                    output_min=r(omin), output_max=r(omax))
            o = np.zeros(128, np.float32)
            lib.pl_tone_map_generate(o.ctypes.data_as(C.c_void_p), C.byref(p))
            out[f"tone/{name.decode()}/{imax}"] = o

    # ---- gamut mapping (small lattice) ----
    for name in GAMUT_NAMES:
        p = GMP(function=lib.pl_find_gamut_map_function(name),
                input_gamut=lib.pl_raw_primaries_get(6).contents,
                output_gamut=lib.pl_raw_primaries_get(3).contents,
                min_luma=lib.pl_hdr_rescale(HDR_NITS, HDR_PQ, 0.005),
                max_luma=lib.pl_hdr_rescale(HDR_NITS, HDR_PQ, 1000.0),
                constants=GMC(*GMC_DEFAULT), lut_size_I=12, lut_size_C=8, lut_size_h=32,
                lut_stride=3)
        o = np.zeros(12 * 8 * 32 * 3, np.float32)
        lib.pl_gamut_map_generate(o.ctypes.data_as(C.c_void_p), C.byref(p))
        out[f"gamut/{name.decode()}"] = o

    # ---- matrices ----
    mats = []
    for p in range(1, 18):
        pr = lib.pl_raw_primaries_get(p)
        for fn in ("pl_get_rgb2xyz_matrix", "pl_get_xyz2rgb_matrix", "pl_ipt_rgb2lms", "pl_ipt_lms2rgb"):
            mats += m3(getattr(lib, fn)(pr))
    out["matrices/primaries"] = np.array(mats, np.float32)
    mats = []
    for sysid in range(14):
        if sysid == 8:
            continue
        for levels in (1, 2):
            for bits in ((8, 8, 0), (16, 10, 6), (10, 10, 0)):
                rr = Repr(sys=sysid, levels=levels, bits=Bits(*bits))
                tr = lib.pl_color_repr_decode(C.byref(rr), None)
                mats += m3(tr.mat) + list(tr.c)
    out["matrices/repr_decode"] = np.array(mats, np.float32)

    # ---- CPU transfer functions ----
    vals = []
    xs = np.linspace(-0.05, 1.1, 47, dtype=np.float32)
    for trc in range(18):
        cs = Csp(primaries=6 if trc in (12, 13) else 3, transfer=trc)
        cs.hdr.min_luma, cs.hdr.max_luma = 0.005, 1000.0 if trc in (12, 13) else 203.0
        for fn in ("pl_color_linearize", "pl_color_delinearize"):
            for x in xs:
                a = (C.c_float * 3)(x, x * 0.5, 0.25)
                getattr(lib, fn)(C.byref(cs), a)
                vals += list(a)
    out["trc/cpu"] = np.array(vals, np.float32)
    return out
