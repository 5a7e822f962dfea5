"""CPU-only checks: arithmetic identities the kernels rely on, and the oracle pinned against
the reference's own known answers / the real reference build where it is present."""
import ctypes as C

import numpy as np
import pytest

import orc
import util


def test_fast_unorm_decode_is_the_correctly_rounded_quotient():
    # devmath.hiph plh_un8 / plh_un16: mul + 2 fma == IEEE division, for every code value
    assert orc.lib().orc_check_unorm_decode(8) == 0
    assert orc.lib().orc_check_unorm_decode(16) == 0


@pytest.mark.parametrize("size", [2, 8, 16, 64, 256])
def test_dither_index_shortcut_is_exact(size):
    # colorops.hiph dither_bias: (int)(fract((n + .5) / size) * size) == n & (size - 1)
    assert orc.lib().orc_check_dither_index(size, 1 << 15) == 0


def test_cfg1_ewa_lanczos_cpu_reference_case():
    """BASELINE configs[0]: pl_filter_sample EWA-Lanczos weights on a synthetic frame, pure CPU.
    Direct evaluation (every tap = pl_filter_sample) vs the 256-entry LUT + lerp the GPU path
    uses: the LUT resampler must agree with the direct one to the LUT's interpolation error."""
    rng = np.random.default_rng(0)
    src = rng.random((48, 48)).astype(np.float32)
    direct, taps_d = orc.ewa_resample_r32f(orc.ewa_lanczos(), src, 96, 96, use_lut=False)
    lut, taps_l = orc.ewa_resample_r32f(orc.ewa_lanczos(), src, 96, 96, use_lut=True)
    assert taps_d == taps_l and 25 < taps_d < 45    # taps per pixel inside the radius (~30)
    assert np.abs(direct - lut).max() < 2e-4
    # a constant image is reproduced exactly by a normalised filter
    flat, _ = orc.ewa_resample_r32f(orc.ewa_lanczos(), np.full((32, 32), 0.25, np.float32),
                                    64, 64, use_lut=True)
    assert np.abs(flat - 0.25).max() < 1e-6


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
def test_oracle_filter_sample_matches_reference_pl_filter_sample():
    import ref_structs as R
    from libplacebo_amd import _capi as capi
    ref = R.declare(orc.ref())
    ref.pl_filter_sample.restype = C.c_double
    ref.pl_filter_sample.argtypes = [C.POINTER(capi.FilterConfig), C.c_double]
    ref.pl_find_filter_config.restype = C.POINTER(capi.FilterConfig)
    ref.pl_find_filter_config.argtypes = [C.c_char_p, C.c_int]
    orc.lib().orc_filter_sample.restype = C.c_double
    for name, mk in (("ewa_lanczos", orc.ewa_lanczos), ("lanczos", orc.lanczos),
                     ("mitchell", orc.mitchell), ("bilinear", orc.triangle)):
        cfg = ref.pl_find_filter_config(name.encode(), 1)
        f = mk()
        for x in np.linspace(0, 3.5, 141):
            a = ref.pl_filter_sample(cfg, float(x))
            b = orc.lib().orc_filter_sample(C.byref(f), C.c_double(float(x)))
            assert a == b, (name, x, a, b)


def test_oracle_samplers_basic_invariants():
    img = orc.tex_decode(util.random_rgba16(24, 16, seed=3), "rgba16")
    # identity fetches return the texels
    assert np.array_equal(orc.sample_simple(img, orc.S_NEAREST, 24, 16), img)
    assert np.array_equal(orc.sample_simple(img, orc.S_BILINEAR, 24, 16), img)
    # a constant image survives every sampler
    flat = np.full((16, 24, 4), 0.375, np.float32)
    for kind in (orc.S_BILINEAR, orc.S_BICUBIC, orc.S_HERMITE, orc.S_GAUSSIAN):
        out = orc.sample_simple(flat, kind, 37, 29)
        assert np.abs(out - 0.375).max() < 1e-6
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
    assert np.abs(orc.sample_polar(flat, w, r, rz, 48, 32) - 0.375).max() < 1e-6
    rows, n, _, _ = orc.filter_generate_ortho(orc.lanczos())
    assert np.abs(orc.sample_ortho(flat, rows, n, 0, 48, 16) - 0.375).max() < 1e-6


def test_oracle_error_diffusion_preserves_the_mean():
    img = orc.tex_decode(util.chirp_rgba16(64, 48), "rgba16")
    out = orc.error_diffusion(img, 3, 2, 4, [[0, 0, 0, 2, 0], [0, 1, 1, 0, 0], [0, 0, 0, 0, 0]])
    q = out[..., :3] * 7
    assert np.abs(q - np.round(q)).max() < 1e-5
    assert abs(out[..., :3].mean() - img[..., :3].mean()) < 2e-3


def test_oracle_peak_detection_matches_the_reference_formula():
    """src/tests/gpu_tests.c:753-785: avg / max PQ luminance of the (x, y, 0) ramp through
    gamma 2.2, against the closed form."""
    W = H = 16
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.zeros((H, W, 4), np.float32)
    img[..., 0], img[..., 1], img[..., 3] = (x + 0.5) / W, (y + 0.5) / H, 1.0
    luma_c = (0.212639, 0.715169, 0.072192)
    buf = orc.detect_peak(img, 6, 0.0, 1.0, luma_c, black_cutoff=0.0)   # TRC gamma22
    wg_count, wg_active, sum_pq, max_pq = buf[0:12], buf[12:24], buf[24:36], buf[36:48]
    assert wg_count.sum() == 1 and wg_active.sum() == 1
    lin = luma_c[0] * img[..., 0] ** 2.2 + luma_c[1] * img[..., 1] ** 2.2
    m1, m2, c1, c2, c3 = 2610 / 16384, 2523 / 32, 3424 / 4096, 2413 / 128, 2392 / 128
    yv = (lin * 203 / 10000) ** m1
    pq = ((c1 + c2 * yv) / (1 + c3 * yv)) ** m2
    assert abs(max_pq.max() / 16383.0 - pq.max()) <= 1e-4
    assert abs(sum_pq.sum() / 16383.0 - pq.mean()) <= 1e-3


def test_multi_gpu_sharding_plan():
    """Streams are independent units: stream s runs on rank s % world."""
    for world in (1, 2, 4, 8):
        owners = [s % world for s in range(16)]
        for r in range(world):
            assert owners.count(r) == 16 // world


def test_icc_entry_points_of_a_build_without_lcms():
    """pl_icc_open / _update / _close as the reference compiled without LittleCMS has them
    (src/shaders/icc.c:802-836): open fails, update clears the object and fails, close is a no-op;
    the default parameters are the header's PL_ICC_DEFAULTS."""
    import ctypes as C
    import libplacebo_amd as pl
    L = pl.lib()

    class IccParams(C.Structure):
        _fields_ = [("intent", C.c_int), ("size_r", C.c_int), ("size_g", C.c_int), ("size_b", C.c_int),
                    ("max_luma", C.c_float), ("force_bpc", C.c_bool), ("cache", C.c_void_p),
                    ("cache_priv", C.c_void_p), ("cache_save", C.c_void_p), ("cache_load", C.c_void_p)]

    d = IccParams.in_dll(L, "pl_icc_default_params")
    assert (d.intent, d.size_r, d.max_luma, d.force_bpc) == (1, 0, 203.0, False)   # RELATIVE_COLORIMETRIC
    L.pl_icc_open.restype = C.c_void_p
    L.pl_icc_open.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.pl_icc_open(None, None, None) is None
    L.pl_icc_update.restype = C.c_bool
    L.pl_icc_update.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p]
    obj = C.c_void_p(0x1234)
    assert L.pl_icc_update(None, C.byref(obj), None, None) is False and not obj.value
    L.pl_icc_close.argtypes = [C.POINTER(C.c_void_p)]
    L.pl_icc_close(C.byref(obj))
    assert not obj.value


def test_shader_variable_helpers_match_the_reference():
    """pl_var_<type>() / pl_var_glsl_types / pl_var_glsl_type_name and the host / std140 / std430
    layouts (src/gpu.c:745-948) against the reference's own gpu.c (oracle/_ref/libplref_gpu.so):
    every GLSL type, array lengths 1 and 3, every starting offset 0..19; memcpy_layout moves a
    mat3 from its host layout into its std140 one column by column."""
    import ctypes as C
    import os
    import numpy as np
    import libplacebo_amd as pl
    path = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "libplref_gpu.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libplref_gpu.so not built")
    ref, our = C.CDLL(path), pl.lib()

    class Var(C.Structure):
        _fields_ = [("name", C.c_char_p), ("type", C.c_int), ("dim_v", C.c_int), ("dim_m", C.c_int),
                    ("dim_a", C.c_int)]

    class Named(C.Structure):
        _fields_ = [("glsl_name", C.c_char_p), ("var", Var)]

    class Layout(C.Structure):
        _fields_ = [("offset", C.c_size_t), ("stride", C.c_size_t), ("size", C.c_size_t)]

    def tup(v):
        return (v.type, v.dim_v, v.dim_m, v.dim_a)

    names = []
    for lib in (ref, our):
        lib.pl_var_glsl_type_name.restype = C.c_char_p
        lib.pl_var_glsl_type_name.argtypes = [Var]
        for f in ("pl_var_host_layout", "pl_std140_layout", "pl_std430_layout"):
            getattr(lib, f).restype = Layout
            getattr(lib, f).argtypes = [C.c_size_t, C.POINTER(Var)]
        tab = (Named * 21).in_dll(lib, "pl_var_glsl_types")
        assert not tab[20].glsl_name
        names.append([(tab[i].glsl_name, tup(tab[i].var)) for i in range(20)])
    assert names[0] == names[1] and len(set(n for n, _ in names[0])) == 20
    for glsl, (ty, v, m, a) in names[0]:
        ctor = "pl_var_" + glsl.decode()
        for lib in (ref, our):
            getattr(lib, ctor).restype = Var
            getattr(lib, ctor).argtypes = [C.c_char_p]
        rv, ov = getattr(ref, ctor)(b"x"), getattr(our, ctor)(b"x")
        assert tup(rv) == tup(ov) == (ty, v, m, 1) and ov.name == b"x"
        assert ref.pl_var_glsl_type_name(ov) == our.pl_var_glsl_type_name(ov) == glsl
        for dim_a in (1, 3):
            ov.dim_a = dim_a
            for off in range(20):
                for f in ("pl_var_host_layout", "pl_std140_layout", "pl_std430_layout"):
                    r, o = getattr(ref, f)(off, C.byref(ov)), getattr(our, f)(off, C.byref(ov))
                    assert (r.offset, r.stride, r.size) == (o.offset, o.stride, o.size), (glsl, dim_a, off, f)
    odd = Var(b"m", 3, 2, 3, 1)      # mat3x2: a name, no constructor
    assert ref.pl_var_glsl_type_name(odd) == our.pl_var_glsl_type_name(odd) == b"mat3x2"
    assert our.pl_var_glsl_type_name(Var(b"n", 1, 2, 2, 1)) is None     # no integer matrices
    # a few layouts spelled out (GLSL 4.60 section 7.6.2.2)
    mat3 = our.pl_var_mat3(b"m")
    h, s140 = our.pl_var_host_layout(0, C.byref(mat3)), our.pl_std140_layout(4, C.byref(mat3))
    assert (h.stride, h.size) == (12, 36) and (s140.offset, s140.stride, s140.size) == (16, 16, 48)
    vec3 = our.pl_var_vec3(b"v")
    s = our.pl_std430_layout(4, C.byref(vec3))
    assert (s.offset, s.stride, s.size) == (16, 12, 12)
    our.memcpy_layout.argtypes = [C.c_void_p, Layout, C.c_void_p, Layout]
    src = np.arange(9, dtype=np.float32)
    dst = np.full(16 + 48 // 4, -1.0, np.float32)
    our.memcpy_layout(dst.ctypes.data, s140, src.ctypes.data, h)
    got = dst[4:16].reshape(3, 4)
    assert np.array_equal(got[:, :3], src.reshape(3, 3)) and np.all(got[:, 3] == -1) and np.all(dst[:4] == -1)
    our.pl_desc_access_glsl_name.restype = C.c_char_p
    assert [our.pl_desc_access_glsl_name(i) for i in range(3)] == [b"", b"readonly", b"writeonly"]


def test_unorm16_to_f16_by_product():
    """k_polar_mx's tile decode (mx_un16_for_f16): a 16-bit unorm code enters the rgba16hf tile as
    f16(fp32(v) * fp32(1 / 65535)) -- conversion and one product -- where the reference's value is
    f16(fp32(v / 65535)) (texture decode, then PASS A's rgba16hf store: renderer.c:2064). Behind the
    f16 rounding the two are the same half for every one of the 65536 codes."""
    v = np.arange(65536, dtype=np.uint32).astype(np.float32)
    quotient = (v / np.float32(65535)).astype(np.float16)
    product = (v * np.float32(1.0 / 65535.0)).astype(np.float16)
    assert np.array_equal(quotient.view(np.uint16), product.view(np.uint16))
    # (and the quotient is the fp32 quotient: float64 division rounded once gives the same floats)
    assert np.array_equal((np.arange(65536) / 65535.0).astype(np.float32), v / np.float32(65535))
