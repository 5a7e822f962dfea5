"""Checker-side restatement of the *parameter inference* of pl_shader_color_map_ex
(reference src/shaders/colorspace.c:1612-1790): which tone/gamut parameters and LUTs
a (src, dst) colour-space pair resolves to. All maths is delegated to the real
reference CPU half (oracle/_ref/libplref.so) when it is present, else to the
product's Tier-0 (which tests/test_tier0_ref.py pins bit-exactly against it).

Test infrastructure only.
"""
import ctypes as C

import numpy as np

import orc
import ref_structs as R


def _cpu():
    if orc.have_ref():
        lib = orc.ref()
    else:
        import libplacebo_amd as pl
        lib = C.CDLL(pl._capi.LIB_PATH) if hasattr(pl._capi, "LIB_PATH") else pl.lib()
    return R.declare(lib)


class NLP(C.Structure):
    _fields_ = [("color", C.POINTER(R.Csp)), ("metadata", C.c_int), ("scaling", C.c_int),
                ("out_min", C.POINTER(C.c_float)), ("out_max", C.POINTER(C.c_float)),
                ("out_avg", C.POINTER(C.c_float))]


def nominal(lib, csp, metadata, scaling):
    mn, mx, av = C.c_float(), C.c_float(), C.c_float()
    p = NLP(C.pointer(csp), metadata, scaling, C.pointer(mn), C.pointer(mx), C.pointer(av))
    lib.pl_color_space_nominal_luma_ex(C.byref(p))
    return mn.value, mx.value, av.value


def make_csp(primaries, transfer, max_luma=0.0, min_luma=0.0, max_cll=0.0):
    c = R.Csp()
    c.primaries, c.transfer = primaries, transfer
    c.hdr.max_luma, c.hdr.min_luma, c.hdr.max_cll = max_luma, min_luma, max_cll
    return c


def resolve(src, dst, tone=b"spline", gamut=b"perceptual", lut_size=256,
            lut3d=(48, 32, 256), metadata=0):
    """-> dict(src, dst (inferred), lin/delin args, oracle color_map kwargs)"""
    lib = _cpu()
    lib.pl_color_space_infer_map(C.byref(src), C.byref(dst))

    tp = R.TMP()
    tp.function = lib.pl_find_tone_map_function(tone)
    tp.constants = R.TMC(*R.TMC_DEFAULT)
    tp.input_scaling = tp.output_scaling = R.HDR_PQ
    tp.lut_size = lut_size
    tp.hdr = src.hdr
    tp.input_min, tp.input_max, tp.input_avg = nominal(lib, src, metadata, R.HDR_PQ)
    tp.output_min, tp.output_max, _ = nominal(lib, dst, 2, R.HDR_PQ)
    lib.pl_tone_map_params_infer(C.byref(tp))
    if abs(tp.input_max - tp.output_max) < 1e-6:
        tp.output_max = tp.input_max
    if abs(tp.input_min - tp.output_min) < 1e-6:
        tp.output_min = tp.input_min
    tp.output_max = min(tp.output_max, tp.input_max)

    gp = R.GMP()
    gp.function = lib.pl_find_gamut_map_function(gamut)
    gp.constants = R.GMC(*R.GMC_DEFAULT)
    gp.input_gamut, gp.output_gamut = src.hdr.prim, dst.hdr.prim
    gp.lut_size_I, gp.lut_size_C, gp.lut_size_h = lut3d
    gp.lut_stride = 3
    gp.min_luma, gp.max_luma, _ = nominal(lib, dst, 2, R.HDR_PQ)
    lib.pl_primaries_compatible.restype = C.c_bool
    lib.pl_primaries_clip.restype = R.Prim
    if gamut in (b"perceptual", b"saturation"):   # .bidirectional (gamut_mapping.c:744,862)
        # bidirectional mappers: clip the target gamut to the source unless expanding
        if lib.pl_primaries_compatible(C.byref(gp.input_gamut), C.byref(gp.output_gamut)):
            gp.output_gamut = lib.pl_primaries_clip(C.byref(gp.output_gamut),
                                                    C.byref(gp.input_gamut))

    lib.pl_tone_map_params_noop.restype = C.c_bool
    lib.pl_gamut_map_params_noop.restype = C.c_bool
    need_tone = not lib.pl_tone_map_params_noop(C.byref(tp))
    need_gamut = not lib.pl_gamut_map_params_noop(C.byref(gp))

    kw = dict(rgb2lms=R.m3(lib.pl_ipt_rgb2lms(lib.pl_raw_primaries_get(src.primaries))),
              lms2rgb=R.m3(lib.pl_ipt_lms2rgb(lib.pl_raw_primaries_get(dst.primaries))))
    if need_tone:
        lut = np.zeros(lut_size, np.float32)
        lib.pl_tone_map_generate(lut.ctypes.data_as(C.c_void_p), C.byref(tp))
        rng = tp.input_max - tp.input_min
        one = np.float32(1.0)
        kw.update(tone_mode=2, tone_lut=lut,
                  tone_p=(float(one / np.float32(rng)),
                          float(np.float32(-tp.input_min) / np.float32(rng)), 0, 0))
    if need_gamut:
        n = lut3d[0] * lut3d[1] * lut3d[2]
        tmp = np.zeros((n, 3), np.float32)
        lib.pl_gamut_map_generate(tmp.ctypes.data_as(C.c_void_p), C.byref(gp))
        packed = np.zeros((n, 4), np.uint16)
        u16 = np.float32(65535.0)
        half = np.float32(32767.0)
        packed[:, 0] = np.floor(tmp[:, 0] * u16 + np.float32(0.5))
        packed[:, 1] = np.floor(tmp[:, 1] * u16 + half + np.float32(0.5))
        packed[:, 2] = np.floor(tmp[:, 2] * u16 + half + np.float32(0.5))
        rng = np.float32(gp.max_luma) - np.float32(gp.min_luma)
        kw.update(gamut_lut=packed, gamut_size=lut3d,
                  gamut_scale=float(np.float32(1.0) / rng),
                  gamut_offset=float(np.float32(-gp.min_luma) / rng))

    def luma(csp):
        m = lib.pl_get_rgb2xyz_matrix(lib.pl_raw_primaries_get(csp.primaries))
        return [m.m[1][0], m.m[1][1], m.m[1][2]]

    smin, smax, _ = nominal(lib, src, 2, R.HDR_NORM)
    dmin, dmax, _ = nominal(lib, dst, 2, R.HDR_NORM)
    return dict(src=src, dst=dst, need_tone=need_tone, need_gamut=need_gamut, kw=kw,
                lin=(int(src.transfer), smin, smax, luma(src)),
                delin=(int(dst.transfer), dmin, dmax, luma(dst)), tone=tp, gamut=gp)


def apply(img, r, lowres=None, strength=0.0, prelinearized=False):
    """Full oracle colour-map pipeline on float32 rgba `img` (in place). lowres: per-pixel
    low-frequency luma for the contrast recovery (orc.feature_luma). prelinearized: `img` is
    already in linear light (pl_color_map_args.prelinearized)."""
    if not prelinearized:
        orc.linearize(img, *r["lin"])
    if r["need_tone"] or r["need_gamut"]:
        kw = dict(r["kw"])
        if lowres is not None:
            kw.update(lowres=lowres, cr_strength=strength,
                      cr_out=(r["tone"].output_min, r["tone"].output_max))
        orc.color_map(img, **kw)
    else:
        raise NotImplementedError("matrix-only fast path: use orc.op_affine")
    orc.delinearize(img, *r["delin"])
    return img
