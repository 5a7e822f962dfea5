"""ctypes mirrors of the colour / tone / gamut structs shared by the product's
Tier-0 (include/libplacebo/*.h) and the real reference (same layouts)."""
import ctypes as C


class XY(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class Prim(C.Structure):
    _fields_ = [("red", XY), ("green", XY), ("blue", XY), ("white", XY)]


class Bez(C.Structure):
    _fields_ = [("target_luma", C.c_float), ("knee_x", C.c_float), ("knee_y", C.c_float),
                ("anchors", C.c_float * 15), ("num_anchors", C.c_uint8)]


class Hdr(C.Structure):
    _fields_ = [("prim", Prim), ("min_luma", C.c_float), ("max_luma", C.c_float),
                ("max_cll", C.c_float), ("max_fall", C.c_float), ("scene_max", C.c_float * 3),
                ("scene_avg", C.c_float), ("ootf", Bez), ("max_pq_y", C.c_float),
                ("avg_pq_y", C.c_float)]


class Csp(C.Structure):
    _fields_ = [("primaries", C.c_int), ("transfer", C.c_int), ("hdr", Hdr)]


class M3(C.Structure):
    _fields_ = [("m", (C.c_float * 3) * 3)]


class T3(C.Structure):
    _fields_ = [("mat", M3), ("c", C.c_float * 3)]


class Bits(C.Structure):
    _fields_ = [("sample_depth", C.c_int), ("color_depth", C.c_int), ("bit_shift", C.c_int)]


class Repr(C.Structure):
    _fields_ = [("sys", C.c_int), ("levels", C.c_int), ("alpha", C.c_int), ("bits", Bits),
                ("dovi", C.c_void_p)]


class Adj(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("brightness", "contrast", "saturation", "hue", "gamma",
                                         "temperature")]


class TMC(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "knee_adaptation", "knee_minimum", "knee_maximum", "knee_default", "knee_offset",
        "slope_tuning", "slope_offset", "spline_contrast", "reinhard_contrast", "linear_knee",
        "exposure")]


TMC_DEFAULT = (0.4, 0.1, 0.8, 0.4, 1.0, 1.5, 0.2, 0.5, 0.5, 0.3, 1.0)


class TMP(C.Structure):
    _fields_ = [("function", C.c_void_p), ("constants", TMC), ("input_scaling", C.c_int),
                ("output_scaling", C.c_int), ("lut_size", C.c_size_t), ("input_min", C.c_float),
                ("input_max", C.c_float), ("input_avg", C.c_float), ("output_min", C.c_float),
                ("output_max", C.c_float), ("hdr", Hdr), ("param", C.c_float)]


class GMC(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("perceptual_deadzone", "perceptual_strength",
                                         "colorimetric_gamma", "softclip_knee", "softclip_desat")]


GMC_DEFAULT = (0.30, 0.80, 1.80, 0.70, 0.35)


class GMP(C.Structure):
    _fields_ = [("function", C.c_void_p), ("input_gamut", Prim), ("output_gamut", Prim),
                ("min_luma", C.c_float), ("max_luma", C.c_float), ("constants", GMC),
                ("lut_size_I", C.c_int), ("lut_size_C", C.c_int), ("lut_size_h", C.c_int),
                ("lut_stride", C.c_int), ("chroma_margin", C.c_float)]


HDR_NORM, HDR_SQRT, HDR_NITS, HDR_PQ = 0, 1, 2, 3
TONE_NAMES = [b"clip", b"st2094-40", b"st2094-10", b"bt2390", b"bt2446a", b"spline", b"reinhard",
              b"mobius", b"hable", b"gamma", b"linear", b"linearlight"]
GAMUT_NAMES = [b"clip", b"perceptual", b"softclip", b"relative", b"saturation", b"absolute",
               b"desaturate", b"darken", b"highlight", b"linear"]


def declare(lib):
    lib.pl_raw_primaries_get.restype = C.POINTER(Prim)
    for fn in ("pl_get_rgb2xyz_matrix", "pl_get_xyz2rgb_matrix", "pl_ipt_rgb2lms",
               "pl_ipt_lms2rgb", "pl_get_color_mapping_matrix"):
        getattr(lib, fn).restype = M3
    lib.pl_color_repr_decode.restype = T3
    lib.pl_hdr_rescale.restype = C.c_float
    lib.pl_hdr_rescale.argtypes = [C.c_int, C.c_int, C.c_float]
    lib.pl_color_repr_normalize.restype = C.c_float
    lib.pl_tone_map_sample.restype = C.c_float
    lib.pl_find_tone_map_function.restype = C.c_void_p
    lib.pl_find_tone_map_function.argtypes = [C.c_char_p]
    lib.pl_find_gamut_map_function.restype = C.c_void_p
    lib.pl_find_gamut_map_function.argtypes = [C.c_char_p]
    return lib


def m3(m):
    return [m.m[i][j] for i in range(3) for j in range(3)]
