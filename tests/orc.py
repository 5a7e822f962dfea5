"""ctypes binding of the CPU oracle (oracle/libploracle.so) and of the real
reference CPU half (oracle/_ref/libplref.so) — test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libploracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libplref.so")

FMT = {"r8": 1, "rg8": 2, "rgba8": 3, "r16": 4, "rg16": 5, "rgba16": 6,
       "r16hf": 7, "rg16hf": 8, "rgba16hf": 9, "r32f": 10, "rg32f": 11, "rgba32f": 12}
S_NEAREST, S_BILINEAR, S_BICUBIC, S_HERMITE, S_GAUSSIAN, S_OVERSAMPLE = 1, 2, 3, 4, 5, 6
K = dict(box=0, triangle=1, hann=2, gaussian=3, sinc=4, jinc=5, cubic=6,
         spline16=7, spline36=8, spline64=9, none=-1)

_f32p = C.POINTER(C.c_float)


class Src(C.Structure):
    _fields_ = [("tex", _f32p), ("w", C.c_int), ("h", C.c_int),
                ("rect", C.c_float * 4), ("address_mode", C.c_int),
                ("rx", C.c_int), ("ry", C.c_int), ("rw", C.c_int), ("rh", C.c_int)]


class Region:
    """Texels [x0, x0+w) x [y0, y0+h) of a (full_w x full_h) texture: lets the oracle evaluate
    output windows of a full-size pass without the whole intermediate image."""

    def __init__(self, data, x0, y0, full_w, full_h):
        self.data = np.ascontiguousarray(data, np.float32)
        self.x0, self.y0, self.full_w, self.full_h = x0, y0, full_w, full_h


class window:
    """with orc.window(x0, y0, w, h): samplers evaluate only that window of their pass (and
    return w x h); dither takes its fragment coordinates from it."""

    def __init__(self, x0, y0, w, h):
        self.win = (x0, y0, w, h)

    def __enter__(self):
        lib().orc_set_window(*self.win)
        _WIN.append(self.win)
        return self

    def __exit__(self, *exc):
        _WIN.pop()
        lib().orc_set_window(*(_WIN[-1] if _WIN else (0, 0, 0, 0)))


_WIN = []


def _out_shape(out_w, out_h):
    return (_WIN[-1][3], _WIN[-1][2], 4) if _WIN else (out_h, out_w, 4)


class OrcFilter(C.Structure):
    _fields_ = [("kernel", C.c_int), ("window", C.c_int), ("kparams", C.c_double * 2),
                ("kradius", C.c_float), ("wradius", C.c_float), ("resizable", C.c_int),
                ("radius", C.c_float), ("clamp", C.c_float), ("blur", C.c_float),
                ("taper", C.c_float)]


JINC_R3 = 3.2383154841662362076499


def ewa_lanczos(blur=0.0):
    return OrcFilter(kernel=K["jinc"], window=K["jinc"], kradius=1.2196698912665045,
                     wradius=1.2196698912665045, resizable=1, radius=JINC_R3, blur=blur)


def lanczos(blur=0.0):
    return OrcFilter(kernel=K["sinc"], window=K["sinc"], kradius=1.0, wradius=1.0,
                     resizable=1, radius=3.0, blur=blur)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_SO)
        L.orc_round_f16.restype = C.c_float
        L.orc_round_f16.argtypes = [C.c_float]
        L.orc_filter_sample.restype = C.c_double
        L.orc_filter_sample.argtypes = [C.POINTER(OrcFilter), C.c_double]
        L.orc_ewa_resample_r32f.restype = C.c_double
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def tex_decode(arr, fmt):
    """Texture fetch of every texel -> float32 RGBA image (h, w, 4)."""
    arr = np.ascontiguousarray(arr)
    h, w = arr.shape[:2]
    out = np.empty((h, w, 4), np.float32)
    lib().orc_tex_decode(_p(arr), FMT[fmt], w, h, C.c_size_t(arr.strides[0]), _p(out))
    return out


def tex_encode(img, fmt):
    from libplacebo_amd import _FMT_DTYPES
    dt, nc = _FMT_DTYPES[fmt]
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[:2]
    out = np.zeros((h, w, nc), dt)
    lib().orc_tex_encode(_p(img), w, h, FMT[fmt], _p(out), C.c_size_t(out.strides[0]))
    return out


def _src(img, rect, address_mode):
    if isinstance(img, Region):
        d = img.data
        s = Src(tex=d.ctypes.data_as(_f32p), w=img.full_w, h=img.full_h,
                address_mode=address_mode, rx=img.x0, ry=img.y0, rw=d.shape[1], rh=d.shape[0])
        w, h = img.full_w, img.full_h
        img = d
    else:
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape[:2]
        s = Src(tex=img.ctypes.data_as(_f32p), w=w, h=h, address_mode=address_mode)
    rect = rect if rect is not None else (0, 0, w, h)
    s.rect = (C.c_float * 4)(*rect)
    return s, img


def sample_simple(img, kind, out_w, out_h, rect=None, scale=1.0, address_mode=0,
                  threshold=0.0):
    s, keep = _src(img, rect, address_mode)
    out = np.empty(_out_shape(out_w, out_h), np.float32)
    rw = abs(s.rect[2] - s.rect[0]); rh = abs(s.rect[3] - s.rect[1])
    lib().orc_sample_simple(C.byref(s), kind, C.c_float(scale), C.c_float(out_w / rw),
                            C.c_float(out_h / rh), C.c_float(threshold), out_w, out_h, _p(out))
    return out


def sample_polar(img, lut, radius, radius_zero, out_w, out_h, rect=None, scale=1.0,
                 antiring=0.0, gather_order=False, mask=0xF, address_mode=0):
    s, keep = _src(img, rect, address_mode)
    lut = np.ascontiguousarray(lut, np.float32)
    assert lut.size == 256
    out = np.empty(_out_shape(out_w, out_h), np.float32)
    lib().orc_sample_polar(C.byref(s), _p(lut), C.c_float(radius), C.c_float(radius_zero),
                           C.c_float(antiring), int(gather_order), C.c_float(scale),
                           C.c_uint(mask), out_w, out_h, _p(out))
    return out


def op_scale(img, s):
    s = (C.c_float * 4)(*([s] * 4 if np.isscalar(s) else s))
    lib().orc_op_scale(_p(img), C.c_size_t(img.size // 4), s)
    return img


def op_quant_f16(img):
    lib().orc_op_quant_f16(_p(img), C.c_size_t(img.size // 4))
    return img


def dither(img, matrix, depth, method=0, gamma=1.0, temporal=False, frame_index=0):
    h, w = img.shape[:2]
    size = 16
    m = None
    if matrix is not None:
        m = np.ascontiguousarray(matrix, np.float32)
        size = int(round(m.size ** 0.5))
    lib().orc_dither(_p(img), w, h, _p(m) if m is not None else None, size, method, depth,
                     C.c_float(gamma), int(temporal), frame_index)
    return img


def filter_generate_polar(f, cutoff=1e-3, n=256):
    w = np.empty(n, np.float32)
    r, rz = C.c_float(), C.c_float()
    lib().orc_filter_generate_polar(C.byref(f), C.c_float(cutoff), n, _p(w), C.byref(r), C.byref(rz))
    return w, r.value, rz.value


def filter_generate_ortho(f, cutoff=0.0, n=256, align=4):
    buf = np.empty(n * 64, np.float32)
    r, rz = C.c_float(), C.c_float()
    row_size = lib().orc_filter_generate_ortho(C.byref(f), C.c_float(cutoff), n, align, _p(buf),
                                               buf.size, C.byref(r), C.byref(rz))
    assert row_size > 0
    stride = (row_size + align - 1) // align * align
    return buf[:n * stride].reshape(n, stride).copy(), row_size, r.value, rz.value


def ewa_resample_r32f(f, src, dw, dh, use_lut):
    src = np.ascontiguousarray(src, np.float32)
    sh, sw = src.shape
    dst = np.empty((dh, dw), np.float32)
    taps = lib().orc_ewa_resample_r32f(C.byref(f), _p(src), sw, sh, dw, dh, int(use_lut), _p(dst))
    return dst, taps


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    """The real reference CPU half (filters.c, tone_mapping.c, ... compiled from
    /root/reference by oracle/build_ref.sh)."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
    return _ref


# ---- colour stages -----------------------------------------------------------------
def _luma(l):
    return (C.c_float * 3)(*(l if l is not None else (0.2126, 0.7152, 0.0722)))


def linearize(img, trc, csp_min, csp_max, luma=None):
    lib().orc_linearize(_p(img), C.c_size_t(img.size // 4), trc, C.c_float(csp_min),
                        C.c_float(csp_max), _luma(luma))
    return img


def delinearize(img, trc, csp_min, csp_max, luma=None):
    lib().orc_delinearize(_p(img), C.c_size_t(img.size // 4), trc, C.c_float(csp_min),
                          C.c_float(csp_max), _luma(luma))
    return img


def sigmoid(img, center=0.75, slope=6.5, inverse=False):
    lib().orc_sigmoid(_p(img), C.c_size_t(img.size // 4), C.c_float(center), C.c_float(slope),
                      int(inverse))
    return img


def alpha(img, mode):
    lib().orc_alpha(_p(img), C.c_size_t(img.size // 4), mode)
    return img


def op_affine(img, m9, c3):
    lib().orc_op_affine(_p(img), C.c_size_t(img.size // 4), (C.c_float * 9)(*m9),
                        (C.c_float * 3)(*c3))
    return img


class PeakBuf(C.Structure):
    _fields_ = [("frame_wg_count", C.c_uint32 * 12), ("frame_wg_active", C.c_uint32 * 12),
                ("frame_sum_pq", C.c_uint32 * 12), ("frame_max_pq", C.c_uint32 * 12),
                ("frame_hist", (C.c_uint32 * 64) * 12)]


def detect_peak(img_padded, trc, csp_min, csp_max, luma, black_cutoff=1.0, use_hist=False):
    img = np.ascontiguousarray(img_padded, np.float32)
    ph, pw = img.shape[:2]
    assert pw % 16 == 0 and ph % 16 == 0
    out = PeakBuf()
    lib().orc_detect_peak(_p(img), pw, ph, trc, C.c_float(csp_min), C.c_float(csp_max),
                          _luma(luma), C.c_float(black_cutoff), int(use_hist), C.byref(out))
    return np.frombuffer(bytes(out), np.uint32).copy()


class ColorMap(C.Structure):
    _fields_ = [("rgb2lms", C.c_float * 9), ("lms2rgb", C.c_float * 9), ("tone_mode", C.c_int),
                ("tone_p", C.c_float * 4), ("tone_lut", C.c_void_p), ("tone_lut_size", C.c_int),
                ("gamut_lut", C.c_void_p), ("gamut_size", C.c_int * 3),
                ("gamut_scale", C.c_float), ("gamut_offset", C.c_float),
                ("lowres", C.c_void_p), ("cr_strength", C.c_float), ("cr_out_min", C.c_float),
                ("cr_out_max", C.c_float), ("gamut_tricubic", C.c_int)]


def extract_features(img, klms):
    """pl_shader_extract_features after the linearization; klms = fp32 (203/10000 * rgb2lms)."""
    lib().orc_extract_features(_p(img), C.c_size_t(img.size // 4), (C.c_float * 9)(*klms))
    return img


def feature_luma(fm, out_w, out_h):
    """The contrast recovery's bicubic lookup of the (h x w float32) feature map."""
    fm = np.ascontiguousarray(fm, np.float32)
    out = np.empty((out_h, out_w), np.float32)
    lib().orc_feature_luma(_p(fm), fm.shape[1], fm.shape[0], out_w, out_h, _p(out))
    return out


def color_map(img, rgb2lms, lms2rgb, tone_mode=-1, tone_p=(0, 0, 0, 0), tone_lut=None,
              gamut_lut=None, gamut_size=(48, 32, 256), gamut_scale=0.0, gamut_offset=0.0,
              lowres=None, cr_strength=0.0, cr_out=(0.0, 1.0), gamut_tricubic=False):
    cm = ColorMap(tone_mode=tone_mode, gamut_scale=gamut_scale, gamut_offset=gamut_offset,
                  gamut_tricubic=int(gamut_tricubic))
    cm.rgb2lms = (C.c_float * 9)(*rgb2lms)
    cm.lms2rgb = (C.c_float * 9)(*lms2rgb)
    cm.tone_p = (C.c_float * 4)(*tone_p)
    keep = []
    if tone_lut is not None:
        tl = np.ascontiguousarray(tone_lut, np.float32)
        keep.append(tl)
        cm.tone_lut, cm.tone_lut_size = tl.ctypes.data, tl.size
    if gamut_lut is not None:
        gl = np.ascontiguousarray(gamut_lut, np.uint16)
        keep.append(gl)
        cm.gamut_lut = gl.ctypes.data
        cm.gamut_size = (C.c_int * 3)(*gamut_size)
    if lowres is not None:
        lr = np.ascontiguousarray(lowres, np.float32)
        assert lr.size == img.size // 4
        keep.append(lr)
        cm.lowres = lr.ctypes.data
        cm.cr_strength, cm.cr_out_min, cm.cr_out_max = cr_strength, cr_out[0], cr_out[1]
    lib().orc_color_map(_p(img), C.c_size_t(img.size // 4), C.byref(cm))
    return img


# ---- separable filters / debanding ---------------------------------------------------------
def triangle(blur=0.0):
    return OrcFilter(kernel=K["triangle"], window=K["none"], kradius=1.0, wradius=1.0,
                     resizable=1, radius=1.0, blur=blur)


def mitchell(blur=0.0):
    f = OrcFilter(kernel=K["cubic"], window=K["none"], kradius=2.0, wradius=1.0,
                  resizable=0, radius=2.0, blur=blur)
    # pl_filter_config.params are floats (promoted to double by the kernel)
    f.kparams[0] = f.kparams[1] = float(np.float32(1 / 3.0))
    return f


def ortho_lut_rows(rows, row_size, use_linear):
    """fill_ortho_lut (sampling.c:914-942): the LUT as uploaded."""
    if not use_linear:
        return rows
    out = rows.copy()
    n, stride = rows.shape
    one = np.float32
    i = 0
    while i < row_size:
        w0, w1 = rows[:, i].astype(one), rows[:, i + 1].astype(one)
        out[:, i] = w0 + w1
        out[:, i + 1] = w1 / (w0 + w1)
        i += 2
    for j in range(i, stride):
        out[:, j] = out[:, j - 4] if j >= 4 else 0
    return out


def sample_ortho(img, rows, row_size, direction, out_w, out_h, rect=None, scale=1.0,
                 use_linear=False, use_ar=False, antiring=0.0, mask=0xF, address_mode=0):
    s, keep = _src(img, rect, address_mode)
    rows = np.ascontiguousarray(rows, np.float32)
    out = np.empty(_out_shape(out_w, out_h), np.float32)
    lib().orc_sample_ortho(C.byref(s), _p(rows), row_size, rows.shape[1], direction,
                           int(use_linear), int(use_ar), C.c_float(antiring), C.c_float(scale),
                           C.c_uint(mask), out_w, out_h, _p(out))
    return out


def deband(img, out_w, out_h, iterations=1, threshold=3.0, radius=16.0, grain=4.0,
           grain_neutral=(0, 0, 0), scale=1.0, mask=0x7, frame_index=0, rect=None, address_mode=0):
    s, keep = _src(img, rect, address_mode)
    out = np.empty(_out_shape(out_w, out_h), np.float32)
    lib().orc_deband(C.byref(s), iterations, C.c_float(threshold), C.c_float(radius),
                     C.c_float(grain), (C.c_float * 3)(*grain_neutral), C.c_float(scale),
                     C.c_uint(mask), C.c_uint(frame_index), out_w, out_h, _p(out))
    return out


def error_diffusion(img, depth, shift, divisor, pattern):
    img = np.ascontiguousarray(img, np.float32)
    h, w = img.shape[:2]
    out = np.zeros_like(img)
    pat = ((C.c_int * 5) * 3)(*[(C.c_int * 5)(*row) for row in pattern])
    lib().orc_error_diffusion(_p(img), w, h, depth, shift, divisor, pat, _p(out))
    return out


def custom_lut(img, lut, size):
    """pl_shader_custom_lut: lut = RGB triples (R innermost), size = (n,) or (r, g, b)."""
    lut = np.ascontiguousarray(lut, np.float32)
    size = tuple(size) + (0,) * (3 - len(size))
    lib().orc_custom_lut(_p(img), C.c_size_t(img.size // 4), _p(lut), (C.c_int * 3)(*size))
    return img


class OverlayPart(C.Structure):
    """struct orc_overlay_part: a part as it lands on a plane (pl_oracle.c)"""
    _fields_ = [("x0", C.c_float), ("y0", C.c_float), ("x1", C.c_float), ("y1", C.c_float),
                ("ox", C.c_float), ("oy", C.c_float),
                ("ux", C.c_float), ("uy", C.c_float), ("u0", C.c_float),
                ("vx", C.c_float), ("vy", C.c_float), ("v0", C.c_float),
                ("color", C.c_float * 4)]


OVERLAY_NORMAL, OVERLAY_MONOCHROME = 0, 1


def overlay_fragments(tex, linear, mode, part, w, h):
    """fragments of one overlay part over a w x h plane: (color, coverage, mask)"""
    tex = np.ascontiguousarray(tex, dtype=np.float32)
    color = np.zeros((h, w, 4), np.float32)
    cov = np.ones((h, w), np.float32)
    mask = np.zeros((h, w), np.uint8)
    lib().orc_overlay_fragments(_p(tex), tex.shape[1], tex.shape[0], int(linear), mode,
                                C.byref(part), w, h, _p(color), _p(cov), _p(mask))
    return color, cov, mask


def blend(dst, src, mask, factors, enable=True, fixed_point=True):
    """the blend unit over the masked pixels of `dst` (in place; not yet rounded to a format)"""
    assert dst.dtype == np.float32 and dst.flags.c_contiguous
    src = np.ascontiguousarray(src, dtype=np.float32)
    f = (C.c_int * 4)(*factors)
    m = np.ascontiguousarray(mask, dtype=np.uint8) if mask is not None else None
    lib().orc_blend(_p(dst), _p(src), _p(m) if m is not None else None,
                    dst.shape[0] * dst.shape[1], f, int(enable), int(fixed_point))
    return dst


DEINT_WEAVE, DEINT_BOB, DEINT_YADIF, DEINT_BWDIF = 0, 1, 2, 3


def deinterlace(cur, prev, next_, field, first_field, algo, skip_spatial_check=False,
                comp_mask=0xf):
    """pl_shader_deinterlace over decoded frames (h, w, 4); prev / next_ may be None;
    field / first_field: 0 none, 1 top, 2 bottom"""
    cur = np.ascontiguousarray(cur, dtype=np.float32)
    h, w = cur.shape[:2]
    keep = [cur]
    ptr = []
    for a in (prev, next_):
        if a is None:
            ptr.append(None)
        else:
            a = np.ascontiguousarray(a, dtype=np.float32)
            assert a.shape == cur.shape
            keep.append(a)
            ptr.append(_p(a))
    out = np.empty_like(cur)
    lib().orc_deinterlace(_p(cur), ptr[0], ptr[1], w, h, field, first_field, algo,
                          int(skip_spatial_check), comp_mask, _p(out))
    return out


def distort(img, tf, out_w, out_h, bicubic=False, alpha_mode=0, address_mode=0):
    """pl_shader_distort: tf = canvas -> texture coordinates (m00, m01, m10, m11, c0, c1)"""
    src, keep = _src(img, None, address_mode)
    out = np.empty((out_h, out_w, 4), np.float32)
    t = (C.c_float * 6)(*tf)
    lib().orc_distort(C.byref(src), t, int(bicubic), int(alpha_mode), out_w, out_h, _p(out))
    return out


class DoviComp(C.Structure):
    """struct orc_dovi_comp (pl_reshape_data with ints)"""
    _fields_ = [("num_pivots", C.c_int), ("pivots", C.c_float * 9), ("method", C.c_int * 8),
                ("poly_coeffs", (C.c_float * 3) * 8), ("mmr_order", C.c_int * 8),
                ("mmr_constant", C.c_float * 8), ("mmr_coeffs", ((C.c_float * 7) * 3) * 8)]


def dovi_reshape(img, comps):
    """pl_shader_dovi_reshape in place; comps = (DoviComp * 3)"""
    assert img.dtype == np.float32 and img.flags.c_contiguous
    lib().orc_dovi_reshape(_p(img), C.c_size_t(img.size // 4), comps)
    return img


def dovi_lms(img, m9):
    assert img.dtype == np.float32 and img.flags.c_contiguous
    m = (C.c_float * 9)(*m9)
    lib().orc_dovi_lms(_p(img), C.c_size_t(img.size // 4), m)
    return img
