"""libplacebo_amd — Python harness over the C-ABI of libplacebo_hip.so.

The product is the C/HIP library (include/libplacebo/*.h); this package only
binds it with ctypes so tests and bench.py can drive it, and mirrors the
reference's object model (pl_gpu / pl_tex / pl_shader / pl_dispatch) 1:1.
There is no Python or CPU implementation of any stage in here: without the
built library (and, for anything that launches work, a HIP device) calls fail.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import BuildError  # noqa: F401

_lib = None


def lib():
    """The loaded libplacebo_hip.so (raises BuildError if it was not built)."""
    global _lib
    if _lib is None:
        _lib = capi.declare(capi.load())
    return _lib


# numpy dtype <-> pl_fmt name
_FMT_DTYPES = {
    "r8": (np.uint8, 1), "rg8": (np.uint8, 2), "rgba8": (np.uint8, 4),
    "r16": (np.uint16, 1), "rg16": (np.uint16, 2), "rgba16": (np.uint16, 4),
    "r16hf": (np.float16, 1), "rg16hf": (np.float16, 2), "rgba16hf": (np.float16, 4),
    "r32f": (np.float32, 1), "rg32f": (np.float32, 2), "rgba32f": (np.float32, 4),
}

ADDRESS_CLAMP, ADDRESS_REPEAT, ADDRESS_MIRROR = 0, 1, 2
DITHER_BLUE_NOISE, DITHER_ORDERED_LUT, DITHER_ORDERED_FIXED, DITHER_WHITE_NOISE = 0, 1, 2, 3
FILTER_UPSCALING, FILTER_DOWNSCALING, FILTER_FRAME_MIXING = 1, 2, 4


def filter_config(name, usage=FILTER_UPSCALING):
    """Look up one of the built-in scaler configs (pl_find_filter_config)."""
    p = lib().pl_find_filter_config(name.encode(), usage)
    if not p:
        raise KeyError(name)
    return p.contents


# enum values (include/libplacebo/colorspace.h)
PRIM = dict(unknown=0, bt601_525=1, bt601_625=2, bt709=3, bt470m=4, ebu3213=5, bt2020=6,
            apple=7, adobe=8, prophoto=9, cie1931=10, dci_p3=11, display_p3=12, v_gamut=13,
            s_gamut=14, film_c=15, aces_ap0=16, aces_ap1=17)
TRC = dict(unknown=0, bt1886=1, srgb=2, linear=3, gamma18=4, gamma20=5, gamma22=6, gamma24=7,
           gamma26=8, gamma28=9, prophoto=10, st428=11, pq=12, hlg=13, vlog=14, slog1=15,
           slog2=16, scrgb=17)
SYS = dict(unknown=0, bt601=1, bt709=2, smpte240m=3, bt2020nc=4, bt2020c=5, bt2100pq=6,
           bt2100hlg=7, dolbyvision=8, ycgco=9, ycgco_re=10, ycgco_ro=11, rgb=12, xyz=13)
LEVELS = dict(unknown=0, limited=1, full=2)
ALPHA = dict(unknown=0, independent=1, premultiplied=2, none=3)
HDR_NORM, HDR_SQRT, HDR_NITS, HDR_PQ = 0, 1, 2, 3


def color_space(primaries="bt709", transfer="bt1886", min_luma=0.0, max_luma=0.0, **hdr):
    cs = capi.ColorSpace(primaries=PRIM[primaries], transfer=TRC[transfer])
    cs.hdr.min_luma, cs.hdr.max_luma = min_luma, max_luma
    for k, v in hdr.items():
        setattr(cs.hdr, k, v)
    return cs


def color_repr(sys="rgb", levels="full", alpha="unknown", sample_depth=0, color_depth=0,
               bit_shift=0):
    return capi.ColorRepr(sys=SYS[sys], levels=LEVELS[levels], alpha=ALPHA[alpha],
                          bits=capi.BitEncoding(sample_depth, color_depth, bit_shift))


def peak_detect_params(smoothing_period=20.0, scene_threshold_low=1.0, scene_threshold_high=3.0,
                       percentile=100.0, black_cutoff=1.0, allow_delayed=False):
    return capi.PeakDetectParams(smoothing_period, scene_threshold_low, scene_threshold_high,
                                 percentile, black_cutoff, allow_delayed, 0.0)


def color_map_params(tone="spline", gamut="perceptual", **kw):
    p = capi.ColorMapParams(
        gamut_mapping=lib().pl_find_gamut_map_function(gamut.encode()),
        gamut_constants=capi.GamutMapConstants(*capi.GAMUT_MAP_CONSTANTS),
        lut3d_size=(C.c_int * 3)(48, 32, 256),
        tone_mapping_function=lib().pl_find_tone_map_function(tone.encode()),
        tone_constants=capi.ToneMapConstants(*capi.TONE_MAP_CONSTANTS),
        metadata=0, lut_size=256, contrast_smoothness=3.5,
        visualize_rect=capi.Rect2df(0, 0, 1, 1))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def hip_device_count():
    return lib().pl_hip_device_count()


class Texture:
    def __init__(self, gpu, ptr, owned=True):
        self.gpu, self.ptr, self.owned = gpu, ptr, owned

    @property
    def w(self):
        return self.ptr.contents.params.w

    @property
    def h(self):
        return self.ptr.contents.params.h

    @property
    def fmt_name(self):
        return self.ptr.contents.params.format.contents.name.decode()

    def device_ptr(self):
        pitch = C.c_size_t()
        p = lib().pl_hip_tex_ptr(self.ptr, C.byref(pitch))
        return p, pitch.value

    def upload(self, arr):
        dt, nc = _FMT_DTYPES[self.fmt_name]
        arr = np.ascontiguousarray(arr, dtype=dt).reshape(self.h, self.w, nc)
        xp = capi.TexTransferParams(tex=self.ptr, ptr=arr.ctypes.data)
        if not lib().pl_tex_upload(self.gpu.gpu, C.byref(xp)):
            raise RuntimeError("pl_tex_upload failed")

    def blit_from(self, src):
        """whole-texture 1:1 copy of `src` into this texture, queued on the stream (no host sync)"""
        bp = capi.TexBlitParams(src=src.ptr, dst=self.ptr,
                                src_rc=capi.Rect3d(0, 0, 0, src.w, src.h, 1),
                                dst_rc=capi.Rect3d(0, 0, 0, self.w, self.h, 1))
        lib().pl_tex_blit(self.gpu.gpu, C.byref(bp))

    def download(self):
        dt, nc = _FMT_DTYPES[self.fmt_name]
        out = np.empty((self.h, self.w, nc), dtype=dt)
        xp = capi.TexTransferParams(tex=self.ptr, ptr=out.ctypes.data)
        if not lib().pl_tex_download(self.gpu.gpu, C.byref(xp)):
            raise RuntimeError("pl_tex_download failed")
        return out

    def destroy(self):
        if self.ptr:
            p = self.ptr
            lib().pl_tex_destroy(self.gpu.gpu, C.byref(p))
            self.ptr = None


class ShaderObj:
    """A pl_shader_obj slot (persistent LUT / filter state)."""

    def __init__(self):
        self.slot = C.c_void_p(None)

    def destroy(self):
        lib().pl_shader_obj_destroy(C.byref(self.slot))


FIELD_NONE, FIELD_TOP, FIELD_BOTTOM = 0, 1, 2
DEINTERLACE_WEAVE, DEINTERLACE_BOB, DEINTERLACE_YADIF, DEINTERLACE_BWDIF = 0, 1, 2, 3


class Shader:
    """A pl_shader obtained from pl_dispatch_begin."""

    def __init__(self, gpu):
        self.gpu = gpu
        self.sh = C.c_void_p(lib().pl_dispatch_begin(gpu.dp))
        self._keep = []

    def _src(self, tex, rect=None, new_w=0, new_h=0, components=0, scale=0.0,
             address_mode=ADDRESS_CLAMP, component_mask=0):
        s = capi.SampleSrc(tex=tex.ptr, address_mode=address_mode, components=components,
                           component_mask=component_mask, new_w=new_w, new_h=new_h, scale=scale)
        if rect is not None:
            s.rect = capi.Rect2df(*rect)
        return s

    def sample(self, kind, tex, **kw):
        s = self._src(tex, **{k: v for k, v in kw.items() if k != "threshold"})
        if kind == "oversample":
            ok = lib().pl_shader_sample_oversample(self.sh, C.byref(s), kw.get("threshold", 0.0))
        else:
            ok = getattr(lib(), f"pl_shader_sample_{kind}")(self.sh, C.byref(s))
        return ok

    def sample_polar(self, tex, cfg, lut_obj, antiring=0.0, no_compute=False, no_widening=False,
                     **kw):
        s = self._src(tex, **kw)
        fp = capi.SampleFilterParams(filter=cfg, antiring=antiring, no_compute=no_compute,
                                     no_widening=no_widening,
                                     lut=C.pointer(lut_obj.slot))
        self._keep.append(fp)
        return lib().pl_shader_sample_polar(self.sh, C.byref(s), C.byref(fp))

    def sample_ortho(self, tex, cfg, lut_obj, antiring=0.0, no_widening=False, **kw):
        s = self._src(tex, **kw)
        fp = capi.SampleFilterParams(filter=cfg, antiring=antiring, no_widening=no_widening,
                                     lut=C.pointer(lut_obj.slot))
        self._keep.append(fp)
        return lib().pl_shader_sample_ortho2(self.sh, C.byref(s), C.byref(fp))

    def deband(self, tex, iterations=1, threshold=3.0, radius=16.0, grain=4.0,
               grain_neutral=(0.0, 0.0, 0.0), **kw):
        s = self._src(tex, **kw)
        dp = capi.DebandParams(iterations, threshold, radius, grain,
                               (C.c_float * 3)(*grain_neutral))
        self._keep.append(dp)
        lib().pl_shader_deband(self.sh, C.byref(s), C.byref(dp))
        return not self.failed()

    def dither(self, depth, state_obj, method=DITHER_BLUE_NOISE, lut_size=6, temporal=False,
               transfer=3):
        dp = capi.DitherParams(method=method, lut_size=lut_size, temporal=temporal,
                               transfer=transfer)
        lib().pl_shader_dither(self.sh, depth, C.byref(state_obj.slot) if state_obj else None,
                               C.byref(dp))

    def error_diffusion(self, src_tex, dst_tex, depth, kernel="sierra-lite"):
        k = lib().pl_find_error_diffusion_kernel(kernel.encode())
        if not k:
            raise KeyError(kernel)
        ep = capi.ErrorDiffusionParams(C.cast(src_tex.ptr, C.c_void_p), C.cast(dst_tex.ptr, C.c_void_p),
                                       depth, k)
        self._keep.append(ep)
        return lib().pl_shader_error_diffusion(self.sh, C.byref(ep))

    # ---- colour stages (shaders/colorspace.h) --------------------------------------------
    def decode_color(self, repr_, adjustment=None):
        lib().pl_shader_decode_color(self.sh, C.byref(repr_),
                                     C.byref(adjustment) if adjustment else None)

    def encode_color(self, repr_):
        lib().pl_shader_encode_color(self.sh, C.byref(repr_))

    def set_alpha(self, repr_, mode):
        lib().pl_shader_set_alpha(self.sh, C.byref(repr_), mode)

    def linearize(self, csp):
        lib().pl_shader_linearize(self.sh, C.byref(csp))

    def delinearize(self, csp):
        lib().pl_shader_delinearize(self.sh, C.byref(csp))

    def sigmoidize(self, center=0.75, slope=6.5, inverse=False):
        sp = capi.SigmoidParams(center, slope)
        fn = lib().pl_shader_unsigmoidize if inverse else lib().pl_shader_sigmoidize
        fn(self.sh, C.byref(sp))

    def detect_peak(self, csp, state_obj, **kw):
        pp = peak_detect_params(**kw)
        return lib().pl_shader_detect_peak(self.sh, csp, C.byref(state_obj.slot), C.byref(pp))

    def custom_lut(self, lut, state_obj):
        self._keep.append(lut)
        lib().pl_shader_custom_lut(self.sh, C.byref(lut), C.byref(state_obj.slot))

    def cone_distort(self, csp, cone_params):
        lib().pl_shader_cone_distort(self.sh, csp, C.byref(cone_params))

    def extract_features(self, csp):
        lib().pl_shader_extract_features(self.sh, csp)

    def deinterlace(self, cur, prev=None, next_=None, field=FIELD_TOP, first_field=FIELD_TOP,
                    algo=DEINTERLACE_YADIF, skip_spatial_check=False, component_mask=0):
        src = capi.DeinterlaceSource(field=field, first_field=first_field,
                                     component_mask=component_mask)
        src.cur.top = cur.ptr
        if prev is not None:
            src.prev.top = prev.ptr
        if next_ is not None:
            src.next.top = next_.ptr
        params = capi.DeinterlaceParams(algo, skip_spatial_check)
        lib().pl_shader_deinterlace(self.sh, C.byref(src), C.byref(params))

    def distort(self, tex, out_w, out_h, params):
        self._keep.append(params)
        lib().pl_shader_distort(self.sh, tex.ptr, out_w, out_h, C.byref(params))

    def color_map(self, src, dst, state_obj=None, params=None, prelinearized=False,
                  feature_map=None):
        params = params or color_map_params()
        args = capi.ColorMapArgs(src=src, dst=dst, prelinearized=prelinearized,
                                 state=C.pointer(state_obj.slot) if state_obj else None,
                                 feature_map=feature_map.ptr if feature_map else None)
        self._keep.append((params, args))
        lib().pl_shader_color_map_ex(self.sh, C.byref(params), C.byref(args))

    def listing(self):
        res = lib().pl_shader_finalize(self.sh)
        return res.contents.glsl.decode() if res else None

    def failed(self):
        return lib().pl_shader_is_failed(self.sh)

    def finish(self, target, rect=None, timer=None, blend_params=None):
        dp = capi.DispatchParams(shader=C.pointer(self.sh), target=target.ptr, timer=timer)
        if rect is not None:
            dp.rect = capi.Rect2d(*rect)
        if blend_params is not None:
            dp.blend_params = C.pointer(blend_params)
        return lib().pl_dispatch_finish(self.gpu.dp, C.byref(dp))

    def compute(self, width=0, height=0, timer=None):
        cp = capi.DispatchComputeParams(shader=C.pointer(self.sh), width=width, height=height,
                                        timer=timer)
        return lib().pl_dispatch_compute(self.gpu.dp, C.byref(cp))

    def abort(self):
        lib().pl_dispatch_abort(self.gpu.dp, C.byref(self.sh))


class HipGpu:
    """pl_hip backend + a pl_dispatch, as a context manager."""

    def __init__(self, device=0, stream=None, log_level=3, max_shmem_size=0, async_measure=None):
        """async_measure: None = the library's default (pl_hip_default_params: on)"""
        L = lib()
        self._msgs = []

        def _cb(_priv, level, msg):
            self._msgs.append((level, msg.decode(errors="replace")))

        self._cb = capi.LOG_CB(_cb)
        lp = capi.LogParams(log_cb=self._cb, log_priv=None, log_level=log_level)
        self.log = C.c_void_p(L.pl_log_create_365(365, C.byref(lp)))
        if async_measure is None:
            async_measure = capi.HipParams.in_dll(L, "pl_hip_default_params").async_measure
        hp = capi.HipParams(device=device, stream=stream, max_shmem_size=max_shmem_size,
                            async_measure=bool(async_measure))
        self.hip = L.pl_hip_create(self.log, C.byref(hp))
        if not self.hip:
            raise RuntimeError("pl_hip_create failed: " + "; ".join(m for _, m in self._msgs))
        self.gpu = self.hip.contents.gpu
        self.dp = C.c_void_p(L.pl_dispatch_create(self.log, self.gpu))

    messages = property(lambda self: list(self._msgs))

    def fmt(self, name):
        f = lib().pl_find_named_fmt(self.gpu, name.encode())
        if not f:
            raise KeyError(name)
        return f

    def tex_create(self, w, h, fmt, data=None):
        dt, nc = _FMT_DTYPES[fmt]
        tp = capi.TexParams(w=w, h=h, format=self.fmt(fmt), sampleable=True, renderable=True,
                            storable=True, blit_src=True, blit_dst=True, host_writable=True,
                            host_readable=True)
        arr = None
        if data is not None:
            arr = np.ascontiguousarray(data, dtype=dt).reshape(h, w, nc)
            tp.initial_data = arr.ctypes.data
        t = lib().pl_tex_create(self.gpu, C.byref(tp))
        if not t:
            raise RuntimeError("pl_tex_create failed: " + "; ".join(m for _, m in self._msgs[-3:]))
        return Texture(self, t)

    def tex_wrap(self, device_ptr, w, h, fmt, row_pitch=0):
        wp = capi.HipWrapParams(ptr=device_ptr, width=w, height=h, row_pitch=row_pitch,
                                format=self.fmt(fmt))
        t = lib().pl_hip_wrap(self.gpu, C.byref(wp))
        if not t:
            raise RuntimeError("pl_hip_wrap failed")
        return Texture(self, t, owned=False)

    def begin(self):
        return Shader(self)

    def reset_frame(self):
        lib().pl_dispatch_reset_frame(self.dp)

    def finish(self):
        lib().pl_gpu_finish(self.gpu)

    def timer(self):
        return C.c_void_p(lib().pl_timer_create(self.gpu))

    def timer_query(self, t):
        return lib().pl_timer_query(self.gpu, t)

    def close(self):
        L = lib()
        if self.dp:
            L.pl_dispatch_destroy(C.byref(self.dp))
        if self.hip:
            h = self.hip
            L.pl_hip_destroy(C.byref(h))
            self.hip = None
        if self.log:
            L.pl_log_destroy(C.byref(self.log))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- renderer.h ----------------------------------------------------------------------------
def frame(tex, repr_=None, color=None, crop=None, components=None, mapping=None,
          flipped=False):
    """A single-plane pl_frame around `tex` (a Texture)."""
    f = capi.Frame(num_planes=1)
    pl_ = f.planes[0]
    pl_.texture = tex.ptr
    comps = components or tex.ptr.contents.params.format.contents.num_components
    pl_.components = comps
    pl_.flipped = flipped
    m = list(mapping) if mapping is not None else list(range(comps))
    for c in range(4):
        pl_.component_mapping[c] = m[c] if c < len(m) else -1
    f.repr = repr_ if repr_ is not None else color_repr("rgb", "full")
    f.color = color if color is not None else color_space("bt709", "srgb")
    if crop is not None:
        f.crop = capi.Rect2df(*crop)
    return f


OVERLAY_NORMAL, OVERLAY_MONOCHROME = 0, 1
OVERLAY_COORDS_AUTO, OVERLAY_COORDS_SRC_FRAME, OVERLAY_COORDS_SRC_CROP = 0, 1, 2
OVERLAY_COORDS_DST_FRAME, OVERLAY_COORDS_DST_CROP = 3, 4
BLEND_ZERO, BLEND_ONE, BLEND_SRC_ALPHA, BLEND_ONE_MINUS_SRC_ALPHA = 0, 1, 2, 3


def distort_params(mat=((1, 0), (0, 1)), c=(0, 0), unscaled=False, constrain=False, bicubic=False,
                   address_mode=ADDRESS_CLAMP, alpha_mode=0):
    p = capi.DistortParams(unscaled=unscaled, constrain=constrain, bicubic=bicubic,
                           address_mode=address_mode, alpha_mode=alpha_mode)
    for i in range(2):
        for j in range(2):
            p.transform.m[i][j] = mat[i][j]
        p.transform.c[i] = c[i]
    return p


def overlay(tex, parts, mode=OVERLAY_NORMAL, coords=OVERLAY_COORDS_AUTO, repr_=None, color=None):
    """pl_overlay around `tex`; parts = [(src rect, dst rect, rgba or None), ...]"""
    arr = (capi.OverlayPart * max(len(parts), 1))()
    for i, (src, dst, rgba) in enumerate(parts):
        arr[i].src = capi.Rect2df(*src)
        arr[i].dst = capi.Rect2df(*dst)
        for c in range(4):
            arr[i].color[c] = rgba[c] if rgba is not None else 0.0
    o = capi.Overlay(tex=tex.ptr, mode=mode, coords=coords, parts=arr, num_parts=len(parts))
    o.repr = repr_ if repr_ is not None else color_repr("rgb", "full", alpha="independent")
    o.color = color if color is not None else color_space("bt709", "srgb")
    o._keep = arr
    return o


def set_overlays(frame_, overlays):
    """attach a list of pl_overlay to a pl_frame (kept alive on the frame)"""
    arr = (capi.Overlay * max(len(overlays), 1))(*overlays)
    frame_.overlays = C.cast(arr, C.c_void_p)
    frame_.num_overlays = len(overlays)
    frame_._keep_overlays = (arr, list(overlays))
    return frame_


FMT_UNORM, FMT_SNORM, FMT_UINT, FMT_SINT, FMT_FLOAT = 1, 2, 3, 4, 5


def plane_data(arr, comp_bits, comp_map=None, comp_pad=None, fmt_type=FMT_UNORM, swapped=False,
               row_stride=0, pixel_stride=0):
    """pl_plane_data (utils/upload.h) describing the numpy array `arr` (H x W x bytes...).
    The array is kept alive by the returned struct (`_keep`)."""
    arr = np.ascontiguousarray(arr)
    d = capi.PlaneData(type=fmt_type, width=arr.shape[1], height=arr.shape[0], swapped=swapped)
    n = len(comp_bits)
    for c in range(4):
        d.component_size[c] = comp_bits[c] if c < n else 0
        d.component_pad[c] = comp_pad[c] if comp_pad and c < n else 0
        d.component_map[c] = (comp_map[c] if comp_map else c) if c < n else 0
    d.pixel_stride = pixel_stride or arr.strides[1]
    d.row_stride = row_stride
    d.pixels = arr.ctypes.data
    d._keep = arr
    return d


def upload_plane(gpu, data, tex=None):
    """pl_upload_plane: returns (capi.Plane, Texture); `tex` is reused when compatible."""
    t = tex.ptr if tex is not None else C.POINTER(capi.Tex)()
    out = capi.Plane()
    if not lib().pl_upload_plane(gpu.gpu, C.byref(out), C.byref(t), C.byref(data)):
        raise RuntimeError("pl_upload_plane failed")
    if tex is not None:
        tex.ptr = t
        return out, tex
    return out, Texture(gpu, t)


def recreate_plane(gpu, data, tex=None):
    """pl_recreate_plane: a renderable plane texture matching `data` (no upload)."""
    t = tex.ptr if tex is not None else C.POINTER(capi.Tex)()
    out = capi.Plane()
    if not lib().pl_recreate_plane(gpu.gpu, C.byref(out), C.byref(t), C.byref(data)):
        raise RuntimeError("pl_recreate_plane failed")
    if tex is not None:
        tex.ptr = t
        return out, tex
    return out, Texture(gpu, t)


CONE_L, CONE_M, CONE_S = 1, 2, 4
VISION = ("normal", "protanomaly", "protanopia", "deuteranomaly", "deuteranopia", "tritanomaly",
          "tritanopia", "monochromacy", "achromatopsia")


def cone_params(cones_or_preset, strength=0.0):
    """pl_cone_params: a preset name from VISION (pl_vision_*) or (cones mask, strength)."""
    if isinstance(cones_or_preset, str):
        src = capi.ConeParams.in_dll(lib(), f"pl_vision_{cones_or_preset}")
        return capi.ConeParams(src.cones, src.strength)
    return capi.ConeParams(int(cones_or_preset), float(strength))


LUT_UNKNOWN, LUT_NATIVE, LUT_NORMALIZED, LUT_CONVERSION = 0, 1, 2, 3


def custom_lut(data, size, shaper_in=None, shaper_out=None, signature=None):
    """pl_custom_lut around a float32 array of RGB triples (R innermost); size = (n,) or
    (r, g, b). The array is kept alive on the returned struct."""
    arr = np.ascontiguousarray(data, np.float32).reshape(-1, 3)
    size = tuple(size) + (0,) * (3 - len(size))
    lut = capi.CustomLut(size=(C.c_int * 3)(*size),
                         data=arr.ctypes.data_as(C.POINTER(C.c_float)),
                         signature=signature if signature is not None
                         else hash(arr.tobytes()) & 0xffffffffffffffff)
    for name, m in (("shaper_in", shaper_in), ("shaper_out", shaper_out)):
        if m is not None:
            mm = capi.Matrix3x3()
            for i in range(3):
                for j in range(3):
                    mm.m[i][j] = m[i][j]
            setattr(lut, name, mm)
    lut._keep = arr
    return lut


def parse_cube(text, log=None):
    """pl_lut_parse_cube: returns a pointer (free with lib().pl_lut_free) or None."""
    raw = text.encode() if isinstance(text, str) else text
    p = lib().pl_lut_parse_cube(log, raw, len(raw))
    return p if p else None


def frame_mix(frames, signatures, timestamps, vsync_duration):
    """pl_frame_mix around parallel python lists (kept alive on the returned struct)."""
    n = len(frames)
    ptrs = (C.POINTER(capi.Frame) * max(n, 1))(*[C.pointer(f) for f in frames])
    sigs = (C.c_uint64 * max(n, 1))(*signatures)
    ts = (C.c_float * max(n, 1))(*timestamps)
    m = capi.FrameMix(num_frames=n, frames=ptrs, signatures=sigs, timestamps=ts,
                      vsync_duration=vsync_duration)
    m._keep = (ptrs, sigs, ts, list(frames))
    return m


QUEUE_OK, QUEUE_EOF, QUEUE_MORE, QUEUE_ERR = 0, 1, 2, -1


class Queue:
    """pl_queue (utils/frame_queue.h): feed it (pl_frame, pts) pairs, ask it for the pl_frame_mix
    of a vsync. Frames are handed over as ready pl_frame structs ("pass-through" map)."""

    def __init__(self, gpu):
        self.gpu = gpu
        self.q = C.c_void_p(lib().pl_queue_create(gpu.gpu))
        assert self.q
        self.frames = {}       # frame_data id -> Frame
        self.unmapped = []     # ids the queue is done with
        self.uploads = []      # per uploaded plane: did the queue hand in a recycled texture?
        self._next = 1
        self._map = capi.QUEUE_MAP_FN(self._on_map)
        self._unmap = capi.QUEUE_UNMAP_FN(self._on_unmap)
        self._discard = capi.QUEUE_DISCARD_FN(self._on_discard)

    def _on_map(self, gpu, tex, src, out):
        item = self.frames[src.contents.frame_data]
        if isinstance(item, capi.Frame):
            C.memmove(out, C.byref(item), C.sizeof(capi.Frame))
            return True
        # host picture: (list of pl_plane_data, repr, color) uploaded into the queue's own
        # texture slots, which it recycles from frame to frame (what a software decoder does)
        planes, repr_, color = item
        f = capi.Frame(num_planes=len(planes), repr=repr_, color=color)
        slots = C.cast(tex, C.POINTER(C.POINTER(capi.Tex)))
        for i, data in enumerate(planes):
            slot = C.cast(C.addressof(slots.contents) + i * C.sizeof(C.c_void_p),
                          C.POINTER(C.POINTER(capi.Tex)))
            self.uploads.append(bool(slot.contents))      # True = a recycled texture was handed in
            if not lib().pl_upload_plane(self.gpu.gpu, C.byref(f.planes[i]), slot, C.byref(data)):
                return False
        C.memmove(out, C.byref(f), C.sizeof(capi.Frame))
        return True

    def _on_unmap(self, gpu, frame, src):
        self.unmapped.append(src.contents.frame_data)

    def _on_discard(self, src):
        self.unmapped.append(src.contents.frame_data)

    def push(self, frame, pts, duration=0.0, first_field=0, block_ns=None):
        """pl_queue_push / pl_queue_push_block; frame None = EOF. Returns the frame's id."""
        if frame is None:
            lib().pl_queue_push(self.q, None)
            return None
        ident, self._next = self._next, self._next + 1
        self.frames[ident] = frame
        src = capi.SourceFrame(pts=pts, duration=duration, first_field=first_field,
                               frame_data=ident, map=self._map, unmap=self._unmap,
                               discard=self._discard)
        if block_ns is None:
            lib().pl_queue_push(self.q, C.byref(src))
        elif not lib().pl_queue_push_block(self.q, block_ns, C.byref(src)):
            del self.frames[ident]
            return None
        return ident

    def update(self, pts, radius=0.0, vsync_duration=0.0, drift_compensation=1e-3,
               interpolation_threshold=1e-6, timeout=0):
        """pl_queue_update -> (status, pl_frame_mix); the mix is valid until the next call."""
        p = capi.QueueParams(pts=pts, radius=radius, vsync_duration=vsync_duration,
                             drift_compensation=drift_compensation,
                             interpolation_threshold=interpolation_threshold, timeout=timeout)
        mix = capi.FrameMix()
        return lib().pl_queue_update(self.q, C.byref(mix), C.byref(p)), mix

    def reset(self):
        lib().pl_queue_reset(self.q)

    def num_frames(self):
        return lib().pl_queue_num_frames(self.q)

    def estimate_fps(self):
        return lib().pl_queue_estimate_fps(self.q)

    def estimate_vps(self):
        return lib().pl_queue_estimate_vps(self.q)

    def destroy(self):
        lib().pl_queue_destroy(C.byref(self.q))


def frame_mix_radius(params):
    """pl_frame_mix_radius (a static inline of renderer.h): the mixer kernel's radius, 0 for
    oversampling / no mixer"""
    if not params.frame_mixer or not params.frame_mixer.contents.kernel:
        return 0.0
    return params.frame_mixer.contents.kernel.contents.radius


def render_params(preset="fast", **kw):
    """pl_render_{fast,default,high_quality}_params with overrides; pointer fields accept
    ctypes structs (kept alive on the returned object)."""
    src = capi.RenderParams.in_dll(lib(), f"pl_render_{preset}_params")
    p = capi.RenderParams()
    C.memmove(C.byref(p), C.byref(src), C.sizeof(p))
    p._keep = []
    for k, v in kw.items():
        if isinstance(v, C.Structure):
            p._keep.append(v)
            v = C.pointer(v)
        elif v is None:
            ftype = dict(capi.RenderParams._fields_)[k]
            v = ftype()
        setattr(p, k, v)
    return p


class Renderer:
    def __init__(self, gpu):
        self.gpu = gpu
        self.rr = C.c_void_p(lib().pl_renderer_create(gpu.log, gpu.gpu))
        assert self.rr

    def render(self, image, target, params=None):
        return lib().pl_render_image(self.rr, C.byref(image), C.byref(target),
                                     C.byref(params) if params is not None else None)

    def render_mix(self, frames, signatures, timestamps, vsync_duration, target, params=None):
        """pl_render_image_mix over parallel lists of frames / signatures / timestamps."""
        return lib().pl_render_image_mix(self.rr, C.byref(frame_mix(frames, signatures, timestamps,
                                                                   vsync_duration)),
                                         C.byref(target),
                                         C.byref(params) if params is not None else None)

    def errors(self):
        return lib().pl_renderer_get_errors(self.rr).errors

    def destroy(self):
        if self.rr:
            lib().pl_renderer_destroy(C.byref(self.rr))
