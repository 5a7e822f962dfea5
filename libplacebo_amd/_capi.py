"""ctypes declarations for the C-ABI of libplacebo_hip.so.

One Python class per public struct of include/libplacebo/*.h (same field
order), and argtypes/restype for every entry point the harness uses. This is
plumbing only: all arithmetic happens in the C/HIP library.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PL_HIP_LIB: load another build of the library (A/B kernel experiments, tools/ab.sh)
_DEFAULT_LIB_PATH = os.path.join(_HERE, "libplacebo_hip.so")
LIB_PATH = os.environ.get("PL_HIP_LIB") or _DEFAULT_LIB_PATH


class BuildError(RuntimeError):
    pass


def load():
    if not os.path.exists(LIB_PATH):
        raise BuildError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C libplacebo_amd/csrc`). There is no Python/CPU fallback.")
    _warn_if_stale()
    return C.CDLL(LIB_PATH)  # RTLD_LOCAL: never interpose on the checker libraries


def _warn_if_stale():
    """A library older than its sources measures (and tests) the previous kernels: say so."""
    import glob
    import sys
    if LIB_PATH != _DEFAULT_LIB_PATH:
        return
    here = os.path.dirname(os.path.abspath(__file__))
    srcs = [f for pat in ("csrc/host/*.[ch]", "csrc/hip/*.hip", "csrc/hip/*.hiph", "csrc/hip/*.h")
            for f in glob.glob(os.path.join(here, pat))]
    try:
        built = os.path.getmtime(LIB_PATH)
        newer = [os.path.basename(f) for f in srcs if os.path.getmtime(f) > built + 1.0]
    except OSError:
        return
    if newer:
        print(f"libplacebo_amd: WARNING: {os.path.basename(LIB_PATH)} is older than "
              f"{', '.join(sorted(newer)[:4])}{' ...' if len(newer) > 4 else ''}: rebuild "
              "(make -C libplacebo_amd/csrc) before measuring", file=sys.stderr)


# ---- common.h ---------------------------------------------------------------
class Rect2d(C.Structure):
    _fields_ = [("x0", C.c_int), ("y0", C.c_int), ("x1", C.c_int), ("y1", C.c_int)]


class Rect2df(C.Structure):
    _fields_ = [("x0", C.c_float), ("y0", C.c_float), ("x1", C.c_float), ("y1", C.c_float)]


class Rect3d(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("x0", "y0", "z0", "x1", "y1", "z1")]


class Matrix3x3(C.Structure):
    _fields_ = [("m", (C.c_float * 3) * 3)]


class Transform3x3(C.Structure):
    _fields_ = [("mat", Matrix3x3), ("c", C.c_float * 3)]


# ---- log.h --------------------------------------------------------------------
LOG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_char_p)


class LogParams(C.Structure):
    _fields_ = [("log_cb", LOG_CB), ("log_priv", C.c_void_p), ("log_level", C.c_int)]


# ---- filters.h ------------------------------------------------------------------
class FilterFunction(C.Structure):
    _fields_ = [("name", C.c_char_p), ("radius", C.c_float), ("resizable", C.c_bool),
                ("tunable", C.c_bool * 2), ("params", C.c_float * 2),
                ("weight", C.c_void_p), ("opaque", C.c_bool)]


class FilterConfig(C.Structure):
    _fields_ = [("name", C.c_char_p), ("description", C.c_char_p),
                ("allowed", C.c_int), ("recommended", C.c_int),
                ("kernel", C.POINTER(FilterFunction)), ("window", C.POINTER(FilterFunction)),
                ("radius", C.c_float), ("params", C.c_float * 2), ("wparams", C.c_float * 2),
                ("clamp", C.c_float), ("blur", C.c_float), ("taper", C.c_float),
                ("polar", C.c_bool), ("antiring", C.c_float)]


class FilterParams(C.Structure):
    _fields_ = [("config", FilterConfig), ("lut_entries", C.c_int), ("cutoff", C.c_float),
                ("max_row_size", C.c_int), ("row_stride_align", C.c_int),
                ("filter_scale", C.c_float)]


class Filter(C.Structure):
    _fields_ = [("params", FilterParams), ("radius", C.c_float), ("radius_zero", C.c_float),
                ("weights", C.POINTER(C.c_float)), ("row_size", C.c_int),
                ("insufficient", C.c_bool), ("row_stride", C.c_int),
                ("radius_cutoff", C.c_float)]


# ---- gpu.h / hip.h ----------------------------------------------------------------
class FmtPlane(C.Structure):
    _fields_ = [("format", C.c_void_p), ("shift_x", C.c_uint8), ("shift_y", C.c_uint8)]


class Fmt(C.Structure):
    _fields_ = [("name", C.c_char_p), ("signature", C.c_uint64), ("type", C.c_int),
                ("caps", C.c_int), ("num_components", C.c_int),
                ("component_depth", C.c_int * 4), ("internal_size", C.c_size_t),
                ("planes", FmtPlane * 4), ("num_planes", C.c_int),
                ("opaque", C.c_bool), ("emulated", C.c_bool), ("texel_size", C.c_size_t),
                ("texel_align", C.c_size_t), ("host_bits", C.c_int * 4),
                ("sample_order", C.c_int * 4), ("gatherable", C.c_bool),
                ("glsl_type", C.c_char_p), ("glsl_format", C.c_char_p),
                ("fourcc", C.c_uint32), ("modifiers", C.c_void_p), ("num_modifiers", C.c_int)]


class GlslVersion(C.Structure):
    _fields_ = [("version", C.c_int), ("gles", C.c_bool), ("vulkan", C.c_bool),
                ("compute", C.c_bool), ("max_shmem_size", C.c_size_t),
                ("max_group_threads", C.c_uint32), ("max_group_size", C.c_uint32 * 3),
                ("subgroup_size", C.c_uint32), ("min_gather_offset", C.c_int16),
                ("max_gather_offset", C.c_int16)]


class GpuLimits(C.Structure):
    _fields_ = [("thread_safe", C.c_bool), ("callbacks", C.c_bool),
                ("max_buf_size", C.c_size_t), ("max_ubo_size", C.c_size_t),
                ("max_ssbo_size", C.c_size_t), ("max_vbo_size", C.c_size_t),
                ("max_mapped_size", C.c_size_t), ("max_buffer_texels", C.c_uint64),
                ("host_cached", C.c_bool), ("host_ptr_slow", C.c_bool),
                ("max_mapped_vram", C.c_size_t), ("align_host_ptr", C.c_size_t),
                ("max_tex_1d_dim", C.c_uint32), ("max_tex_2d_dim", C.c_uint32),
                ("max_tex_3d_dim", C.c_uint32), ("blittable_1d_3d", C.c_bool),
                ("buf_transfer", C.c_bool), ("align_tex_xfer_pitch", C.c_size_t),
                ("align_tex_xfer_offset", C.c_size_t), ("max_variable_comps", C.c_size_t),
                ("max_constants", C.c_size_t), ("array_size_constants", C.c_bool),
                ("max_pushc_size", C.c_size_t), ("align_vertex_stride", C.c_size_t),
                ("max_dispatch", C.c_uint32 * 3),
                ("fragment_queues", C.c_uint32), ("compute_queues", C.c_uint32)]


class PciAddress(C.Structure):
    _fields_ = [("domain", C.c_uint32), ("bus", C.c_uint32), ("device", C.c_uint32),
                ("function", C.c_uint32)]


class HandleCaps(C.Structure):
    _fields_ = [("tex", C.c_uint64), ("buf", C.c_uint64), ("sync", C.c_uint64)]


class SharedMem(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("size", C.c_size_t), ("offset", C.c_size_t),
                ("drm_format_mod", C.c_uint64), ("stride_w", C.c_size_t),
                ("stride_h", C.c_size_t), ("plane", C.c_uint)]


class Gpu(C.Structure):
    _fields_ = [("log", C.c_void_p), ("glsl", GlslVersion), ("limits", GpuLimits),
                ("export_caps", HandleCaps), ("import_caps", HandleCaps),
                ("uuid", C.c_uint8 * 16), ("formats", C.POINTER(C.POINTER(Fmt))),
                ("num_formats", C.c_int), ("pci", PciAddress)]


class Hip(C.Structure):
    _fields_ = [("gpu", C.POINTER(Gpu)), ("device", C.c_int), ("stream", C.c_void_p),
                ("arch", C.c_char_p), ("compute_units", C.c_int)]


class HipParams(C.Structure):
    _fields_ = [("device", C.c_int), ("stream", C.c_void_p), ("max_shmem_size", C.c_size_t),
                ("async_measure", C.c_bool)]


class HipWrapParams(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_int), ("height", C.c_int),
                ("row_pitch", C.c_size_t), ("format", C.POINTER(Fmt))]


class TexParams(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("d", C.c_int), ("format", C.POINTER(Fmt)),
                ("sampleable", C.c_bool), ("renderable", C.c_bool), ("storable", C.c_bool),
                ("blit_src", C.c_bool), ("blit_dst", C.c_bool), ("host_writable", C.c_bool),
                ("host_readable", C.c_bool), ("export_handle", C.c_int),
                ("import_handle", C.c_int), ("shared_mem", SharedMem),
                ("initial_data", C.c_void_p),
                ("user_data", C.c_void_p), ("debug_tag", C.c_char_p)]


class Tex(C.Structure):
    _fields_ = [("params", TexParams), ("planes", C.c_void_p * 4), ("parent", C.c_void_p),
                ("shared_mem", SharedMem), ("sampler_type", C.c_int)]


class TexTransferParams(C.Structure):
    _fields_ = [("tex", C.POINTER(Tex)), ("rc", Rect3d), ("row_pitch", C.c_size_t),
                ("depth_pitch", C.c_size_t), ("timer", C.c_void_p), ("callback", C.c_void_p),
                ("priv", C.c_void_p), ("buf", C.c_void_p), ("buf_offset", C.c_size_t),
                ("ptr", C.c_void_p), ("no_import", C.c_bool)]


class TexBlitParams(C.Structure):
    _fields_ = [("src", C.POINTER(Tex)), ("dst", C.POINTER(Tex)), ("src_rc", Rect3d),
                ("dst_rc", Rect3d), ("sample_mode", C.c_int)]


# ---- shaders ----------------------------------------------------------------------
class ShaderParams(C.Structure):
    _fields_ = [("id", C.c_uint8), ("gpu", C.c_void_p), ("index", C.c_uint8),
                ("glsl", GlslVersion), ("dynamic_constants", C.c_bool)]


class ShaderInfo(C.Structure):
    _fields_ = [("params", ShaderParams), ("steps", C.POINTER(C.c_char_p)),
                ("num_steps", C.c_int), ("description", C.c_char_p)]


class ShaderRes(C.Structure):
    _fields_ = [("info", C.POINTER(ShaderInfo)), ("glsl", C.c_char_p), ("name", C.c_char_p),
                ("input", C.c_int), ("output", C.c_int), ("compute_group_size", C.c_int * 2),
                ("compute_shmem", C.c_size_t),
                ("vertex_attribs", C.c_void_p), ("num_vertex_attribs", C.c_int),
                ("variables", C.c_void_p), ("num_variables", C.c_int),
                ("descriptors", C.c_void_p), ("num_descriptors", C.c_int),
                ("constants", C.c_void_p), ("num_constants", C.c_int),
                ("params", ShaderParams), ("steps", C.POINTER(C.c_char_p)),
                ("num_steps", C.c_int), ("description", C.c_char_p)]


class SampleSrc(C.Structure):
    _fields_ = [("tex", C.POINTER(Tex)), ("rect", Rect2df), ("address_mode", C.c_int),
                ("tex_w", C.c_int), ("tex_h", C.c_int), ("format", C.c_int),
                ("sampler", C.c_int), ("mode", C.c_int), ("sampled_w", C.c_float),
                ("sampled_h", C.c_float), ("components", C.c_int),
                ("component_mask", C.c_uint8), ("new_w", C.c_int), ("new_h", C.c_int),
                ("scale", C.c_float)]


class DebandParams(C.Structure):
    _fields_ = [("iterations", C.c_int), ("threshold", C.c_float), ("radius", C.c_float),
                ("grain", C.c_float), ("grain_neutral", C.c_float * 3)]


class SampleFilterParams(C.Structure):
    _fields_ = [("filter", FilterConfig), ("antiring", C.c_float), ("no_compute", C.c_bool),
                ("no_widening", C.c_bool), ("lut", C.POINTER(C.c_void_p)),
                ("lut_entries", C.c_int), ("cutoff", C.c_float)]


class DitherParams(C.Structure):
    _fields_ = [("method", C.c_int), ("lut_size", C.c_int), ("temporal", C.c_bool),
                ("transfer", C.c_int)]


class ErrorDiffusionKernel(C.Structure):
    _fields_ = [("name", C.c_char_p), ("description", C.c_char_p), ("shift", C.c_int),
                ("pattern", (C.c_int * 5) * 3), ("divisor", C.c_int)]


class ErrorDiffusionParams(C.Structure):
    _fields_ = [("input_tex", C.c_void_p), ("output_tex", C.c_void_p), ("new_depth", C.c_int),
                ("kernel", C.POINTER(ErrorDiffusionKernel))]


class DispatchComputeParams(C.Structure):
    _fields_ = [("shader", C.POINTER(C.c_void_p)), ("dispatch_size", C.c_int * 3),
                ("width", C.c_int), ("height", C.c_int), ("timer", C.c_void_p)]


class BlendParams(C.Structure):
    _fields_ = [("src_rgb", C.c_int), ("dst_rgb", C.c_int),
                ("src_alpha", C.c_int), ("dst_alpha", C.c_int)]


class DispatchParams(C.Structure):
    _fields_ = [("shader", C.POINTER(C.c_void_p)), ("target", C.POINTER(Tex)),
                ("rect", Rect2d), ("blend_params", C.POINTER(BlendParams)), ("timer", C.c_void_p)]


# ---- colorspace.h / tone_mapping.h / gamut_mapping.h --------------------------------
class CieXy(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float)]


class ConeParams(C.Structure):
    _fields_ = [("cones", C.c_int), ("strength", C.c_float)]


class RawPrimaries(C.Structure):
    _fields_ = [("red", CieXy), ("green", CieXy), ("blue", CieXy), ("white", CieXy)]


class HdrBezier(C.Structure):
    _fields_ = [("target_luma", C.c_float), ("knee_x", C.c_float), ("knee_y", C.c_float),
                ("anchors", C.c_float * 15), ("num_anchors", C.c_uint8)]


class HdrMetadata(C.Structure):
    _fields_ = [("prim", RawPrimaries), ("min_luma", C.c_float), ("max_luma", C.c_float),
                ("max_cll", C.c_float), ("max_fall", C.c_float), ("scene_max", C.c_float * 3),
                ("scene_avg", C.c_float), ("ootf", HdrBezier), ("max_pq_y", C.c_float),
                ("avg_pq_y", C.c_float)]


class ColorSpace(C.Structure):
    _fields_ = [("primaries", C.c_int), ("transfer", C.c_int), ("hdr", HdrMetadata)]


class BitEncoding(C.Structure):
    _fields_ = [("sample_depth", C.c_int), ("color_depth", C.c_int), ("bit_shift", C.c_int)]


class ReshapeData(C.Structure):
    _fields_ = [("num_pivots", C.c_uint8), ("pivots", C.c_float * 9), ("method", C.c_uint8 * 8),
                ("poly_coeffs", (C.c_float * 3) * 8), ("mmr_order", C.c_uint8 * 8),
                ("mmr_constant", C.c_float * 8), ("mmr_coeffs", ((C.c_float * 7) * 3) * 8)]


class DoviMetadata(C.Structure):
    _fields_ = [("nonlinear_offset", C.c_float * 3), ("nonlinear", (C.c_float * 3) * 3),
                ("linear", (C.c_float * 3) * 3), ("comp", ReshapeData * 3)]


class ColorRepr(C.Structure):
    _fields_ = [("sys", C.c_int), ("levels", C.c_int), ("alpha", C.c_int),
                ("bits", BitEncoding), ("dovi", C.c_void_p)]


class ColorAdjustment(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("brightness", "contrast", "saturation", "hue", "gamma",
                                         "temperature")]


class ToneMapConstants(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "knee_adaptation", "knee_minimum", "knee_maximum", "knee_default", "knee_offset",
        "slope_tuning", "slope_offset", "spline_contrast", "reinhard_contrast", "linear_knee",
        "exposure")]


TONE_MAP_CONSTANTS = (0.4, 0.1, 0.8, 0.4, 1.0, 1.5, 0.2, 0.5, 0.5, 0.3, 1.0)


class GamutMapConstants(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("perceptual_deadzone", "perceptual_strength",
                                         "colorimetric_gamma", "softclip_knee", "softclip_desat")]


GAMUT_MAP_CONSTANTS = (0.30, 0.80, 1.80, 0.70, 0.35)


class SigmoidParams(C.Structure):
    _fields_ = [("center", C.c_float), ("slope", C.c_float)]


class PeakDetectParams(C.Structure):
    _fields_ = [("smoothing_period", C.c_float), ("scene_threshold_low", C.c_float),
                ("scene_threshold_high", C.c_float), ("percentile", C.c_float),
                ("black_cutoff", C.c_float), ("allow_delayed", C.c_bool),
                ("minimum_peak", C.c_float)]


class ColorMapParams(C.Structure):
    _fields_ = [("gamut_mapping", C.c_void_p), ("gamut_constants", GamutMapConstants),
                ("lut3d_size", C.c_int * 3), ("lut3d_tricubic", C.c_bool),
                ("gamut_expansion", C.c_bool), ("tone_mapping_function", C.c_void_p),
                ("tone_constants", ToneMapConstants), ("inverse_tone_mapping", C.c_bool),
                ("metadata", C.c_int), ("lut_size", C.c_int), ("contrast_recovery", C.c_float),
                ("contrast_smoothness", C.c_float), ("force_tone_mapping_lut", C.c_bool),
                ("visualize_lut", C.c_bool), ("visualize_rect", Rect2df),
                ("visualize_hue", C.c_float), ("visualize_theta", C.c_float),
                ("show_clipping", C.c_bool), ("tone_mapping_mode", C.c_int),
                ("tone_mapping_param", C.c_float), ("tone_mapping_crosstalk", C.c_float),
                ("intent", C.c_int), ("gamut_mode", C.c_int), ("hybrid_mix", C.c_float)]


class ColorMapArgs(C.Structure):
    _fields_ = [("src", ColorSpace), ("dst", ColorSpace), ("prelinearized", C.c_bool),
                ("state", C.POINTER(C.c_void_p)), ("feature_map", C.POINTER(Tex))]


# ---- renderer.h ----------------------------------------------------------------------------
class Plane(C.Structure):
    _fields_ = [("texture", C.POINTER(Tex)), ("address_mode", C.c_int), ("flipped", C.c_bool),
                ("components", C.c_int), ("component_mapping", C.c_int * 4),
                ("shift_x", C.c_float), ("shift_y", C.c_float)]


class PlaneData(C.Structure):  # utils/upload.h
    _fields_ = [("type", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("component_size", C.c_int * 4), ("component_pad", C.c_int * 4),
                ("component_map", C.c_int * 4), ("pixel_stride", C.c_size_t),
                ("row_stride", C.c_size_t), ("swapped", C.c_bool), ("pixels", C.c_void_p),
                ("buf", C.c_void_p), ("buf_offset", C.c_size_t), ("callback", C.c_void_p),
                ("priv", C.c_void_p)]


class CustomLut(C.Structure):  # shaders/lut.h
    _fields_ = [("signature", C.c_uint64), ("size", C.c_int * 3), ("data", C.POINTER(C.c_float)),
                ("shaper_in", Matrix3x3), ("shaper_out", Matrix3x3),
                ("repr_in", ColorRepr), ("repr_out", ColorRepr),
                ("color_in", ColorSpace), ("color_out", ColorSpace)]


class IccProfile(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("signature", C.c_uint64)]


class Av1GrainData(C.Structure):
    _fields_ = [("num_points_y", C.c_int), ("points_y", (C.c_uint8 * 2) * 14),
                ("chroma_scaling_from_luma", C.c_bool), ("num_points_uv", C.c_int * 2),
                ("points_uv", ((C.c_uint8 * 2) * 10) * 2), ("scaling_shift", C.c_int),
                ("ar_coeff_lag", C.c_int), ("ar_coeffs_y", C.c_int8 * 24),
                ("ar_coeffs_uv", (C.c_int8 * 25) * 2), ("ar_coeff_shift", C.c_int),
                ("grain_scale_shift", C.c_int), ("uv_mult", C.c_int8 * 2),
                ("uv_mult_luma", C.c_int8 * 2), ("uv_offset", C.c_int16 * 2),
                ("overlap", C.c_bool)]


class H274GrainData(C.Structure):
    _fields_ = [("model_id", C.c_int), ("blending_mode_id", C.c_int),
                ("log2_scale_factor", C.c_int), ("component_model_present", C.c_bool * 3),
                ("num_intensity_intervals", C.c_uint16 * 3), ("num_model_values", C.c_uint8 * 3),
                ("intensity_interval_lower_bound", C.c_void_p * 3),
                ("intensity_interval_upper_bound", C.c_void_p * 3),
                ("comp_model_value", C.c_void_p * 3)]


class _GrainUnion(C.Union):
    _fields_ = [("av1", Av1GrainData), ("h274", H274GrainData)]


class FilmGrainData(C.Structure):
    _fields_ = [("type", C.c_int), ("seed", C.c_uint64), ("params", _GrainUnion)]


class Transform2x2(C.Structure):
    _fields_ = [("m", (C.c_float * 2) * 2), ("c", C.c_float * 2)]


class DistortParams(C.Structure):
    _fields_ = [("transform", Transform2x2), ("unscaled", C.c_bool), ("constrain", C.c_bool),
                ("bicubic", C.c_bool), ("address_mode", C.c_int), ("alpha_mode", C.c_int)]


class DeinterlaceParams(C.Structure):
    _fields_ = [("algo", C.c_int), ("skip_spatial_check", C.c_bool)]


class FieldPair(C.Structure):
    _fields_ = [("top", C.POINTER(Tex))]


class DeinterlaceSource(C.Structure):
    _fields_ = [("prev", FieldPair), ("cur", FieldPair), ("next", FieldPair),
                ("field", C.c_int), ("first_field", C.c_int), ("component_mask", C.c_uint8)]


class OverlayPart(C.Structure):
    _fields_ = [("src", Rect2df), ("dst", Rect2df), ("color", C.c_float * 4)]


class Overlay(C.Structure):
    _fields_ = [("tex", C.POINTER(Tex)), ("mode", C.c_int), ("coords", C.c_int),
                ("repr", ColorRepr), ("color", ColorSpace),
                ("parts", C.POINTER(OverlayPart)), ("num_parts", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("num_planes", C.c_int), ("planes", Plane * 4),
                ("field", C.c_int), ("first_field", C.c_int),
                ("prev", C.c_void_p), ("next", C.c_void_p),
                ("acquire", C.c_void_p), ("release", C.c_void_p),
                ("repr", ColorRepr), ("color", ColorSpace),
                ("icc", C.c_void_p), ("profile", IccProfile),
                ("lut", C.POINTER(CustomLut)), ("lut_type", C.c_int), ("crop", Rect2df),
                ("rotation", C.c_int), ("pixel_aspect_ratio", C.c_float),
                ("overlays", C.c_void_p), ("num_overlays", C.c_int),
                ("film_grain", FilmGrainData), ("user_data", C.c_void_p)]


class FrameMix(C.Structure):
    _fields_ = [("num_frames", C.c_int), ("frames", C.POINTER(C.POINTER(Frame))),
                ("signatures", C.POINTER(C.c_uint64)), ("timestamps", C.POINTER(C.c_float)),
                ("vsync_duration", C.c_float)]


class SourceFrame(C.Structure):
    """struct pl_source_frame (utils/frame_queue.h)"""


QUEUE_MAP_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(SourceFrame),
                           C.POINTER(Frame))
QUEUE_UNMAP_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(Frame), C.POINTER(SourceFrame))
QUEUE_DISCARD_FN = C.CFUNCTYPE(None, C.POINTER(SourceFrame))
SourceFrame._fields_ = [("pts", C.c_double), ("duration", C.c_float), ("first_field", C.c_int),
                        ("frame_data", C.c_void_p), ("map", QUEUE_MAP_FN),
                        ("unmap", QUEUE_UNMAP_FN), ("discard", QUEUE_DISCARD_FN)]


class QueueParams(C.Structure):
    """struct pl_queue_params (utils/frame_queue.h)"""


QUEUE_GET_FN = C.CFUNCTYPE(C.c_int, C.POINTER(SourceFrame), C.POINTER(QueueParams))
QueueParams._fields_ = [("pts", C.c_double), ("radius", C.c_float), ("vsync_duration", C.c_float),
                        ("drift_compensation", C.c_float), ("interpolation_threshold", C.c_float),
                        ("timeout", C.c_uint64), ("get_frame", QUEUE_GET_FN), ("priv", C.c_void_p)]


class RenderParams(C.Structure):
    _fields_ = [("upscaler", C.POINTER(FilterConfig)), ("downscaler", C.POINTER(FilterConfig)),
                ("plane_upscaler", C.POINTER(FilterConfig)),
                ("plane_downscaler", C.POINTER(FilterConfig)),
                ("antiringing_strength", C.c_float), ("frame_mixer", C.POINTER(FilterConfig)),
                ("deband_params", C.POINTER(DebandParams)),
                ("sigmoid_params", C.POINTER(SigmoidParams)),
                ("color_adjustment", C.POINTER(ColorAdjustment)),
                ("peak_detect_params", C.POINTER(PeakDetectParams)),
                ("color_map_params", C.POINTER(ColorMapParams)),
                ("dither_params", C.POINTER(DitherParams)),
                ("error_diffusion", C.POINTER(ErrorDiffusionKernel)),
                ("cone_params", C.POINTER(ConeParams)), ("blend_params", C.POINTER(BlendParams)),
                ("deinterlace_params", C.POINTER(DeinterlaceParams)),
                ("distort_params", C.POINTER(DistortParams)),
                ("hooks", C.c_void_p), ("num_hooks", C.c_int), ("lut", C.POINTER(CustomLut)),
                ("lut_type", C.c_int), ("background", C.c_int), ("border", C.c_int),
                ("background_color", C.c_float * 3), ("background_transparency", C.c_float),
                ("tile_colors", (C.c_float * 3) * 2), ("tile_size", C.c_int),
                ("blur_radius", C.c_float), ("corner_rounding", C.c_float),
                ("skip_anti_aliasing", C.c_bool), ("preserve_mixing_cache", C.c_bool),
                ("skip_caching_single_frame", C.c_bool), ("disable_linear_scaling", C.c_bool),
                ("disable_builtin_scalers", C.c_bool), ("correct_subpixel_offsets", C.c_bool),
                ("force_dither", C.c_bool), ("disable_dither_gamma_correction", C.c_bool),
                ("disable_fbos", C.c_bool), ("force_low_bit_depth_fbos", C.c_bool),
                ("dynamic_constants", C.c_bool), ("info_callback", C.c_void_p),
                ("info_priv", C.c_void_p),
                ("allow_delayed_peak_detect", C.c_bool), ("icc_params", C.c_void_p),
                ("ignore_icc_profiles", C.c_bool), ("lut_entries", C.c_int),
                ("polar_cutoff", C.c_float), ("skip_target_clearing", C.c_bool),
                ("blend_against_tiles", C.c_bool)]


class DispatchInfo(C.Structure):
    _fields_ = [("shader", C.POINTER(ShaderInfo)), ("signature", C.c_uint64),
                ("samples", C.c_uint64 * 256), ("num_samples", C.c_int), ("last", C.c_uint64),
                ("peak", C.c_uint64), ("average", C.c_uint64)]


class RenderInfo(C.Structure):
    _fields_ = [("pass_", C.POINTER(DispatchInfo)), ("stage", C.c_int), ("index", C.c_int),
                ("count", C.c_int)]


RENDER_INFO_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(RenderInfo))


class RenderErrors(C.Structure):
    _fields_ = [("errors", C.c_int), ("disabled_hooks", C.c_void_p),
                ("num_disabled_hooks", C.c_int)]


def declare(lib):
    """Attach argtypes/restypes."""
    P = C.POINTER
    vp = C.c_void_p

    def fn(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)
        return f

    fn("pl_log_create_365", vp, C.c_int, P(LogParams))
    fn("pl_log_destroy", None, P(vp))

    fn("pl_filter_generate", P(Filter), vp, P(FilterParams))
    fn("pl_filter_free", None, P(P(Filter)))
    fn("pl_filter_sample", C.c_double, P(FilterConfig), C.c_double)
    fn("pl_find_filter_config", P(FilterConfig), C.c_char_p, C.c_int)
    fn("pl_filter_radius_bound", C.c_float, P(FilterConfig))
    fn("pl_generate_bayer_matrix", None, vp, C.c_int)
    fn("pl_generate_blue_noise", None, vp, C.c_int)

    fn("pl_hip_device_count", C.c_int)
    fn("pl_hip_create", P(Hip), vp, P(HipParams))
    fn("pl_hip_destroy", None, P(P(Hip)))
    fn("pl_hip_wrap", P(Tex), P(Gpu), P(HipWrapParams))
    fn("pl_hip_tex_ptr", vp, P(Tex), P(C.c_size_t))
    fn("pl_find_named_fmt", P(Fmt), P(Gpu), C.c_char_p)
    fn("pl_find_fmt", P(Fmt), P(Gpu), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
    fn("pl_tex_create", P(Tex), P(Gpu), P(TexParams))
    fn("pl_tex_destroy", None, P(Gpu), P(P(Tex)))
    fn("pl_tex_upload", C.c_bool, P(Gpu), P(TexTransferParams))
    fn("pl_tex_blit", None, P(Gpu), P(TexBlitParams))
    fn("pl_tex_download", C.c_bool, P(Gpu), P(TexTransferParams))
    fn("pl_tex_clear", None, P(Gpu), P(Tex), P(C.c_float))
    fn("pl_gpu_finish", None, P(Gpu))
    fn("pl_timer_create", vp, P(Gpu))
    fn("pl_timer_destroy", None, P(Gpu), P(vp))
    fn("pl_timer_query", C.c_uint64, P(Gpu), vp)

    fn("pl_dispatch_create", vp, vp, P(Gpu))
    fn("pl_dispatch_destroy", None, P(vp))
    fn("pl_dispatch_reset_frame", None, vp)
    fn("pl_dispatch_begin", vp, vp)
    fn("pl_dispatch_finish", C.c_bool, vp, P(DispatchParams))
    fn("pl_dispatch_abort", None, vp, P(vp))
    fn("pl_dispatch_compute", C.c_bool, vp, P(DispatchComputeParams))
    fn("pl_shader_finalize", P(ShaderRes), vp)
    fn("pl_shader_is_failed", C.c_bool, vp)
    fn("pl_shader_obj_destroy", None, P(vp))

    for n in ("direct", "nearest", "bilinear", "bicubic", "hermite", "gaussian"):
        fn(f"pl_shader_sample_{n}", C.c_bool, vp, P(SampleSrc))
    fn("pl_shader_sample_oversample", C.c_bool, vp, P(SampleSrc), C.c_float)
    fn("pl_shader_sample_polar", C.c_bool, vp, P(SampleSrc), P(SampleFilterParams))
    fn("pl_shader_sample_ortho2", C.c_bool, vp, P(SampleSrc), P(SampleFilterParams))
    fn("pl_shader_deband", None, vp, P(SampleSrc), P(DebandParams))
    fn("pl_shader_dither", None, vp, C.c_int, P(vp), P(DitherParams))
    fn("pl_shader_error_diffusion", C.c_bool, vp, P(ErrorDiffusionParams))
    fn("pl_find_error_diffusion_kernel", P(ErrorDiffusionKernel), C.c_char_p)
    fn("pl_error_diffusion_shmem_req", C.c_size_t, P(ErrorDiffusionKernel), C.c_int)

    fn("pl_renderer_create", vp, vp, P(Gpu))
    fn("pl_renderer_destroy", None, P(vp))
    fn("pl_renderer_get_errors", RenderErrors, vp)
    fn("pl_renderer_reset_errors", None, vp, P(RenderErrors))
    fn("pl_render_image", C.c_bool, vp, P(Frame), P(Frame), P(RenderParams))
    fn("pl_frames_infer", None, vp, P(Frame), P(Frame))
    fn("pl_render_image_mix", C.c_bool, vp, P(FrameMix), P(Frame), P(RenderParams))
    fn("pl_queue_create", vp, vp)
    fn("pl_queue_destroy", None, P(vp))
    fn("pl_queue_reset", None, vp)
    fn("pl_queue_push", None, vp, P(SourceFrame))
    fn("pl_queue_push_block", C.c_bool, vp, C.c_uint64, P(SourceFrame))
    fn("pl_queue_update", C.c_int, vp, P(FrameMix), P(QueueParams))
    fn("pl_queue_estimate_fps", C.c_float, vp)
    fn("pl_queue_estimate_vps", C.c_float, vp)
    fn("pl_queue_num_frames", C.c_int, vp)
    fn("pl_queue_pts_offset", C.c_double, vp)
    fn("pl_queue_peek", C.c_bool, vp, C.c_int, P(SourceFrame))
    fn("pl_frames_infer_mix", None, vp, P(FrameMix), P(Frame), P(Frame))
    fn("pl_frame_mix_current", P(Frame), P(FrameMix))
    fn("pl_frame_mix_nearest", P(Frame), P(FrameMix))
    fn("pl_frame_set_chroma_location", None, P(Frame), C.c_int)
    fn("pl_plane_data_from_mask", None, P(PlaneData), P(C.c_uint64))
    fn("pl_plane_data_from_comps", None, P(PlaneData), P(C.c_int), P(C.c_int))
    fn("pl_plane_data_align", C.c_bool, P(PlaneData), P(BitEncoding))
    fn("pl_plane_find_fmt", P(Fmt), P(Gpu), P(C.c_int), P(PlaneData))
    fn("pl_upload_plane", C.c_bool, P(Gpu), P(Plane), P(P(Tex)), P(PlaneData))
    fn("pl_recreate_plane", C.c_bool, P(Gpu), P(Plane), P(P(Tex)), P(PlaneData))
    fn("pl_renderer_get_hdr_metadata", C.c_bool, vp, P(HdrMetadata))
    fn("pl_renderer_flush_cache", None, vp)
    fn("pl_hip_renderer_tone_map_state", vp, vp)
    fn("pl_shader_set_alpha", None, vp, P(ColorRepr), C.c_int)
    fn("pl_shader_decode_color", None, vp, P(ColorRepr), P(ColorAdjustment))
    fn("pl_shader_encode_color", None, vp, P(ColorRepr))
    fn("pl_shader_linearize", None, vp, P(ColorSpace))
    fn("pl_shader_delinearize", None, vp, P(ColorSpace))
    fn("pl_shader_sigmoidize", None, vp, P(SigmoidParams))
    fn("pl_shader_unsigmoidize", None, vp, P(SigmoidParams))
    fn("pl_shader_detect_peak", C.c_bool, vp, ColorSpace, P(vp), P(PeakDetectParams))
    fn("pl_get_detected_hdr_metadata", C.c_bool, vp, P(HdrMetadata))
    fn("pl_reset_detected_peak", None, vp)
    fn("pl_hip_peak_buffer", vp, vp, P(C.c_size_t))
    fn("pl_shader_color_map_ex", None, vp, P(ColorMapParams), P(ColorMapArgs))
    fn("pl_shader_extract_features", None, vp, ColorSpace)
    fn("pl_shader_cone_distort", None, vp, ColorSpace, P(ConeParams))
    fn("pl_get_cone_matrix", Matrix3x3, P(ConeParams), P(RawPrimaries))
    fn("pl_lut_parse_cube", P(CustomLut), vp, C.c_char_p, C.c_size_t)
    fn("pl_lut_free", None, P(P(CustomLut)))
    fn("pl_shader_custom_lut", None, vp, P(CustomLut), P(vp))
    fn("pl_find_tone_map_function", vp, C.c_char_p)
    fn("pl_find_gamut_map_function", vp, C.c_char_p)
    fn("pl_color_space_nominal_luma_ex", None, vp)
    fn("pl_color_space_infer", None, P(ColorSpace))
    fn("pl_color_space_infer_map", None, P(ColorSpace), P(ColorSpace))
    fn("pl_raw_primaries_get", P(RawPrimaries), C.c_int)
    fn("pl_get_rgb2xyz_matrix", Matrix3x3, P(RawPrimaries))
    fn("pl_ipt_rgb2lms", Matrix3x3, P(RawPrimaries))
    fn("pl_ipt_lms2rgb", Matrix3x3, P(RawPrimaries))
    fn("pl_color_repr_decode", Transform3x3, P(ColorRepr), P(ColorAdjustment))
    fn("pl_color_repr_normalize", C.c_float, P(ColorRepr))
    fn("pl_hdr_rescale", C.c_float, C.c_int, C.c_int, C.c_float)
    fn("pl_get_color_mapping_matrix", Matrix3x3, P(RawPrimaries), P(RawPrimaries), C.c_int)
    fn("pl_matrix3x3_apply", None, P(Matrix3x3), P(C.c_float))
    fn("pl_transform3x3_apply", None, P(Transform3x3), P(C.c_float))
    fn("pl_transform3x3_invert", None, P(Transform3x3))
    fn("pl_color_linearize", None, P(ColorSpace), P(C.c_float))
    fn("pl_color_delinearize", None, P(ColorSpace), P(C.c_float))
    fn("pl_frame_clear_rgba", None, P(Gpu), P(Frame), P(C.c_float))
    fn("pl_frame_clear_tiles", None, P(Gpu), P(Frame), vp, C.c_int)
    return lib


# C aggregate -> ctypes mirror; tests/test_abi_layout.py checks each against the layout table
# the C probe prints for include/ (which in turn equals the reference's).
MIRRORS = {
    "struct pl_rect2d": Rect2d, "struct pl_rect2df": Rect2df, "struct pl_rect3d": Rect3d,
    "struct pl_matrix3x3": Matrix3x3, "struct pl_transform3x3": Transform3x3,
    "struct pl_log_params": LogParams, "struct pl_filter_function": FilterFunction,
    "struct pl_filter_config": FilterConfig, "struct pl_filter_params": FilterParams,
    "struct pl_filter_t": Filter, "struct pl_fmt_plane": FmtPlane, "struct pl_fmt_t": Fmt,
    "struct pl_glsl_version": GlslVersion, "struct pl_gpu_limits": GpuLimits,
    "struct pl_gpu_pci_address": PciAddress, "struct pl_gpu_handle_caps": HandleCaps,
    "struct pl_shared_mem": SharedMem, "struct pl_gpu_t": Gpu, "struct pl_hip_t": Hip,
    "struct pl_hip_params": HipParams, "struct pl_hip_wrap_params": HipWrapParams,
    "struct pl_tex_params": TexParams, "struct pl_tex_t": Tex,
    "struct pl_tex_transfer_params": TexTransferParams, "struct pl_shader_params": ShaderParams,
    "struct pl_shader_info_t": ShaderInfo, "struct pl_shader_res": ShaderRes,
    "struct pl_sample_src": SampleSrc, "struct pl_deband_params": DebandParams,
    "struct pl_sample_filter_params": SampleFilterParams, "struct pl_dither_params": DitherParams,
    "struct pl_error_diffusion_kernel": ErrorDiffusionKernel,
    "struct pl_error_diffusion_params": ErrorDiffusionParams,
    "struct pl_dispatch_compute_params": DispatchComputeParams,
    "struct pl_dispatch_params": DispatchParams, "struct pl_cie_xy": CieXy,
    "struct pl_cone_params": ConeParams, "struct pl_raw_primaries": RawPrimaries,
    "struct pl_hdr_bezier": HdrBezier, "struct pl_hdr_metadata": HdrMetadata,
    "struct pl_color_space": ColorSpace, "struct pl_bit_encoding": BitEncoding,
    "struct pl_color_repr": ColorRepr, "struct pl_color_adjustment": ColorAdjustment,
    "struct pl_tone_map_constants": ToneMapConstants,
    "struct pl_gamut_map_constants": GamutMapConstants, "struct pl_sigmoid_params": SigmoidParams,
    "struct pl_peak_detect_params": PeakDetectParams, "struct pl_color_map_params": ColorMapParams,
    "struct pl_color_map_args": ColorMapArgs, "struct pl_plane": Plane,
    "struct pl_plane_data": PlaneData, "struct pl_custom_lut": CustomLut,
    "struct pl_icc_profile": IccProfile, "struct pl_av1_grain_data": Av1GrainData,
    "struct pl_h274_grain_data": H274GrainData, "struct pl_film_grain_data": FilmGrainData,
    "struct pl_frame": Frame, "struct pl_frame_mix": FrameMix,
    "struct pl_source_frame": SourceFrame, "struct pl_queue_params": QueueParams,
    "struct pl_render_params": RenderParams, "struct pl_dispatch_info": DispatchInfo,
    "struct pl_render_info": RenderInfo, "struct pl_render_errors": RenderErrors,
}
