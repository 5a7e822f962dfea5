"""The renderer's planner (libplacebo_amd/csrc/host/render_plan.c) on descriptions alone: no GPU,
no shader recording. For each BASELINE.json configuration the plan must show the pass
structure the reference would build (src/renderer.c), and the rect arithmetic
must follow the reference's rounding rules (fix_refs_and_rects :3068-3159)."""
import ctypes as C

import pytest

import libplacebo_amd as pl
from libplacebo_amd import _capi as capi


@pytest.fixture(scope="module")
def L(built):
    lib = pl.lib()
    lib.plh_test_format.restype = C.POINTER(capi.Fmt)
    lib.plh_test_format.argtypes = [C.c_char_p]
    lib.plh_test_plan.restype = C.c_size_t
    return lib


class FakeTex:
    """a pl_tex_t with nothing but params (what the planner may look at)"""

    def __init__(self, L, w, h, fmt, storable=True):
        self.t = capi.Tex()
        self.t.params.w, self.t.params.h = w, h
        self.t.params.format = L.plh_test_format(fmt.encode())
        assert self.t.params.format, fmt
        self.t.params.sampleable = True
        self.t.params.storable = storable
        self.ptr = C.pointer(self.t)


def plan(L, image, target, params, fbos=True, shmem=160 * 1024):
    buf = C.create_string_buffer(4096)
    L.plh_test_plan(C.byref(image), C.byref(target), C.byref(params) if params else None,
                    C.c_bool(fbos), C.c_size_t(shmem), buf, C.c_size_t(len(buf)))
    return buf.value.decode()


HDR = dict(color=None)


def frames(L, src, dst, src_fmt="rgba16", dst_fmt="rgba16", icsp=None, tcsp=None, trepr=None,
           crop=None, tcrop=None, comps=3):
    s, d = FakeTex(L, *src, src_fmt), FakeTex(L, *dst, dst_fmt)
    image = pl.frame(s, components=comps, color=icsp, crop=crop)
    target = pl.frame(d, color=tcsp, repr_=trepr, crop=tcrop)
    image._keep, target._keep = s, d
    return image, target


def test_cfg2_bilinear_is_one_deferred_pass(L):
    image, target = frames(L, (1920, 1080), (3840, 2160))
    text = plan(L, image, target, pl.render_params("fast"))
    assert "scale: deferred to the output pass (builtin)" in text, text
    assert "output plane 0: store 0,0-3840,2160 dither none/0" in text, text
    assert "peak:" not in text


def test_cfg3_polar_upscale_with_dither(L):
    image, target = frames(L, (1920, 1080), (3840, 2160),
                           trepr=pl.color_repr("rgb", "full", sample_depth=16, color_depth=10,
                                               bit_shift=6))
    params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"),
                              dither_params=capi.DitherParams(method=0, lut_size=6))
    text = plan(L, image, target, params)
    assert "scale: polar up -> 3840x2160" in text, text
    assert "dither ordered/10" in text and "scale 1/1.00096" in text, text


def test_default_preset_is_sigmoidized_two_pass_lanczos(L):
    image, target = frames(L, (1920, 1080), (3840, 2160),
                           icsp=pl.color_space("bt709", "bt1886"),
                           tcsp=pl.color_space("bt709", "bt1886"))
    text = plan(L, image, target, pl.render_params("default"))
    assert "scale: separable up sigmoid two-pass -> 3840x2160" in text, text


def test_cfg4_hdr_peak_measured_without_scaling(L):
    image, target = frames(L, (3840, 2160), (3840, 2160), icsp=pl.color_space("bt2020", "pq"),
                           tcsp=pl.color_space("bt709", "bt1886"))
    text = plan(L, image, target, pl.render_params("default"))
    assert "scale: none" in text and "peak: measured after scaling" in text, text
    assert "PQ" in text and "BT.1886" in text.split("colour:")[1], text
    # SDR input: nothing to measure
    image, target = frames(L, (3840, 2160), (3840, 2160))
    assert "peak: skipped (image is not HDR)" in plan(L, image, target, pl.render_params("default"))


def test_cfg5_downscale_is_linear_light_polar_with_peak_after(L):
    image, target = frames(L, (7680, 4320), (3840, 2160), icsp=pl.color_space("bt2020", "pq"),
                           tcsp=pl.color_space("bt709", "bt1886"))
    params = pl.render_params("high_quality", downscaler=pl.filter_config("ewa_lanczos", 2))
    text = plan(L, image, target, params)
    assert "plane 0: role 4 (reference) deband -> 7680x4320 as is" in text, text
    assert "scale: polar down linear -> 3840x2160" in text, text
    assert text.index("scale:") < text.index("peak: measured after scaling"), text
    assert "contrast recovery: feature map 1098x618" in text, text
    assert "(prelinearized)" in text


def test_upscale_measures_the_peak_first(L):
    image, target = frames(L, (1920, 1080), (3840, 2160), icsp=pl.color_space("bt2020", "pq"),
                           tcsp=pl.color_space("bt709", "bt1886"))
    params = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"))
    text = plan(L, image, target, params)
    assert text.index("peak: measured before scaling") < text.index("scale: polar up"), text
    assert "sigmoid" not in text    # never for HDR


def test_no_fbos_means_direct_sampling(L):
    image, target = frames(L, (1920, 1080), (3840, 2160))
    params = pl.render_params("default")
    text = plan(L, image, target, params, fbos=False)
    assert "no intermediate format" in text, text


@pytest.mark.parametrize("crop,tcrop,want_src,want_dst", [
    # half-pixel target rect: rounded outwards to even, source shifted by the same fraction
    ((0, 0, 100, 100), (10.4, 20.6, 110.4, 120.6), (-0.4, 0.4, 99.6, 100.4), (10, 21, 110, 121)),
    # horizontally flipped source: the flip moves to the target rect
    ((100, 0, 0, 100), (0, 0, 200, 200), (0, 0, 100, 100), (200, 0, 0, 200)),
    # target rect beyond the texture: clipped, source shrinks proportionally
    ((0, 0, 100, 100), (-50, 0, 150, 200), (25, 0, 100, 100), (0, 0, 150, 200)),
])
def test_rect_fitting(L, crop, tcrop, want_src, want_dst):
    image, target = frames(L, (100, 100), (200, 200), crop=crop, tcrop=tcrop)
    text = plan(L, image, target, pl.render_params("fast"))
    line = text.splitlines()[0]
    nums = [float(x) for x in line.replace("geometry: src ", "").replace(" dst ", ",")
            .replace(" rot ", ",").replace("-", ",-").replace(",,", ",").split(",") if x]
    # parse robustly: "src a,b-c,d dst e,f-g,h rot r"
    import re
    m = re.match(r"geometry: src (\S+),(\S+?)-(-?[\d.]+),(\S+) dst (-?\d+),(-?\d+)-(-?\d+),(-?\d+)", line)
    assert m, line
    got_src = tuple(float(m.group(i)) for i in (1, 2, 3, 4))
    got_dst = tuple(int(m.group(i)) for i in (5, 6, 7, 8))
    assert got_dst == want_dst, line
    assert all(abs(a - b) < 1e-4 for a, b in zip(got_src, want_src)), line


def test_planar_target_and_rotation(L):
    s = FakeTex(L, 64, 48, "rgba16")
    y, uv = FakeTex(L, 48, 64, "r8"), FakeTex(L, 24, 32, "rg8")
    image = pl.frame(s, components=3)
    image.rotation = 1
    target = capi.Frame(num_planes=2)
    target.planes[0] = capi.Plane(texture=y.ptr, components=1)
    target.planes[1] = capi.Plane(texture=uv.ptr, components=2)
    for c in range(4):
        target.planes[0].component_mapping[c] = [0, -1, -1, -1][c]
        target.planes[1].component_mapping[c] = [1, 2, -1, -1][c]
    target.repr = pl.color_repr("bt709", "limited")
    target.color = pl.color_space("bt709", "bt1886")
    text = plan(L, image, target, pl.render_params("fast"))
    assert "rot 1" in text.splitlines()[0], text
    # a quarter turn = transposed stores + one flipped axis
    assert "output plane 0: store 48,0-0,64" in text and "transposed" in text, text
    assert "output plane 1: store 24,0-0,32" in text, text


def test_invalid_frames_are_reported(L):
    image, target = frames(L, (16, 16), (16, 16))
    target.num_planes = 5
    assert plan(L, image, target, None).startswith("invalid: invalid number of planes")


def test_which_pending_images_are_fused_into_a_polar_scaler(L):
    """rp_fuse_into_polar: the pending image (plane fetch + colour ops) becomes the fused PASS A of
    a polar main scaler -- always for a downscale (the intermediate would be the large side;
    k_polar_mxd linearises while it stages), for an upscale only when its ops need no
    transcendental: pl_render_default_params on SDR video records LINEARIZE + SIGMOIDIZE there, and
    two passes (k_pass_chain, then the polar kernel with the chain epilogue) take 0.063 ms where the
    fused launch took 0.113 (profiles/r04_44_default_preset_ewa_fusion.txt). Anti-ringing has no
    fused form; PL_HIP_NO_FUSION=0 / 1 overrides (tests compare the structures)."""
    NONE, UP, DOWN = 0, 1, 2
    f = L.plh_test_fuse_into_polar
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int]
    assert f(UP, 1, 0.0, -1) and not f(UP, 0, 0.0, -1)          # lite / LINEARIZE + SIGMOIDIZE
    assert f(DOWN, 1, 0.0, -1) and f(DOWN, 0, 0.0, -1)
    assert not f(NONE, 1, 0.0, -1)
    assert not f(UP, 1, 0.8, -1) and not f(DOWN, 0, 0.8, 0)     # anti-ringing: never
    assert f(UP, 0, 0.0, 0)                                     # PL_HIP_NO_FUSION=0: fuse anyway
    assert not f(UP, 1, 0.0, 1) and not f(DOWN, 1, 0.0, 1)      # =1: the reference's structure
