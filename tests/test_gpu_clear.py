"""pl_frame_clear_rgba / pl_frame_clear_tiles (src/renderer.c:4116-4199) and the background / border
modes that use them (PL_CLEAR_COLOR / PL_CLEAR_TILES, :2491-2553, :2717-2756).

The colour handed to these functions is sRGB; it is translated into the frame's colour space
(translate_srgb_color, :2555-2584) and encoded with the inverse of the frame's pl_color_repr. The
expected texel values are computed here from the REFERENCE's own CPU code where it exists on this
machine (oracle/_ref/libplref.so: pl_color_linearize, pl_get_color_mapping_matrix,
pl_color_delinearize, pl_color_repr_decode), otherwise from the product's Tier-0 functions, which
tests/test_tier0_ref.py holds to it bit for bit."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu


def tier0():
    """(library, ColorSpace type, ColorRepr type) for the expected values"""
    return pl.lib()


def expected_rgb(color, csp):
    """translate_srgb_color: gamma-like targets are taken to be the curve the colour was given in"""
    L = tier0()
    src = pl.color_space("bt709", "srgb")
    gamma_like = int(csp.transfer) in (pl.TRC["bt1886"], pl.TRC["srgb"], pl.TRC["gamma22"])
    if gamma_like:
        src.transfer = csp.transfer
        src.hdr.min_luma = csp.hdr.min_luma
    else:
        src.hdr.min_luma = 1e-6         # PL_COLOR_HDR_BLACK
    v = (C.c_float * 3)(*color)
    L.pl_color_linearize(C.byref(src), v)
    m = L.pl_get_color_mapping_matrix(L.pl_raw_primaries_get(int(src.primaries)),
                                      L.pl_raw_primaries_get(int(csp.primaries)), 1)
    L.pl_matrix3x3_apply(C.byref(m), v)
    L.pl_color_delinearize(C.byref(csp), v)
    return np.array(list(v), np.float32)


def encoded(rgb, repr_):
    L = tier0()
    r2 = capi.ColorRepr()
    C.memmove(C.byref(r2), C.byref(repr_), C.sizeof(r2))
    tr = L.pl_color_repr_decode(C.byref(r2), None)
    L.pl_transform3x3_invert(C.byref(tr))
    v = (C.c_float * 3)(*rgb)
    L.pl_transform3x3_apply(C.byref(tr), v)
    return np.array(list(v), np.float32)


def clear_rgba(gpu, f, rgba):
    pl.lib().pl_frame_clear_rgba(gpu.gpu, C.byref(f), (C.c_float * 4)(*rgba))


@pytest.mark.parametrize("case", ["srgb", "bt1886", "bt2020_pq", "dcip3_gamma22"])
def test_clear_rgba_translates_the_colour(gpu, case):
    csp = {"srgb": pl.color_space("bt709", "srgb"), "bt1886": pl.color_space("bt709", "bt1886"),
           "bt2020_pq": pl.color_space("bt2020", "pq"),
           "dcip3_gamma22": pl.color_space("display_p3", "gamma22")}[case]
    pl.lib().pl_color_space_infer(C.byref(csp))
    tex = gpu.tex_create(24, 10, "rgba32f")
    f = pl.frame(tex, color=csp)
    rgba = (0.8, 0.3, 0.1, 0.6)
    clear_rgba(gpu, f, rgba)
    got = tex.download()
    want = expected_rgb(rgba[:3], csp)
    assert np.all(got[..., 3] == np.float32(0.6))
    assert np.allclose(got[0, 0, :3], want, rtol=0, atol=2e-7), (got[0, 0], want)
    assert np.all(got[..., :3] == got[0, 0, :3])
    if case in ("srgb", "bt1886"):
        # no primaries to map, the curve undone by itself: the colour comes back as it went in
        assert np.allclose(got[0, 0, :3], rgba[:3], atol=1e-6)
    else:
        assert np.abs(got[0, 0, :3] - np.float32(rgba[:3])).max() > 0.01
    tex.destroy()


def test_clear_rgba_planar_ycbcr_and_premultiplied_alpha(gpu):
    """a limited-range BT.709 Y + CbCr frame (10 bit in 16, shifted): every plane gets its own
    components of the ENCODED colour, scaled into the container; a premultiplied RGBA frame gets
    rgb * alpha"""
    w, h = 16, 8
    y = gpu.tex_create(w, h, "r16")
    uv = gpu.tex_create(w // 2, h // 2, "rg16")
    f = capi.Frame(num_planes=2)
    f.planes[0].texture, f.planes[0].components = y.ptr, 1
    f.planes[1].texture, f.planes[1].components = uv.ptr, 2
    for c in range(4):
        f.planes[0].component_mapping[c] = 0 if c == 0 else -1
        f.planes[1].component_mapping[c] = (1, 2, -1, -1)[c]
    f.repr = pl.color_repr("bt709", "limited", sample_depth=16, color_depth=10, bit_shift=6)
    f.color = pl.color_space("bt709", "bt1886")
    pl.lib().pl_color_space_infer(C.byref(f.color))
    rgba = (0.9, 0.2, 0.4, 1.0)
    clear_rgba(gpu, f, rgba)
    enc = encoded(expected_rgb(rgba[:3], f.color), f.repr)
    gy, guv = y.download(), uv.download()
    want = np.rint(np.clip(enc.astype(np.float64), 0, 1) * 65535).astype(np.int64)
    assert np.all(np.abs(gy[..., 0].astype(np.int64) - want[0]) <= 1), (gy[0, 0], want)
    assert np.all(np.abs(guv[..., 0].astype(np.int64) - want[1]) <= 1)
    assert np.all(np.abs(guv[..., 1].astype(np.int64) - want[2]) <= 1)
    # ten-bit codes shifted into the container: Y of a saturated red-ish colour, limited range
    assert 64 << 6 <= gy[0, 0, 0] <= 940 << 6 and abs(int(gy[0, 0, 0]) - int(want[0])) <= 1
    y.destroy(); uv.destroy()

    t = gpu.tex_create(8, 8, "rgba32f")
    f = pl.frame(t, color=pl.color_space("bt709", "srgb"), repr_=pl.color_repr("rgb", "full", alpha="premultiplied"))
    clear_rgba(gpu, f, (0.5, 0.25, 1.0, 0.5))
    got = t.download()
    assert np.allclose(got[3, 3], (0.25, 0.125, 0.5, 0.5), atol=1e-7), got[3, 3]
    t.destroy()


def tiles_expected(w, h, period_x, period_y, c0, c1):
    yy, xx = np.mgrid[0:h, 0:w]
    kx, ky = np.float32(1.0 / period_x), np.float32(1.0 / period_y)
    ox, oy = np.float32(xx + 0.5) * kx, np.float32(yy + 0.5) * ky
    tx, ty = (ox - np.floor(ox)) < 0.5, (oy - np.floor(oy)) < 0.5
    return np.where((tx == ty)[..., None], np.float32(c0), np.float32(c1))


def test_clear_tiles_planes_follow_their_subsampling(gpu):
    L = pl.lib()
    w, h = 64, 32
    y = gpu.tex_create(w, h, "r32f")
    uv = gpu.tex_create(w // 2, h // 2, "rg32f")
    f = capi.Frame(num_planes=2)
    f.planes[0].texture, f.planes[0].components = y.ptr, 1
    f.planes[1].texture, f.planes[1].components = uv.ptr, 2
    for c in range(4):
        f.planes[0].component_mapping[c] = 0 if c == 0 else -1
        f.planes[1].component_mapping[c] = (1, 2, -1, -1)[c]
    f.repr = pl.color_repr("bt709", "full")
    f.color = pl.color_space("bt709", "srgb")
    colors = ((C.c_float * 3) * 2)((1.0, 0.0, 0.0), (0.0, 0.0, 1.0))
    L.pl_frame_clear_tiles(gpu.gpu, C.byref(f), colors, 8)
    e0, e1 = encoded((1.0, 0.0, 0.0), f.repr), encoded((0.0, 0.0, 1.0), f.repr)
    gy, guv = y.download(), uv.download()
    wy = tiles_expected(w, h, 8, 8, [e0[0]], [e1[0]])
    wuv = tiles_expected(w // 2, h // 2, 4, 4, e0[1:3], e1[1:3])      # half-size planes: period 4
    assert np.allclose(gy, wy, atol=1e-6) and np.allclose(guv, wuv, atol=1e-6)
    assert len(np.unique(gy)) == 2
    y.destroy(); uv.destroy()


@pytest.mark.parametrize("mode", ["color", "tiles"])
def test_border_of_a_cropped_target(gpu, mode):
    """a target whose crop leaves a border: PL_CLEAR_COLOR fills it with the background colour
    (translated: a BT.1886 target keeps the value), PL_CLEAR_TILES with the tile pattern"""
    import util
    sw, sh = 32, 24
    img = util.chirp_rgba16(sw, sh)
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(64, 48, "rgba16")
    rr = pl.Renderer(gpu)
    csp = pl.color_space("bt709", "bt1886")
    params = pl.render_params("fast")
    params.background_color = (C.c_float * 3)(0.2, 0.4, 0.6)
    params.border = {"color": 0, "tiles": 1}[mode]      # enum pl_clear_mode
    params.tile_size = 8
    target = pl.frame(dst, color=csp, crop=(16.0, 12.0, 48.0, 36.0))
    assert rr.render(pl.frame(src, components=3, color=csp), target, params), gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    inside = np.zeros((48, 64), bool)
    inside[12:36, 16:48] = True
    assert np.array_equal(got[12:36, 16:48, :3], img[..., :3])      # 1:1 inside the crop
    if mode == "color":
        want = np.rint(np.float32([0.2, 0.4, 0.6]).astype(np.float64) * 65535)
        assert np.all(np.abs(got[~inside][:, :3].astype(np.float64) - want) <= 1), got[0, 0]
    else:
        t = tiles_expected(64, 48, 8, 8, [0.93] * 3, [0.87] * 3)
        want = np.rint(t.astype(np.float64) * 65535)
        assert np.all(np.abs(got[..., :3].astype(np.float64) - want)[~inside] <= 1)
    rr.destroy(); src.destroy(); dst.destroy()


def test_blend_against_tiles(gpu):
    """an image with alpha over PL_CLEAR_TILES (renderer.c:2734-2756): rgb * a + (1 - a) * tile,
    alpha 1 -- through the full interpreter (BLEND_TILES is not a `lite` op)"""
    w, h = 48, 32
    rng = np.random.default_rng(4)
    img = rng.integers(0, 65536, (h, w, 4)).astype(np.uint16)
    src = gpu.tex_create(w, h, "rgba16", img)
    dst = gpu.tex_create(w, h, "rgba32f")
    rr = pl.Renderer(gpu)
    csp = pl.color_space("bt709", "srgb")
    params = pl.render_params("fast")
    params.background = 1       # PL_CLEAR_TILES
    params.tile_size = 16
    image = pl.frame(src, components=4, color=csp, repr_=pl.color_repr("rgb", "full", alpha="independent"))
    assert rr.render(image, pl.frame(dst, color=csp), params), gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    a = orc.tex_decode(img, "rgba16").astype(np.float64)
    tile = tiles_expected(w, h, 16, 16, [0.93] * 3, [0.87] * 3).astype(np.float64)
    want = a[..., :3] * a[..., 3:4] + (1.0 - a[..., 3:4]) * tile
    assert np.abs(got[..., :3] - want).max() < 2e-6, np.abs(got[..., :3] - want).max()
    assert np.all(got[..., 3] == 1.0)
    rr.destroy(); src.destroy(); dst.destroy()


def test_tiles_under_a_cropped_image_continue_the_border(gpu):
    """PL_CLEAR_TILES as border AND as background of a transparent image on a cropped target whose
    origin is not a multiple of the tile period: one seamless pattern in absolute target
    coordinates (the reference's output pass is a raster pass: gl_FragCoord is the target pixel,
    src/renderer.c:2745; pl_frame_clear_tiles: :4153)."""
    sw, sh = 40, 24
    rng = np.random.default_rng(9)
    img = rng.integers(0, 65536, (sh, sw, 4)).astype(np.uint16)
    img[..., 3] = 0             # fully transparent: the frame IS the pattern
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(64, 48, "rgba16")
    rr = pl.Renderer(gpu)
    csp = pl.color_space("bt709", "bt1886")
    params = pl.render_params("fast")
    params.border = 1           # PL_CLEAR_TILES
    params.background = 1
    params.tile_size = 8
    x0, y0 = 13, 7              # (neither a multiple of 8 nor of 4)
    target = pl.frame(dst, color=csp, crop=(float(x0), float(y0), float(x0 + sw), float(y0 + sh)))
    image = pl.frame(src, components=4, color=csp, repr_=pl.color_repr("rgb", "full", alpha="independent"))
    assert rr.render(image, target, params), gpu.messages[-4:]
    assert rr.errors() == 0
    got = dst.download()
    t = tiles_expected(64, 48, 8, 8, [0.93] * 3, [0.87] * 3)
    want = np.rint(t.astype(np.float64) * 65535)
    err = np.abs(got[..., :3].astype(np.float64) - want)
    assert err.max() <= 1, (err.max(), np.argwhere(err > 1)[:4])
    rr.destroy(); src.destroy(); dst.destroy()
