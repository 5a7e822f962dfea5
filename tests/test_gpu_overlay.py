"""Overlays (pl_frame.overlays: subtitles, on-screen display) and blended output
(pl_render_params.blend_params, pl_dispatch_params.blend_params) against the oracle.

The reference draws overlay parts as triangles through the rasteriser and blends with the fixed-
function unit (src/renderer.c:811-1020); neither runs on its CPU-only dummy backend and its tests
hold no images of them, so the oracle here is the restatement of that geometry
(oracle/pl_oracle.c: orc_overlay_fragments, orc_blend) fed with part placements computed in THIS
file from the reference's transform chain (:833-893, :2918-2934), not taken from the product.
The frame under the overlays is the product's own render without them (the renderer's stages have
their own parity tests): what is held to the oracle here is everything the overlays add."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu

ALPHA_OVERLAY = (pl.BLEND_SRC_ALPHA, pl.BLEND_ONE_MINUS_SRC_ALPHA,
                 pl.BLEND_ONE, pl.BLEND_ONE_MINUS_SRC_ALPHA)
PREMUL_OVERLAY = (pl.BLEND_ONE, pl.BLEND_ONE_MINUS_SRC_ALPHA,
                  pl.BLEND_ONE, pl.BLEND_ONE_MINUS_SRC_ALPHA)
# (a target whose alpha mode is left open counts as premultiplied: say which one is meant)
INDEPENDENT = pl.color_repr("rgb", "full", alpha="independent")


@pytest.fixture()
def rr(gpu):
    r = pl.Renderer(gpu)
    yield r
    r.destroy()


# ---- the reference's transform chain, restated (pl_transform2x2: p' = M p + c) ----------------

def tf_identity():
    return np.eye(2), np.zeros(2)


def tf_rmul(a, b):
    """b := a o b (pl_transform2x2_rmul)"""
    return a[0] @ b[0], a[0] @ b[1] + a[1]


def tf_apply(tf, x, y):
    p = tf[0] @ np.array([x, y], dtype=np.float64) + tf[1]
    return float(p[0]), float(p[1])


def plane_shift(plane_w, plane_h, ref_w, ref_h, shift_x=0.0, shift_y=0.0):
    """tscale of pass_output_target (:2918-2934)"""
    rx, ry = plane_w / ref_w, plane_h / ref_h
    rrx = round(rx) if rx >= 1 else 1.0 / round(1.0 / rx)
    rry = round(ry) if ry >= 1 else 1.0 / round(1.0 / ry)
    return np.diag([rrx, rry]), np.array([-shift_x, -shift_y])


def image_to_target(crop, dst, quarter_turn=False):
    """src_to_dst of draw_overlays (:833-850); crop / dst = (x0, y0, x1, y1)"""
    rx = (dst[2] - dst[0]) / (crop[2] - crop[0])
    ry = (dst[3] - dst[1]) / (crop[3] - crop[1])
    m, c = np.diag([rx, ry]), np.array([dst[0] - rx * crop[0], dst[1] - ry * crop[1]])
    if quarter_turn:
        m, c = np.array([[0.0, ry], [rx, 0.0]]), c[::-1].copy()
    return m, c


def place(part, tf, tex_w, tex_h):
    """vertices of EMIT_VERT (:898-918) -> the part on the plane, as the oracle takes it"""
    src, dst, rgba = part
    p00, p10 = tf_apply(tf, dst[0], dst[1]), tf_apply(tf, dst[2], dst[1])
    p01, p11 = tf_apply(tf, dst[0], dst[3]), tf_apply(tf, dst[2], dst[3])
    q = orc.OverlayPart(x0=min(p00[0], p11[0]), x1=max(p00[0], p11[0]),
                        y0=min(p00[1], p11[1]), y1=max(p00[1], p11[1]),
                        ox=p00[0], oy=p00[1], u0=src[0] / tex_w, v0=src[1] / tex_h)
    du, dv = (src[2] - src[0]) / tex_w, (src[3] - src[1]) / tex_h
    swapped = tf[0][0, 0] == 0 and tf[0][1, 1] == 0
    if swapped:     # the texture's x runs along the plane's y
        q.uy, q.vx = du / (p10[1] - p00[1]), dv / (p01[0] - p00[0])
    else:
        q.ux, q.vy = du / (p10[0] - p00[0]), dv / (p01[1] - p00[1])
    for c in range(4):
        q.color[c] = rgba[c] if rgba is not None else 0.0
    return q


def draw(plane, fmt, tex, parts, tf, mode=orc.OVERLAY_NORMAL, linear=True, premul=False,
         color_fn=None, swizzle=None):
    """one overlay over `plane` (decoded float image of a `fmt` texture), part by part: fragments,
    colour function, glyph coverage, swizzle, blend, rounding through the format"""
    h, w = plane.shape[:2]
    fixed = fmt in ("r8", "rg8", "rgba8", "r16", "rg16", "rgba16")
    for part in parts:
        q = place(part, tf, tex.shape[1], tex.shape[0])
        color, cov, mask = orc.overlay_fragments(tex, linear, mode, q, w, h)
        if color_fn is not None:
            color = color_fn(color)
        if mode == orc.OVERLAY_MONOCHROME:
            if premul:
                color[..., :3] *= cov[..., None]
            color[..., 3] *= cov
        if swizzle is not None:     # swizzle_color with force_alpha (:791-808)
            out = np.zeros_like(color)
            out[..., 3] = 1.0
            for c, m in enumerate(swizzle):
                out[..., c] = color[..., m]
            out[..., 3] = color[..., 3]
            color = out
        orc.blend(plane, color, mask, PREMUL_OVERLAY if premul else ALPHA_OVERLAY, True, fixed)
        plane[...] = orc.tex_decode(orc.tex_encode(plane, fmt), fmt)
    return plane


def bitmap_rgba8(w, h, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    img[..., 3] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    img[: h // 4, :, 3] = 255       # an opaque band and a transparent one
    img[-(h // 4):, :, 3] = 0
    return img


def glyphs_r8(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = (127.5 + 127.5 * np.sin(xx * 0.9 + seed) * np.cos(yy * 0.7)).astype(np.uint8)
    img[rng.integers(0, h, 8), rng.integers(0, w, 8)] = 255
    return img[..., None]


def test_target_overlay_bitmap_parts_in_drawing_order(gpu, rr):
    """PL_OVERLAY_NORMAL over an rgba8 target in target coordinates: scaled, overlapping (drawn
    in order, each over the rounded result of the one before), cut by the target's edge, mirrored,
    and one part that covers no pixel centre. Bit-exact."""
    w, h = 96, 64
    src = gpu.tex_create(w, h, "rgba16", util.chirp_rgba16(w, h))
    dst = gpu.tex_create(w, h, "rgba8")
    bmp = bitmap_rgba8(32, 24, 3)
    otex = gpu.tex_create(32, 24, "rgba8", bmp)
    image, target = pl.frame(src, components=3), pl.frame(dst, repr_=INDEPENDENT)
    params = pl.render_params("fast", dither_params=None)
    assert rr.render(image, target, params), gpu.messages[-4:]
    base = dst.download()

    parts = [
        ((0, 0, 32, 24), (8, 4, 40, 28), None),                 # 1:1
        ((0, 0, 32, 24), (24.25, 16.5, 88.25, 64.5), None),     # 2x, overlaps the first, leaves the target
        ((4, 2, 20, 10), (-6, -3, 10, 5), None),                # a sub-rect, cut by the top-left corner
        ((0, 0, 32, 24), (80, 40, 48, 16), None),               # mirrored in x and y
        ((0, 0, 32, 24), (50.6, 2, 50.9, 30), None),            # thinner than a pixel: nothing
    ]
    pl.set_overlays(target, [pl.overlay(otex, parts)])
    assert rr.render(image, target, params), gpu.messages[-4:]
    assert rr.errors() == 0, gpu.messages[-4:]
    got = dst.download()

    ref = draw(orc.tex_decode(base, "rgba8"), "rgba8", orc.tex_decode(bmp, "rgba8"), parts,
               tf_identity())
    ref8 = orc.tex_encode(ref, "rgba8")
    assert np.array_equal(got, ref8), util.diff_stats(got, ref8)
    assert not np.array_equal(got, base)
    for t in (src, dst, otex):
        t.destroy()


@pytest.mark.parametrize("premul", [False, True])
def test_monochrome_glyphs_over_a_16_bit_target(gpu, rr, premul):
    """PL_OVERLAY_MONOCHROME: the part's colour, its alpha (a premultiplied target: all of it)
    times the glyph texture's red channel; rgba16 target with an alpha channel. Bit-exact."""
    w, h = 80, 48
    src = gpu.tex_create(w, h, "rgba16", util.chirp_rgba16(w, h, alpha=40000))
    dst = gpu.tex_create(w, h, "rgba16")
    atlas = glyphs_r8(24, 16, 1)
    otex = gpu.tex_create(24, 16, "r8", atlas)
    image = pl.frame(src, repr_=pl.color_repr("rgb", "full", alpha="independent"))
    target = pl.frame(dst, repr_=pl.color_repr("rgb", "full",
                                               alpha="premultiplied" if premul else "independent"))
    params = pl.render_params("fast", dither_params=None, background_transparency=1.0)
    assert rr.render(image, target, params), gpu.messages[-4:]
    base = dst.download()

    parts = [
        ((0, 0, 12, 16), (4, 4, 16, 20), (1.0, 0.5, 0.25, 1.0)),
        ((12, 0, 24, 16), (14, 6, 26, 22), (0.0, 1.0, 0.5, 0.75)),      # kerned into the first
        ((0, 0, 24, 16), (30, 8, 78, 40), (0.25, 0.25, 1.0, 0.5)),      # 2x
    ]
    pl.set_overlays(target, [pl.overlay(otex, parts, mode=pl.OVERLAY_MONOCHROME)])
    assert rr.render(image, target, params), gpu.messages[-4:]
    assert rr.errors() == 0, gpu.messages[-4:]
    got = dst.download()

    def encode(color):      # pl_shader_encode_color: premultiplied targets get rgb * a
        if premul:
            color = color.copy()
            color[..., :3] *= color[..., 3:4]
        return color

    ref = draw(orc.tex_decode(base, "rgba16"), "rgba16", orc.tex_decode(atlas, "r8"), parts,
               tf_identity(), mode=orc.OVERLAY_MONOCHROME, premul=premul, color_fn=encode)
    ref16 = orc.tex_encode(ref, "rgba16")
    assert np.array_equal(got, ref16), util.diff_stats(got, ref16)
    assert not np.array_equal(got, base)
    for t in (src, dst, otex):
        t.destroy()


def test_image_overlay_follows_crop_scale_and_rotation(gpu, rr):
    """An IMAGE's overlays are given in its own coordinates (SRC_FRAME: texels of the frame,
    SRC_CROP: relative to its crop) and land where the image does: cropped, scaled 2x, turned by
    90 degrees. Bit-exact (every factor is a power of two)."""
    sw, sh = 64, 40
    src = gpu.tex_create(sw, sh, "rgba16", util.chirp_rgba16(sw, sh))
    bmp = bitmap_rgba8(16, 16, 5)
    otex = gpu.tex_create(16, 16, "rgba8", bmp)
    crop = (8, 4, 56, 36)                               # 48 x 32 of the image
    for rotation, tw, th, dst_rect in ((0, 128, 96, (16, 8, 112, 72)),
                                       (1, 96, 128, (8, 16, 72, 112))):
        dst = gpu.tex_create(tw, th, "rgba8")
        image = pl.frame(src, components=3, crop=crop)
        image.rotation = rotation
        target = pl.frame(dst, crop=dst_rect, repr_=INDEPENDENT)
        params = pl.render_params("fast", dither_params=None)
        assert rr.render(image, target, params), gpu.messages[-4:]
        base = dst.download()

        frame_parts = [((0, 0, 16, 16), (12, 8, 28, 24), None)]
        crop_parts = [((0, 0, 16, 16), (20, 10, 36, 26), None)]
        pl.set_overlays(image, [pl.overlay(otex, frame_parts, coords=pl.OVERLAY_COORDS_SRC_FRAME),
                                pl.overlay(otex, crop_parts, coords=pl.OVERLAY_COORDS_SRC_CROP)])
        assert rr.render(image, target, params), gpu.messages[-4:]
        assert rr.errors() == 0, gpu.messages[-4:]
        got = dst.download()

        # the target rect counter-rotated into the image's orientation (fix_refs_and_rects)
        if rotation:
            # pl_rect2df_rotate(dst, -1) on a tw x th target: (x, y) -> (y, x) here, with flips that
            # the reference keeps in the rect; take them from the product-independent statement:
            # a clockwise quarter turn maps image x to target y and image y to target -x
            d = (dst_rect[1], dst_rect[2], dst_rect[3], dst_rect[0])
        else:
            d = dst_rect
        s2d = image_to_target(crop, d, quarter_turn=bool(rotation))
        ref = orc.tex_decode(base, "rgba8")
        tex = orc.tex_decode(bmp, "rgba8")
        draw(ref, "rgba8", tex, frame_parts, s2d)
        draw(ref, "rgba8", tex, crop_parts,
             tf_rmul(s2d, (np.eye(2), np.array([crop[0], crop[1]], dtype=np.float64))))
        ref8 = orc.tex_encode(ref, "rgba8")
        assert np.array_equal(got, ref8), (rotation, util.diff_stats(got, ref8))
        assert not np.array_equal(got, base)
        dst.destroy()
    src.destroy(); otex.destroy()


def test_overlay_on_subsampled_planes(gpu, rr):
    """A planar 4:2:0 target: every plane gets the overlay through its own transform (subsampling
    ratio, chroma sample position: tscale, :2918-2934) and its own swizzle. An opaque flat-coloured
    bitmap: the covered texels of each plane are exactly the ones the oracle covers, and they hold
    the encoded colour (within a code: the YCbCr matrix is float arithmetic)."""
    w, h = 64, 48
    src = gpu.tex_create(w, h, "rgba16", util.chirp_rgba16(w, h))
    ty, tuv = gpu.tex_create(w, h, "r8"), gpu.tex_create(w // 2, h // 2, "rg8")
    image = pl.frame(src, components=3, color=pl.color_space("bt709", "bt1886"))
    f = capi.Frame(num_planes=2)
    f.planes[0].texture, f.planes[0].components = ty.ptr, 1
    f.planes[1].texture, f.planes[1].components = tuv.ptr, 2
    for c in range(4):
        f.planes[0].component_mapping[c] = 0 if c == 0 else -1
        f.planes[1].component_mapping[c] = c + 1 if c < 2 else -1
    f.repr = pl.color_repr("bt709", "limited")
    f.color = pl.color_space("bt709", "bt1886")
    pl.lib().pl_frame_set_chroma_location.argtypes = [C.POINTER(capi.Frame), C.c_int]
    pl.lib().pl_frame_set_chroma_location(C.byref(f), 1)    # left: shift_x = -0.5
    params = pl.render_params("fast", dither_params=None)
    assert rr.render(image, f, params), gpu.messages[-4:]
    base_y, base_uv = ty.download(), tuv.download()

    rgb = (200, 40, 90)
    bmp = np.zeros((8, 8, 4), np.uint8)
    bmp[..., :3] = rgb
    bmp[..., 3] = 255
    otex = gpu.tex_create(8, 8, "rgba8", bmp)
    parts = [((0, 0, 8, 8), (10, 6, 42, 31), None), ((0, 0, 8, 8), (37, 20, 60, 44), None)]
    pl.set_overlays(f, [pl.overlay(otex, parts, color=pl.color_space("bt709", "bt1886"))])
    assert rr.render(image, f, params), gpu.messages[-4:]
    assert rr.errors() == 0, gpu.messages[-4:]
    got_y, got_uv = ty.download(), tuv.download()

    r, g, b = (np.float64(v) / 255 for v in rgb)
    luma = 0.2126 * r + 0.7152 * g + 0.0722 * b
    want = (16 + 219 * luma, 128 + 224 * (b - luma) / 1.8556, 128 + 224 * (r - luma) / 1.5748)
    sx, sy = f.planes[1].shift_x, f.planes[1].shift_y
    assert (sx, sy) == (-0.5, 0.0)
    for got, base, fmt, tf, chans in (
            (got_y, base_y, "r8", plane_shift(w, h, w, h), (0,)),
            (got_uv, base_uv, "rg8", plane_shift(w // 2, h // 2, w, h, sx, sy), (1, 2))):
        ph, pw = base.shape[:2]
        covered = np.zeros((ph, pw), bool)
        for part in parts:
            q = place(part, tf, 8, 8)
            covered |= orc.overlay_fragments(orc.tex_decode(bmp, "rgba8"), True, orc.OVERLAY_NORMAL,
                                             q, pw, ph)[2].astype(bool)
        assert covered.any() and not covered.all()
        assert np.array_equal(got[~covered], base[~covered])
        for k, ch in enumerate(chans):
            assert np.abs(got[covered][:, k].astype(np.float64) - want[ch]).max() <= 1.0, fmt
    for t in (src, ty, tuv, otex):
        t.destroy()


def test_overlay_colour_is_mapped_to_an_hdr_target(gpu, rr):
    """An sRGB overlay over a PQ / BT.2020 target goes through the stateless colour map of
    draw_overlays (linear tone mapping, saturation gamut mapping: `osd_params`, :963-967). The
    fragments are recorded by hand through the pl_shader_* API (each stage has its own parity
    test) into an rgba32f texture, the oracle blends them: the renderer adds nothing but glue."""
    w, h = 64, 48
    hdr = pl.color_space("bt2020", "pq")
    src = gpu.tex_create(w, h, "rgba16", util.chirp_rgba16(w, h))
    dst = gpu.tex_create(w, h, "rgba16")
    bmp = bitmap_rgba8(16, 12, 9)
    otex = gpu.tex_create(16, 12, "rgba8", bmp)
    image, target = pl.frame(src, components=3, color=hdr), pl.frame(dst, color=hdr, repr_=INDEPENDENT)
    params = pl.render_params("fast", dither_params=None)
    assert rr.render(image, target, params), gpu.messages[-4:]
    base = dst.download()
    parts = [((0, 0, 16, 12), (20, 10, 36, 22), None)]     # 1:1: the fragments are the texels
    pl.set_overlays(target, [pl.overlay(otex, parts)])
    assert rr.render(image, target, params), gpu.messages[-4:]
    assert rr.errors() == 0, gpu.messages[-4:]
    got = dst.download()

    frag_tex = gpu.tex_create(16, 12, "rgba32f")
    sh = gpu.begin()
    assert sh.sample("direct", otex)
    sh.decode_color(pl.color_repr("rgb", "full", alpha="independent"))
    sh.color_map(pl.color_space("bt709", "srgb"), target.color,
                 params=pl.color_map_params(tone="linear", gamut="saturation"))
    sh.encode_color(pl.color_repr("rgb", "full"))
    assert sh.finish(frag_tex), gpu.messages[-4:]
    frags = frag_tex.download()

    ref = orc.tex_decode(base, "rgba16")
    color = np.zeros_like(ref)
    mask = np.zeros((h, w), np.uint8)
    color[10:22, 20:36] = frags
    mask[10:22, 20:36] = 1
    orc.blend(ref, color, mask, ALPHA_OVERLAY, True, True)
    ref16 = orc.tex_encode(ref, "rgba16")
    assert np.array_equal(got, ref16), util.diff_stats(got, ref16)
    # and the map did something: an sRGB white is nowhere near PQ full scale
    assert frags[..., :3].max() < 0.7
    for t in (src, dst, otex, frag_tex):
        t.destroy()


def test_target_overlays_without_an_image(gpu, rr):
    """pl_render_image(NULL image): the target is cleared and its overlays are drawn (:3397-3424)."""
    w, h = 48, 32
    dst = gpu.tex_create(w, h, "rgba8")
    bmp = bitmap_rgba8(16, 8, 11)
    otex = gpu.tex_create(16, 8, "rgba8", bmp)
    target = pl.frame(dst, repr_=INDEPENDENT)
    params = pl.render_params("fast", background_color=(C.c_float * 3)(0.25, 0.5, 0.75))
    assert pl.lib().pl_render_image(rr.rr, None, C.byref(target), C.byref(params))
    base = dst.download()
    parts = [((0, 0, 16, 8), (4, 4, 36, 20), None)]
    pl.set_overlays(target, [pl.overlay(otex, parts, coords=pl.OVERLAY_COORDS_DST_CROP)])
    assert pl.lib().pl_render_image(rr.rr, None, C.byref(target), C.byref(params))
    assert rr.errors() == 0, gpu.messages[-4:]
    got = dst.download()
    ref = draw(orc.tex_decode(base, "rgba8"), "rgba8", orc.tex_decode(bmp, "rgba8"), parts,
               tf_identity())
    ref8 = orc.tex_encode(ref, "rgba8")
    assert np.array_equal(got, ref8), util.diff_stats(got, ref8)
    dst.destroy(); otex.destroy()


@pytest.mark.parametrize("fmt", ["rgba8", "rgba16", "rgba16hf"])
def test_dispatch_with_blend_params(gpu, fmt):
    """pl_dispatch_finish(.blend_params): the pass's colour meets the target's content in the
    blend unit -- unrounded, clamped for a fixed-point target, into a flipped sub-rect."""
    w, h = 40, 24
    img = util.chirp_rgba16(w, h)
    img[..., 3] = (np.arange(w, dtype=np.uint32) * 65535 // (w - 1)).astype(np.uint16)[None, :]
    src = gpu.tex_create(w, h, "rgba16", img)
    under = orc.tex_encode(orc.tex_decode(util.random_rgba16(64, 48, seed=4), "rgba16"), fmt)
    dst = gpu.tex_create(64, 48, fmt, under)
    sh = gpu.begin()
    assert sh.sample("direct", src)
    pl.lib().plh_test_op_scale(sh.sh, C.c_float(1.25))  # leaves [0, 1]: the clamp matters
    blend = capi.BlendParams(*ALPHA_OVERLAY)
    assert sh.finish(dst, rect=(52, 6, 12, 30), blend_params=blend), gpu.messages[-4:]
    got = dst.download()

    frag = orc.sample_simple(orc.tex_decode(img, "rgba16"), orc.S_NEAREST, w, h)
    frag = (frag * np.float32(1.25))[:, ::-1]           # x1 < x0: mirrored
    ref = orc.tex_decode(under, fmt)
    color = np.zeros_like(ref)
    mask = np.zeros(ref.shape[:2], np.uint8)
    color[6:30, 12:52] = frag
    mask[6:30, 12:52] = 1
    orc.blend(ref, color, mask, ALPHA_OVERLAY, True, fmt != "rgba16hf")
    want = orc.tex_encode(ref, fmt)
    assert np.array_equal(got, want), util.diff_stats(got, want)
    src.destroy(); dst.destroy()


def test_render_with_blend_params(gpu, rr):
    """pl_render_params.blend_params: the image keeps its alpha up to the blend unit (a blended
    output counts as having one, src/renderer.c:2713), colour scale and swizzle leave it alone
    (:2911-2917), and the frame lands OVER what the target held."""
    w, h = 48, 32
    img = util.chirp_rgba16(w, h)
    img[..., 3] = (np.arange(h, dtype=np.uint32) * 65535 // (h - 1)).astype(np.uint16)[:, None]
    src = gpu.tex_create(w, h, "rgba16", img)
    under = util.random_rgba16(w, h, seed=8)
    dst = gpu.tex_create(w, h, "rgba16", under)
    image = pl.frame(src, repr_=pl.color_repr("rgb", "full", alpha="independent"))
    target = pl.frame(dst, components=3)
    params = pl.render_params("fast", dither_params=None, background_transparency=1.0,
                              skip_target_clearing=True,
                              blend_params=capi.BlendParams(*ALPHA_OVERLAY))
    assert rr.render(image, target, params), gpu.messages[-4:]
    assert rr.errors() == 0, gpu.messages[-4:]
    got = dst.download()
    ref = orc.tex_decode(under, "rgba16")
    orc.blend(ref, orc.tex_decode(img, "rgba16"), None, ALPHA_OVERLAY, True, True)
    want = orc.tex_encode(ref, "rgba16")
    assert np.array_equal(got, want), util.diff_stats(got, want)
    src.destroy(); dst.destroy()


def test_many_glyphs_and_large_part_lists(gpu, rr):
    """2000 small parts over a 4K target (a screenful of text): the host's tile binning against
    the oracle walking the parts one by one over the region they cover. Bit-exact."""
    w, h = 3840, 2160
    dst = gpu.tex_create(w, h, "rgba8")
    target = pl.frame(dst, repr_=INDEPENDENT)
    params = pl.render_params("fast")
    assert pl.lib().pl_render_image(rr.rr, None, C.byref(target), C.byref(params))
    atlas = glyphs_r8(64, 64, 2)
    otex = gpu.tex_create(64, 64, "r8", atlas)
    rng = np.random.default_rng(0)
    parts = []
    x0r, y0r, x1r, y1r = 1800, 1000, 2200, 1200     # the oracle's window
    for i in range(2000):
        inside = i % 10 == 0
        x = rng.integers(x0r, x1r - 40) if inside else rng.integers(-20, w)
        y = rng.integers(y0r, y1r - 40) if inside else rng.integers(-20, h)
        gx, gy = rng.integers(0, 4) * 16, rng.integers(0, 4) * 16
        parts.append(((gx, gy, gx + 16, gy + 16), (x + 0.5, y + 0.25, x + 24.5, y + 32.25),
                      tuple(rng.random(3)) + (0.75,)))
    base = dst.download()
    pl.set_overlays(target, [pl.overlay(otex, parts, mode=pl.OVERLAY_MONOCHROME)])
    assert pl.lib().pl_render_image(rr.rr, None, C.byref(target), C.byref(params))
    assert rr.errors() == 0, gpu.messages[-4:]
    got = dst.download()

    # the window, with every part that touches it, in order (coordinates relative to the window)
    tex = orc.tex_decode(atlas, "r8")
    win = orc.tex_decode(np.ascontiguousarray(base[y0r:y1r, x0r:x1r]), "rgba8")
    near = [p for p in parts if p[1][2] > x0r and p[1][0] < x1r and p[1][3] > y0r and p[1][1] < y1r]
    assert len(near) >= 200
    draw(win, "rgba8", tex, near, (np.eye(2), np.array([-x0r, -y0r], dtype=np.float64)),
         mode=orc.OVERLAY_MONOCHROME)
    want = orc.tex_encode(win, "rgba8")
    assert np.array_equal(got[y0r:y1r, x0r:x1r], want), util.diff_stats(got[y0r:y1r, x0r:x1r], want)
    assert (got != base).any(axis=2).sum() > 500000
    dst.destroy(); otex.destroy()


def test_mixed_frames_carry_their_own_overlays(gpu, rr):
    """pl_render_image_mix: a frame's overlays are drawn onto its cached intermediate (the f16
    image the blend reads, src/renderer.c:3855-3877), so they are mixed with the frame's weight:
    two frames with the overlay at different places give two ghosts. Same chain and bar as
    test_gpu_mix.py (2 LSB of 16 bit), with the oracle's overlay drawn into each f16 frame."""
    import test_gpu_mix as tm
    from test_gpu_color import nominal, luma_coeffs
    csp = pl.color_space("bt709", "bt1886")
    imgs, texs, frames = tm.sources(gpu, 2)
    bmp = bitmap_rgba8(16, 12, 7)
    otex = gpu.tex_create(16, 12, "rgba8", bmp)
    places = [[((0, 0, 16, 12), (6, 4, 38, 28), None)], [((0, 0, 16, 12), (20, 16, 52, 40), None)]]
    for f, parts in zip(frames, places):
        pl.set_overlays(f, [pl.overlay(otex, parts, color=csp)])
    cfg = tm.mixer("linear")
    ts = [-0.7, 0.3]
    ok, got = tm.run_mix(gpu, rr, frames, ts, pl.render_params("fast", frame_mixer=cfg))
    assert ok and rr.errors() == 0, gpu.messages[-4:]
    cfg.blur = cfg.blur or 1.0
    weights = [np.float32(pl.lib().pl_filter_sample(C.byref(cfg), t)) for t in ts]

    full = pl.color_space("bt709", "bt1886")
    pl.lib().pl_color_space_infer(C.byref(full))
    mn, mx = nominal(full)
    luma = luma_coeffs(full.primaries)
    wsum = np.float32(sum(weights))
    acc = np.zeros((tm.H, tm.W, 4), np.float32)
    tex = orc.tex_decode(bmp, "rgba8")
    for img, parts, wgt in zip(imgs, places, weights):
        f = orc.tex_decode(img, "rgba16")
        f[..., 3] = 1.0
        f = orc.op_quant_f16(f)                                 # the cached rgba16hf frame
        draw(f, "rgba16hf", tex, parts, tf_identity())          # ... with its overlay in it
        f = orc.linearize(f, int(full.transfer), mn, mx, luma)
        acc += (wgt / wsum) * f
    ref16 = orc.tex_encode(orc.delinearize(acc, int(full.transfer), mn, mx, luma), "rgba16")
    d = np.abs(got.astype(np.int64) - ref16.astype(np.int64))[..., :3]
    assert d.max() <= 2, (int(d.max()), float(d.mean()))
    # without the overlays the same mix is far away, where either one lies
    for f in frames:
        f.num_overlays = 0
    ok, plain = tm.run_mix(gpu, rr, frames, ts, pl.render_params("fast", frame_mixer=cfg),
                           sigs=[2000, 2001])
    assert ok
    for (_, (x0, y0, x1, y1), _), in places:
        assert np.abs(got[y0:y1, x0:x1, :3].astype(np.int64) - plain[y0:y1, x0:x1, :3]).max() > 2000
    for t in texs + [otex]:
        t.destroy()


def test_overlay_layouts_are_reused_and_replaced(gpu, rr):
    """The dispatch keeps the binned parts of the overlays it drew on the device (subtitles stay
    for seconds) and replaces the least recently used one: ten different overlays shown in turn,
    three times round -- more than it keeps -- and every frame is the oracle's."""
    w, h = 64, 48
    dst = gpu.tex_create(w, h, "rgba8")
    bmp = bitmap_rgba8(16, 12, 13)
    otex = gpu.tex_create(16, 12, "rgba8", bmp)
    tex = orc.tex_decode(bmp, "rgba8")
    target = pl.frame(dst, repr_=INDEPENDENT)
    params = pl.render_params("fast")
    assert pl.lib().pl_render_image(rr.rr, None, C.byref(target), C.byref(params))
    base = dst.download()
    lists = [[((0, 0, 16, 12), (2 + 5 * k, 3 + 2 * k, 18 + 5 * k, 15 + 2 * k), None),
              ((0, 0, 16, 12), (40 - 3 * k, 30 - k, 56 - 3 * k, 42 - k), None)] for k in range(10)]
    want = [orc.tex_encode(draw(orc.tex_decode(base, "rgba8"), "rgba8", tex, parts, tf_identity()),
                           "rgba8") for parts in lists]
    for turn in range(3):
        for k, parts in enumerate(lists):
            pl.set_overlays(target, [pl.overlay(otex, parts)])
            assert pl.lib().pl_render_image(rr.rr, None, C.byref(target), C.byref(params))
            got = dst.download()
            assert np.array_equal(got, want[k]), (turn, k, util.diff_stats(got, want[k]))
    assert rr.errors() == 0
    dst.destroy(); otex.destroy()
