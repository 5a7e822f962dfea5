"""The oracle's restatement of overlay rasterisation and of the blend unit, on cases small enough
to work out by hand (the reference holds no vectors for either: its tests never draw an overlay
with a real backend, src/tests/gpu_tests.c). What is pinned here: the rasteriser's top-left rule
(Vulkan spec 25.8, OpenGL 4.6 14.6.1: a pixel is covered when its centre is inside; centres ON an
edge belong to the primitive whose interior lies to their right / below), attribute interpolation
on a parallelogram being affine, and the blend equations of gpu.h pl_blend_params."""
import numpy as np

import orc


def checker(w, h):
    tex = np.zeros((h, w, 4), np.float32)
    tex[..., 0] = np.arange(w, dtype=np.float32)[None, :] / w
    tex[..., 1] = np.arange(h, dtype=np.float32)[:, None] / h
    tex[..., 3] = 1.0
    return tex


def test_top_left_rule_and_affine_coordinates():
    tex = checker(4, 2)
    # texture (0,0)-(4,2) onto plane [1, 3) x [0.5, 2.5): half size, origin (1, 0.5)
    q = orc.OverlayPart(x0=1.0, y0=0.5, x1=3.0, y1=2.5, ox=1.0, oy=0.5,
                        ux=1.0 / 2.0, u0=0.0, vy=1.0 / 2.0, v0=0.0)
    color, cov, mask = orc.overlay_fragments(tex, False, orc.OVERLAY_NORMAL, q, 5, 4)
    want = np.zeros((4, 5), np.uint8)
    want[0:2, 1:3] = 1      # centres 1.5, 2.5 (3.5 is outside); 0.5 is ON the top edge: inside;
    assert np.array_equal(mask, want)       # 2.5 is ON the bottom edge: outside
    # centre (1.5, 0.5): u = 0.25 -> texel 1 of 4; v = 0 -> texel 0
    assert color[0, 1, 0] == np.float32(0.25) and color[0, 1, 1] == 0.0
    # centre (2.5, 1.5): u = 0.75 -> texel 3; v = 0.5 -> texel 1
    assert color[1, 2, 0] == np.float32(0.75) and color[1, 2, 1] == np.float32(0.5)
    assert np.all(cov == 1.0)


def test_mirrored_and_turned_parts():
    tex = checker(4, 4)
    # mirrored in x: vertex (x0) lands on the right edge, u falls as x grows
    q = orc.OverlayPart(x0=0.0, y0=0.0, x1=4.0, y1=4.0, ox=4.0, oy=0.0,
                        ux=-1.0 / 4.0, u0=0.0, vy=1.0 / 4.0, v0=0.0)
    color, _, mask = orc.overlay_fragments(tex, False, orc.OVERLAY_NORMAL, q, 4, 4)
    assert mask.all()
    assert np.array_equal(color[0, :, 0], tex[0, ::-1, 0])
    # a quarter turn: the texture's x runs down the plane, its y across
    q = orc.OverlayPart(x0=0.0, y0=0.0, x1=4.0, y1=4.0, ox=0.0, oy=0.0,
                        uy=1.0 / 4.0, u0=0.0, vx=1.0 / 4.0, v0=0.0)
    color, _, _ = orc.overlay_fragments(tex, False, orc.OVERLAY_NORMAL, q, 4, 4)
    assert np.array_equal(color[..., 0], tex[..., 0].T) and np.array_equal(color[..., 1], tex[..., 1].T)


def test_monochrome_takes_the_parts_colour_and_the_textures_red():
    tex = checker(2, 2)
    tex[..., 0] = [[0.25, 0.5], [0.75, 1.0]]
    q = orc.OverlayPart(x0=0.0, y0=0.0, x1=2.0, y1=2.0, ox=0.0, oy=0.0,
                        ux=0.5, u0=0.0, vy=0.5, v0=0.0)
    for c, v in enumerate((0.1, 0.2, 0.3, 0.4)):
        q.color[c] = v
    color, cov, mask = orc.overlay_fragments(tex, False, orc.OVERLAY_MONOCHROME, q, 2, 2)
    assert mask.all()
    assert np.allclose(color, np.float32([0.1, 0.2, 0.3, 0.4]), rtol=0, atol=0)
    assert np.array_equal(cov, tex[..., 0])
    # bilinear at a texel centre is the texel
    assert np.array_equal(orc.overlay_fragments(tex, True, orc.OVERLAY_MONOCHROME, q, 2, 2)[1], cov)


def test_blend_equations():
    ZERO, ONE, SA, OMSA = 0, 1, 2, 3
    dst = np.float32([[[0.0, 0.5, 1.0, 1.0], [0.25, 0.25, 0.25, 0.5]]])
    src = np.float32([[[1.0, 0.5, 0.0, 0.5], [2.0, -1.0, 0.5, 0.5]]])
    # pl_alpha_overlay on a fixed-point target: the fragment is clamped first
    out = orc.blend(dst.copy(), src, None, (SA, OMSA, ONE, OMSA), True, True)
    assert np.array_equal(out[0, 0], np.float32([0.5, 0.5, 0.5, 1.0]))
    assert np.array_equal(out[0, 1], np.float32([0.625, 0.125, 0.375, 0.75]))
    # a float target: no clamp
    out = orc.blend(dst.copy(), src, None, (SA, OMSA, ONE, OMSA), True, False)
    assert np.array_equal(out[0, 1], np.float32([1.125, -0.375, 0.375, 0.75]))
    # premultiplied source: ONE / OMSA
    out = orc.blend(dst.copy(), src, None, (ONE, OMSA, ONE, OMSA), True, True)
    assert np.array_equal(out[0, 0], np.float32([1.0, 0.75, 0.5, 1.0]))
    # factors ZERO / ONE leave the target alone; the mask does too
    assert np.array_equal(orc.blend(dst.copy(), src, None, (ZERO, ONE, ZERO, ONE), True, True), dst)
    m = np.uint8([[0, 1]])
    out = orc.blend(dst.copy(), src, m, (ONE, ZERO, ONE, ZERO), True, True)
    assert np.array_equal(out[0, 0], dst[0, 0]) and np.array_equal(out[0, 1], np.float32([1, 0, 0.5, 0.5]))
    # blending off: the (clamped) fragment replaces the target
    out = orc.blend(dst.copy(), src, None, (ZERO, ZERO, ZERO, ZERO), False, True)
    assert np.array_equal(out[0, 1], np.float32([1, 0, 0.5, 0.5]))
