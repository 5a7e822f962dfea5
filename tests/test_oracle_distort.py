"""The oracle's pl_shader_distort (src/shaders/sampling.c:1106-1217) on hand-checkable cases: upstream
only dispatches it (no vectors), so what pins the restatement is geometry -- the identity map returns
the image, a quarter turn is a quarter turn, the canvas' y axis points up, the alpha fade is one texel
wide."""
import numpy as np

import orc


def image(n, seed=0):
    rng = np.random.default_rng(seed)
    img = rng.random((n, n, 4), dtype=np.float32)
    img[..., 3] = 1.0
    return img


def test_identity_and_quarter_turn():
    n = 8
    img = image(n)
    # canvas (x, y) in [-1, 1]^2, y up -> texture (u, v) in [0, 1]^2, v down: u = x / 2 + 1/2, v = 1/2 - y / 2
    ident = [0.5, 0.0, 0.0, -0.5, 0.5, 0.5]
    assert np.array_equal(orc.distort(img, ident, n, n), img)
    # rotating the picture by +90 degrees (counter-clockwise, y up): the canvas point p shows the
    # texture at R^-1 p: (x, y) -> (y, -x) -> u = y / 2 + 1/2, v = 1/2 + x / 2
    turn = [0.0, 0.5, 0.5, 0.0, 0.5, 0.5]
    assert np.array_equal(orc.distort(img, turn, n, n), np.rot90(img, 1))
    # bicubic at texel centres is NOT the texel (a B-spline smooths), but it is symmetric
    smooth = orc.distort(img, ident, n, n, bicubic=True)
    assert not np.array_equal(smooth, img)
    assert np.allclose(orc.distort(img[:, ::-1], ident, n, n, bicubic=True), smooth[:, ::-1], atol=1e-6)


def test_alpha_fades_over_one_texel_outside():
    n = 8
    img = image(n, 1)
    # the picture at half size in the middle of the canvas: u = x + 1/2, v = 1/2 - y
    half = [1.0, 0.0, 0.0, -1.0, 0.5, 0.5]
    out = orc.distort(img, half, 2 * n, 2 * n, alpha_mode=1)
    a = out[..., 3]
    assert np.all(a[:3] == 0) and np.all(a[:, :3] == 0) and np.all(a[-3:] == 0)
    assert np.all(a[6:10, 6:10] == 1.0)
    # one output per texel here: column 4 is the first inside, its centre half a texel from the
    # edge: border = smoothstep(0, pt, pt / 2) = 1/2; column 5 is a texel and a half inside: 1
    assert a[8, 3] == 0.0 and a[8, 4] == 0.5 and a[8, 5] == 1.0
    assert a[4, 4] == 0.25                                      # a corner: both axes
    # premultiplied: colour fades with it
    pre = orc.distort(img, half, 2 * n, 2 * n, alpha_mode=2)
    assert np.allclose(pre[..., :3], out[..., :3] * a[..., None], atol=1e-7)
