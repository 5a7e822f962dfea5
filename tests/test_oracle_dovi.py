"""The oracle's Dolby Vision reshaping (src/shaders/colorspace.c:51-271) on values worked by hand:
the piece is chosen by the inner pivots, a polynomial piece is a quadratic, an MMR piece is the
constant plus, per order, the dot products with (s0 s1 s2) and the cross terms (s0 s1, s0 s2,
s1 s2, s0 s1 s2) raised to that order, every curve sees the CLAMPED INPUT colour (not the other
components' results), and the result is clamped to the outer pivots."""
import numpy as np

import orc


def comps():
    return (orc.DoviComp * 3)()


def px(r, g, b):
    return np.array([[[r, g, b, 0.5]]], np.float32)


def test_polynomial_pieces_and_pivots():
    c = comps()
    c[0].num_pivots = 4
    for k, v in enumerate((0.1, 0.4, 0.8, 0.9)):
        c[0].pivots[k] = v
    for i, co in enumerate(((0.5, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0))):
        for k, v in enumerate(co):
            c[0].poly_coeffs[i][k] = v
    # piece 0 below the first inner pivot (0.4): constant 0.5
    assert orc.dovi_reshape(px(0.2, 0.3, 0.3), c)[0, 0, 0] == 0.5
    # exactly ON an inner pivot: the upper piece (s >= pivot)
    assert orc.dovi_reshape(px(0.4, 0.3, 0.3), c)[0, 0, 0] == np.float32(0.4)
    assert orc.dovi_reshape(px(0.5, 0.3, 0.3), c)[0, 0, 0] == 0.5         # piece 1: identity
    # piece 2: s^2 = 0.7225 ... clamped to the outer pivots [0.1, 0.9]
    assert orc.dovi_reshape(px(0.85, 0, 0), c)[0, 0, 0] == np.float32(0.85) * np.float32(0.85)
    assert orc.dovi_reshape(px(0.99, 0, 0), c)[0, 0, 0] == np.float32(0.9)
    # the input is clamped to [0, 1] before anything else; untouched components pass through
    out = orc.dovi_reshape(px(-3.0, 7.0, -1.0), c)
    assert out[0, 0, 0] == 0.5 and out[0, 0, 1] == 7.0 and out[0, 0, 2] == -1.0 and out[0, 0, 3] == 0.5


def test_mmr_terms_by_order():
    s0, s1, s2 = 0.5, 0.25, 0.75
    for order in (1, 2, 3):
        c = comps()
        c[1].num_pivots = 2
        c[1].pivots[0], c[1].pivots[1] = -1000.0, 1000.0
        c[1].method[0] = 1
        c[1].mmr_order[0] = order
        c[1].mmr_constant[0] = 0.125
        for j in range(3):              # (unused orders must not be read)
            for k in range(7):
                c[1].mmr_coeffs[0][j][k] = (1, 2, 4, 8, 16, 32, 64)[k] if j < order else 1e6
        got = orc.dovi_reshape(px(s0, s1, s2), c)[0, 0, 1]
        x = (s0 * s1, s0 * s2, s1 * s2, s0 * s1 * s2)
        want = 0.125
        for j in range(order):
            p = j + 1
            want += 1 * s0 ** p + 2 * s1 ** p + 4 * s2 ** p
            want += 8 * x[0] ** p + 16 * x[1] ** p + 32 * x[2] ** p + 64 * x[3] ** p
        assert abs(got - want) < 1e-4 * want, (order, got, want)


def test_curves_read_the_input_not_each_other():
    c = comps()
    for ch in range(3):
        c[ch].num_pivots = 2
        c[ch].pivots[0], c[ch].pivots[1] = 0.0, 1.0
        c[ch].method[0] = 1
        c[ch].mmr_order[0] = 1
        c[ch].mmr_coeffs[0][0][(ch + 1) % 3] = 1.0      # every component takes its neighbour's INPUT
    out = orc.dovi_reshape(px(0.1, 0.2, 0.3), c)[0, 0]
    assert np.allclose(out[:3], [0.2, 0.3, 0.1], atol=1e-7)


def test_lms_tail_is_the_identity_for_the_identity_matrix():
    img = np.zeros((1, 5, 4), np.float32)
    img[0, :, 0] = [0.1, 0.3, 0.5, 0.7, 0.9]
    img[0, :, 1] = 0.4
    img[0, :, 2] = 0.6
    out = orc.dovi_lms(img.copy(), [1, 0, 0, 0, 1, 0, 0, 0, 1])
    # PQ OETF(EOTF(x)) = x up to the "%f"-printed constants (c1, c2, c3 no longer satisfy
    # c1 = c3 - c2 + 1 exactly: 5e-6 at the top of the range)
    assert np.abs(out - img).max() < 1e-5
