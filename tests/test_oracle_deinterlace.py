"""The oracle's restatement of pl_shader_deinterlace (src/shaders/deinterlacing.c) on cases that can
be worked out by hand. The reference's tests only check that the shader dispatches
(src/tests/gpu_tests.c:875-880) and time it (src/tests/bench.c:314-366): there are no vectors, so
what pins the restatement is what the algorithms promise -- the shown field passes through, a
still scene comes back as it went in, bob doubles rows, yadif follows an edge, bwdif's intra
filter is the cubic it says it is."""
import numpy as np

import orc

TOP, BOTTOM = 1, 2


def frame(w, h, seed):
    rng = np.random.default_rng(seed)
    return rng.random((h, w, 4), dtype=np.float32)


def test_shown_field_passes_through_and_weave_is_the_frame():
    cur, prev, nxt = frame(9, 8, 1), frame(9, 8, 2), frame(9, 8, 3)
    for algo in (orc.DEINT_WEAVE, orc.DEINT_BOB, orc.DEINT_YADIF, orc.DEINT_BWDIF):
        for field in (TOP, BOTTOM):
            out = orc.deinterlace(cur, prev, nxt, field, TOP, algo)
            kept = slice(0, None, 2) if field == TOP else slice(1, None, 2)
            assert np.array_equal(out[kept], cur[kept]), (algo, field)
    assert np.array_equal(orc.deinterlace(cur, prev, nxt, TOP, TOP, orc.DEINT_WEAVE), cur)
    assert np.array_equal(orc.deinterlace(cur, prev, nxt, 0, TOP, orc.DEINT_YADIF), cur)


def test_component_mask_leaves_the_initial_colour():
    cur = frame(5, 6, 4)
    out = orc.deinterlace(cur, None, None, TOP, TOP, orc.DEINT_BOB, comp_mask=0b0101)
    assert np.all(out[..., 1] == 0.0) and np.all(out[..., 3] == 1.0)
    assert np.array_equal(out[::2, :, 0], cur[::2, :, 0])


def test_bob_doubles_the_rows_of_the_shown_field():
    cur = frame(4, 8, 5)
    out = orc.deinterlace(cur, None, None, TOP, TOP, orc.DEINT_BOB)
    assert np.array_equal(out[1::2], cur[0::2])                 # the row above
    out = orc.deinterlace(cur, None, None, BOTTOM, TOP, orc.DEINT_BOB)
    assert np.array_equal(out[0:-1:2], cur[1::2])               # the row below


def test_a_still_scene_comes_back():
    cur = frame(12, 10, 6)
    for field in (TOP, BOTTOM):
        for first in (TOP, BOTTOM):
            # bwdif: no motion at all (diff == 0): the average of the second neighbours, which are
            # the frame itself
            out = orc.deinterlace(cur, cur, cur, field, first, orc.DEINT_BWDIF)
            assert np.array_equal(out, cur)
    # yadif on a vertical ramp: the spatial prediction is the row between, which is what is there
    ramp = np.zeros((10, 12, 4), np.float32)
    ramp[...] = (np.arange(10, dtype=np.float32) / 16)[:, None, None]
    out = orc.deinterlace(ramp, ramp, ramp, TOP, TOP, orc.DEINT_YADIF)
    assert np.array_equal(out[1:-1], ramp[1:-1])
    # (row 9's neighbour below is the mirror of row 8: the clamp keeps it within what moved: 0)
    assert np.array_equal(out[-1], ramp[-1])


def test_yadif_follows_a_moving_diagonal_edge():
    # Rows above / below the rebuilt row 1: an edge that moves two columns per row pair. At column
    # 4 the vertical neighbours are 0 (above) and 1 (below), the straight average would smear the
    # edge. Scores, (above, below) columns: straight (3,3) (4,4) (5,5): 1 + 1 + 0 = 2 - 1/255;
    # leaning one way (2,4) (3,5) (4,6): 3, no; the other way (4,2) (5,3) (6,4): 0 + 0 + 0 = 0,
    # better; one step further (5,1) (6,2) (7,3): 1 + 1 + 0 = 2, not better. So the prediction is
    # (above[5] + below[3]) / 2 = 1: the edge passes through column 4 of the rebuilt row -- where
    # the temporal clamp lets it: the rebuilt row's own field changes between prev and next.
    cur = np.zeros((3, 9, 4), np.float32)
    cur[0, 5:] = 1.0
    cur[2, 3:] = 1.0
    prev, nxt = cur.copy(), cur.copy()
    prev[1, :] = 0.0
    nxt[1, :] = 1.0          # row 1 went from black to white: |D - I| / 2 = 0.5, p2 = 0.5
    out = orc.deinterlace(cur, prev, nxt, TOP, TOP, orc.DEINT_YADIF, skip_spatial_check=True)
    # field == first_field: second neighbours are (prev, cur): D = prev(1) = 0, I = cur(1) = 0
    # -> p2 = 0, tdiff0 = 0; tdiff1 = (|A - F| + |B - G|) / 2 with A, B = prev rows 0, 2 = F, G: 0;
    # tdiff2 = (|K - F| + |G - L|) / 2 with K, L = next rows 0, 2: 0 -> clamp to 0
    assert out[1, 4, 0] == 0.0
    # the second field of the frame (field != first): second neighbours are (cur, next):
    # D = cur(1) = 0, I = next(1) = 1 -> p2 = 0.5, diff = 0.5: [0, 1] -> the prediction (1) stands
    out = orc.deinterlace(cur, prev, nxt, TOP, BOTTOM, orc.DEINT_YADIF, skip_spatial_check=True)
    assert out[1, 4, 0] == 1.0
    assert out[1, 0, 0] == 0.0 and out[1, 8, 0] == 1.0       # flat areas: the plain average


def test_bwdif_intra_is_its_cubic():
    # without the frame it needs, bwdif filters the current field only:
    # (5077 (c + e) - 981 (top + bottom)) / 8192 over rows -3 -1 +1 +3
    img = np.zeros((8, 1, 4), np.float32)
    img[:, 0, 0] = [0.0, 9.0, 0.25, 9.0, 0.5, 9.0, 1.0, 9.0]      # odd rows: the other field
    out = orc.deinterlace(img, None, None, TOP, TOP, orc.DEINT_BWDIF)
    want = np.float32(5077 / 8192) * np.float32(0.25 + 0.5) - np.float32(981 / 8192) * np.float32(0.0 + 1.0)
    assert out[3, 0, 0] == np.float32(want)
    # a ramp stays a ramp: (5077 - 981) * 2 / 8192 = 1
    ramp = np.zeros((12, 1, 4), np.float32)
    ramp[:, 0, 0] = np.arange(12, dtype=np.float32) / 16
    out = orc.deinterlace(ramp, None, None, TOP, TOP, orc.DEINT_BWDIF)
    assert np.allclose(out[3:-3:2, 0, 0], ramp[3:-3:2, 0, 0], rtol=0, atol=1e-7)


def test_bwdif_needs_the_frame_on_the_far_side_only():
    # first field of a frame: its second neighbours are (prev, cur) -> without prev: intra;
    # second field: (cur, next) -> without next: intra; the other missing frame is replaced by cur
    cur, other = frame(6, 10, 7), frame(6, 10, 8)
    intra = orc.deinterlace(cur, None, None, TOP, TOP, orc.DEINT_BWDIF)
    assert np.array_equal(orc.deinterlace(cur, None, other, TOP, TOP, orc.DEINT_BWDIF), intra)
    assert np.array_equal(orc.deinterlace(cur, other, None, TOP, BOTTOM, orc.DEINT_BWDIF), intra)
    full = orc.deinterlace(cur, other, None, TOP, TOP, orc.DEINT_BWDIF)
    assert not np.array_equal(full, intra)
    assert np.array_equal(full, orc.deinterlace(cur, other, cur, TOP, TOP, orc.DEINT_BWDIF))
