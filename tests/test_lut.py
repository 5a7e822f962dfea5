"""Custom LUTs: pl_lut_parse_cube, pl_shader_custom_lut and the renderer's
three LUT slots.

CPU: the reference's own .cube samples (src/tests/lut.c:6-60) with the values its parser
produces (num - min) / (max - min) in float, plus malformed files.
GPU: the CUSTOM_LUT op against the oracle's restatement of the GLSL (1D linear :731-745,
3D tetrahedral :762-809) -- bit-exact, there is no transcendental in it -- and pl_render_image
with LUTs against the same pipeline composed by hand."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
from libplacebo_amd import _capi as capi

CUBE_1D = """TITLE "1D LUT example"
LUT_1D_SIZE 11
# Random comment
0.0 0.0 0.0
0.1 0.1 0.1
0.2 0.2 0.2
0.3 0.3 0.3
0.4 0.4 0.4
0.5 0.5 0.5
0.6 0.6 0.6
0.7 0.7 0.7
0.8 0.8 0.8
0.9 0.9 0.9
0.10 0.10 0.10
"""

CUBE_3D = "LUT_3D_SIZE 3\nTITLE \"3D LUT example\"\n" + "".join(
    f"{r / 2:.1f} {g / 2:.1f} {b / 2:.1f}   \n" for b in range(3) for g in range(3) for r in range(3))

CUBE_DOMAIN = """LUT_1D_SIZE 3
TITLE "custom domain"
DOMAIN_MAX 255 255 255
0 0 0
128 128 128
255 255 255
"""


def parsed(text):
    p = pl.parse_cube(text)
    if p is None:
        return None
    lut = p.contents
    n = lut.size[0] * max(lut.size[1], 1) * max(lut.size[2], 1)
    data = np.ctypeslib.as_array(lut.data, shape=(n * 3,)).copy().reshape(n, 3)
    res = (list(lut.size), data, lut.signature)
    pl.lib().pl_lut_free(C.byref(p))
    assert not p
    return res


def test_parse_reference_samples(built):
    size, data, sig = parsed(CUBE_1D)
    assert size == [11, 0, 0]
    want = np.float32([0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.10])
    assert np.array_equal(data, np.repeat(want[:, None], 3, axis=1))
    assert sig != 0 and sig == parsed(CUBE_1D)[2] and sig != parsed(CUBE_1D + " ")[2]

    size, data, _ = parsed(CUBE_3D)
    assert size == [3, 3, 3]
    grid = np.float32([[r / 2, g / 2, b / 2] for b in range(3) for g in range(3) for r in range(3)])
    assert np.array_equal(data, grid)

    size, data, _ = parsed(CUBE_DOMAIN)
    assert size == [3, 0, 0]
    want = (np.float32([0, 128, 255]) - np.float32(0)) / (np.float32(255) - np.float32(0))
    assert np.array_equal(data[:, 0], want) and np.array_equal(data[:, 2], want)


def test_parse_rejects_malformed_files(built):
    assert parsed("TITLE \"no size\"\n0 0 0\n") is None
    assert parsed("LUT_1D_SIZE 3\n0 0 0\n1 1 1\n") is None                  # truncated
    assert parsed("LUT_1D_SIZE 2\n0 0 0\n1 x 1\n") is None                  # garbage
    assert parsed("LUT_3D_SIZE 0\n") is None
    assert parsed("LUT_3D_SIZE 2000\n") is None
    assert parsed("LUT_1D_SIZE 2\nDOMAIN_MIN 1 1 1\nDOMAIN_MAX 1 1 1\n0 0 0\n1 1 1\n") is None
    # scientific notation, signs, trailing data and DOMAIN_MIN are fine
    size, data, _ = parsed("LUT_1D_SIZE 2\nDOMAIN_MIN -1 -1 -1\n-1 -1.0 -1e0\n+1 1.0e0 1\n# tail")
    assert size == [2, 0, 0] and np.array_equal(data, np.float32([[0, 0, 0], [1, 1, 1]]))


# --------------------------------------------------------------------------- GPU

def run_lut(gpu, src, lut):
    from test_gpu_color import run_ops
    state = pl.ShaderObj()
    out = run_ops(gpu, src, lambda sh: sh.custom_lut(lut, state))
    state.destroy()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(2, 2, 2), (5, 7, 3), (17, 17, 17), (33, 33, 33)])
def test_3d_lut_tetrahedral_bit_exact(gpu, size):
    rng = np.random.default_rng(sum(size))
    n = size[0] * size[1] * size[2]
    data = rng.random((n, 3)).astype(np.float32)
    src = rng.random((40, 56, 4)).astype(np.float32)
    src[0, :8, :3] = [[0, 0, 0], [1, 1, 1], [1, 0, 0], [0, 1, 0], [0, 0, 1], [.5, .5, .5],
                      [-0.3, 1.7, 0.5], [0.25, 0.25, 0.75]]      # corners, ties, out of range
    got = run_lut(gpu, src, pl.custom_lut(data, size))
    ref = orc.custom_lut(src.copy(), data, size)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(got[..., 3], src[..., 3])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 11, 256, 4096])
def test_1d_lut_linear_bit_exact(gpu, n):
    rng = np.random.default_rng(n)
    data = np.sort(rng.random((n, 3)).astype(np.float32), axis=0)
    src = rng.random((32, 32, 4)).astype(np.float32) * 1.2 - 0.1
    got = run_lut(gpu, src, pl.custom_lut(data, (n,)))
    ref = orc.custom_lut(src.copy(), data, (n,))
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_identity_lut_and_shapers(gpu):
    rng = np.random.default_rng(1)
    src = rng.random((16, 16, 4)).astype(np.float32)
    size = (9, 9, 9)
    grid = np.float32([[r / 8, g / 8, b / 8] for b in range(9) for g in range(9) for r in range(9)])
    got = run_lut(gpu, src, pl.custom_lut(grid, size))
    assert np.abs(got - src).max() <= 2e-7       # the identity, up to the blend's rounding
    # shaper matrices: swap R and B going in, scale going out
    swap = [[0, 0, 1], [0, 1, 0], [1, 0, 0]]
    half = [[0.5, 0, 0], [0, 0.5, 0], [0, 0, 0.5]]
    got = run_lut(gpu, src, pl.custom_lut(grid, size, shaper_in=swap, shaper_out=half))
    want = src.copy()
    want[..., :3] = src[..., 2::-1] * 0.5
    assert np.abs(got - want).max() <= 2e-7
    # invalid dimensions fail the shader
    sh = gpu.begin()
    t = gpu.tex_create(4, 4, "rgba32f", np.zeros((4, 4, 4), np.float32))
    assert sh.sample("nearest", t)
    state = pl.ShaderObj()
    sh.custom_lut(pl.custom_lut(grid[:6], (3, 2)), state)
    assert sh.failed()
    sh.abort(); state.destroy(); t.destroy()


@pytest.mark.gpu
def test_renderer_lut_slots(gpu):
    """params->lut (NATIVE, between image and target colour space), image LUT (NATIVE: on the
    raw samples) and target LUT (NATIVE: after encoding) against hand-composed pipelines."""
    import util
    sw, sh_ = 48, 32
    img = util.chirp_rgba16(sw, sh_)
    src = gpu.tex_create(sw, sh_, "rgba16", img)
    dst = gpu.tex_create(sw, sh_, "rgba16")
    rng = np.random.default_rng(9)
    size = (9, 9, 9)
    data = np.clip(np.float32([[r / 8, g / 8, b / 8] for b in range(9) for g in range(9)
                               for r in range(9)]) ** 1.3 + rng.normal(0, 0.01, (729, 3)), 0, 1)
    lut = pl.custom_lut(data.astype(np.float32), size)
    f = orc.tex_decode(img, "rgba16")
    f[..., 3] = 1.0
    want = orc.tex_encode(orc.custom_lut(f.copy(), data.astype(np.float32), size), "rgba16")

    def render(image, target, params):
        rr = pl.Renderer(gpu)
        assert rr.render(image, target, params), gpu.messages[-4:]
        assert rr.errors() == 0
        out = dst.download()
        rr.destroy()
        return out

    plain = render(pl.frame(src, components=3), pl.frame(dst), pl.render_params("fast"))
    assert np.array_equal(plain[..., :3], img[..., :3])
    # 1. main LUT, PL_LUT_NATIVE, same colour space on both sides: image -> LUT -> target
    params = pl.render_params("fast", lut=lut, lut_type=pl.LUT_NATIVE)
    got = render(pl.frame(src, components=3), pl.frame(dst), params)
    assert np.array_equal(got, want)
    # 2. the same LUT attached to the image (acts on the raw samples before decoding)
    image = pl.frame(src, components=3)
    image.lut = C.pointer(lut)
    image.lut_type = pl.LUT_NATIVE
    got = render(image, pl.frame(dst), pl.render_params("fast"))
    assert np.array_equal(got, want)
    # 3. ... and to the target (acts on the encoded samples)
    target = pl.frame(dst)
    target.lut = C.pointer(lut)
    target.lut_type = pl.LUT_NATIVE
    got = render(pl.frame(src, components=3), target, pl.render_params("fast"))
    assert np.array_equal(got, want)
    # 4. a CONVERSION LUT on the image replaces the YCbCr decoding
    ycc = pl.frame(src, components=3, repr_=pl.color_repr("bt709", "limited"))
    ycc.lut = C.pointer(lut)
    ycc.lut_type = pl.LUT_CONVERSION
    got = render(ycc, pl.frame(dst), pl.render_params("fast"))
    assert np.array_equal(got, want)        # no matrix: the LUT output is taken as RGB
    src.destroy(); dst.destroy()


@pytest.mark.gpu
def test_renderer_main_lut_types(gpu):
    """params->lut as PL_LUT_CONVERSION (replaces the image -> target conversion) and
    PL_LUT_NORMALIZED (applied to linear light, scaled by the nominal peak)."""
    import util
    from test_gpu_color import nominal, luma_coeffs
    sw, sh_ = 48, 32
    img = util.chirp_rgba16(sw, sh_)
    src = gpu.tex_create(sw, sh_, "rgba16", img)
    dst = gpu.tex_create(sw, sh_, "rgba16")
    size = (9, 9, 9)
    grid = np.float32([[r / 8, g / 8, b / 8] for b in range(9) for g in range(9) for r in range(9)])
    data = (grid ** 1.2).astype(np.float32)
    lut = pl.custom_lut(data, size)
    csp = pl.color_space("bt709", "bt1886")

    def render(params):
        rr = pl.Renderer(gpu)
        assert rr.render(pl.frame(src, components=3, color=csp), pl.frame(dst, color=csp), params)
        assert rr.errors() == 0
        out = dst.download()
        rr.destroy()
        return out

    f = orc.tex_decode(img, "rgba16")
    f[..., 3] = 1.0
    # CONVERSION: the LUT output is the target signal
    got = render(pl.render_params("fast", lut=lut, lut_type=pl.LUT_CONVERSION))
    want = orc.tex_encode(orc.custom_lut(f.copy(), data, size), "rgba16")
    assert np.array_equal(got, want)
    # NORMALIZED: linearize -> / peak -> LUT -> * peak -> delinearize (SDR: peak = 1)
    got = render(pl.render_params("fast", lut=lut, lut_type=pl.LUT_NORMALIZED))
    c = pl.color_space("bt709", "bt1886")
    pl.lib().pl_color_space_infer(C.byref(c))
    mn, mx = nominal(c)
    luma = luma_coeffs(c.primaries)
    lin = orc.linearize(f.copy(), int(c.transfer), mn, mx, luma)
    lin = orc.custom_lut(lin, data, size)
    want = orc.tex_encode(orc.delinearize(lin, int(c.transfer), mn, mx, luma), "rgba16")
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))[..., :3]
    assert d.max() <= 3, int(d.max())
    # identity LUT in NORMALIZED mode: the frame survives the round trip
    ident = pl.custom_lut(grid, size)
    got = render(pl.render_params("fast", lut=ident, lut_type=pl.LUT_NORMALIZED))
    d = np.abs(got.astype(np.int64) - img.astype(np.int64))[..., :3]
    assert d.max() <= 3, int(d.max())
    src.destroy(); dst.destroy()
