"""pl_shader_dither, every method and option (reference src/shaders/dithering.c:109-273):
blue noise / Bayer LUTs of several sizes, the 16x16 bit-twiddled ordered pattern, white noise
(pcg3d, src/shaders.c:965-998), temporal rotation / reseeding over eight frame indices, and the
gamma-aware path for depths <= 4. Bit-exact against the oracle except where pow() decides
(gamma-aware path): there a vanishing fraction of pixels may land on the neighbouring level."""
import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util

pytestmark = pytest.mark.gpu

W, H = 200, 120


def ramp_image(w=W, h=H, seed=5):
    """smooth ramps + noise, float32 in [0, 1] (+ a few values outside it)"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, 4), np.float32)
    img[..., 0] = x / (w - 1)
    img[..., 1] = y / (h - 1)
    img[..., 2] = rng.random((h, w), np.float32)
    img[..., 3] = 0.5 + 0.5 * np.sin(0.05 * (x + y))
    img[0, :8, :3] = [-0.01, 1.01, 0.5]
    return img


def matrix_for(method, lut_size):
    if method == pl.DITHER_BLUE_NOISE:
        return util.blue_noise(pl, 1 << lut_size)
    if method == pl.DITHER_ORDERED_LUT:
        m = np.empty((1 << lut_size) ** 2, np.float32)
        pl.lib().pl_generate_bayer_matrix(m.ctypes.data, 1 << lut_size)
        return m.reshape(1 << lut_size, 1 << lut_size)
    return None


ORC_METHOD = {pl.DITHER_BLUE_NOISE: 0, pl.DITHER_ORDERED_LUT: 0, pl.DITHER_ORDERED_FIXED: 1,
              pl.DITHER_WHITE_NOISE: 2}


def run_dither(gpu, img, depth, method, lut_size=6, temporal=False, transfer=0, frames=0,
               state=None):
    """direct fetch of an rgba32f texture + dither -> rgba32f; returns (image, listing fields)"""
    h, w = img.shape[:2]
    src = gpu.tex_create(w, h, "rgba32f", img)
    dst = gpu.tex_create(w, h, "rgba32f")
    own = state is None
    state = state or pl.ShaderObj()
    for _ in range(frames):
        gpu.reset_frame()
    s = gpu.begin()
    assert s.sample("direct", src)
    util.srand(1)
    s.dither(depth, state if method in (pl.DITHER_BLUE_NOISE, pl.DITHER_ORDERED_LUT) else None,
             method=method, lut_size=lut_size, temporal=temporal, transfer=transfer)
    line = [l for l in s.listing().splitlines() if l.startswith("dither(")][0]
    info = dict(kv.split("=") for kv in line[len("dither("):-1].split(", "))
    assert s.finish(dst), gpu.messages[-3:]
    out = dst.download()
    src.destroy(); dst.destroy()
    if own:
        state.destroy()
    return out, info


@pytest.mark.parametrize("depth", [8, 10, 6])
@pytest.mark.parametrize("method,lut_size", [
    (pl.DITHER_BLUE_NOISE, 6), (pl.DITHER_BLUE_NOISE, 4), (pl.DITHER_ORDERED_LUT, 6),
    (pl.DITHER_ORDERED_LUT, 3), (pl.DITHER_ORDERED_FIXED, 0), (pl.DITHER_WHITE_NOISE, 0)])
def test_dither_methods_bit_exact(gpu, depth, method, lut_size):
    img = ramp_image()
    got, info = run_dither(gpu, img, depth, method, lut_size or 6)
    assert int(info["method"]) == method
    ref = img.copy()
    orc.dither(ref, matrix_for(method, lut_size), depth, method=ORC_METHOD[method])
    assert np.array_equal(got, ref), util.diff_stats(got.view(np.int32), ref.view(np.int32))
    # and it is a dither: every in-range value sits on a level of the target depth
    lv = got[4:, :, :3] * ((1 << depth) - 1)
    assert np.abs(lv - np.rint(lv)).max() < 1e-3
    assert 0.2 < (got[4:, :, 2] > img[4:, :, 2]).mean() < 0.8


def test_white_noise_plane_covers_the_padded_rows(gpu):
    """White noise is evaluated into a plane `side` floats wide (dispatch.c: realize_white_noise).
    The kernels fetch the dither value of the lanes that pad the rect to whole tiles BEFORE the
    store guard drops them, so the plane must reach past the pass' height: 1920 x 804 (804 is not
    a multiple of the 8-row tiles; side = 2048, and a plane of exactly 804 rows would end on a page
    boundary) and an EWA pass whose 64-row tiles overhang by 60 rows. Bit-exact as ever."""
    w, h = 1920, 804
    img = ramp_image(w, h)
    got, _ = run_dither(gpu, img, 8, pl.DITHER_WHITE_NOISE)
    ref = img.copy()
    orc.dither(ref, None, 8, method=ORC_METHOD[pl.DITHER_WHITE_NOISE])
    assert np.array_equal(got, ref)
    # behind the polar kernels (2x: k_polar_mx / k_polar_pp tiles, 2:1 down: k_polar_mxd)
    from libplacebo_amd import _capi as capi
    for (sw, sh), (dw, dh), key in (((480, 201), (960, 402), "upscaler"), ((1920, 804), (960, 402), "downscaler")):
        src16 = util.chirp_rgba16(sw, sh)
        for mfma in ("1", "0"):
            import os
            old = os.environ.get("PL_HIP_POLAR_MFMA")
            os.environ["PL_HIP_POLAR_MFMA"] = mfma
            try:
                with pl.HipGpu(0) as g:
                    src = g.tex_create(sw, sh, "rgba16", src16)
                    dst = g.tex_create(dw, dh, "rgba16")
                    rr = pl.Renderer(g)
                    params = pl.render_params("fast", dither_params=capi.DitherParams(
                        method=pl.DITHER_WHITE_NOISE, lut_size=6, transfer=0),
                        disable_linear_scaling=True, **{key: pl.filter_config("ewa_lanczos")})
                    tgt = pl.frame(dst, repr_=pl.color_repr("rgb", "full", sample_depth=16, color_depth=8, bit_shift=8))
                    assert rr.render(pl.frame(src, components=3), tgt, params), g.messages[-3:]
                    assert rr.errors() == 0
                    out = dst.download()
                    low = out[..., :3] & 0xff      # 8-bit codes in the upper byte (+-1: k / 255 / scale is not exact)
                    assert out[..., :3].std() > 1000 and np.all((low <= 1) | (low == 255))
                    rr.destroy(); src.destroy(); dst.destroy()
            finally:
                if old is None:
                    os.environ.pop("PL_HIP_POLAR_MFMA", None)
                else:
                    os.environ["PL_HIP_POLAR_MFMA"] = old


@pytest.mark.parametrize("method", [pl.DITHER_BLUE_NOISE, pl.DITHER_ORDERED_LUT,
                                    pl.DITHER_ORDERED_FIXED, pl.DITHER_WHITE_NOISE])
def test_dither_temporal(gpu, method):
    """temporal: the pattern is rotated / mirrored by the frame index mod 8 (LUT, ordered) or
    reseeded by the frame index (white noise); frames differ from one another"""
    img = ramp_image()
    seen = []
    for frame in range(9):
        got, info = run_dither(gpu, img, 8, method, temporal=True, frames=1)
        index = int(info["index"])
        ref = img.copy()
        orc.dither(ref, matrix_for(method, 6), 8, method=ORC_METHOD[method], temporal=True,
                   frame_index=index)
        assert np.array_equal(got, ref), (frame, index, util.diff_stats(got.view(np.int32),
                                                                         ref.view(np.int32)))
        seen.append(got)
    assert sum(not np.array_equal(seen[0], s) for s in seen[1:8]) >= 6
    # non-temporal white noise does not depend on the frame index
    if method == pl.DITHER_WHITE_NOISE:
        a, _ = run_dither(gpu, img, 8, method, frames=1)
        b, _ = run_dither(gpu, img, 8, method, frames=1)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("depth", [1, 2, 3, 4])
@pytest.mark.parametrize("trc", ["bt1886", "srgb", "linear", "gamma22"])
@pytest.mark.parametrize("method", [pl.DITHER_BLUE_NOISE, pl.DITHER_ORDERED_FIXED,
                                    pl.DITHER_WHITE_NOISE])
def test_dither_low_depth_gamma_aware(gpu, depth, trc, method):
    """depth <= 4 with a non-linear transfer: levels are chosen in linear light
    (dithering.c:241-266); `linear` takes the plain floor path"""
    img = np.clip(ramp_image(), 0.0, 1.0)
    got, info = run_dither(gpu, img, depth, method, transfer=pl.TRC[trc])
    gamma = float(info["gamma"])
    assert (gamma == 1.0) == (trc == "linear")
    ref = img.copy()
    orc.dither(ref, matrix_for(method, 6), depth, method=ORC_METHOD[method], gamma=gamma)
    scale = (1 << depth) - 1
    same = got == ref
    if gamma == 1.0:
        assert same.all()
    else:
        # pow() decides `offset > bias`: device pow vs libm may put a pixel whose offset equals
        # the bias to ~1e-7 on the other level
        assert same.mean() >= 0.9995, same.mean()
        assert np.abs(got - ref).max() <= 1.0 / scale + 1e-6
        # gamma-aware means: the mean LINEAR light of a flat mid-grey patch is preserved
        flat = np.full((64, 64, 4), 0.4, np.float32)
        out, _ = run_dither(gpu, flat, depth, method, transfer=pl.TRC[trc])
        assert abs((out[..., 0].astype(np.float64) ** gamma).mean() - 0.4 ** gamma) < 0.02
    lv = got * scale
    assert np.abs(lv - np.rint(lv)).max() < 1e-4


@pytest.mark.parametrize("method,temporal", [(pl.DITHER_ORDERED_FIXED, False),
                                             (pl.DITHER_WHITE_NOISE, True),
                                             (pl.DITHER_BLUE_NOISE, True),
                                             (pl.DITHER_ORDERED_LUT, False)])
@pytest.mark.parametrize("scaler", ["polar", "ortho", "bilinear"])
def test_dither_behind_every_scaler_kernel(gpu, method, temporal, scaler):
    """the fused scaler kernels carry their own fragment-coordinate bookkeeping: the dither
    pattern must land on the same pixels behind each of them"""
    sw, sh, dw, dh = 96, 54, 192, 108
    srcimg = util.chirp_rgba16(sw, sh)
    t = gpu.tex_create(sw, sh, "rgba16", srcimg)
    d = gpu.tex_create(dw, dh, "rgba16")
    lut, state = pl.ShaderObj(), pl.ShaderObj()
    gpu.reset_frame(); gpu.reset_frame(); gpu.reset_frame()
    s = gpu.begin()
    tex = orc.tex_decode(srcimg, "rgba16")
    if scaler == "polar":
        assert s.sample_polar(t, pl.filter_config("ewa_lanczos"), lut, new_w=dw, new_h=dh)
        w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
        ref = orc.sample_polar(tex, w, r, rz, dw, dh)
    elif scaler == "bilinear":
        assert s.sample("bilinear", t, new_w=dw, new_h=dh)
        ref = orc.sample_simple(tex, orc.S_BILINEAR, dw, dh)
    else:
        assert s.sample("bicubic", t, new_w=dw, new_h=dh)
        ref = orc.sample_simple(tex, orc.S_BICUBIC, dw, dh)
    util.srand(1)
    s.dither(8, state if method in (pl.DITHER_BLUE_NOISE, pl.DITHER_ORDERED_LUT) else None,
             method=method, temporal=temporal, transfer=0)
    line = [l for l in s.listing().splitlines() if l.startswith("dither(")][0]
    index = int(line.split("index=")[1].rstrip(")"))
    assert s.finish(d), gpu.messages[-3:]
    got = d.download()
    orc.dither(ref, matrix_for(method, 6), 8, method=ORC_METHOD[method], temporal=temporal,
               frame_index=index)
    assert np.array_equal(got, orc.tex_encode(ref, "rgba16")), util.diff_stats(
        got, orc.tex_encode(ref, "rgba16"))
    for o in (t, d, lut, state):
        o.destroy()
