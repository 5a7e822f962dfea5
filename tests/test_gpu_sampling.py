"""HIP sampling / EWA / dither stages vs the CPU oracle, through the C-ABI.

Bar: these stages contain no transcendental, so the HIP output must
be *bit-identical* to the oracle after the target format's conversion.
"""
import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util

pytestmark = pytest.mark.gpu


def run_simple(g, kind, src, fmt, dw, dh, out_fmt="rgba16", rect=None, **kw):
    h, w = src.shape[:2]
    t = g.tex_create(w, h, fmt, src)
    d = g.tex_create(dw, dh, out_fmt)
    sh = g.begin()
    assert sh.sample(kind, t, new_w=dw, new_h=dh, rect=rect, **kw), g.messages[-3:]
    assert sh.finish(d), g.messages[-3:]
    out = d.download()
    t.destroy(); d.destroy()
    return out


KINDS = {"nearest": orc.S_NEAREST, "bilinear": orc.S_BILINEAR, "bicubic": orc.S_BICUBIC,
         "hermite": orc.S_HERMITE, "oversample": orc.S_OVERSAMPLE}


@pytest.mark.parametrize("kind", list(KINDS))
@pytest.mark.parametrize("shape", [((64, 48), (128, 96)), ((61, 37), (97, 111)),
                                   ((80, 60), (40, 30))])
def test_simple_samplers_bit_exact(gpu, kind, shape):
    (sw, sh_), (dw, dh) = shape
    src = util.random_rgba16(sw, sh_, seed=sw + dh)
    got = run_simple(gpu, kind, src, "rgba16", dw, dh)
    ref = orc.tex_encode(orc.sample_simple(orc.tex_decode(src, "rgba16"), KINDS[kind], dw, dh),
                         "rgba16")
    mx, n = util.diff_stats(got, ref)
    assert mx == 0, f"{kind} {shape}: max diff {mx}, {n} texels"


def test_gaussian_within_1lsb(gpu):
    # exp() on the GPU is the native v_exp_f32; the oracle uses libm
    src = util.random_rgba16(64, 48)
    got = run_simple(gpu, "gaussian", src, "rgba16", 128, 96)
    ref = orc.tex_encode(orc.sample_simple(orc.tex_decode(src, "rgba16"), orc.S_GAUSSIAN, 128, 96),
                         "rgba16")
    mx, _ = util.diff_stats(got, ref)
    assert mx <= 1


@pytest.mark.parametrize("fmt", ["r8", "rg8", "rgba8", "r16", "rg16", "rgba16",
                                 "r16hf", "rgba16hf", "r32f", "rgba32f"])
def test_formats_roundtrip_and_bilinear(gpu, fmt):
    dt, nc = pl._FMT_DTYPES[fmt]
    rng = np.random.default_rng(7)
    if np.issubdtype(dt, np.integer):
        src = rng.integers(0, np.iinfo(dt).max + 1, (33, 47, nc)).astype(dt)
    else:
        src = rng.random((33, 47, nc)).astype(dt)
    got = run_simple(gpu, "bilinear", src, fmt, 94, 66, out_fmt=fmt)
    ref = orc.tex_encode(orc.sample_simple(orc.tex_decode(src, fmt), orc.S_BILINEAR, 94, 66), fmt)
    assert np.array_equal(got.view(np.uint8), ref.view(np.uint8))


@pytest.mark.parametrize("rect", [(0, 0, 64, 48), (64, 0, 0, 48), (0, 48, 64, 0),
                                  (3.25, 1.5, 60.75, 40.0), (-4, -4, 70, 50)])
@pytest.mark.parametrize("mode", [pl.ADDRESS_CLAMP, pl.ADDRESS_REPEAT, pl.ADDRESS_MIRROR])
def test_rects_flips_address_modes(gpu, rect, mode):
    src = util.random_rgba16(64, 48, seed=3)
    got = run_simple(gpu, "bilinear", src, "rgba16", 100, 70, rect=rect, address_mode=mode)
    ref = orc.tex_encode(orc.sample_simple(orc.tex_decode(src, "rgba16"), orc.S_BILINEAR, 100, 70,
                                           rect=rect, address_mode=mode), "rgba16")
    assert np.array_equal(got, ref)


def polar_pipeline(g, src, dw, dh, cfg_name="ewa_lanczos", comps=3, dither_depth=None,
                   out_fmt="rgba16", antiring=0.0, rect=None, **kw):
    """PASS A (texture -> rgba16hf FBO) + polar pass [+ dither], like the
    reference's pass_scale_main (renderer.c:2064-2075)."""
    h, w = src.shape[:2]
    t = g.tex_create(w, h, "rgba16", src)
    fbo = g.tex_create(w, h, "rgba16hf")
    d = g.tex_create(dw, dh, out_fmt)
    lut, dstate = pl.ShaderObj(), pl.ShaderObj()
    a = g.begin()
    assert a.sample("direct", t)
    assert a.finish(fbo)
    b = g.begin()
    cfg = pl.filter_config(cfg_name)
    assert b.sample_polar(fbo, cfg, lut, antiring=antiring, new_w=dw, new_h=dh, rect=rect,
                          components=comps, **kw), g.messages[-3:]
    listing = None
    if dither_depth:
        util.srand(1)
        b.dither(dither_depth, dstate)
    assert b.finish(d), g.messages[-3:]
    out = d.download()
    for o in (t, fbo, d, lut, dstate):
        o.destroy()
    return out


def polar_oracle(src, dw, dh, f, comps=3, dither_depth=None, out_fmt="rgba16", antiring=0.0,
                 rect=None, gather=None, matrix=None):
    tex = orc.tex_decode(src, "rgba16")
    h_, w_ = tex.shape[:2]
    # PASS A: pl_shader_sample_direct -> rgba16hf FBO
    img = orc.op_quant_f16(orc.sample_simple(tex, orc.S_BILINEAR, w_, h_))
    w, radius, radius_zero = orc.filter_generate_polar(f)
    gather = (not radius < 6.0) if gather is None else gather
    out = orc.sample_polar(img, w, radius, radius_zero, dw, dh, mask=(1 << comps) - 1,
                           antiring=antiring, rect=rect, gather_order=gather)
    if dither_depth:
        orc.dither(out, matrix, dither_depth)
    return orc.tex_encode(out, out_fmt)


@pytest.mark.parametrize("shape", [((96, 64), (192, 128)), ((50, 41), (133, 87)),
                                   ((64, 64), (96, 96))])
@pytest.mark.parametrize("comps", [3, 4])
def test_polar_upscale_bit_exact(gpu, shape, comps):
    (sw, sh_), (dw, dh) = shape
    src = util.random_rgba16(sw, sh_, seed=11)
    got = polar_pipeline(gpu, src, dw, dh, comps=comps)
    ref = polar_oracle(src, dw, dh, orc.ewa_lanczos(), comps=comps)
    util.assert_polar_equal(got, ref)


def test_polar_downscale_widened_gather_order(gpu):
    # ratio 0.5 -> blur 2 -> radius 6.3 >= 6: the reference switches to the
    # gather formulation, whose tap order we follow (sampling.c:671-674)
    src = util.random_rgba16(128, 96, seed=5)
    got = polar_pipeline(gpu, src, 64, 48)
    ref = polar_oracle(src, 64, 48, orc.ewa_lanczos(blur=2.0))
    util.assert_polar_equal(got, ref)


def test_polar_antiring(gpu):
    src = util.random_rgba16(64, 48, seed=9)
    got = polar_pipeline(gpu, src, 128, 96, antiring=0.8)
    ref = polar_oracle(src, 128, 96, orc.ewa_lanczos(), antiring=0.8)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("rect", [(96, 0, 0, 64), (0, 64, 96, 0), (10.5, 7.25, 80.0, 60.5)])
def test_polar_rects(gpu, rect):
    src = util.random_rgba16(96, 64, seed=13)
    got = polar_pipeline(gpu, src, 160, 120, rect=rect)
    ref = polar_oracle(src, 160, 120, orc.ewa_lanczos(), rect=rect)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("depth,fmt", [(8, "rgba8"), (10, "rgba16"), (6, "rgba8")])
def test_polar_plus_blue_noise_dither(gpu, depth, fmt):
    src = util.chirp_rgba16(120, 68)
    mat = util.blue_noise(pl)
    got = polar_pipeline(gpu, src, 240, 136, dither_depth=depth, out_fmt=fmt)
    ref = polar_oracle(src, 240, 136, orc.ewa_lanczos(), dither_depth=depth, out_fmt=fmt,
                       matrix=mat)
    # (one step of the target's depth in its container: rgba8 holds 8 and 6 bits, rgba16 10 bits,
    # unshifted -- pl_shader_dither scales to the full container)
    util.assert_polar_equal(got, ref, step={8: 1, 10: 65, 6: 5}[depth])
