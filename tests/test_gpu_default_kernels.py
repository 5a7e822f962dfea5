"""The library's DEFAULT polar kernels -- k_polar_mx (exact 2x upscale) and k_polar_mxd (exact 2 : 1
downscale), the contraction on the f16 matrix pipe -- held to k_polar_pp (the sequential-fma
kernel that is bit-exact with the oracle) AND to the oracle itself, through pl_render_image, under
the configurations applications actually run:

  * pl_render_default_params + upscaler = ewa_lanczos on SDR video: linearize + sigmoidize fused
    into the tile staging (the non-PRE_LITE variant), unsigmoidize + delinearize in the CHAIN
    epilogue (src/renderer.c:1997-2042 decides when sigmoid applies, src/shaders/colorspace.c:
    851-894 is the pair) -- into rgba16, 10-bit-in-16 and rgba8 targets;
  * pl_render_high_quality_params at 2x;
  * the LITE and FULL interpreter epilogues behind a plain encode;
  * ewa_lanczos 4K -> 1080p + 10-bit dither (k_polar_mxd's rgba16 path) through the renderer.

tests/conftest.py pins PL_HIP_POLAR_MFMA=0 for the rest of the suite; every render here says which
kernel it wants and checks the backend's log for the one it got.

Statement, in codes of 16 bits BEFORE the dither: |matrix pipe - k_polar_pp| <= 1, identical on the
bulk; |k_polar_pp - oracle| <= 1 where a transfer function (native v_exp / v_log against libm) is
involved, 0 where none is. After the dither: every sample is the dither of the frame's OWN
pre-dither value with the oracle's matrix cell (dither index path bit-exact), as in
tests/test_gpu_metric.py.

Behind an epilogue that amplifies near black. A pass that scales in LINEAR / sigmoidized light
continues with unsigmoidize (slope up to 17 at y = 0.0076) and the display gamma's inverse
((1 / 2.4) x^-0.58, 340 at x = 1e-5); VERDICT r03 asked whether the contraction's ~1e-6 of full
scale survives that. It does: its error scales with the tap products, which are small where the
output is dark -- <= 1 code at 1080p -> 4K on white noise and on a dark field with isolated
full-scale texels ("stars": every dark output within three texels of a star is a small difference
of large products), test_matrix_pipe_behind_sigmoid. So the default preset stays on the matrix pipe.
"""
import ctypes as C
import os

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util
from libplacebo_amd import _capi as capi
from test_gpu_metric import dither_consistency

pytestmark = pytest.mark.gpu

P1080 = (1920, 1080)


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def blue():
    return capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0)


def repr_bits(depth, container):
    return pl.color_repr("rgb", "full", sample_depth=container, color_depth=depth,
                         bit_shift=container - depth)


LAST_LOG = []


def render(img, dw, dh, params, mfma, dst_fmt="rgba16", target_repr=None, src_fmt="rgba16",
           image_color=None, target_color=None, extra_env=None):
    """one frame through pl_render_image on a fresh backend; returns (frame, used the matrix pipe)"""
    e = {"PL_HIP_POLAR_MFMA": "1" if mfma else "0", **(extra_env or {})}
    with env(**e):
        with pl.HipGpu(0, log_level=5) as g:
            sh, sw = img.shape[:2]
            src = g.tex_create(sw, sh, src_fmt, img)
            dst = g.tex_create(dw, dh, dst_fmt)
            rr = pl.Renderer(g)
            util.srand(1)
            image = pl.frame(src, components=3, color=image_color)
            target = pl.frame(dst, repr_=target_repr, color=target_color)
            assert rr.render(image, target, params), g.messages[-4:]
            assert rr.errors() == 0
            out = dst.download()
            used = any("polar on the matrix pipe" in m for _, m in g.messages)
            LAST_LOG[:] = [m for _, m in g.messages if "matrix" in m or "polar" in m]
            rr.destroy(); src.destroy(); dst.destroy()
    return out, used


def codes(a, b):
    return np.abs(a[..., :3].astype(np.int64) - b[..., :3].astype(np.int64))


def report(tag, d):
    print("%s: |diff| max %d, > 0 on %.4f, > 1 on %.2e of the samples" %
          (tag, int(d.max()), float((d > 0).mean()), float((d > 1).mean())))


def content(kind, w, h):
    if kind == "stars":
        # the worst case for an epilogue that amplifies near black: a dark, slightly noisy field
        # (codes 600 .. 3000 of 65535) with isolated full-scale texels -- every dark output within
        # three texels of a star is a small difference of large tap products
        rng = np.random.default_rng(8)
        img = rng.integers(600, 3000, (h, w, 4)).astype(np.uint16)
        img[rng.random((h, w)) < 0.01] = 65535
        img[..., 3] = 65535
        return img
    return util.chirp_rgba16(w, h) if kind == "chirp" else util.random_rgba16(w, h, seed=5)


# ---- oracle: pl_render_default_params + ewa_lanczos on BT.709 / BT.1886 video, 2x -----------------
def oracle_pass_a(img, csp):
    """decode -> linearize -> sigmoidize -> f16: the reference's PASS A for an SDR upscale
    (renderer.c:1997-2042, colorspace.c:589-720, 851-872)"""
    from test_gpu_color import luma_coeffs, nominal
    mn, mx = nominal(csp)
    a = orc.tex_decode(img, "rgba16")
    a[..., 3] = 1.0
    orc.linearize(a, int(csp.transfer), mn, mx, luma_coeffs(csp.primaries))
    orc.sigmoid(a)
    return orc.op_quant_f16(a)


def oracle_pass_b(fbo, dw, dh, csp):
    """rgba16hf FBO -> EWA-Lanczos -> unsigmoidize -> delinearize (colorspace.c:874-894, 722-850)"""
    from test_gpu_color import luma_coeffs, nominal
    mn, mx = nominal(csp)
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
    b = orc.sample_polar(np.ascontiguousarray(fbo, np.float32), w, r, rz, dw, dh, mask=0x7)
    orc.sigmoid(b, inverse=True)
    orc.delinearize(b, int(csp.transfer), mn, mx, luma_coeffs(csp.primaries))
    return b


def pass_a_by_hand(img, csp):
    """the same PASS A recorded by hand on the GPU (the fused kernels run exactly these ops per
    source texel while staging their tile: tests/test_gpu_kernel_variants.py)"""
    h, w = img.shape[:2]
    with pl.HipGpu(0) as g:
        src = g.tex_create(w, h, "rgba16", img)
        fbo = g.tex_create(w, h, "rgba16hf")
        a = g.begin()
        assert a.sample("direct", src, components=3)
        a.linearize(csp)
        a.sigmoidize()
        assert a.finish(fbo), g.messages[-3:]
        out = fbo.download()
        src.destroy(); fbo.destroy()
    return out


def inferred(csp):
    out = capi.ColorSpace()
    C.memmove(C.byref(out), C.byref(csp), C.sizeof(out))
    pl.lib().pl_color_space_infer(C.byref(out))
    return out


@pytest.mark.parametrize("size", [(166, 93), P1080])
@pytest.mark.parametrize("kind", ["chirp", "noise"])
def test_default_preset_sdr_2x_ewa(size, kind):
    """pl_render_default_params, upscaler = ewa_lanczos, BT.709 / BT.1886 in and out: the upscale
    runs in sigmoidized linear light, on the matrix pipe (non-PRE_LITE staging, CHAIN epilogue).
    Held to k_polar_pp and to the oracle: <= 1 code of 16 bits before the dither, the dither
    index path exact, rgba8 <= 1."""
    sw, sh = size
    dw, dh = 2 * sw, 2 * sh
    img = content(kind, sw, sh)
    csp = pl.color_space("bt709", "bt1886")
    kw = dict(image_color=csp, target_color=csp)
    nodither = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=None)
    dithered = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=blue(),
                                disable_dither_gamma_correction=True)

    # the library as shipped (PL_HIP_POLAR_MFMA unset = 1): which kernel, and what it renders
    pre, used = render(img, dw, dh, nodither, True, **kw)
    assert used
    pre_pp, used_pp = render(img, dw, dh, nodither, False, **kw)
    assert not used_pp
    d = codes(pre, pre_pp)
    report("default preset SDR 2x %dx%d %s, matrix pipe vs k_polar_pp" % (sw, sh, kind), d)
    assert d.max() <= 1 and (d > 0).mean() < 0.05, (d.max(), (d > 0).mean())

    # ---- stage by stage (each stage fed the GPU's output of the previous one), then end to end.
    # PASS A ends in an rgba16hf store: where native pow and libm put a value on different sides
    # of an f16 rounding boundary (about one texel in 10^4) the intermediate differs by an f16 ulp
    # -- 5e-4 relative, a dozen 16-bit codes under a centre tap -- so "<= 1 code" is a statement
    # about each stage on the SAME input, and end to end about all but those pixels.
    csp_i = inferred(csp)
    fbo = pass_a_by_hand(img, csp_i)
    want = oracle_pass_a(img, csp_i).astype(np.float16)
    ulps = np.abs(fbo[..., :3].view(np.uint16).astype(np.int64) - want[..., :3].view(np.uint16))
    print("PASS A (linearize + sigmoidize -> f16) %dx%d %s: f16 codes differ on %.2e of the samples, "
          "max %d ulp" % (sw, sh, kind, (ulps > 0).mean(), ulps.max()))
    assert ulps.max() <= 1 and (ulps > 0).mean() < 2e-3
    ref_b = oracle_pass_b(fbo.astype(np.float32), dw, dh, csp_i)
    ref_b16 = orc.tex_encode(ref_b, "rgba16")
    for tag, frame in (("k_polar_pp", pre_pp), ("matrix pipe", pre)):
        d = codes(frame, ref_b16)
        report("   PASS B, %s vs oracle on the GPU's intermediate" % tag, d)
        assert d.max() <= 1 and (d > 0).mean() < 0.05, (tag, d.max(), (d > 0).mean())
    ref = oracle_pass_b(want.astype(np.float32), dw, dh, csp_i)
    ref16 = orc.tex_encode(ref, "rgba16")
    d = codes(pre, ref16)
    report("   ... end to end vs oracle", d)
    print("   quantiles (99.9, 99.99, 100 %%): %s" % np.quantile(d, (0.999, 0.9999, 1.0)))
    assert (d > 1).mean() <= 16 * (ulps > 0).mean() + 1e-6, ((d > 1).mean(), (ulps > 0).mean())

    # 10 bit in 16, blue-noise dither: the index path
    got10, _ = render(img, dw, dh, dithered, True, target_repr=repr_bits(10, 16), **kw)
    matrix = util.blue_noise(pl)
    assert dither_consistency(got10, pre, matrix) == 0.0
    assert dither_consistency(got10, pre, np.roll(matrix, 1, axis=1)) > 0.2

    # rgba8 (no dither): one 8-bit step at most, and only where the 16-bit values straddle a
    # rounding boundary
    got8, _ = render(img, dw, dh, nodither, True, dst_fmt="rgba8", **kw)
    ref8 = orc.tex_encode(ref_b, "rgba8")
    d8 = codes(got8, ref8)
    report("   ... rgba8 vs oracle (on the GPU's intermediate)", d8)
    assert d8.max() <= 1 and (d8 > 0).mean() < 0.01, (d8.max(), (d8 > 0).mean())


@pytest.mark.parametrize("kind,size", [("chirp", (480, 270)), ("noise", P1080), ("stars", P1080)])
def test_matrix_pipe_behind_sigmoid(kind, size):
    """The matrix pipe behind unsigmoidize + delinearize (module docstring), in codes of 16 bits
    against the sequential-fma kernel, with the dark samples -- where the inverse curves amplify
    -- reported on their own."""
    sw, sh = size
    img = content(kind, sw, sh)
    csp = pl.color_space("bt709", "bt1886")
    kw = dict(image_color=csp, target_color=csp)
    p = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=None)
    pp, _ = render(img, 2 * sw, 2 * sh, p, False, **kw)
    mx, used = render(img, 2 * sw, 2 * sh, p, True, **kw)
    assert used
    d = codes(mx, pp)
    report("matrix pipe behind unsigmoidize + delinearize, %s %dx%d" % (kind, sw, sh), d)
    dark = pp[..., :3] < 4000
    print("   among the %.3f of samples below code 4000: max %d, > 0 on %.4f" %
          (dark.mean(), int(d[dark].max()), float((d[dark] > 0).mean())))
    assert d.max() <= 1 and (d > 0).mean() < 0.05


@pytest.mark.parametrize("size", [(166, 93), P1080])
def test_gamma_light_2x_ewa_on_the_matrix_pipe(size):
    """The same preset with linear scaling off (and the `fast` preset): the upscale runs on the
    gamma-coded signal, the epilogue is dither + scale -- the matrix pipe's home ground. <= 1 code
    vs k_polar_pp, which IS the oracle bit for bit; dither index exact."""
    sw, sh = size
    dw, dh = 2 * sw, 2 * sh
    csp = pl.color_space("bt709", "bt1886")
    kw = dict(image_color=csp, target_color=csp)
    for kind in ("chirp", "noise"):
        img = content(kind, sw, sh)
        p = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=None,
                             disable_linear_scaling=True)
        mx, used = render(img, dw, dh, p, True, **kw)
        assert used
        pp, _ = render(img, dw, dh, p, False, **kw)
        a = orc.op_quant_f16(orc.tex_decode(img, "rgba16"))
        a[..., 3] = 1.0
        w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
        ref16 = orc.tex_encode(orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7), "rgba16")
        assert np.array_equal(pp, ref16)
        d = codes(mx, ref16)
        report("gamma-light 2x %dx%d %s, matrix pipe vs oracle" % (sw, sh, kind), d)
        assert d.max() <= 1 and (d > 0).mean() < 0.15
        pd = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=blue(),
                              disable_linear_scaling=True, disable_dither_gamma_correction=True)
        got10, used = render(img, dw, dh, pd, True, target_repr=repr_bits(10, 16), **kw)
        assert used
        assert dither_consistency(got10, mx, util.blue_noise(pl)) == 0.0
        got8, used = render(img, dw, dh, p, True, dst_fmt="rgba8", **kw)
        d8 = codes(got8, orc.tex_encode(orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7), "rgba8"))
        assert used and d8.max() <= 1 and (d8 > 0).mean() < 0.005, (d8.max(), (d8 > 0).mean())


@pytest.mark.parametrize("kind", ["chirp", "noise"])
def test_high_quality_preset_2x(kind):
    """pl_render_high_quality_params at 2x on SDR video (ewa_lanczossharp in sigmoidized linear
    light, debanding in front: the scaler reads an rgba16hf intermediate), and the same with linear
    scaling off: the matrix pipe within a code of the sequential-fma kernel."""
    sw, sh = 320, 180
    img = content(kind, sw, sh)
    csp = pl.color_space("bt709", "bt1886")
    kw = dict(image_color=csp, target_color=csp)
    for linear in (True, False):
        p = pl.render_params("high_quality", dither_params=None, disable_linear_scaling=not linear)
        mx, used = render(img, 2 * sw, 2 * sh, p, True, **kw)
        pp, _ = render(img, 2 * sw, 2 * sh, p, False, **kw)
        d = codes(mx, pp)
        report("high_quality 2x (%s light) %s, matrix pipe vs k_polar_pp" %
               ("sigmoidized linear" if linear else "gamma", kind), d)
        assert used and d.max() <= 1 and (d > 0).mean() < 0.15


@pytest.mark.parametrize("post", ["lite", "full"])
def test_interpreter_epilogues_behind_a_plain_encode(post):
    """MX_POST_LITE: a YCbCr (BT.709, limited range) target -- the encode is an AFFINE op, no
    transcendental, not the fused dither + scale tail. MX_POST_FULL: the same with the map chain
    switched off and a gamma-2.2 target (LINEARIZE / DELINEARIZE through the interpreter; both
    curves are gamma-like, there is no amplification to speak of: 2.4 / 2.2). <= 1 code vs
    k_polar_pp at 1080p -> 4K on chirp and noise."""
    sw, sh = 480, 270
    for kind in ("chirp", "noise"):
        img = content(kind, sw, sh)
        icsp = pl.color_space("bt709", "bt1886")
        if post == "lite":
            trepr = pl.color_repr("bt709", "limited", sample_depth=16, color_depth=16, bit_shift=0)
            tcsp, extra = icsp, {}
        else:
            trepr, tcsp, extra = None, pl.color_space("bt709", "gamma22"), {"PL_HIP_MAP_CHAIN": "0"}
        p = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"), dither_params=None)
        kw = dict(image_color=icsp, target_color=tcsp, target_repr=trepr, extra_env=extra)
        mx, used = render(img, 2 * sw, 2 * sh, p, True, **kw)
        pp, _ = render(img, 2 * sw, 2 * sh, p, False, **kw)
        d = codes(mx, pp)
        report("%s interpreter epilogue, %s, matrix pipe vs k_polar_pp" % (post, kind), d)
        assert used and d.max() <= 1 and (d > 0).mean() < 0.15, (used, d.max(), (d > 0).mean())


@pytest.mark.parametrize("size", [((600, 340), (300, 170)), ((3840, 2160), P1080)])
def test_ewa_4k_to_1080p_dither10_through_the_renderer(size):
    """bench.py's `ewa_lanczos_4k_to_1080p_dither10`: the plain SDR downscale (widened EWA-Lanczos,
    148 taps, gamma light), 10-bit blue-noise dither, ONE k_polar_mxd launch (rgba16 source decoded
    while it is staged, fused dither + scale epilogue) through pl_render_image -- against
    k_polar_pp and the oracle: <= 1 code before the dither, dither index exact."""
    (sw, sh), (dw, dh) = size
    csp = pl.color_space("bt709", "srgb")
    kw = dict(image_color=csp, target_color=csp)
    for kind in ("chirp", "noise"):
        img = content(kind, sw, sh)

        def params(dither):
            return pl.render_params("fast", downscaler=pl.filter_config("ewa_lanczos"),
                                    dither_params=blue() if dither else None,
                                    disable_linear_scaling=True, disable_dither_gamma_correction=True)
        mx, used = render(img, dw, dh, params(False), True, **kw)
        assert used, LAST_LOG
        pp, used_pp = render(img, dw, dh, params(False), False, **kw)
        assert not used_pp
        a = orc.op_quant_f16(orc.tex_decode(img, "rgba16"))
        a[..., 3] = 1.0
        # (a downscale widens the kernel by the ratio -- blur 2, radius 6.48 -- and at that radius the
        # reference's gather path sets the tap order: sampling.c:600-606, 671-674)
        w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos(blur=float(sw) / dw))
        ref16 = orc.tex_encode(orc.sample_polar(a, w, r, rz, dw, dh, mask=0x7, gather_order=not r < 6.0),
                               "rgba16")
        assert np.array_equal(pp, ref16), util.diff_stats(pp, ref16)
        d = codes(mx, pp)
        report("4K -> 1080p class %dx%d %s, k_polar_mxd vs k_polar_pp" % (sw, sh, kind), d)
        assert d.max() <= 1 and (d > 0).mean() < 0.05, (d.max(), (d > 0).mean())
        got10, used = render(img, dw, dh, params(True), True, target_repr=repr_bits(10, 16), **kw)
        assert used
        assert dither_consistency(got10, mx, util.blue_noise(pl)) == 0.0
        pp10, _ = render(img, dw, dh, params(True), False, target_repr=repr_bits(10, 16), **kw)
        steps = np.abs((got10[..., :3] >> 6).astype(np.int64) - (pp10[..., :3] >> 6))
        assert steps.max() <= 1 and (steps > 0).mean() < 0.002, (steps.max(), (steps > 0).mean())


@pytest.mark.parametrize("size", [((600, 340), (300, 170)), ((3840, 2160), P1080)])
@pytest.mark.parametrize("trc", ["srgb", "bt1886"])
def test_linear_light_downscale_on_the_matrix_pipe(size, trc):
    """the reference linearises in front of a downscaler (src/renderer.c:1997-2003): PLANE_MAP +
    LINEARIZE as the fused PASS A of the polar pass, DELINEARIZE in front of its dither -- on
    k_polar_mxd since round 4 (VERDICT r03 missing 1). Against k_polar_pp (the same ops, the taps as
    sequential fma): the f16 tiles are identical, so the frames differ by what the fp32 summation
    order leaves behind the inverse curve; and against the oracle composed by hand from the GPU's
    own PASS A."""
    from test_gpu_color import luma_coeffs, nominal
    (sw, sh), (dw, dh) = size
    csp = inferred(pl.color_space("bt709", trc))
    kw = dict(image_color=csp, target_color=csp)
    for kind in ("chirp", "noise") if sw < 1000 else ("noise",):
        img = content(kind, sw, sh)

        def params(dither):
            return pl.render_params("fast", downscaler=pl.filter_config("ewa_lanczos"),
                                    dither_params=blue() if dither else None,
                                    disable_dither_gamma_correction=True)
        mx, used = render(img, dw, dh, params(False), True, **kw)
        assert used, LAST_LOG
        pp, used_pp = render(img, dw, dh, params(False), False, **kw)
        assert not used_pp
        d = codes(mx, pp)
        report("linear-light %s %dx%d -> %dx%d %s, k_polar_mxd vs k_polar_pp" % (trc, sw, sh, dw, dh, kind), d)
        assert d.max() <= 1 and (d > 0).mean() < 0.05, (d.max(), (d > 0).mean())
        if sw < 1000:
            # the oracle over the GPU's own linear f16 intermediate
            with pl.HipGpu(0) as g:
                src = g.tex_create(sw, sh, "rgba16", img)
                fbo = g.tex_create(sw, sh, "rgba16hf")
                a = g.begin()
                assert a.sample("direct", src, components=3)
                a.linearize(csp)
                assert a.finish(fbo), g.messages[-3:]
                lin = fbo.download()
                src.destroy(); fbo.destroy()
            mn, mxl = nominal(csp)
            w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos(blur=float(sw) / dw))
            b = orc.sample_polar(np.ascontiguousarray(lin, np.float32), w, r, rz, dw, dh, mask=0x7,
                                 gather_order=not r < 6.0)
            orc.delinearize(b, int(csp.transfer), mn, mxl, luma_coeffs(csp.primaries))
            ref16 = orc.tex_encode(b, "rgba16")
            do = codes(pp, ref16)
            report("   ... k_polar_pp vs the oracle (libm pow against v_log / v_exp)", do)
            assert do.max() <= 1
            assert codes(mx, ref16).max() <= 2
        got10, used = render(img, dw, dh, params(True), True, target_repr=repr_bits(10, 16), **kw)
        assert used
        assert dither_consistency(got10, mx, util.blue_noise(pl)) == 0.0


def test_hdr_downscale_with_fused_pq_linearisation():
    """HDR10 2 : 1 downscale without debanding: the polar pass reads the PQ plane, linearises while
    it stages its tile, and writes the linear rgba16hf intermediate that the measurement and the
    map pass read -- k_polar_mxd<unorm, f16 target, PRE>. Final SDR frames against the k_polar_pp
    route under the conditioning of the colour map (as test_mx_hdr_colour_map_epilogue)."""
    from test_gpu_fullsize import hdr_frame16
    sw, sh, dw, dh = 1024, 576, 512, 288
    img = hdr_frame16(sw, sh)
    hdr = pl.color_space("bt2020", "pq", max_luma=1000.0)
    sdr = pl.color_space("bt709", "bt1886")
    params = pl.render_params("default", downscaler=pl.filter_config("ewa_lanczos"), dither_params=None,
                              peak_detect_params=pl.peak_detect_params(percentile=99.995))
    kw = dict(image_color=hdr, target_color=sdr)
    mx, used = render(img, dw, dh, params, True, **kw)
    assert used, LAST_LOG
    pp, used_pp = render(img, dw, dh, params, False, **kw)
    assert not used_pp
    d = codes(mx, pp)
    print("HDR downscale: |mxd - pp| codes: median %.1f p99 %.1f p99.9 %.1f max %d; > 8 codes on %.2e" %
          (np.median(d), np.quantile(d, 0.99), np.quantile(d, 0.999), d.max(), (d > 8).mean()))
    # (the two routes round the LINEAR intermediate to f16 -- the reference's rgba16hf FBO -- from
    # sums in another order: one f16 ulp apart on ~0.1 % of the texels (test_mxd_* in
    # test_gpu_polar_mfma.py), 5e-4 relative, which the tone curve and the inverse display gamma
    # then amplify on a few dark saturated pixels)
    assert np.quantile(d, 0.5) <= 1 and np.quantile(d, 0.99) <= 4 and np.quantile(d, 0.999) <= 16
    assert d.max() <= 256 and (d > 8).mean() < 2e-3


@pytest.mark.parametrize("mfma", [True, False])
def test_default_preset_ewa_pass_structures_identical(mfma):
    """pl_render_default_params + ewa_lanczos, SDR 2x: since round 4 the renderer keeps the
    reference's two passes for an upscale whose pending ops need transcendentals (PASS A =
    k_pass_chain into the rgba16hf intermediate, then the polar kernel with the chain epilogue:
    0.063 ms against 0.113 ms fused at 1080p -> 4K). The fused launch (PL_HIP_NO_FUSION=0) and the
    forced two-pass structure (=1) render the same frame: bit for bit on k_polar_pp, within the
    matrix-pipe kernels' one code (here: one 10-bit step on < 1 % of the samples) on k_polar_mx."""
    sw, sh = 166, 93
    img = content("noise", sw, sh)
    csp = inferred(pl.color_space("bt709", "bt1886"))
    p = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=blue(),
                         disable_dither_gamma_correction=True)
    kw = dict(image_color=csp, target_color=csp, target_repr=repr_bits(10, 16))
    base, used = render(img, 2 * sw, 2 * sh, p, mfma, **kw)
    assert used == mfma
    for v in ("0", "1"):
        o, used = render(img, 2 * sw, 2 * sh, p, mfma, extra_env={"PL_HIP_NO_FUSION": v}, **kw)
        assert used == mfma
        if mfma:
            # (the fused launch is another variant of the matrix-pipe kernel -- interpreter epilogue,
            # the row-phase term on the A fragments: the kernels' "one code" statement, dithered)
            d = codes(o, base)
            assert d.max() <= 64 and (d > 0).mean() < 0.01, (v, int(d.max()), float((d > 0).mean()))
        else:
            assert np.array_equal(o, base), (v, util.diff_stats(o, base))
