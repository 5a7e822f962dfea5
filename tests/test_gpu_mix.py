"""pl_render_image_mix (reference src/renderer.c:3612-4028).

Parity statement: a mix is `delinearize(sum_i w_i/W * linearize(f16(frame_i)))`, frames rendered
like pl_render_image renders them (pinned in test_gpu_renderer.py), weights from the mixer
(`pl_filter_sample`, pinned against the reference in test_tier0_ref.py) or from the visible
fraction of the vsync (oversampling). The oracle supplies the transfer functions; the bar is
2 LSB of 16 bit (two pow() evaluations per channel). The single-frame paths must equal
pl_render_image bit for bit."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util
from libplacebo_amd import _capi as capi
from test_gpu_color import nominal, luma_coeffs

pytestmark = pytest.mark.gpu

W, H = 64, 48


@pytest.fixture()
def rr(gpu):
    r = pl.Renderer(gpu)
    yield r
    r.destroy()


def sources(gpu, n):
    """n different frames: the chirp, rolled and rescaled"""
    base = util.chirp_rgba16(W, H).astype(np.float64)
    imgs, texs, frames = [], [], []
    for i in range(n):
        img = np.roll(base, 5 * i, axis=1) * (1.0 - 0.12 * i)
        img[..., 3] = 65535
        img = img.astype(np.uint16)
        t = gpu.tex_create(W, H, "rgba16", img)
        imgs.append(img); texs.append(t)
        frames.append(pl.frame(t, components=3, color=pl.color_space("bt709", "bt1886")))
    return imgs, texs, frames


def mixer(name):
    cfg = capi.FilterConfig()
    C.memmove(C.byref(cfg), C.byref(pl.filter_config(name, pl.FILTER_FRAME_MIXING)), C.sizeof(cfg))
    return cfg


def expected_mix(imgs, weights):
    csp = pl.color_space("bt709", "bt1886")
    pl.lib().pl_color_space_infer(C.byref(csp))
    mn, mx = nominal(csp)
    luma = luma_coeffs(csp.primaries)
    wsum = np.float32(sum(np.float32(w) for w in weights))
    acc = np.zeros((H, W, 4), np.float32)
    for img, w in zip(imgs, weights):
        f = orc.tex_decode(img, "rgba16")
        f[..., 3] = 1.0
        f = orc.op_quant_f16(f)                                  # the cached rgba16hf frame
        f = orc.linearize(f, int(csp.transfer), mn, mx, luma)
        acc += (np.float32(w) / wsum) * f
    acc = orc.delinearize(acc, int(csp.transfer), mn, mx, luma)
    return acc


def run_mix(gpu, rr, frames, ts, params, vsync=1.0, sigs=None):
    dst = gpu.tex_create(W, H, "rgba16")
    target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"))
    sigs = sigs or [1000 + i for i in range(len(frames))]
    ok = rr.render_mix(frames, sigs, ts, vsync, target, params)
    out = dst.download() if ok else None
    dst.destroy()
    return ok, out


@pytest.mark.parametrize("name,ts", [("linear", [-0.7, 0.3]), ("hermite", [-0.25, 0.75]),
                                      ("mitchell", [-1.6, -0.6, 0.4, 1.4]),
                                      ("catmull_rom", [-1.1, -0.1, 0.9, 1.9])])
def test_mix_weights_from_the_filter(gpu, rr, name, ts):
    imgs, texs, frames = sources(gpu, len(ts))
    cfg = mixer(name)
    params = pl.render_params("fast", frame_mixer=cfg)
    ok, got = run_mix(gpu, rr, frames, ts, params)
    assert ok and rr.errors() == 0, gpu.messages[-4:]
    cfg.blur = cfg.blur or 1.0
    weights = [pl.lib().pl_filter_sample(C.byref(cfg), t) for t in ts]
    assert abs(sum(weights)) > 0.5
    use = [(i, w) for i, w in zip(imgs, weights) if abs(w) > 1e-3]
    ref = expected_mix([i for i, _ in use], [w for _, w in use])
    ref16 = orc.tex_encode(ref, "rgba16")
    d = np.abs(got.astype(np.int64) - ref16.astype(np.int64))[..., :3]
    assert d.max() <= 2, (name, int(d.max()), float(d.mean()))
    assert np.all(got[..., 3] == 65535)
    # ... and it is a real blend, not one of the inputs
    for img in imgs:
        assert np.abs(got[..., :3].astype(np.int64) - img[..., :3]).max() > 500
    for t in texs:
        t.destroy()


def test_mix_oversample_weights_are_visible_fractions(gpu, rr):
    imgs, texs, frames = sources(gpu, 3)
    params = pl.render_params("fast", frame_mixer=mixer("oversample"))
    # vsync [0, 1]: frame 0 visible on [0, 0.4], frame 1 on [0.4, 1], frame 2 starts at 1.4
    ok, got = run_mix(gpu, rr, frames, [-0.6, 0.4, 1.4], params)
    assert ok and rr.errors() == 0
    ref16 = orc.tex_encode(expected_mix(imgs[:2], [0.4, 0.6]), "rgba16")
    d = np.abs(got.astype(np.int64) - ref16.astype(np.int64))[..., :3]
    assert d.max() <= 2, int(d.max())
    # a shorter vsync that frame 0 covers entirely: nothing to mix -> the frame itself, exactly
    ok, got = run_mix(gpu, rr, frames, [-0.6, 0.4, 1.4], params, vsync=0.25)
    assert ok
    f = orc.tex_decode(imgs[0], "rgba16"); f[..., 3] = 1.0
    assert np.array_equal(got, orc.tex_encode(orc.op_quant_f16(f), "rgba16"))
    for t in texs:
        t.destroy()


@pytest.mark.parametrize("with_mixer", [False, True])
@pytest.mark.parametrize("skip_cache", [True, False])
def test_single_frame_paths(gpu, rr, with_mixer, skip_cache):
    """One frame to show: with skip_caching_single_frame the call IS pl_render_image; without,
    the frame takes the cache detour (rendered to the rgba16hf cache texture first, output
    stage afterwards: renderer.c:3760-3770), which may move a dithered code by one step."""
    imgs, texs, frames = sources(gpu, 3)
    dither = capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0)
    kw = dict(upscaler=pl.filter_config("ewa_lanczos"), dither_params=dither,
              disable_dither_gamma_correction=True)
    ten = pl.color_repr("rgb", "full", sample_depth=16, color_depth=10, bit_shift=6)
    dst = gpu.tex_create(2 * W, 2 * H, "rgba16")
    target = pl.frame(dst, repr_=ten, color=pl.color_space("bt709", "bt1886"))
    util.srand(1)
    assert rr.render(frames[1], target, pl.render_params("fast", **kw))
    want = dst.download()
    kw["skip_caching_single_frame"] = skip_cache
    if with_mixer:      # a mixer, but only one frame in the mix
        params = pl.render_params("fast", frame_mixer=mixer("mitchell"), **kw)
        fr, ts, sg = [frames[1]], [0.3], [7]
    else:               # no mixer: the nearest frame (|0.3| < |-0.7|)
        params = pl.render_params("fast", **kw)
        fr, ts, sg = frames, [-0.7, 0.3, 1.3], [6, 7, 8]
    r2 = pl.Renderer(gpu)
    util.srand(1)
    assert r2.render_mix(fr, sg, ts, 1.0, target, params)
    got = dst.download()
    assert r2.errors() == 0
    r2.destroy()
    if skip_cache:
        assert np.array_equal(got, want)
    else:
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        # (the f16 rounding of the cached frame moves ~1/12 of the dither decisions by one code)
        assert d.max() <= 64 and (d > 0).mean() < 0.15, (int(d.max()), float((d > 0).mean()))
    dst.destroy()
    for t in texs:
        t.destroy()


def test_rendered_frames_are_cached_by_signature(gpu, rr):
    imgs, texs, frames = sources(gpu, 3)
    params = pl.render_params("fast", frame_mixer=mixer("linear"))
    ok, a = run_mix(gpu, rr, frames[:2], [-0.7, 0.3], params, sigs=[11, 12])
    assert ok
    # Overwrite the source of frame 12: a cached frame does not notice
    texs[1].upload(np.zeros_like(imgs[1]))
    # next vsync: frame 11 falls out of the radius, 12 is reused, 13 is new
    ok, b = run_mix(gpu, rr, frames[1:], [-0.2, 0.8], params, sigs=[12, 13])
    assert ok
    ref16 = orc.tex_encode(expected_mix(imgs[1:], [0.8, 0.2]), "rgba16")
    assert np.abs(b.astype(np.int64) - ref16.astype(np.int64))[..., :3].max() <= 2
    # different params invalidate the cache (strict reuse): frame 12 is rendered again, from
    # the overwritten source
    adj = capi.ColorAdjustment(brightness=0.0, contrast=1.0, saturation=1.0, hue=0.0, gamma=1.0,
                               temperature=0.0)
    params2 = pl.render_params("fast", frame_mixer=mixer("linear"), color_adjustment=adj,
                               skip_anti_aliasing=True)
    ok, c = run_mix(gpu, rr, frames[1:], [-0.2, 0.8], params2, sigs=[12, 13])
    assert ok
    black = np.zeros_like(imgs[1]); black[..., 3] = 65535
    ref16 = orc.tex_encode(expected_mix([black, imgs[2]], [0.8, 0.2]), "rgba16")
    assert np.abs(c.astype(np.int64) - ref16.astype(np.int64))[..., :3].max() <= 2
    # pl_renderer_flush_cache drops the frames as well
    texs[2].upload(np.zeros_like(imgs[2]))
    pl.lib().pl_renderer_flush_cache(rr.rr)
    ok, d = run_mix(gpu, rr, frames[1:], [-0.2, 0.8], params2, sigs=[12, 13])
    assert ok and d[..., :3].max() == 0
    for t in texs:
        t.destroy()


def test_rejections_and_fallbacks(gpu, rr):
    imgs, texs, frames = sources(gpu, 6)
    params = pl.render_params("fast", frame_mixer=mixer("linear"))
    ok, _ = run_mix(gpu, rr, frames[:2], [0.3, -0.7], params)           # unsorted
    assert not ok
    ok, _ = run_mix(gpu, rr, frames[:2], [-0.7, 0.3], params, vsync=0.0)
    assert not ok
    # an empty mix is pl_render_image without an image: the target is cleared to the background
    params_bg = pl.render_params("fast", frame_mixer=mixer("linear"))
    params_bg.background_color = (C.c_float * 3)(1.0, 0.0, 0.5)
    ok, got = run_mix(gpu, rr, [], [], params_bg)
    assert ok
    assert np.all(got[..., 0] == 65535) and np.all(got[..., 1] == 0) and np.all(got[..., 3] == 65535)
    assert abs(int(got[0, 0, 2]) - 32768) <= 600                        # (sRGB 0.5 in BT.1886)
    # six frames fit one pass ...
    cfg = mixer("lanczos")                                              # radius 3
    params = pl.render_params("fast", frame_mixer=cfg)
    ts = [-2.7, -1.7, -0.7, 0.3, 1.3, 2.3]
    ok, got = run_mix(gpu, rr, frames, ts, params)
    assert ok and rr.errors() == 0
    cfg.blur = cfg.blur or 1.0
    ws = [pl.lib().pl_filter_sample(C.byref(cfg), t) for t in ts]
    ref16 = orc.tex_encode(expected_mix(imgs, ws), "rgba16")
    assert np.abs(got.astype(np.int64) - ref16.astype(np.int64))[..., :3].max() <= 3
    # ... eight do not: the nearest frame is rendered instead (a warning, no error flag)
    imgs8, texs8, frames8 = sources(gpu, 8)
    params = pl.render_params("fast", frame_mixer=mixer("spline64"))    # radius 4
    ts = [-3.7, -2.7, -1.7, -0.7, 0.3, 1.3, 2.3, 3.3]
    ok, got = run_mix(gpu, rr, frames8, ts, params)
    assert ok and rr.errors() == 0
    assert np.array_equal(got[..., :3], imgs8[4][..., :3])       # (1:1 passthrough of frame 4)
    assert any("could not be recorded" in m for _, m in gpu.messages[-12:])
    for t in texs8:
        t.destroy()
    for t in texs:
        t.destroy()


@pytest.mark.parametrize("name,ts", [("linear", [-0.7, 0.3]), ("mitchell", [-1.6, -0.6, 0.4, 1.4]),
                                      ("oversample", [-0.6, 0.4, 1.4])])
@pytest.mark.parametrize("dither", [False, True])
def test_blend_kernel_equals_the_interpreter(gpu, name, ts, dither, monkeypatch):
    """k_pass_mix (round 4: the blending pass as straight-line code -- per cached frame a fetch,
    LINEARIZE and MIX_ADD, then MIX_END, DELINEARIZE and the fused epilogue) against the same
    pass through the op interpreter (PL_HIP_PASS_NATIVE=0: k_pass_generic<.., MIX>): the same
    device functions in the same order, so the same frame bit for bit -- two and four frames,
    with and without the dither, and the last column of an odd width."""
    global W
    outs = []
    old_w = W
    try:
        W = 63
        for native in ("1", "0"):
            monkeypatch.setenv("PL_HIP_PASS_NATIVE", native)
            r = pl.Renderer(gpu)
            imgs, texs, frames = sources(gpu, len(ts))
            kw = dict(frame_mixer=mixer(name))
            if dither:
                kw.update(dither_params=capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0),
                          disable_dither_gamma_correction=True)
            params = pl.render_params("fast", **kw)
            ok, got = run_mix(gpu, r, frames, ts, params)
            assert ok and r.errors() == 0, gpu.messages[-4:]
            outs.append(got)
            r.destroy()
            for t in texs:
                t.destroy()
    finally:
        W = old_w
    assert np.array_equal(outs[0], outs[1]), util.diff_stats(outs[0], outs[1])
