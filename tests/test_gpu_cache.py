"""pl_gpu_set_cache: the two expensive host-generated tables of the render path -- the
blue-noise dither matrix (void-and-cluster, O(size^4)) and the gamut-mapping 3D-LUT -- are
memoised through the user's pl_cache with the reference's sh_lut protocol (src/shaders/lut.c:
329,478-486,600; signatures dithering.c:158, colorspace.c:991-1001), survive a save / load
round trip, and a renderer built on a warm cache produces the same frame bit for bit."""
import ctypes as C
import time

import numpy as np
import pytest

import libplacebo_amd as pl
import util
from libplacebo_amd import _capi as capi
from test_cache import Params, bind

pytestmark = pytest.mark.gpu

PL_LOG_DEBUG = 5


def render_once(g, img, w, h):
    src = g.tex_create(w, h, "rgba16", img)
    dst = g.tex_create(w, h, "rgba16")
    rr = pl.Renderer(g)
    target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"),
                      repr_=pl.color_repr("rgb", "full", sample_depth=16, color_depth=8))
    util.srand(1)
    t0 = time.perf_counter()
    assert rr.render(pl.frame(src, components=3, color=pl.color_space("bt2020", "pq")), target,
                     pl.render_params("default")), g.messages[-4:]
    dt = time.perf_counter() - t0
    assert rr.errors() == 0
    out = dst.download()
    rr.destroy(); src.destroy(); dst.destroy()
    return out, dt


def test_dither_matrix_and_gamut_lut_are_memoised():
    lib = bind(pl.lib())
    w, h = 160, 90
    img = util.chirp_rgba16(w, h)
    cache = lib.pl_cache_create(C.byref(Params()))

    with pl.HipGpu(0, log_level=PL_LOG_DEBUG) as g:
        pl.lib().pl_gpu_set_cache(g.gpu, C.c_void_p(cache))
        cold, t_cold = render_once(g, img, w, h)
        msgs = [m for _, m in g.messages]
        assert any("Generated dither matrix" in m for m in msgs), msgs[-10:]
        assert any("Generated gamut LUT" in m for m in msgs)
        assert lib.pl_cache_objects(cache) == 2
        # 64x64 floats + 48x32x256 rgba16 texels
        assert lib.pl_cache_size(cache) == 64 * 64 * 4 + 48 * 32 * 256 * 8
        n = lib.pl_cache_save(cache, None, 0)
        stream = C.create_string_buffer(n)
        assert lib.pl_cache_save(cache, stream, n) == n

    # a new process would start here: fresh cache object restored from the stream, fresh GPU
    warm_cache = lib.pl_cache_create(C.byref(Params()))
    assert lib.pl_cache_load(warm_cache, stream, n) == 2
    assert lib.pl_cache_signature(warm_cache) == lib.pl_cache_signature(cache)
    with pl.HipGpu(0, log_level=PL_LOG_DEBUG) as g:
        pl.lib().pl_gpu_set_cache(g.gpu, C.c_void_p(warm_cache))
        warm, t_warm = render_once(g, img, w, h)
        msgs = [m for _, m in g.messages]
        assert any("Re-using cached dither matrix" in m for m in msgs), msgs[-10:]
        assert any("Re-using cached gamut LUT" in m for m in msgs)
        assert not any("Generated" in m for m in msgs)
        assert lib.pl_cache_objects(warm_cache) == 2        # taken out and handed back
    assert np.array_equal(cold, warm)
    print("first frame: cold %.1f ms, warm cache %.1f ms" % (t_cold * 1e3, t_warm * 1e3))

    # without a cache everything still works (and generates)
    with pl.HipGpu(0, log_level=PL_LOG_DEBUG) as g:
        plain, _ = render_once(g, img, w, h)
        assert any("Generated dither matrix" in m for _, m in g.messages)
    assert np.array_equal(plain, cold)
    for c in (cache, warm_cache):
        cc = C.c_void_p(c)
        lib.pl_cache_destroy(C.byref(cc))


def test_renderer_and_dispatch_save_load_are_the_gpu_cache():
    """pl_renderer_save / _load and pl_dispatch_save / _load (deprecated front ends, src/renderer.c:
    184-192, src/dispatch.c:1624-1632): the bytes pl_cache_save of the gpu's cache writes, and a
    renderer on another gpu loads them into ITS cache."""
    lib = bind(pl.lib())
    L = pl.lib()
    L.pl_renderer_save.restype = C.c_size_t
    L.pl_renderer_save.argtypes = [C.c_void_p, C.c_char_p]
    L.pl_renderer_load.argtypes = [C.c_void_p, C.c_char_p]
    L.pl_dispatch_save.restype = C.c_size_t
    L.pl_dispatch_save.argtypes = [C.c_void_p, C.c_char_p]
    L.pl_dispatch_load.argtypes = [C.c_void_p, C.c_char_p]
    w, h = 96, 64
    img = util.chirp_rgba16(w, h)
    cache = lib.pl_cache_create(C.byref(Params()))
    with pl.HipGpu(0) as g:
        L.pl_gpu_set_cache(g.gpu, C.c_void_p(cache))
        render_once(g, img, w, h)
        rr = pl.Renderer(g)
        n = L.pl_renderer_save(rr.rr, None)
        assert n == lib.pl_cache_save(cache, None, 0) and n > 64 * 64 * 4
        a, b, c = (C.create_string_buffer(n) for _ in range(3))
        assert L.pl_renderer_save(rr.rr, a) == n and lib.pl_cache_save(cache, b, n) == n
        assert L.pl_dispatch_save(g.dp, None) == n and L.pl_dispatch_save(g.dp, c) == n
        rr.destroy()
        assert a.raw == b.raw == c.raw
    other = lib.pl_cache_create(C.byref(Params()))
    with pl.HipGpu(0) as g:
        L.pl_gpu_set_cache(g.gpu, C.c_void_p(other))
        rr = pl.Renderer(g)
        L.pl_renderer_load(rr.rr, a)
        assert lib.pl_cache_objects(other) == 2
        assert lib.pl_cache_signature(other) == lib.pl_cache_signature(cache)
        rr.destroy()
    third = lib.pl_cache_create(C.byref(Params()))
    with pl.HipGpu(0) as g:
        L.pl_gpu_set_cache(g.gpu, C.c_void_p(third))
        L.pl_dispatch_load(g.dp, a)
        assert lib.pl_cache_objects(third) == 2
    cv = C.c_void_p(third)
    lib.pl_cache_destroy(C.byref(cv))
    for cc in (cache, other):
        cv = C.c_void_p(cc)
        lib.pl_cache_destroy(C.byref(cv))
