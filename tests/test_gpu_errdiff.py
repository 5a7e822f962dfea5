"""Error diffusion (pl_shader_error_diffusion) on the GPU against the oracle: integer error
transport + a handful of exactly rounded float ops -> bit-exact."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
import util

pytestmark = pytest.mark.gpu

KERNELS = ["simple", "false-fs", "sierra-lite", "floyd-steinberg", "atkinson",
           "jarvis-judice-ninke", "stucki", "burkes", "sierra-2", "sierra-3"]


def kernel_of(name):
    k = pl.lib().pl_find_error_diffusion_kernel(name.encode())
    assert k, name
    k = k.contents
    return k.shift, k.divisor, [[k.pattern[y][x] for x in range(5)] for y in range(3)]


@pytest.mark.parametrize("name", KERNELS)
def test_error_diffusion_bit_exact(gpu, name):
    w, h, depth = 96, 70, 4
    img = util.chirp_rgba16(w, h)
    src = gpu.tex_create(w, h, "rgba16", img)
    dst = gpu.tex_create(w, h, "rgba16")
    sh = gpu.begin()
    assert sh.error_diffusion(src, dst, depth, name), gpu.messages[-3:]
    assert sh.compute(), gpu.messages[-3:]
    got = dst.download()
    shift, divisor, pattern = kernel_of(name)
    ref = orc.error_diffusion(orc.tex_decode(img, "rgba16"), depth, shift, divisor, pattern)
    ref16 = orc.tex_encode(ref, "rgba16")
    assert np.array_equal(got, ref16), util.diff_stats(got, ref16)
    # it is a dither: every value is a multiple of 1/15, mean is preserved
    q = orc.tex_decode(got, "rgba16")[..., :3] * 15
    assert np.abs(q - np.round(q)).max() < 1e-3
    assert abs(q.mean() / 15 - orc.tex_decode(img, "rgba16")[..., :3].mean()) < 2e-3
    src.destroy(); dst.destroy()


@pytest.mark.parametrize("h", [1100, 2160])
def test_error_diffusion_tall_frames_multi_step(gpu, h):
    """height > 1024: several sequential steps per sheared column; 2160 rows needs > 64 KiB of
    LDS with a 3-row kernel (the case the reference downgrades to ordered dither)."""
    w, depth = 48, 8
    img = util.random_rgba16(w, h, seed=h)
    src = gpu.tex_create(w, h, "rgba16", img)
    dst = gpu.tex_create(w, h, "rgba16")
    sh = gpu.begin()
    assert sh.error_diffusion(src, dst, depth, "sierra-3"), gpu.messages[-3:]
    assert sh.compute(), gpu.messages[-3:]
    got = dst.download()
    shift, divisor, pattern = kernel_of("sierra-3")
    ref = orc.tex_encode(orc.error_diffusion(orc.tex_decode(img, "rgba16"), depth, shift, divisor,
                                             pattern), "rgba16")
    assert np.array_equal(got, ref), util.diff_stats(got, ref)
    src.destroy(); dst.destroy()


def test_error_diffusion_shmem_req(gpu):
    k = pl.lib().pl_find_error_diffusion_kernel(b"sierra-lite")
    shift, divisor, pattern = kernel_of("sierra-lite")
    cols = 1 + max(dx - 2 + dy * shift for dy in range(3) for dx in range(5) if pattern[dy][dx])
    assert pl.lib().pl_error_diffusion_shmem_req(k, 1080) == (1080 + 2) * cols * 4
