"""Shared helpers for the parity tests: synthetic frames and
the libc rand() seeding the blue-noise generator needs to be reproducible."""
import ctypes as C

import numpy as np

_libc = C.CDLL("libc.so.6")


def srand(seed=1):
    _libc.srand(seed)


def chirp_rgba16(w, h, alpha=65535):
    """The reference's bench pattern (src/tests/bench.c:32-51): per-channel radial
    sine chirp 0.5*sin(f_k r^2)+0.5, f_G = f_R/phi, f_B = f_G/phi, as RGBA16."""
    yc, xc = h / 2.0, w / 2.0
    kx = 0.5 * np.pi / xc / xc if False else None  # (kept simple below)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r2 = (x - xc) ** 2 + (y - yc) ** 2
    phi = (1 + 5 ** 0.5) / 2
    f_r = 0.1 * np.pi * 0.5 / np.sqrt(xc * xc + yc * yc)
    out = np.empty((h, w, 4), np.uint16)
    for k, f in enumerate((f_r, f_r / phi, f_r / phi / phi)):
        out[..., k] = np.rint((0.5 * np.sin(f * r2) + 0.5) * 65535).astype(np.uint16)
    out[..., 3] = alpha
    return out


def random_rgba16(w, h, seed=1):
    return np.random.default_rng(seed).integers(0, 65536, (h, w, 4), dtype=np.uint16)


def blue_noise(pl, size=64, seed=1):
    srand(seed)
    m = np.empty(size * size, np.float32)
    pl.lib().pl_generate_blue_noise(m.ctypes.data, size)
    return m.reshape(size, size)


def diff_stats(a, b):
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    return int(d.max()), int((d > 0).sum())


def hip_runtime():
    """ctypes handle of the HIP runtime instance libplacebo_hip.so itself is bound to (found in
    /proc/self/maps; dlopen of an already mapped path returns that very instance). Loading
    "libamdhip64.so" by name can pick up another copy (torch bundles one), whose streams and
    allocations are not interchangeable with ours."""
    import libplacebo_amd as pl
    pl.lib()
    paths = []
    for ln in open("/proc/self/maps"):
        if "libamdhip64" in ln:
            path = ln.split()[-1]
            if path not in paths:
                paths.append(path)
    assert paths, "no HIP runtime mapped"
    lib = C.CDLL(paths[0])
    lib._paths = paths
    return lib
