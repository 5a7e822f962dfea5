"""Shared helpers for the parity tests: synthetic frames and
the libc rand() seeding the blue-noise generator needs to be reproducible."""
import ctypes as C

import os

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
    # torch (if some test imported it) maps its own bundled copy; ours is the system one
    mine = [p for p in paths if "/torch/" not in p] or paths
    lib = C.CDLL(mine[0])
    lib._paths = paths
    return lib


def assert_colormap_parity(got, ref, truth=None, scale=65535.0,
                           quantiles=(0.5, 0.9, 0.99, 0.999, 1.0), per_sample=True):
    """Parity statement for stages that go through the PQ / IPT colour-mapping chain, in 16-bit
    code values (`got`, `ref`, `truth`: same shape, RGB in the first three components; float
    images in [0, 1] or integer code values with scale = 1).

    The chain is ill-conditioned in fp32 on bright saturated colours: the float-libm oracle (a
    straightforward evaluation of the reference's formulas) is itself up to ~10^2 codes away
    from a float64 evaluation there (tests/colormap_f64.py). The kernels evaluate the PQ pair in
    a well-conditioned form (csrc/hip/cmfast.hiph) and sit much closer to float64 than the oracle
    does, so the statement has two halves:

    * vs float64 (`truth`): at EVERY quantile, maximum included, the GPU may be at most one code
      further from float64 than the oracle is. This is the parity bar.
    * vs the oracle: at every quantile the two may differ by what BOTH are away from float64
      (1.5 x the sum of their quantiles + 2 codes; at the median: the sum + half a code -- the
      oracle's own median error is about one code, so "identical on the bulk" holds only while
      the rounding falls that way) -- without `truth`, by the oracle's typical own error: half
      of the samples within 1 code, 90 % within 3, 99 % within 40."""
    # (compared as a 16-bit unorm target would store them: clipped to [0, 1])
    top = 65535.0
    g = np.clip(np.asarray(got, np.float64)[..., :3].reshape(-1) * scale, 0.0, top)
    o = np.clip(np.asarray(ref, np.float64)[..., :3].reshape(-1) * scale, 0.0, top)
    d = np.abs(g - o)
    if truth is None:
        assert np.quantile(d, 0.5) <= 1.0 and np.quantile(d, 0.9) <= 3.0 and \
            np.quantile(d, 0.99) <= 40.0, np.quantile(d, (0.5, 0.9, 0.99, 1.0))
        return
    t = np.clip(np.asarray(truth, np.float64)[..., :3].reshape(-1), 0.0, 1.0) * 65535.0
    eg, eo = np.abs(g - t), np.abs(o - t)
    assert np.quantile(d, 0.5) <= np.quantile(eg, 0.5) + np.quantile(eo, 0.5) + 0.5, \
        (np.quantile(d, 0.5), np.quantile(eg, 0.5), np.quantile(eo, 0.5))
    for q in quantiles:
        qg, qo = np.quantile(eg, q), np.quantile(eo, q)
        assert qg <= qo + 1.0, ("GPU further from float64 than the oracle", q, qg, qo)
        # (quantiles are not sub-additive sample by sample: allow half as much again)
        assert np.quantile(d, q) <= 1.5 * (qg + qo) + 2.0, (q, np.quantile(d, q), qg, qo)
    # The per-sample form (a quantile cannot see a small set of samples that
    # are individually wrong). Where the oracle can be trusted -- it lies within one code of
    # float64: three quarters of the samples of the HDR frames -- the GPU is held to it sample by
    # sample; the rest are the samples on which the fp32 oracle itself is off (the ill-conditioned
    # bright saturated colours), and there the GPU is held to float64: not further from it than the
    # oracle is, plus one code. Measured at full size (profiles/r06_04_colormap_per_sample_stats.txt;
    # the metric's frame, 24.9 M samples: the oracle within one code of float64 on 74.8 %; there
    # |GPU - oracle| > 1 code on 1523 samples = 8e-5, > 2 on 181 = 1e-5, maximum 7; 198 samples
    # = 3e-5 of the rest where the GPU is the further one): the GPU has a few outliers OF ITS OWN,
    # up to 6 codes from float64 where the oracle happens to be accurate -- the lookups' cell
    # boundaries fall elsewhere under its reformulated arithmetic. The statement held per sample
    # is therefore: within a code and a half of the oracle on all but 2e-4 of the trusted samples,
    # within two and a half on all but 5e-5, never more than 10 codes; further from float64 than
    # the oracle + 1 on at most 1e-4 of the others. (`per_sample=False`: end-to-end tests whose GPU
    # and oracle map DIFFERENT intermediate images -- a few f16 codes of an earlier stage -- state
    # the quantiles only; their stage-by-stage twins carry the per-sample form.)
    if not per_sample:
        return
    st = colormap_per_sample(g, o, t)
    print("colour-map parity, per sample: oracle within 1 code of float64 on %.4f of %d samples; there "
          "|GPU - oracle| max %.2f (> 1.5 codes on %d, > 2.5 on %d); elsewhere GPU further from float64 than "
          "the oracle + 1 on %d" % (st["covered"], st["n"], st["max_covered"], st["over1"], st["over2"],
                                   st["worse_elsewhere"]))
    if os.environ.get("PL_PARITY_REPORT_ONLY"):
        print("   ", st)
        return
    ncov = st["covered"] * st["n"]
    # (small frames: a handful of samples is the resolution of the count)
    assert st["over1"] <= max(20, 2e-4 * ncov), st
    assert st["over2"] <= max(3, 5e-5 * ncov), st
    assert st["max_covered"] <= 10.0, st
    assert st["worse_elsewhere"] <= max(3, 1e-4 * (st["n"] - ncov)), st


def colormap_per_sample(g, o, t, trust=1.0):
    """Per-sample statistics behind assert_colormap_parity (g, o, t: GPU, oracle, float64 in codes).
    `covered`: the samples whose oracle value lies within `trust` codes of float64. On those
    |g - o| <= |g - t| + |o - t|: at most `trust` + the GPU's own distance from float64, which the
    conditioning argument (cmfast.hiph) puts below one code on the bulk. `over1` / `over2`: the
    covered samples more than 1.5 / 2.5 codes from the oracle (stored codes: 2 / 3 or more)."""
    eg, eo, d = np.abs(g - t), np.abs(o - t), np.abs(g - o)
    cov = eo <= trust
    return {
        "n": int(d.size), "covered": float(cov.mean()),
        "max_covered": float(d[cov].max()) if cov.any() else 0.0,
        "over1": int((d[cov] > 1.5).sum()), "over2": int((d[cov] > 2.5).sum()),
        "worse_elsewhere": int((eg[~cov] > eo[~cov] + 1.0).sum()),
        "max_gpu_vs_f64_covered": float(eg[cov].max()) if cov.any() else 0.0,
    }


def polar_exact():
    """Does this run pin the sequential-fma polar kernel (tests/conftest.py's default)? With
    PL_HIP_POLAR_MFMA=1 the whole suite runs on the library's defaults -- the matrix-pipe kernels for
    the exact ratios -- whose statement is "never more than one code from k_polar_pp"."""
    import os
    return os.environ.get("PL_HIP_POLAR_MFMA", "1") == "0"


def assert_polar_equal(got, ref, step=1, max_frac=0.05, what=None):
    """`got == ref` bit for bit where the polar pass runs on k_polar_pp; where the matrix-pipe
    kernels may have run (PL_HIP_POLAR_MFMA=1: ADVICE r03, the suite on the library's defaults) at
    most `step` apart -- one code of the target's depth (64 for 10 bits in 16), one f16 ulp for
    half-float arrays, 4e-6 for fp32 -- on at most `max_frac` of the samples."""
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if polar_exact():
        assert np.array_equal(got, ref), (what, diff_stats(got, ref))
        return
    if got.dtype == np.float32:
        d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
        assert d.max() <= 4e-6 * step, (what, float(d.max()))
        return
    if got.dtype == np.float16:
        # one ulp -- or, for values so small that an ulp is less than the contraction's absolute
        # error (1.3e-6 of the data's scale: dark linear-light texels), that absolute error
        d = np.abs(got.view(np.int16).astype(np.int64) - ref.view(np.int16).astype(np.int64))
        scale = max(1.0, float(np.abs(ref.astype(np.float32)).max()))
        small = np.abs(got.astype(np.float32) - ref.astype(np.float32)) <= 4e-6 * scale
        d = np.where(small & (d > 1), 1, d)
    else:
        d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
    assert d.max() <= step and (d > 0).mean() <= max_frac, (what, int(d.max()), float((d > 0).mean()))
