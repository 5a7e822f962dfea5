"""plh_match_map_chain (csrc/hip/fastepi.hiph): which recorded op lists run as straight-line code
instead of through the op interpreter. Host logic only -- no GPU: the matcher is
called through a test hook on op lists given by their kinds. The shapes are the ones the renderer
records (tests/test_gpu_kernel_variants.py renders them and compares the kernels bit for bit)."""
import ctypes as C

import pytest

import libplacebo_amd as pl

# enum plh_op_kind (csrc/hip/plh_device.h)
SCALE, AFFINE, LIN, DELIN, SIG, UNSIG = 1, 2, 3, 4, 5, 6
DITHER, RGB2IPT, TONE, GAMUT, IPT2RGB, PEAK, PMAP, FETCH = 12, 20, 21, 22, 23, 24, 25, 27
RGBA16, RGBA16F = 6, 9
FIELDS = ("enabled lin in tone gamut out delin contrast_recovery unsig sig pmap tail "
          "epi_enabled epi_dither epi_scale").split()


def match(kinds, flags=None, num_pre=0, dst=RGBA16, transpose=0, cr=0, pmap=0, f16=0):
    n = len(kinds)
    flags = flags or [0] * n
    out = (C.c_int * 15)()
    fn = pl.lib().plh_test_match_chain
    fn.restype = C.c_int
    fn((C.c_int * n)(*kinds), (C.c_int * n)(*flags), n, num_pre, dst, transpose, cr, pmap, f16, out)
    return dict(zip(FIELDS, out))


def test_hdr_map_pass_as_the_metric_records_it():
    m = match([LIN, RGB2IPT, TONE, GAMUT, IPT2RGB, DELIN, DITHER, SCALE])
    assert m["enabled"] and (m["lin"], m["in"], m["tone"], m["gamut"], m["out"], m["delin"]) == (0, 1, 2, 3, 4, 5)
    assert m["tail"] == 6 and m["epi_dither"] and m["epi_scale"] and m["unsig"] == m["sig"] == m["pmap"] == -1
    # configs[3]: no dither; configs[4]: no linearize (the downscale ran in linear light)
    assert match([LIN, RGB2IPT, TONE, GAMUT, IPT2RGB, DELIN, SCALE])["enabled"]
    m = match([RGB2IPT, TONE, GAMUT, IPT2RGB, DELIN, DITHER, SCALE])
    assert m["enabled"] and m["lin"] == -1 and m["in"] == 0
    # tone map or gamut map alone
    assert match([RGB2IPT, TONE, IPT2RGB, DELIN])["gamut"] == -1
    assert match([RGB2IPT, GAMUT, IPT2RGB, DELIN])["tone"] == -1


def test_sdr_preset_shapes():
    m = match([UNSIG, DELIN, DITHER, SCALE])            # last scaler pass of pl_render_default_params
    assert m["enabled"] and m["unsig"] == 0 and m["delin"] == 1 and m["in"] == -1 and m["tail"] == 2
    m = match([PMAP, LIN, SIG], dst=RGBA16F, pmap=1, f16=1)    # its first pass, into the intermediate
    assert m["enabled"] and (m["pmap"], m["lin"], m["sig"]) == (0, 1, 2) and not m["epi_enabled"]
    assert not match([PMAP, LIN, SIG], dst=RGBA16F)["enabled"]             # (a kernel without those variants)
    assert not match([PMAP, LIN, SIG, DITHER], dst=RGBA16F, pmap=1, f16=1)["enabled"]   # no epilogue into f16


def test_what_stays_on_the_interpreter():
    hdr = [LIN, RGB2IPT, TONE, GAMUT, IPT2RGB, DELIN, DITHER, SCALE]
    assert not match(hdr, flags=[0, 0, 1, 0, 0, 0, 0, 0])["enabled"]           # contrast recovery ...
    m = match(hdr, flags=[0, 0, 1, 0, 0, 0, 0, 0], cr=1)                         # ... unless the kernel has it
    assert m["enabled"] and m["contrast_recovery"]
    # the tricubic lookup is not part of the fused map: the chain ends in front of it
    assert not match(hdr, flags=[0, 0, 0, 1, 0, 0, 0, 0])["enabled"]
    assert not match(hdr, flags=[0, 0, 0, 0, 0, 0, 1, 0])["enabled"]           # gamma-aware / ordered dither
    assert not match(hdr, flags=[0, 0, 0, 0, 0, 0, 0, 1])["enabled"]           # per-channel scale
    assert not match(hdr, transpose=1)["enabled"]
    assert not match(hdr, dst=RGBA16F)["enabled"]
    assert not match([LIN, RGB2IPT, TONE, GAMUT, DELIN])["enabled"]             # colour map cut short
    assert not match([LIN, AFFINE, DELIN, DITHER])["enabled"]                   # an op the chain does not know
    assert not match([DITHER, SCALE])["enabled"]                                # nothing of the chain: plain epilogue
    assert not match([PMAP, DITHER, SCALE], pmap=1)["enabled"]
    assert not match([PMAP, LIN, SIG], flags=[1, 0, 0], dst=RGBA16F, pmap=1, f16=1)["enabled"]   # a real swizzle


def test_fused_pre_ops_are_not_the_chains_business():
    # ops in front of num_pre_ops belong to the scaler's tile staging (the reference's PASS A)
    m = match([PMAP, UNSIG, DELIN, DITHER, SCALE], num_pre=1)
    assert m["enabled"] and m["unsig"] == 1 and m["tail"] == 3 and m["pmap"] == -1


# ---- the blending pass of pl_render_image_mix (k_pass_mix, k_pass.hip) ---------------------------
MIX_ADD, MIX_END, NEAREST, BILINEAR = 28, 29, 1, 2


def match_mix(kinds, dst=RGBA16, sampler=NEAREST):
    n = len(kinds)
    out = (C.c_int * 11)()
    fn = pl.lib().plh_test_match_mix
    fn.restype = C.c_int
    ok = fn((C.c_int * n)(*kinds), n, dst, sampler, out)
    return bool(ok), dict(frames=out[0], delin=out[1], epi=out[2], lin=list(out[3:7]), fetch=list(out[7:11]))


def test_blending_pass_shapes():
    # what render_mix.c records for two cached frames into a 10-bit target (tools/r04_51.sh)
    ok, m = match_mix([LIN, MIX_ADD, FETCH, LIN, MIX_ADD, MIX_END, DELIN, DITHER, SCALE])
    assert ok and m["frames"] == 2 and m["lin"][:2] == [0, 3] and m["fetch"][:2] == [-1, 2]
    assert m["delin"] == 6 and m["epi"]
    # four frames (a wider mixer), no curves (a linear target), an rgba16hf target without an epilogue
    ok, m = match_mix([MIX_ADD, FETCH, MIX_ADD, FETCH, MIX_ADD, FETCH, MIX_ADD, MIX_END], dst=RGBA16F)
    assert ok and m["frames"] == 4 and m["delin"] == -1 and m["fetch"] == [-1, 1, 3, 5]
    # (the fused epilogue may be empty: a plain rgba16 store)
    assert match_mix([LIN, MIX_ADD, FETCH, LIN, MIX_ADD, MIX_END, DELIN])[0]
    assert match_mix([LIN, MIX_ADD, FETCH, LIN, MIX_ADD, MIX_END, DELIN, SCALE])[0]
    # not its business: one frame, a resampled first frame, anything else between the ops
    assert not match_mix([LIN, MIX_ADD, MIX_END, DELIN, DITHER, SCALE])[0]
    assert not match_mix([LIN, MIX_ADD, FETCH, LIN, MIX_ADD, MIX_END, DELIN, DITHER, SCALE], sampler=BILINEAR)[0]
    assert not match_mix([LIN, AFFINE, MIX_ADD, FETCH, LIN, MIX_ADD, MIX_END, DELIN, DITHER, SCALE])[0]
    assert not match_mix([LIN, MIX_ADD, FETCH, LIN, MIX_ADD, MIX_END, DELIN, RGB2IPT, DITHER, SCALE])[0]
    assert not match_mix([LIN, MIX_ADD, FETCH, LIN, MIX_ADD])[0]                         # no MIX_END
