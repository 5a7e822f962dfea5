"""plh_colormap_resolve (csrc/host/colormap_plan.c): what a colour-mapping request resolves to,
as one value -- against the same decision taken in Python over the REFERENCE's own CPU build
(tests/colormap_ref.py::resolve; reference src/shaders/colorspace.c:1612-1790), for every tone
and gamut function and for the switches that change the decision (no state object, forced LUT,
inverse tone mapping, gamut expansion, measured source peak)."""
import ctypes as C
import itertools

import pytest

import colormap_ref as cr
import libplacebo_amd as pl
import ref_structs as R
from libplacebo_amd import _capi as capi


class Plan(C.Structure):
    _fields_ = [("src", R.Csp), ("dst", R.Csp), ("identity", C.c_bool), ("tone", R.TMP),
                ("gamut", R.GMP), ("closed_form", C.c_bool), ("need_tone", C.c_bool),
                ("need_gamut", C.c_bool), ("tone_direct", C.c_bool), ("fold_saturation", C.c_bool)]


@pytest.fixture(scope="module")
def lib(built):
    lib = R.declare(pl.lib())
    lib.plh_test_colormap_resolve.restype = None
    return lib


def names(lib):
    tone = {lib.pl_find_tone_map_function(n): n for n in R.TONE_NAMES}
    gamut = {lib.pl_find_gamut_map_function(n): n for n in R.GAMUT_NAMES}
    return tone, gamut


def plan(lib, src, dst, tone, gamut, stateful=True, **kw):
    par = capi.ColorMapParams.in_dll(lib, "pl_color_map_default_params")
    par = capi.ColorMapParams.from_buffer_copy(par)
    par.tone_mapping_function = lib.pl_find_tone_map_function(tone)
    par.gamut_mapping = lib.pl_find_gamut_map_function(gamut)
    for k, v in kw.items():
        setattr(par, k, v)
    out = Plan()
    lib.plh_test_colormap_resolve(C.byref(out), C.byref(par), C.byref(src), C.byref(dst), C.c_bool(stateful))
    return out


def spaces(which):
    if which == "hdr10_to_sdr":
        return (cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=1000.0),
                cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]))
    if which == "hdr10_measured":
        s = cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=4000.0)
        s.hdr.max_pq_y, s.hdr.avg_pq_y = 0.62, 0.31
        return s, cr.make_csp(pl.PRIM["bt709"], pl.TRC["srgb"])
    if which == "sdr_to_hdr10":
        return (cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]),
                cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=1000.0))
    if which == "hlg_to_p3":
        return (cr.make_csp(pl.PRIM["bt2020"], pl.TRC["hlg"]),
                cr.make_csp(pl.PRIM["display_p3"], pl.TRC["bt1886"]))
    if which == "same":
        return (cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]),
                cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]))
    raise KeyError(which)


def copy_csp(c):
    return R.Csp.from_buffer_copy(c)


def assert_same_decision(lib, p, want, tone_name, gamut_name):
    tnames, gnames = names(lib)
    assert not p.identity
    t, wt = p.tone, want["tone"]
    assert tnames[t.function] == tone_name
    for f in ("input_min", "input_max", "input_avg", "output_min", "output_max", "lut_size",
              "input_scaling", "output_scaling"):
        assert getattr(t, f) == getattr(wt, f), (f, getattr(t, f), getattr(wt, f))
    g, wg = p.gamut, want["gamut"]
    assert gnames[g.function] == gamut_name
    for f in ("min_luma", "max_luma", "lut_size_I", "lut_size_C", "lut_size_h", "lut_stride"):
        assert getattr(g, f) == getattr(wg, f), f
    assert bytes(g.input_gamut) == bytes(wg.input_gamut)
    assert bytes(g.output_gamut) == bytes(wg.output_gamut)
    assert bytes(p.src) == bytes(want["src"]) and bytes(p.dst) == bytes(want["dst"])


@pytest.mark.parametrize("which", ["hdr10_to_sdr", "hdr10_measured", "sdr_to_hdr10", "hlg_to_p3"])
def test_every_function_resolves_like_the_reference(lib, which):
    for tone, gamut in itertools.product(R.TONE_NAMES, R.GAMUT_NAMES):
        src, dst = spaces(which)
        want = cr.resolve(copy_csp(src), copy_csp(dst), tone=tone, gamut=gamut)
        p = plan(lib, src, dst, tone, gamut)
        assert_same_decision(lib, p, want, tone, gamut)
        # which steps remain: the `saturation` mapper is a matrix, `clip` / `linear` curves are
        # closed-form -- unless the LUT is forced
        fold = want["need_gamut"] and gamut == b"saturation"
        assert p.closed_form and p.need_tone == want["need_tone"]
        assert p.fold_saturation == fold and p.need_gamut == (want["need_gamut"] and not fold)
        assert p.tone_direct == (want["need_tone"] and tone in (b"clip", b"linear"))
        forced = plan(lib, src, dst, tone, gamut, force_tone_mapping_lut=True)
        assert not forced.closed_form and not forced.tone_direct and not forced.fold_saturation
        assert forced.need_gamut == want["need_gamut"]


def test_equal_spaces_are_an_identity(lib):
    src, dst = spaces("same")
    p = plan(lib, src, dst, b"spline", b"perceptual")
    assert p.identity and not p.need_tone and not p.need_gamut


def test_without_a_state_object_only_closed_form_steps_remain(lib):
    tnames, gnames = names(lib)
    src, dst = spaces("hdr10_to_sdr")
    for tone, gamut in itertools.product([b"clip", b"spline", b"bt2390"], [b"clip", b"perceptual", b"darken"]):
        p = plan(lib, src, dst, tone, gamut, stateful=False, force_tone_mapping_lut=True)
        assert p.closed_form                                     # (the forced LUT has nowhere to live)
        assert tnames[p.tone.function] == (b"clip" if tone == b"clip" else b"linear")
        assert gnames[p.gamut.function] == (b"clip" if gamut == b"clip" else b"saturation")
        assert p.tone_direct == p.need_tone
        assert p.fold_saturation == (gamut != b"clip") and (not p.need_gamut or gamut == b"clip")


def test_inverse_tone_mapping_and_gamut_expansion(lib):
    # SDR -> HDR10: without inverse tone mapping the curve's top stays at the source peak
    src, dst = spaces("sdr_to_hdr10")
    off = plan(lib, src, dst, b"spline", b"perceptual")
    on = plan(lib, src, dst, b"spline", b"perceptual", inverse_tone_mapping=True)
    assert off.tone.output_max == off.tone.input_max and not off.need_tone
    assert on.tone.output_max > on.tone.input_max and on.need_tone
    # a bidirectional mapper towards a WIDER gamut: clipped to the source unless expanding
    assert bytes(off.gamut.output_gamut) == bytes(off.gamut.input_gamut)
    wide = plan(lib, src, dst, b"spline", b"perceptual", gamut_expansion=True)
    assert bytes(wide.gamut.output_gamut) == bytes(wide.dst.hdr.prim) != bytes(wide.gamut.input_gamut)
    # a one-directional mapper is never clipped
    one = plan(lib, src, dst, b"spline", b"darken")
    assert bytes(one.gamut.output_gamut) == bytes(one.dst.hdr.prim)
