"""The structs that cross the C ABI are layout-identical to libplacebo's.

tools/abi_probe.py compiles one C probe that prints sizeof / offsetof of every aggregate the
public headers define. The table for include/ must equal

* tests/golden/abi_layout.json -- the same probe compiled against the reference's own headers
  (`tools/abi_probe.py golden`, committed so that the check also runs where /root/reference is
  absent), and
* the live reference headers, where they are present (which also keeps the golden honest).

The ctypes mirrors the Python harness uses (libplacebo_amd/_capi.py) are checked against the
same table, so a test can never pass on a struct that only the harness and the library agree on.
No GPU needed."""
import ctypes as C
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import abi_probe  # noqa: E402

HAVE_REF = os.path.isdir(os.path.join(abi_probe.REF, "src", "include")) and \
    os.path.exists(os.path.join(ROOT, "oracle", "_ref", "gen", "libplacebo", "config.h"))

# the structs VERDICT r01 measured as mismatching + everything else a caller fills in
MUST_COVER = ["struct pl_frame", "struct pl_render_params", "struct pl_tex_params",
              "struct pl_buf_params", "struct pl_tex_transfer_params", "struct pl_fmt_t",
              "struct pl_gpu_t", "struct pl_gpu_limits", "struct pl_shader_res",
              "struct pl_sample_src", "struct pl_sample_filter_params",
              "struct pl_color_map_params", "struct pl_dispatch_params", "struct pl_plane",
              "struct pl_color_space", "struct pl_color_repr", "struct pl_filter_config",
              "struct pl_dispatch_info", "struct pl_pass_params", "struct pl_pass_run_params",
              "struct pl_tone_map_params", "struct pl_gamut_map_params", "struct pl_cache_obj",
              "struct pl_source_frame", "struct pl_queue_params", "struct pl_plane_data"]


@pytest.fixture(scope="module")
def ours():
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    return abi_probe.run_probe("ours", abi_probe.aggregates("ours"))


@pytest.fixture(scope="module")
def golden():
    return {k: (tuple(v) if isinstance(v, list) else v)
            for k, v in json.load(open(abi_probe.GOLDEN)).items()}


def norm(table):
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in table.items()}


def test_layouts_equal_the_reference_golden(ours, golden):
    ours = norm(ours)
    for name in MUST_COVER:
        assert name in golden, f"{name} missing from the golden table"
    bad = {k: (ours.get(k), v) for k, v in golden.items() if ours.get(k) != v}
    assert not bad, f"{len(bad)} layout mismatches, e.g. {list(bad.items())[:8]}"


def test_no_member_missing_or_extra_vs_golden(ours, golden):
    """Every member the reference declares exists here and vice versa (for the aggregates both
    define); our own aggregates are the pl_hip_* ones only."""
    ours = norm(ours)
    ref_aggs = {k for k in golden if "." not in k}
    for agg in ref_aggs:
        theirs = {k for k in golden if k.startswith(agg + ".")}
        mine = {k for k in ours if k.startswith(agg + ".")}
        assert theirs == mine, (agg, sorted(theirs ^ mine))
    own = {k for k in ours if "." not in k} - ref_aggs
    assert own == {"struct pl_hip_params", "struct pl_hip_t", "struct pl_hip_wrap_params"}, own


@pytest.mark.skipif(not HAVE_REF, reason="reference headers not present")
def test_golden_is_what_the_reference_headers_give(golden):
    live = norm(abi_probe.run_probe("ref"))
    assert live == golden, "tests/golden/abi_layout.json is stale: run tools/abi_probe.py golden"


@pytest.mark.skipif(not HAVE_REF, reason="reference headers not present")
def test_members_textually_equal_reference():
    missing, extra, only_ours = abi_probe.member_diff()
    assert not missing and not extra, (missing, extra)


def test_ctypes_mirrors_match_the_headers(ours):
    import libplacebo_amd._capi as capi
    ours = norm(ours)
    bad = []
    for cname, mirror in capi.MIRRORS.items():
        assert cname in ours, cname
        if C.sizeof(mirror) != ours[cname]:
            bad.append((cname, "sizeof", C.sizeof(mirror), ours[cname]))
        for fname, *_ in mirror._fields_:
            key = f"{cname}.{fname.rstrip('_')}"
            assert key in ours, key
            f = getattr(mirror, fname)
            if (f.offset, f.size) != ours[key]:
                bad.append((key, (f.offset, f.size), ours[key]))
        n_c = sum(1 for k in ours if k.startswith(cname + "."))
        if n_c != len(mirror._fields_):
            bad.append((cname, "member count", len(mirror._fields_), n_c))
    assert not bad, bad[:10]
