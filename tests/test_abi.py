"""The drop-in boundary: the C-ABI shared library loads (every symbol resolved at load time)
and exports every function / object the public headers under include/ declare. No compute
calls, no GPU needed."""
import ctypes as C
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include", "libplacebo")


def declared_symbols():
    syms = {}
    for path in sorted(glob.glob(os.path.join(INCLUDE, "**", "*.h"), recursive=True)):
        text = open(path).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"\bPL_API\b([^;{]*?);", text, flags=re.S):
            decl = " ".join(m.group(1).split())
            if "(" in decl:
                # function: name is the identifier before the first '(' (skip `(*name)` forms)
                head = decl.split("(", 1)[0].strip()
                name = re.findall(r"[A-Za-z_]\w*", head)[-1]
            else:
                # extern object: last identifier (strip array suffixes)
                name = re.findall(r"[A-Za-z_]\w*", decl.split("[", 1)[0])[-1]
            syms[name] = os.path.relpath(path, ROOT)
    return syms


def test_headers_declare_the_expected_entry_points():
    syms = declared_symbols()
    # the hot-path API surface
    for name in ("pl_hip_create", "pl_hip_destroy", "pl_renderer_create", "pl_render_image",
                 "pl_dispatch_begin", "pl_dispatch_finish", "pl_dispatch_compute",
                 "pl_shader_sample_polar", "pl_shader_sample_ortho2", "pl_shader_deband",
                 "pl_shader_dither", "pl_shader_error_diffusion", "pl_shader_detect_peak",
                 "pl_shader_color_map_ex", "pl_shader_linearize", "pl_shader_decode_color",
                 "pl_filter_generate", "pl_tone_map_generate", "pl_gamut_map_generate",
                 "pl_generate_blue_noise", "pl_tex_create", "pl_tex_upload", "pl_tex_download"):
        assert name in syms, name
    assert len(syms) > 150


def test_library_loads_and_exports_every_declared_symbol(built):
    import libplacebo_amd._capi as capi
    # RTLD_NOW: any unresolved reference (a missing kernel launcher, ...) fails here
    lib = C.CDLL(capi.LIB_PATH, mode=os.RTLD_NOW)
    missing = [f"{n} ({h})" for n, h in declared_symbols().items() if not hasattr(lib, n)]
    assert not missing, missing


def test_library_exports_nothing_but_the_api(built):
    """-fvisibility=hidden: only pl_* / plh_test_* leave the library."""
    import libplacebo_amd._capi as capi
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.split()[1:2] and
             ln.split()[-2] in "TDBR"]
    # (memcpy_layout: the one public function of the reference without the prefix, gpu.h:1048)
    stray = [n for n in names if not (n.startswith("pl_") or n.startswith("plh_test_") or n == "memcpy_layout"
                                      or n.startswith("__hip") or n.startswith("_Z")
                                      or n.startswith("__"))]
    assert not stray, stray[:20]
    # device-side launchers must not be part of the ABI
    assert not [n for n in names if n.startswith("plh_launch")]


def test_product_does_not_link_the_oracle(built):
    """The oracle is test infrastructure: the shipped library must not depend on it."""
    import libplacebo_amd._capi as capi
    out = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "ploracle" not in out and "plref" not in out
    for path in glob.glob(os.path.join(ROOT, "libplacebo_amd", "**", "*.*"), recursive=True):
        if path.endswith((".c", ".h", ".hip", ".hiph", ".py")):
            text = open(path, errors="replace").read()
            assert "libploracle" not in text and "libplref" not in text, path
            assert not re.search(r'#include\s*[<"][^>"]*oracle', text), path
            assert not re.search(r"^\s*(import|from)\s+orc\b", text, flags=re.M), path
