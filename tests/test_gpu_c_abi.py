"""The drop-in boundary exercised from C.

tests/c/render_frame.c is compiled twice (tests/c/Makefile, run by __graft_entry__.build()):
against include/ and against the REFERENCE's own headers (only <libplacebo/hip.h> comes from
this repository), both linked against libplacebo_hip.so. Both must render the same frame,
and that frame must equal the oracle's, bit for bit. The second binary is the proof that
structs laid out by libplacebo's declarations are what this library reads."""
import os
import subprocess

import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "c", "build")


def run(binary, out, sw, sh):
    r = subprocess.run([os.path.join(BUILD, binary), out, str(sw), str(sh)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_c_program_renders_through_the_c_abi(gpu, tmp_path):
    sw, sh = 96, 64
    ours = os.path.join(BUILD, "render_frame_ours")
    assert os.path.exists(ours), "tests/c/build/render_frame_ours missing: run build()"
    log = run("render_frame_ours", str(tmp_path / "ours.raw"), sw, sh)
    raw = np.fromfile(tmp_path / "ours.raw", np.uint16)
    n = 2 * sh * 2 * sw * 4
    got = raw[:n].reshape(2 * sh, 2 * sw, 4)
    # pl_frame_clear_rgba / pl_frame_clear_tiles on a 32 x 16 sRGB frame (renderer.c:4116-4199):
    # one colour, alpha included; then 4-texel tiles of the default colours, alpha 1
    cleared, tiles = raw[n:n + 32 * 16 * 4].reshape(16, 32, 4), raw[n + 32 * 16 * 4:].reshape(16, 32, 4)
    want = np.rint(np.float32([0.25, 0.5, 0.75, 0.5]).astype(np.float64) * 65535).astype(np.uint16)
    assert np.all(cleared == want), cleared[0, 0]
    yy, xx = np.mgrid[0:16, 0:32]
    kx = np.float32(1.0 / 4)
    fx = (np.float32(xx + 0.5) * kx) % 1 < 0.5
    fy = (np.float32(yy + 0.5) * kx) % 1 < 0.5
    c0, c1 = (np.uint16(round(np.float32(v) * 65535.0)) for v in (0.93, 0.87))
    assert np.array_equal(tiles[..., 0], np.where(fx == fy, c0, c1)), (tiles[0, :9, 0], c0, c1)
    assert np.all(tiles[..., 1] == tiles[..., 0]) and np.all(tiles[..., 2] == tiles[..., 0])
    assert np.all(tiles[..., 3] == 65535)

    # the same frame through the oracle (pattern = the C program's, bench.c:32-51)
    yc, xc = (sh - 1) / 2.0, (sw - 1) / 2.0
    y, x = np.mgrid[0:sh, 0:sw].astype(np.float64)
    r2 = (x - xc) ** 2 + (y - yc) ** 2
    phi = 1.6180339887498948
    fr = 0.1 * np.pi * 0.5 / np.sqrt(xc * xc + yc * yc)
    img = np.empty((sh, sw, 4), np.uint16)
    for k, f in enumerate((fr, fr / phi, fr / phi / phi)):
        img[..., k] = np.rint(65535.0 * (0.5 * np.sin(f * r2) + 0.5))
    img[..., 3] = 65535
    a = orc.tex_decode(img, "rgba16")
    a[..., 3] = 1.0
    a = orc.op_quant_f16(a)
    w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
    ref = orc.tex_encode(orc.sample_polar(a, w, r, rz, 2 * sw, 2 * sh, mask=0x7), "rgba16")
    import util
    util.assert_polar_equal(got, ref, what=log)

    # and the binary built against libplacebo's own headers
    if not os.path.exists(os.path.join(BUILD, "render_frame_ref")):
        pytest.skip("render_frame_ref was not built (reference headers absent at build time)")
    log2 = run("render_frame_ref", str(tmp_path / "ref.raw"), sw, sh)
    raw2 = np.fromfile(tmp_path / "ref.raw", np.uint16)
    assert np.array_equal(raw2, raw), (log, log2)
    # both report the reference's struct sizes
    assert "sizeof(pl_frame)=736" in log and "sizeof(pl_frame)=736" in log2, (log, log2)


@pytest.mark.parametrize("binary", ["gpu_contract", "gpu_contract_ref", "pass_roundtrip",
                                    "pass_roundtrip_ref"])
def test_gpu_api_contract_from_c(gpu, binary):
    """tests/c/gpu_contract.c: buffer / texture round trips and every API misuse the reference's
    validation front-end rejects (bounds, flags, alignment, overflow); tests/c/pass_roundtrip.c:
    pl_pass_create / pl_pass_run on a recorded shader (raster and compute flavour, reuse, the
    shader freed first) byte-identical to pl_dispatch_finish -- each compiled against this
    repository's headers and against the reference's"""
    exe = os.path.join(BUILD, binary)
    if not os.path.exists(exe):
        if binary.endswith("_ref"):
            pytest.skip("built only where the reference's headers exist")
        pytest.fail(f"tests/c/build/{binary} missing: run build()")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
