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


def test_round5_stages_from_c_against_both_header_sets(gpu, tmp_path):
    """tests/c/frame_stages.c: deinterlacing (bob, bwdif with both neighbours), an overlay, a
    distortion, a blended frame and a Dolby Vision frame through pl_render_image from plain C --
    compiled against include/ and against the reference's own headers (pl_overlay,
    pl_deinterlace_params, pl_distort_params, pl_blend_params, pl_dovi_metadata as libplacebo
    declares them). Identical bytes from both binaries; the frames follow from the inputs."""
    W, H = 64, 48
    outs = {}
    for which in ("ours", "ref"):
        exe = os.path.join(BUILD, f"frame_stages_{which}")
        if not os.path.exists(exe):
            if which == "ref":
                pytest.skip("frame_stages_ref not built (reference headers absent at build time)")
            pytest.fail("tests/c/build/frame_stages_ours missing: run build()")
        path = tmp_path / f"{which}.raw"
        r = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[which] = np.fromfile(path, np.uint16).reshape(6, H, W, 4)
    assert np.array_equal(outs["ours"], outs["ref"])
    bob, bwdif, overlay, turned, blended, dovi = outs["ours"]

    def pattern(t):
        y, x = np.mgrid[0:H, 0:W]
        p = np.empty((H, W, 4), np.uint16)
        p[..., 0] = 1000 * ((x + 3 * t) % 64)
        p[..., 1] = 1300 * (y % 48)
        p[..., 2] = np.where(y % 2, 60000, 5000)
        p[..., 3] = 65535
        return p
    frames = [pattern(t) for t in range(3)]
    # bob: the top field's rows, each twice
    assert np.array_equal(bob[0::2], frames[1][0::2]) and np.array_equal(bob[1::2], frames[1][0::2])
    # bwdif: the oracle's
    dec = [orc.tex_decode(f, "rgba16") for f in frames]
    ref = orc.deinterlace(dec[1], dec[0], dec[2], 1, 1, orc.DEINT_BWDIF, comp_mask=0x7)
    assert np.array_equal(bwdif, orc.tex_encode(ref, "rgba16"))
    # (the blue channel, which is the same in all three frames, comes back as it was: no motion)
    assert np.array_equal(bwdif[..., 2], frames[1][..., 2])
    assert not np.array_equal(bwdif, bob)
    # the overlay: an opaque box of the part's colour, the frame around it
    want = frames[1].copy()
    want[4:12, 8:24] = (65535, 32768, 16384, 65535)
    assert np.array_equal(overlay, want)
    # the half turn, through the renderer's rgba16hf intermediate
    base = orc.op_quant_f16(dec[1].copy())
    assert np.array_equal(turned, orc.tex_encode(base, "rgba16")[::-1, ::-1])
    # the translucent frame over it
    under = orc.tex_decode(turned, "rgba16")
    layer = dec[1].copy()
    layer[..., 3] = np.float32(32768) / np.float32(65535)
    layer = orc.tex_decode(orc.tex_encode(layer, "rgba16"), "rgba16")
    orc.blend(under, layer, None, (2, 3, 1, 3), True, True)
    assert np.array_equal(blended, orc.tex_encode(under, "rgba16"))
    # Dolby Vision with do-nothing metadata: the picture comes back. (The matrix in the C file is the
    # decoder's fixed one inverted to five digits: components near black, where 1e-5 of a
    # neighbouring channel is as much as the channel itself, move by up to a few thousand codes --
    # 5 % of this pattern's columns; the rest by single codes.)
    d = np.abs(dovi[..., :3].astype(np.int64) - frames[1][..., :3].astype(np.int64))
    assert d.mean() < 150 and np.percentile(d, 90) < 200, (float(d.mean()), float(np.percentile(d, 90)))
