"""utils/upload.h: host plane description -> texture format -> upload.

CPU: the reference's own known answers (src/tests/utils.c:9-98) and a fuzz of the pure
helpers against the real reference code (oracle/_ref, built from src/utils/upload.c).
GPU: format matching for the usual video layouts, pl_upload_plane round trips (row stride,
byte-swapped samples) and an NV12 frame rendered from uploaded planes (the shape of
pl_ycbcr_tests, src/tests/gpu_tests.c:1599-1731)."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import orc
from libplacebo_amd import _capi as capi


def from_mask(lib, masks):
    d = capi.PlaneData()
    m = (C.c_uint64 * 4)(*(list(masks) + [0] * (4 - len(masks))))
    lib.pl_plane_data_from_mask(C.byref(d), m)
    return d


def fields(d):
    return (list(d.component_size), list(d.component_pad), list(d.component_map))


def align(lib, d):
    bits = capi.BitEncoding()
    ok = lib.pl_plane_data_align(C.byref(d), C.byref(bits))
    return ok, (bits.sample_depth, bits.color_depth, bits.bit_shift)


# (masks, after from_mask, after align, bits)  -- src/tests/utils.c:44-98
Z = [0, 0, 0, 0]
KNOWN = [
    ([0xFF, 0xFF00, 0xFF0000], ([8, 8, 8, 0], Z, [0, 1, 2, 0]), None, (8, 8, 0)),
    ([0xFF0000, 0xFF00, 0xFF, 0xFF000000], ([8, 8, 8, 8], Z, [2, 1, 0, 3]), None, (8, 8, 0)),
    ([0xFFFF0000, 0xFFFF], ([16, 16, 0, 0], Z, [1, 0, 0, 0]), None, (16, 16, 0)),
    ([0x03FF0000, 0x03FF], ([10, 10, 0, 0], [0, 6, 0, 0], [1, 0, 0, 0]),
     ([16, 16, 0, 0], Z, [1, 0, 0, 0]), (16, 10, 0)),
    ([0xF800, 0x07E0, 0x001F], ([5, 6, 5, 0], Z, [2, 1, 0, 0]), None, (0, 0, 0)),
    ([0xFFFF, 0xFFFF0000, 0xFFFF00000000, 0xFFFF000000000000],
     ([16, 16, 16, 16], Z, [0, 1, 2, 3]), None, (16, 16, 0)),
    ([0xFFC0, 0xFFC00000, 0xFFC000000000], ([10, 10, 10, 0], [6, 6, 6, 0], [0, 1, 2, 0]),
     ([16, 16, 16, 0], Z, [0, 1, 2, 0]), (16, 10, 6)),
]


@pytest.mark.parametrize("case", range(len(KNOWN)))
def test_reference_known_answers(built, case):
    lib = pl.lib()
    masks, want, want_aligned, want_bits = KNOWN[case]
    d = from_mask(lib, masks)
    assert fields(d) == want
    _, bits = align(lib, d)
    assert fields(d) == (want_aligned or want)
    assert bits == want_bits


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built")
def test_pure_helpers_match_reference_code(built):
    our = capi.declare(C.CDLL(capi.LIB_PATH))
    ref = orc.ref()
    for f in ("pl_plane_data_from_comps", "pl_plane_data_align"):
        getattr(ref, f).argtypes = getattr(our, f).argtypes
        getattr(ref, f).restype = getattr(our, f).restype
    rng = np.random.default_rng(7)
    aligned = 0
    for it in range(4000):
        # random non-overlapping bit fields in a 64-bit (sometimes wider) pixel
        n = int(rng.integers(1, 5))
        sizes, shifts, pos = [], [], 0
        for _ in range(n):
            pos += int(rng.choice([0, 0, 0, 1, 2, 4, 6, 8]))
            sz = int(rng.choice([1, 2, 4, 5, 6, 8, 10, 12, 16, 32]))
            sizes.append(sz); shifts.append(pos)
            pos += sz
        order = rng.permutation(n)
        size = [0] * 4; shift = [0] * 4
        slots = rng.permutation(4)[:n]
        for k, c in enumerate(slots):
            size[c] = sizes[order[k]]; shift[c] = shifts[order[k]]
        res = []
        for lib in (ref, our):
            d = capi.PlaneData(pixel_stride=int(rng.choice([0, 8])) if lib is ref else 0)
            res.append(d)
        res[1].pixel_stride = res[0].pixel_stride
        out = []
        for lib, d in zip((ref, our), res):
            lib.pl_plane_data_from_comps(C.byref(d), (C.c_int * 4)(*size), (C.c_int * 4)(*shift))
            f0 = fields(d)
            ok, bits = align(lib, d)
            out.append((f0, ok, bits, fields(d)))
        assert out[0] == out[1], (size, shift, out)
        aligned += out[0][1]
    assert 200 < aligned < 3800     # both outcomes are exercised


# --------------------------------------------------------------------------- GPU

def pd(w, h, bits, stride, cmap=None, pad=None, typ=pl.FMT_UNORM):
    d = capi.PlaneData(type=typ, width=w, height=h, pixel_stride=stride)
    for c, b in enumerate(bits):
        d.component_size[c] = b
        d.component_map[c] = cmap[c] if cmap else c
        d.component_pad[c] = pad[c] if pad else 0
    return d


@pytest.mark.gpu
def test_find_fmt_video_layouts(gpu):
    lib = pl.lib()

    def find(d):
        m = (C.c_int * 4)()
        f = lib.pl_plane_find_fmt(gpu.gpu, m, C.byref(d))
        return (f.contents.name.decode() if f else None), list(m)

    assert find(pd(64, 64, [8], 1)) == ("r8", [0, -1, -1, -1])                  # Y of NV12
    assert find(pd(32, 32, [8, 8], 2, [1, 2])) == ("rg8", [1, 2, -1, -1])        # UV of NV12
    assert find(pd(32, 32, [8, 8], 2, [2, 1])) == ("rg8", [2, 1, -1, -1])        # NV21
    assert find(pd(64, 64, [16], 2)) == ("r16", [0, -1, -1, -1])                 # P016 / yuv420p16
    assert find(pd(64, 64, [8, 8, 8, 8], 4, [2, 1, 0, 3])) == ("rgba8", [2, 1, 0, 3])    # bgra8
    assert find(pd(64, 64, [8, 8, 8], 4, [0, 1, 2], [8, 0, 0])) == ("rgba8", [-1, 0, 1, 2])  # xrgb8
    assert find(pd(64, 64, [8, 8, 8], 4)) == ("rgba8", [0, 1, 2, -1])            # rgbx8
    assert find(pd(64, 64, [16] * 4, 8)) == ("rgba16", [0, 1, 2, 3])
    assert find(pd(8, 8, [32], 4, typ=pl.FMT_FLOAT)) == ("r32f", [0, -1, -1, -1])
    assert find(pd(8, 8, [16] * 4, 8, typ=pl.FMT_FLOAT)) == ("rgba16hf", [0, 1, 2, 3])
    # no such texture formats here: packed 24-bit rgb, 10-bit unaligned, rgb565
    assert find(pd(64, 64, [8, 8, 8], 3))[0] is None
    assert find(pd(64, 64, [10], 2, pad=[6]))[0] is None
    assert find(pd(64, 64, [5, 6, 5], 2))[0] is None
    # ... but the 10-in-16 layout aligns to r16 (what p010 users do first)
    d = pd(64, 64, [10], 2, pad=[6])
    ok, bits = align(lib, d)
    assert ok and bits == (16, 10, 6) and find(d)[0] == "r16"
    # a row stride that is not a multiple of the texel is rejected
    d = pd(64, 64, [16], 2)
    d.row_stride = 129
    assert find(d)[0] is None


@pytest.mark.gpu
def test_upload_plane_round_trips(gpu):
    rng = np.random.default_rng(3)
    # tightly packed rg8
    a = rng.integers(0, 256, (37, 53, 2), dtype=np.uint8)
    plane, tex = pl.upload_plane(gpu, pl.plane_data(a, [8, 8], [1, 2], pixel_stride=2))
    assert tex.fmt_name == "rg8" and plane.components == 2
    assert list(plane.component_mapping) == [1, 2, -1, -1]
    assert np.array_equal(tex.download(), a)
    # the same texture object is reused for compatible data, recreated otherwise
    b = rng.integers(0, 256, (37, 53, 2), dtype=np.uint8)
    ptr0 = C.addressof(tex.ptr.contents)
    pl.upload_plane(gpu, pl.plane_data(b, [8, 8], [1, 2], pixel_stride=2), tex)
    assert C.addressof(tex.ptr.contents) == ptr0 and np.array_equal(tex.download(), b)
    c = rng.integers(0, 65536, (20, 24, 1), dtype=np.uint16)
    pl.upload_plane(gpu, pl.plane_data(c, [16], pixel_stride=2), tex)
    assert tex.fmt_name == "r16" and (tex.w, tex.h) == (24, 20)
    assert np.array_equal(tex.download(), c)
    tex.destroy()

    # padded rows (row_stride) of r16, then the same samples byte-swapped
    full = rng.integers(0, 65536, (16, 40, 1), dtype=np.uint16)
    d = pl.plane_data(full, [16], pixel_stride=2, row_stride=80)
    d.width = 31
    _, t2 = pl.upload_plane(gpu, d)
    assert np.array_equal(t2.download(), full[:, :31])
    sw = full.byteswap()
    d = pl.plane_data(sw, [16], pixel_stride=2, row_stride=80, swapped=True)
    d.width = 31
    pl.upload_plane(gpu, d, t2)
    assert np.array_equal(t2.download(), full[:, :31])
    t2.destroy()

    # exactly one data source; unsupported layouts fail loudly
    d = pl.plane_data(a, [8, 8], pixel_stride=2)
    d.pixels = None
    with pytest.raises(RuntimeError):
        pl.upload_plane(gpu, d)
    with pytest.raises(RuntimeError):
        pl.upload_plane(gpu, pl.plane_data(np.zeros((4, 4, 3), np.uint8), [8, 8, 8], pixel_stride=3))


@pytest.mark.gpu
def test_nv12_from_uploaded_planes_matches_manual_planes(gpu):
    """pl_upload_plane output plugged into pl_frame: same pixels as the hand-built frame."""
    rng = np.random.default_rng(11)
    w, h = 64, 48
    y = rng.integers(16, 236, (h, w, 1), dtype=np.uint8)
    uv = rng.integers(16, 241, (h // 2, w // 2, 2), dtype=np.uint8)

    def render(planes):
        img = capi.Frame(num_planes=2)
        for i, p in enumerate(planes):
            img.planes[i] = p
        img.repr = pl.color_repr("bt709", "limited", sample_depth=8, color_depth=8)
        img.color = pl.color_space("bt709", "bt1886")
        pl.lib().pl_frame_set_chroma_location(C.byref(img), 1)  # left
        # target plane through pl_recreate_plane (renderable rgba16 texture)
        tdesc = capi.PlaneData(type=pl.FMT_UNORM, width=w, height=h, pixel_stride=8)
        for c in range(4):
            tdesc.component_size[c] = 16
            tdesc.component_map[c] = c
        tplane, ttex = pl.recreate_plane(gpu, tdesc)
        tgt = capi.Frame(num_planes=1)
        tgt.planes[0] = tplane
        tgt.repr = pl.color_repr("rgb", "full")
        tgt.color = pl.color_space("bt709", "bt1886")
        rr = pl.Renderer(gpu)
        assert rr.render(img, tgt, pl.render_params("fast"))
        out = ttex.download()
        rr.destroy(); ttex.destroy()
        return out

    p0, t0 = pl.upload_plane(gpu, pl.plane_data(y, [8], [0], pixel_stride=1))
    p1, t1 = pl.upload_plane(gpu, pl.plane_data(uv, [8, 8], [1, 2], pixel_stride=2))
    got = render([p0, p1])

    ty = gpu.tex_create(w, h, "r8", y)
    tuv = gpu.tex_create(w // 2, h // 2, "rg8", uv)
    m0 = capi.Plane(texture=ty.ptr, components=1)
    m1 = capi.Plane(texture=tuv.ptr, components=2)
    for c in range(4):
        m0.component_mapping[c] = [0, -1, -1, -1][c]
        m1.component_mapping[c] = [1, 2, -1, -1][c]
    want = render([m0, m1])
    assert np.array_equal(got, want)
    assert got[..., :3].std() > 1000     # a real picture, not a constant
    for t in (t0, t1, ty, tuv):
        t.destroy()
