"""pl_frame.rotation (reference src/renderer.c:3113-3117, 2787-2793, src/common.c:469-500): the
image is processed in its own orientation and only the final stores are transposed / flipped, so
a rotated render must equal numpy's rotation of the unrotated render (into a target of the
counter-rotated size) bit for bit."""
import numpy as np
import pytest

import libplacebo_amd as pl
import util
from libplacebo_amd import _capi as capi

pytestmark = pytest.mark.gpu


def render(gpu, img, dw, dh, params, rotation=0, target_rotation=0, ten_bit=False, crop=None):
    sh, sw = img.shape[:2]
    src = gpu.tex_create(sw, sh, "rgba16", img)
    dst = gpu.tex_create(dw, dh, "rgba16")
    image = pl.frame(src, components=3, crop=crop)
    image.rotation = rotation
    target = pl.frame(dst, repr_=pl.color_repr("rgb", "full", sample_depth=16, color_depth=10,
                                                bit_shift=6) if ten_bit else None)
    target.rotation = target_rotation
    rr = pl.Renderer(gpu)
    util.srand(1)
    assert rr.render(image, target, params), gpu.messages[-4:]
    assert rr.errors() == 0
    out = dst.download()
    rr.destroy(); src.destroy(); dst.destroy()
    return out


@pytest.mark.parametrize("scaler", [None, "ewa_lanczos", "lanczos"])
@pytest.mark.parametrize("rot", [1, 2, 3, -1, 5])
def test_rotated_render_equals_rotated_unrotated_render(gpu, scaler, rot):
    sw, sh = 80, 48
    img = util.chirp_rgba16(sw, sh)
    kw = dict(upscaler=pl.filter_config(scaler)) if scaler else {}
    params = pl.render_params("fast", **kw)
    dw, dh = 2 * sw, 2 * sh
    base = render(gpu, img, dw, dh, params)
    k = rot % 4
    tw, th = (dh, dw) if k % 2 else (dw, dh)
    got = render(gpu, img, tw, th, params, rotation=rot)
    want = np.rot90(base, k=-k)            # clockwise
    assert got.shape == want.shape
    polar = scaler == "ewa_lanczos"
    same_ = util.assert_polar_equal if polar else (lambda a, b, what=None: np.testing.assert_array_equal(a, b))
    # (a rotated EWA pass is a transposed one: k_polar_pp, where the unrotated frame may come from
    # the matrix pipe)
    same_(got, want, what=(scaler, rot))
    # rotating the target the other way is the same end-to-end rotation
    got2 = render(gpu, img, tw, th, params, rotation=0, target_rotation=-rot)
    same_(got2, want)
    # image and target rotated alike: nothing happens
    same = render(gpu, img, dw, dh, params, rotation=rot, target_rotation=rot)
    same_(same, base)


def test_rotation_with_crop_and_flip(gpu):
    sw, sh = 64, 40
    img = util.chirp_rgba16(sw, sh)
    params = pl.render_params("fast", upscaler=pl.filter_config("mitchell"))
    crop = (50.0, 4.0, 6.0, 36.0)          # flipped in x
    base = render(gpu, img, 90, 70, params, crop=crop)
    got = render(gpu, img, 70, 90, params, rotation=1, crop=crop)
    assert np.array_equal(got, np.rot90(base, k=-1))
