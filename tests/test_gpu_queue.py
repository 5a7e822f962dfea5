"""pl_queue driving pl_render_image_mix on the GPU: the reference's own scenario
(src/tests/gpu_tests.c:1467-1586) — 20 frames at 24 fps shown at 60 Hz, pushed out of order with
a delayed EOF under a radius-2 mixer; pulled through `get_frame`-style feeding under the
oversampling mixer; then the same frames marked interlaced. The queue logic itself is pinned
against the reference's frame_queue.c on the CPU (tests/test_frame_queue.py)."""
import ctypes as C

import numpy as np
import pytest

import libplacebo_amd as pl
import util
from test_gpu_mix import mixer, sources, W, H

pytestmark = pytest.mark.gpu

N = 20
FRAME = 1.0 / 24.0
VSYNC = 1.0 / 60.0


@pytest.fixture()
def rr(gpu):
    r = pl.Renderer(gpu)
    yield r
    r.destroy()


def order():
    # gpu_tests.c:1521-1524: the second half arrives in reverse order
    return [i if i <= 10 else N + 10 - i for i in range(N)]


def test_reference_scenario_mixer_and_delayed_eof(gpu, rr):
    imgs, texs, frames = sources(gpu, 4)
    dst = gpu.tex_create(W, H, "rgba16")
    target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"))
    params = pl.render_params("fast", frame_mixer=mixer("mitchell_clamp"))
    radius = pl.frame_mix_radius(params)
    assert radius == 2.0
    q = pl.Queue(gpu)
    for i in order():
        if q.push(frames[i % 4], i * FRAME, FRAME, block_ns=1) is None:
            q.push(frames[i % 4], i * FRAME, FRAME)     # "push it anyway, for testing"
    assert q.num_frames() == N

    pts, shown, sent_eof, widest = 0.0, 0, False, 0
    while True:
        st, mix = q.update(pts, radius=radius, vsync_duration=VSYNC)
        if st == pl.QUEUE_EOF:
            break
        if st == pl.QUEUE_MORE:
            assert pts > 0.0 and not sent_eof
            q.push(None, 0.0)                           # delayed EOF
            sent_eof = True
            continue
        assert st == pl.QUEUE_OK
        widest = max(widest, mix.num_frames)
        assert lib_render(rr, mix, target, params), gpu.messages[-4:]
        shown += 1
        pts += VSYNC
    assert rr.errors() == 0 and sent_eof
    assert widest >= 5                                  # +-2 source frames around the vsync
    assert shown == pytest.approx(N * FRAME / VSYNC, abs=3)
    assert abs(q.estimate_fps() - 24.0) < 0.01 and abs(q.estimate_vps() - 60.0) < 0.01
    q.destroy()
    assert len(q.unmapped) == N                         # every frame was given back
    for t in texs + [dst]:
        t.destroy()


def lib_render(rr, mix, target, params):
    return pl.lib().pl_render_image_mix(rr.rr, C.byref(mix), C.byref(target), C.byref(params))


def test_oversample_then_interlaced(gpu, rr):
    imgs, texs, frames = sources(gpu, 4)
    dst = gpu.tex_create(W, H, "rgba16")
    target = pl.frame(dst, color=pl.color_space("bt709", "bt1886"))
    params = pl.render_params("fast", frame_mixer=mixer("oversample"))
    assert pl.frame_mix_radius(params) == 0.0
    q = pl.Queue(gpu)

    # fed just in time, as a get_frame callback would: one frame whenever the queue asks for MORE
    fed, pts = 0, 0.0
    while True:
        st, mix = q.update(pts, vsync_duration=VSYNC)
        if st == pl.QUEUE_MORE:
            q.push(frames[fed % 4] if fed < N else None, fed * FRAME, FRAME)
            fed += 1
            continue
        if st == pl.QUEUE_EOF:
            break
        assert st == pl.QUEUE_OK and mix.num_frames <= 2
        assert lib_render(rr, mix, target, params)
        pts += VSYNC
    assert fed == N + 1

    # a frame exactly on the vsync is shown alone: same picture as pl_render_image
    q.reset()
    for i in range(3):
        q.push(frames[i], i * FRAME, FRAME)
    st, mix = q.update(0.0, vsync_duration=VSYNC)
    assert st == pl.QUEUE_OK and lib_render(rr, mix, target, params)
    got = dst.download()
    assert rr.render(frames[0], target, pl.render_params("fast"))
    assert np.abs(got.astype(np.int64) - dst.download()).max() <= 64  # the cache is f16

    # gpu_tests.c:1563-1564: nothing queued, the source is at its end
    q.reset()
    q.push(None, 0.0)
    assert q.update(pts, vsync_duration=VSYNC)[0] == pl.QUEUE_EOF

    # gpu_tests.c:1566-1584: interlaced, out of order; every update is OK until EOF
    q.reset()
    for i in order():
        q.push(frames[i % 4], i * FRAME, FRAME, first_field=1)
    q.push(None, 0.0)
    assert q.num_frames() == 2 * N
    pts, fields = 0.0, set()
    while True:
        st, mix = q.update(pts, vsync_duration=VSYNC)
        if st == pl.QUEUE_EOF:
            break
        assert st == pl.QUEUE_OK
        for i in range(mix.num_frames):
            f = mix.frames[i].contents
            fields.add(f.field)
            assert f.first_field == 1
        assert lib_render(rr, mix, target, params)
        pts += VSYNC
    assert fields == {1, 2} and rr.errors() == 0
    q.destroy()
    for t in texs + [dst]:
        t.destroy()


def test_host_frames_uploaded_in_map_recycle_the_queue_textures(gpu, rr):
    """frame_queue.h: `map` gets four queue-owned texture slots to (re)create its planes in
    (pl_upload_plane); the queue invalidates and recycles them when a frame leaves, and destroys
    them with the queue. The rendered mix must equal the one from pre-uploaded frames."""
    n = 12
    imgs, texs, frames = sources(gpu, 4)
    dst_a, dst_b = gpu.tex_create(W, H, "rgba16"), gpu.tex_create(W, H, "rgba16")
    csp = pl.color_space("bt709", "bt1886")
    ta, tb = pl.frame(dst_a, color=csp), pl.frame(dst_b, color=csp)
    params = pl.render_params("fast", frame_mixer=mixer("oversample"))
    qa, qb = pl.Queue(gpu), pl.Queue(gpu)
    def feed(i):
        if i == n:
            qa.push(None, 0.0); qb.push(None, 0.0)
            return
        qa.push(frames[i % 4], i * FRAME, FRAME)
        host = ([pl.plane_data(imgs[i % 4], (16, 16, 16, 16))], frames[i % 4].repr, frames[i % 4].color)
        qb.push(host, i * FRAME, FRAME)
    pts, compared, fed = 0.0, 0, 0
    while True:
        # a decoder two frames ahead of the display (recycled textures are handed out at push time)
        while fed <= n and fed * FRAME <= pts + 2 * FRAME:
            feed(fed)
            fed += 1
        (sa, ma), (sb, mb) = (q.update(pts, vsync_duration=VSYNC) for q in (qa, qb))
        assert sa == sb
        if sa == pl.QUEUE_EOF:
            break
        assert sa == pl.QUEUE_OK and ma.num_frames == mb.num_frames
        assert lib_render(rr, ma, ta, params) and lib_render(rr, mb, tb, params)
        # (different signatures, so both renders really happen)
        assert np.array_equal(dst_a.download(), dst_b.download())
        compared += 1
        pts += VSYNC
    assert compared > 20 and rr.errors() == 0
    # the first frames got fresh textures, later ones the recycled sets
    assert qb.uploads[:2] == [False, False] and sum(qb.uploads) >= n - 4, qb.uploads
    qa.destroy(); qb.destroy()
    for t in texs + [dst_a, dst_b]:
        t.destroy()
