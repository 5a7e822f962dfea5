"""pl_hip_params.async_measure: the measuring pass of a frame on a second stream (include/libplacebo/hip.h,
gpu_hip.c "two streams"). What it must not change: any pixel, any measurement. The sequences below re-upload
their sources between frames and reuse targets, so that every ordering the two streams need (upload -> measure,
measure -> scale, scale -> measure two frames on, measure -> re-upload) is on the path."""
import os

import numpy as np
import pytest

import libplacebo_amd as pl
from tests import util

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("PL_HIP_ASYNC_MEASURE") == "0",
                                 reason="the environment forces one stream: nothing to compare")]


def hdr_frames(w, h, n, seed):
    """frames with different peaks, so that a measurement read a frame late would show"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        img = util.random_rgba16(w, h, seed=seed + i).astype(np.float64)
        img *= rng.uniform(0.35, 1.0)
        out.append(img.astype(np.uint16))
    return out


def run_sequence(async_measure, frames, upscale, peak, delayed=False, host_syncs=True):
    """host_syncs=False: the sources are resident and copied into place on the stream, nothing is read back
    before the end -- the host never waits for the main stream, so every dependency between the two streams
    has to be expressed on the GPU."""
    h, w = frames[0].shape[:2]
    hdr, sdr = pl.color_space("bt2020", "pq", max_luma=1000.0), pl.color_space("bt709", "bt1886")
    outs = []
    with pl.HipGpu(0, log_level=4, async_measure=async_measure) as g:
        rr = pl.Renderer(g)
        src = [g.tex_create(w, h, "rgba16", frames[0]) for _ in range(2)]
        dst = [g.tex_create(upscale * w, upscale * h, "rgba16") for _ in range(2)]
        pd = pl.peak_detect_params(percentile=99.995, allow_delayed=delayed) if peak else None
        kw = dict(upscaler=pl.filter_config("ewa_lanczos")) if upscale > 1 else {}
        params = pl.render_params("default", peak_detect_params=pd, **kw)
        resident = [] if host_syncs else [g.tex_create(w, h, "rgba16", f) for f in frames]
        if not host_syncs:
            dst = [g.tex_create(upscale * w, upscale * h, "rgba16") for _ in frames]
        for i, f in enumerate(frames):
            s, d = src[i % 2], dst[i % len(dst)]
            if host_syncs:
                s.upload(f)
            else:
                s.blit_from(resident[i])
            assert rr.render(pl.frame(s, components=3, color=hdr), pl.frame(d, color=sdr), params), \
                g.messages[-3:]
            if host_syncs and (i % 3 == 2 or i == len(frames) - 1):
                outs.append(d.download())       # (not every frame: keep the host running ahead)
        if not host_syncs:
            outs = [d.download() for d in dst]
        assert rr.errors() == 0
        second_stream = any("second stream" in m for _, m in g.messages)
        rr.destroy()
        for t in src + dst + resident:
            t.destroy()
    return outs, second_stream


@pytest.mark.parametrize("upscale", [1, 2])
def test_async_measure_renders_the_same_frames(upscale):
    frames = hdr_frames(384, 216, 9, seed=70)
    ref, used_ref = run_sequence(False, frames, upscale, peak=True)
    got, used = run_sequence(True, frames, upscale, peak=True)
    assert used
    assert not used_ref or os.environ.get("PL_HIP_ASYNC_MEASURE")     # (the override forces it on)
    assert len(got) == len(ref)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("delayed", [False, True])
def test_async_measure_without_host_syncs(delayed):
    """Sources written by the main stream right before they are measured on the other one, and rewritten
    while the measurement of the frame before may still be reading them."""
    frames = hdr_frames(640, 360, 8, seed=11) if not delayed else [util.random_rgba16(640, 360, seed=12)] * 8
    ref, _ = run_sequence(False, frames, 2, peak=True, host_syncs=False)
    got, used = run_sequence(True, frames, 2, peak=True, delayed=delayed, host_syncs=False)
    assert used
    # (delayed: which frames already see a measurement is a matter of timing; the last one does)
    for a, b in list(zip(ref, got))[-1 if delayed else 0:]:
        assert np.array_equal(a, b)


def test_async_measure_without_a_measurement_stays_on_one_stream():
    frames = hdr_frames(256, 144, 3, seed=90)
    ref, _ = run_sequence(False, frames, 2, peak=False)
    got, used = run_sequence(True, frames, 2, peak=False)
    assert not used
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)


def test_async_measure_delayed_results_are_consistent():
    """allow_delayed: a frame may go out with the previous measurement (which one is a matter of timing, in
    either mode). With a static scene every measurement is the same, so the output must be too -- after the
    first frame, which may predate any result."""
    frames = [util.random_rgba16(384, 216, seed=33)] * 7
    ref, _ = run_sequence(False, frames, 2, peak=True)
    got, used = run_sequence(True, frames, 2, peak=True, delayed=True)
    assert used
    assert np.array_equal(ref[-1], got[-1])


def test_async_measure_full_size_metric_frame():
    """The metric's geometry (1080p -> 4K): long second pass, so the measurement of frame N+1 really runs
    beside the scaler of frame N."""
    frames = hdr_frames(1920, 1080, 4, seed=5)
    ref, _ = run_sequence(False, frames, 2, peak=True)
    got, used = run_sequence(True, frames, 2, peak=True)
    assert used
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
