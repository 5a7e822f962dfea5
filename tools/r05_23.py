#!/usr/bin/env python3
"""Host cost of one pl_render_image call of the metric's frame when nothing is waited for: the
measurement may lag (allow_delayed), so a call only records and launches. Through ctypes (what
bench.py pays) -- and tests/c/build/bench_frames for the same from C."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import libplacebo_amd as pl

st = bench.Stream(0, "ewa_1080p_to_4k_hdr_tonemap", 10)
from libplacebo_amd import _capi as capi
dither = capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0)
st.params = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=dither,
                             peak_detect_params=pl.peak_detect_params(percentile=99.995, allow_delayed=True))
for _ in range(60):
    st.step()
st.g.finish()
for rep in range(3):
    N = 300
    t0 = time.perf_counter()
    for _ in range(N):
        st.step()
    t1 = time.perf_counter()
    st.g.finish()
    t2 = time.perf_counter()
    print(f"metric, delayed measurement: calls {1e6 * (t1 - t0) / N:.1f} us/frame, total {1e6 * (t2 - t0) / N:.1f} us/frame")
st.close()
