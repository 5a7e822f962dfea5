"""How long does the host side of one pl_render_image call take (no GPU wait)?  Enqueue N frames,
take the time before and after the final sync."""
import sys, time
sys.path[:0] = ["tests", "."]
import bench
import libplacebo_amd as pl
wl = sys.argv[1] if len(sys.argv) > 1 else "ewa_lanczos_1080p_to_4k_dither10"
st = bench.Stream(0, wl, 16)
for _ in range(30):
    st.step()
st.g.finish()
for with_cb in (True, False):
    if not with_cb:
        st.params.info_callback = None
    N = 400
    t0 = time.perf_counter()
    for _ in range(N):
        st.step()
    t1 = time.perf_counter()
    st.g.finish()
    t2 = time.perf_counter()
    print(f"{wl} info_callback={with_cb}: enqueue {1e6 * (t1 - t0) / N:.1f} us/frame, "
          f"total {1e6 * (t2 - t0) / N:.1f} us/frame")
st.close()
