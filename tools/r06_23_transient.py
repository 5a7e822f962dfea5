"""Where do the driver's 20 steps lose 10 % against 300? Host time per frame over 2000 frames after the
bench's own priming, with and without the drain (finish) in front -- per block of 100 frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

st = bench.Stream(0, "ewa_1080p_to_4k_hdr_tonemap", 10)
bench.prime(st)
for drain in (True, False, True):
    for _ in range(50):
        st.step()
    if drain:
        st.g.finish()
    t = [time.perf_counter()]
    for _ in range(2000):
        st.step()
        t.append(time.perf_counter())
    st.g.finish()
    tend = time.perf_counter()
    blocks = [1e6 * (t[i + 100] - t[i]) / 100 for i in range(0, 2000, 100)]
    print("drain" if drain else "no drain", "host us/frame per 100 frames:", " ".join(f"{b:.1f}" for b in blocks),
          f"| first 200 incl. nothing: {1e6 * (t[200] - t[0]) / 200:.1f} | all incl. finish: {1e6 * (tend - t[0]) / 2000:.1f}")
st.close()
