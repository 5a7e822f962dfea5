"""A soak of the metric's stream: 150 000 frames, device memory before / after, render errors, and the
frame time of the first and the last 10 000 (nothing may drift)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import util

hip = util.hip_runtime()
def free_mb():
    f, t = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value / 2**20

st = bench.Stream(0, "ewa_1080p_to_4k_hdr_tonemap", 10)
bench.prime(st)
st.g.finish()
m0 = free_mb()
marks = []
t0 = time.perf_counter()
for i in range(150000):
    st.step()
    if i % 10000 == 9999:
        st.g.finish()
        marks.append(time.perf_counter())
m1 = free_mb()
per = [(b - a) / 10000 * 1e6 for a, b in zip([t0] + marks[:-1], marks)]
print("us per frame per 10 000 frames:", " ".join(f"{p:.1f}" for p in per))
print(f"free device memory: {m0:.0f} MiB before, {m1:.0f} MiB after; render errors {st.rr.errors()}")
assert st.rr.errors() == 0 and abs(m0 - m1) < 64 and max(per) < 1.1 * min(per)
st.close()
print("soak ok")
