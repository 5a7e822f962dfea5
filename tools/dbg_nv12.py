import sys
sys.path[:0] = ["tests", "."]
import bench, libplacebo_amd as pl
orig = pl.HipGpu.__init__
def init(self, device=0, stream=None, log_level=3, max_shmem_size=0):
    orig(self, device, stream, 5, max_shmem_size)
pl.HipGpu.__init__ = init
st = bench.Stream(0, "nv12_1080p_to_4k_ewa_dither10", 2)
st.step()
st.g.finish()
for lev, m in st.g.messages:
    if "polar" in m or "class" in m or "fall" in m:
        print(lev, m[:200])
st.close()
