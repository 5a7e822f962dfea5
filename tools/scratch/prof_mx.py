import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12, 4, 8)
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 512
d = d[:ng].astype(np.int64)
t0 = d[d > 0].min()
names = ["top->issued", "compute", "commit", "redeem", "barrier"]
valid = d[..., 0] > 0
print("groups", ng, "tiles per group: mean %.2f max %d" % (valid[:, :, 0].sum(1).mean(), valid[:, :, 0].sum(1).max()))
for k in range(5):
    dt = (d[..., k + 1] - d[..., k])[valid]
    print("%-12s mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f  (clock ticks)" % (names[k], dt.mean(), np.median(dt), np.quantile(dt, 0.9), dt.max()))
tot = (d[..., 5] - d[..., 0])[valid]
print("tile total   mean %8.0f" % tot.mean())
end = d[..., 5].max() - t0
print("span first stamp -> last stamp: %d ticks" % end)
# per-wave imbalance in compute
c = (d[..., 2] - d[..., 1])
print("compute per wave (mean over groups/its):", [int(c[..., w][valid[..., w]].mean()) for w in range(4)])
# start times of first iteration
st = d[:, 0, 0, 0] - t0
print("first-iteration start: p50 %d max %d" % (np.median(st), st.max()))
last = np.where(valid[:, :, 0], d[:, :, 0, 5], 0).max(1) - t0
print("group finish: p10 %d p50 %d p90 %d max %d" % tuple(np.quantile(last, [0.1, 0.5, 0.9, 1.0])))
