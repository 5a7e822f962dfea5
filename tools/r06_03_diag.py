"""diagnostic: chain vs interpreter behind the matrix-pipe polar kernel (tests/test_gpu_kernel_variants.py::
test_sdr_chain_equals_interpreter[size0-ewa_lanczos])"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import libplacebo_amd as pl
import util
from test_gpu_kernel_variants import render, _env

for size in ((90, 62), (256, 130)):
    sw, sh = size
    img = util.chirp_rgba16(sw, sh)
    params = pl.render_params("default", upscaler=pl.filter_config("ewa_lanczos"))
    outs = {}
    with pl.HipGpu(0) as gpu:
        for chain in ("1", "0"):
            for mfma in ("1", "0"):
                with _env("PL_HIP_MAP_CHAIN", chain), _env("PL_HIP_POLAR_MFMA", mfma):
                    outs[(chain, mfma)] = render(gpu, img, 2 * sw, 2 * sh, params, True, {}).astype(np.int64)
        msgs = [m for m in gpu.messages if "matrix-pipe" in m]
    print(size, msgs[:2])
    for a, b in ((("1", "1"), ("0", "1")), (("1", "0"), ("0", "0")), (("1", "1"), ("1", "0")), (("0", "1"), ("0", "0"))):
        d = outs[a] - outs[b]
        idx = np.argwhere(d != 0)
        print(f"  chain,mfma {a} vs {b}: {len(idx)} samples differ, max |d| {np.abs(d).max()}", idx[:6].tolist(),
              [int(d[tuple(i)]) for i in idx[:6]])
