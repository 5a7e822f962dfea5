"""The reference's OWN GPU tests, executed on the HIP backend.

oracle/_ref/ref_gpu_tests is /root/reference/src/tests/gpu_tests.c compiled in place against the
reference's public headers and linked to libplacebo_hip.so (oracle/ref_tests/build.sh; the wrapper
is what src/tests/vulkan.c is for the Vulkan backend). Its REQUIREs are the reference's, verbatim:
buffer and texture round trips for every format of the backend, the transfer-function round trips
(1e-6, 1e-4 for HDR curves), the BT.2020 -> scRGB golden triples, the colour-system round trips,
the dispatch cache, peak detection against the CPU formula, Dolby Vision, deinterlacing, error
diffusion (pl_shader_tests); every scaler preset (pl_scaler_tests); the parameter sweeps of
pl_render_tests -- every upscaler and downscaler, debanding, sigmoid, colour-map intents, dither
methods, distortion, cone distortion, gamma-aware dithering, HDR tone mapping, inverse tone mapping,
colour adjustment, inferred frames, tile backgrounds, custom LUTs, overlays, rotation, frame mixing
through pl_queue, deinterlacing; the 4:2:0 round trip within 150 LSB16 (pl_ycbcr_tests).
What was cut, and why (GLSL, film grain, hooks), is listed by the build and printed here."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_gpu_tests")
CUTS = os.path.join(ROOT, "oracle", "_ref", "gen", "gpu_tests_cuts.txt")

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binary(built):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/ref_gpu_tests not built (no /root/reference at build time)")
    if os.path.exists(CUTS):
        print("\n" + open(CUTS).read())
    return BIN


@pytest.mark.parametrize("name", ["buffer", "texture", "planar", "shader", "scaler", "render", "ycbcr"])
def test_reference_gpu_tests(binary, name):
    r = subprocess.run([binary, name], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    tail = r.stdout[-1500:] + "\n--- stderr:\n" + r.stderr[-3000:]
    assert r.returncode == 0, f"pl_{name}_tests of the reference failed on the HIP backend:\n{tail}"
    assert f"=== {name}: done" in r.stdout and "every REQUIRE held" in r.stdout, tail
    assert "=== FAILED" not in r.stderr
