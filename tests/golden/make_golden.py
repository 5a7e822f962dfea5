#!/usr/bin/env python3
"""Generates tests/golden/tier0.npz from the REAL reference CPU half (oracle/_ref/libplref.so,
built from /root/reference by oracle/build_ref.sh). Run in the build container:

    make -C oracle && python tests/golden/make_golden.py

The vectors pin, without needing the reference at test time: pl_filter_generate (polar and
separable LUTs, radii), the dither matrices, pl_tone_map_generate, pl_gamut_map_generate,
the colour matrices, pl_color_repr_decode and the CPU transfer functions.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]

import orc  # noqa: E402
import util  # noqa: E402
from golden_cases import *  # noqa: E402,F401,F403
import golden_cases as gc  # noqa: E402


def main():
    assert orc.have_ref(), "oracle/_ref/libplref.so missing: run `make -C oracle` first"
    out = gc.evaluate(gc.RefLib(orc.ref()))
    path = os.path.join(HERE, "tier0.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
