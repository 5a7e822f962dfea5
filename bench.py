#!/usr/bin/env python3
"""bench.py — throughput of the pl_render_image hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`
prints ONE JSON line on rank 0. A "step" is one frame through the hot path on
synthetic input that is already resident in HBM.

Default workload (BASELINE.json configs[2], the one the north-star target is
quoted on): 1920x1080 RGBA16 -> 3840x2160, EWA-Lanczos (Jinc) polar upscale +
blue-noise dither to 10 bit, written as RGBA16. Frames rotate over a pool of
source/target textures larger than the 256 MiB Infinity Cache so that every
frame's compulsory traffic really crosses HBM.

Multi-GPU (--gpus N>1, launched by torch.distributed.run): streams are
independent, one per GPU, no data-path collective -> weak
scaling; value = frames of all ranks / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import libplacebo_amd as pl  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable

WORKLOADS = {
    # name: (src w, src h, dst w, dst h, algorithmic bytes per frame)
    "ewa_lanczos_1080p_to_4k_dither10": (1920, 1080, 3840, 2160,
                                         1920 * 1080 * 8 + 3840 * 2160 * 8),
    "bilinear_1080p_to_4k": (1920, 1080, 3840, 2160, 1920 * 1080 * 8 + 3840 * 2160 * 8),
}


def chirp(w, h):
    import util
    return util.chirp_rgba16(w, h)


class Stream:
    """One independent video stream on one GPU."""

    def __init__(self, device, workload, pool):
        self.g = pl.HipGpu(device)
        self.workload = workload
        sw, sh, dw, dh, _ = WORKLOADS[workload]
        self.dims = (sw, sh, dw, dh)
        frame = chirp(sw, sh)
        self.srcs = [self.g.tex_create(sw, sh, "rgba16", np.roll(frame, 7 * i, axis=1))
                     for i in range(pool)]
        self.fbos = [self.g.tex_create(sw, sh, "rgba16hf") for _ in range(pool)]
        self.dsts = [self.g.tex_create(dw, dh, "rgba16") for _ in range(pool)]
        self.lut, self.dstate = pl.ShaderObj(), pl.ShaderObj()
        self.cfg = pl.filter_config("ewa_lanczos")
        self.pool = pool
        self.i = 0

    def step(self, timer=None):
        g, i = self.g, self.i % self.pool
        sw, sh, dw, dh = self.dims
        self.i += 1
        g.reset_frame()
        if self.workload == "bilinear_1080p_to_4k":
            s = g.begin()
            s.sample("bilinear", self.srcs[i], new_w=dw, new_h=dh)
            assert s.finish(self.dsts[i], timer=timer)
            return
        # PASS A (plane -> rgba16hf FBO), as the reference always does before a
        # complex scaler (renderer.c:2064), then polar + dither into the target
        a = g.begin()
        a.sample("direct", self.srcs[i])
        assert a.finish(self.fbos[i])
        b = g.begin()
        assert b.sample_polar(self.fbos[i], self.cfg, self.lut, new_w=dw, new_h=dh, components=3)
        b.dither(10, self.dstate)
        assert b.finish(self.dsts[i], timer=timer)

    def close(self):
        self.g.finish()
        for t in self.srcs + self.fbos + self.dsts:
            t.destroy()
        self.lut.destroy()
        self.dstate.destroy()
        self.g.close()


def cpu_baseline(workload):
    """The CPU oracle (a scalar port of the reference's algorithm) timed on the
    host, single thread, on a bounded crop of the same workload."""
    import orc
    import util
    sw, sh, dw, dh, _ = WORKLOADS[workload]
    cw, ch = sw, sh  # one whole frame: ~10-20 s of scalar CPU work
    src = chirp(sw, sh)[:ch, :cw]
    tex = orc.tex_decode(src, "rgba16")
    t0 = time.perf_counter()
    if workload.startswith("bilinear"):
        out = orc.sample_simple(tex, orc.S_BILINEAR, cw * 2, ch * 2)
    else:
        img = orc.op_quant_f16(orc.sample_simple(tex, orc.S_BILINEAR, cw, ch))
        w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
        out = orc.sample_polar(img, w, r, rz, cw * 2, ch * 2, mask=0x7)
        orc.dither(out, util.blue_noise(pl), 10)
    orc.tex_encode(out, "rgba16")
    dt = time.perf_counter() - t0
    return {
        "value": round(cw * 2 * ch * 2 / dt / 1e6, 4),
        "unit": "Mpixels/s",
        "cores": 1,
        "kind": "port",
        "sample": f"1 frame {cw}x{ch}->{cw * 2}x{ch * 2} of the same workload, "
                  f"oracle/pl_oracle.c (scalar C, -O2), {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ewa_lanczos_1080p_to_4k_dither10",
                    choices=sorted(WORKLOADS))
    ap.add_argument("--pool", type=int, default=12,
                    help="rotating source/FBO/target textures per stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    def barrier():
        if dist is not None:
            dist.barrier()

    st = Stream(local_rank, args.workload, args.pool)
    sw, sh, dw, dh, alg_bytes = WORKLOADS[args.workload]

    for _ in range(args.warmup):
        st.step()
    st.g.finish()
    torch.cuda.synchronize()
    barrier()

    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.step()
    st.g.finish()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- dominant-kernel time, HIP events on the pass' own stream -----------------
    roofline = None
    if rank == 0:
        timer = st.g.timer()
        samples = []
        for _ in range(64):
            st.step(timer=timer)
            if len(samples) < 64 and (_ % 8) == 7:
                st.g.finish()
                while True:
                    ns = st.g.timer_query(timer)
                    if not ns:
                        break
                    samples.append(ns)
        st.g.finish()
        while True:
            ns = st.g.timer_query(timer)
            if not ns:
                break
            samples.append(ns)
        kern_s = float(np.mean(samples)) * 1e-9
        achieved = alg_bytes / kern_s / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            with open(prof) as f:
                traffic = json.load(f).get(args.workload)
        roofline = {
            "bound": "hbm",
            "kernel": "k_pass_generic" if args.workload.startswith("bilinear") else "k_polar",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "kernel_us": round(kern_s * 1e6, 2),
            "algorithmic_bytes": alg_bytes,
        }

    if rank == 0:
        frames = args.steps * world
        out = {
            "metric": "Mpixels/s (output) EWA-Lanczos 1080p->4K upscale + dither, per-GPU streams",
            "value": round(frames * dw * dh / elapsed / 1e6, 1),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "frames_per_s": round(frames / elapsed, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "src": f"{sw}x{sh} rgba16", "dst": f"{dw}x{dh} rgba16",
                "pool": args.pool,
                "passes": "sample->rgba16hf FBO, polar EWA + dither" if not
                          args.workload.startswith("bilinear") else "bilinear",
                "parallelism": f"{world} independent stream(s), one per GPU",
            },
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out))

    st.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
