#!/usr/bin/env python3
"""bench.py — throughput of the pl_render_image hot path on MI355X.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE
JSON line on rank 0. A "step" is one frame through `pl_render_image` on synthetic input that is
already resident in HBM.

Workloads (BASELINE.json `configs`):
  ewa_lanczos_1080p_to_4k_dither10   configs[2], the default: the configuration the north-star
        target (">= 70 % of HBM roofline on EWA-Lanczos 1080p->4K") and the metric are quoted on.
        1920x1080 RGBA16 -> 3840x2160, EWA-Lanczos (Jinc) polar upscale + blue-noise dither to
        10 bit in an RGBA16 target. One launch per frame: the reference's PASS A (plane ->
        rgba16hf FBO, renderer.c:2064) is folded into the polar kernel as per-source-texel
        pre-ops (same values, PL_HIP_NO_FUSION=1 restores the two-pass structure).
  bilinear_1080p_to_4k               configs[1]: bilinear + sRGB passthrough, one pass.
  lanczos_1080p_to_4k_dither10       the separable (two-pass) Lanczos upscale, same output format
  nv12_1080p_to_4k_ewa_dither10      NV12 source planes (pl_upload_plane layout), EWA-Lanczos 2x
  nv12_1080p_to_4k_default_preset    NV12 source, pl_render_default_params untouched
  default_preset_1080p_to_4k         pl_render_default_params untouched (lanczos, sigmoid, dither)
  hdr10_4k_tonemap                   configs[3]: 4K BT.2020/PQ -> BT.709 SDR, same-frame peak
        detection (histogram) + spline tone mapping + perceptual gamut mapping 3D-LUT.
  ewa_8k_to_4k_deband_tonemap        configs[4], one stream: 8K HDR -> 4K SDR, deband + EWA
        downscale + tone map.
  ewa_1080p_to_4k_hdr_tonemap        both halves of the metric's name in one frame: 1080p HDR10
        -> peak detect -> EWA-Lanczos 2x -> tone/gamut map -> dither -> 4K SDR.
  mix_24_to_60_ewa_1080p_to_4k       SURVEY 8f rank 3: a 24 fps stream shown at 60 Hz through pl_queue
        + pl_render_image_mix (oversampling mixer); a step is one vsync.

The default run (N = 1) also reports the two tone-mapping workloads under "companions" in the same
JSON line (the metric's name reads "EWA-Lanczos 1080p->4K + HDR tonemap"); `value` is the headline
workload alone.

Frames rotate over a pool of source/target textures larger than the 256 MiB Infinity Cache so
that every frame's compulsory traffic really crosses HBM.

Multi-GPU (--gpus N>1, launched by torch.distributed.run): streams are independent, one per
GPU, no data-path collective -> weak scaling; value = frames of all ranks /
max-over-ranks time.
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
from libplacebo_amd import _capi as capi  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable

P1080, P4K, P8K = (1920, 1080), (3840, 2160), (7680, 4320)


def px(dim):
    return dim[0] * dim[1]


WORKLOADS = {
    # name: (src dims, dst dims, algorithmic bytes per frame, dominant pass)
    "ewa_lanczos_1080p_to_4k_dither10": (P1080, P4K, px(P1080) * 8 + px(P4K) * 8, "polar"),
    "bilinear_1080p_to_4k": (P1080, P4K, px(P1080) * 8 + px(P4K) * 8, None),
    # the separable counterpart of the headline (pl_render_default_params' upscaler): two passes
    "lanczos_1080p_to_4k_dither10": (P1080, P4K, px(P1080) * 8 + px(P4K) * 8, "ortho"),
    # real video ingest: NV12 (8-bit 4:2:0, BT.709 limited) -> EWA 2x -> RGB, 10-bit dither
    "nv12_1080p_to_4k_ewa_dither10": (P1080, P4K, px(P1080) * 3 // 2 + px(P4K) * 8, "polar"),
    "nv12_1080p_to_4k_default_preset": (P1080, P4K, px(P1080) * 3 // 2 + px(P4K) * 8, "ortho"),
    # pl_render_default_params as shipped: lanczos in linear, sigmoidized light + dither
    "default_preset_1080p_to_4k": (P1080, P4K, px(P1080) * 8 + px(P4K) * 8, "ortho"),
    "hdr10_4k_tonemap": (P4K, P4K, 3 * px(P4K) * 8, "tone map"),
    "ewa_8k_to_4k_deband_tonemap": (P8K, P4K, px(P8K) * 8 + px(P4K) * 8, "polar"),
    # the metric's two halves in one frame: 1080p HDR10 -> EWA 2x upscale -> tone map -> 4K SDR
    "ewa_1080p_to_4k_hdr_tonemap": (P1080, P4K, 2 * px(P1080) * 8 + px(P4K) * 8, "polar"),
    # 24 fps -> 60 Hz through pl_queue + pl_render_image_mix (oversampling mixer): a step is one
    # vsync; 0.4 source frames per vsync are scaled into the f16 cache, 40 % of the vsyncs blend
    # two cached frames, the others show one (bytes: the output pass, 1.4 x f16 in + rgba16 out on
    # average)
    "mix_24_to_60_ewa_1080p_to_4k": (P1080, P4K, px(P4K) * 8 * 12 // 5, "frame mixing"),
}


def synthetic_frame(workload, w, h):
    import util
    if "hdr" in workload or "tonemap" in workload:
        # PQ-coded BT.2020: the chirp, scaled so that the peak is ~1000 nits (PQ 0.75)
        f = util.chirp_rgba16(w, h).astype(np.float32) * 0.75
        f[..., 3] = 65535
        return f.astype(np.uint16)
    return util.chirp_rgba16(w, h)


class Stream:
    """One independent video stream on one GPU: a pl_hip backend + a pl_renderer."""

    def __init__(self, device, workload, pool):
        self.g = pl.HipGpu(device)
        self.rr = pl.Renderer(self.g)
        self.workload = workload
        (sw, sh), (dw, dh), _, _ = WORKLOADS[workload]
        frame = synthetic_frame(workload, sw, sh)
        self.srcs = [self.g.tex_create(sw, sh, "rgba16", np.roll(frame, 7 * i, axis=1))
                     for i in range(pool)]
        self.dsts = [self.g.tex_create(dw, dh, "rgba16") for _ in range(pool)]
        self.pool = pool
        self.i = 0
        self.pass_ns = {}
        self.nv12 = None
        self.queue = None
        if workload.startswith("nv12"):
            # luma from the chirp's green channel, chroma from (b - g, r - g), 8 bit
            y = (frame[..., 1] >> 8).astype(np.uint8)
            cb = (128 + ((frame[::2, ::2, 2].astype(np.int32) - frame[::2, ::2, 1]) >> 9)).clip(16, 240)
            cr = (128 + ((frame[::2, ::2, 0].astype(np.int32) - frame[::2, ::2, 1]) >> 9)).clip(16, 240)
            uv = np.stack([cb, cr], axis=-1).astype(np.uint8)
            self.nv12 = [(self.g.tex_create(sw, sh, "r8", np.roll(y, 7 * i, axis=1)[..., None]),
                          self.g.tex_create(sw // 2, sh // 2, "rg8", np.roll(uv, 3 * i, axis=1)))
                         for i in range(pool)]

        sdr = pl.color_space("bt709", "srgb")
        hdr = pl.color_space("bt2020", "pq", max_luma=1000.0)
        bt1886 = pl.color_space("bt709", "bt1886")
        dither = capi.DitherParams(method=pl.DITHER_BLUE_NOISE, lut_size=6, transfer=0)
        ten_bit = pl.color_repr("rgb", "full", sample_depth=16, color_depth=10, bit_shift=6)
        if workload == "bilinear_1080p_to_4k":
            self.params = pl.render_params("fast")
            icsp, tcsp, trepr = sdr, sdr, None
        elif workload == "ewa_lanczos_1080p_to_4k_dither10":
            self.params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"),
                                           dither_params=dither,
                                           disable_dither_gamma_correction=True)
            icsp, tcsp, trepr = sdr, sdr, ten_bit
        elif workload == "lanczos_1080p_to_4k_dither10":
            self.params = pl.render_params("fast", upscaler=pl.filter_config("lanczos"),
                                           dither_params=dither,
                                           disable_dither_gamma_correction=True)
            icsp, tcsp, trepr = sdr, sdr, ten_bit
        elif workload == "nv12_1080p_to_4k_ewa_dither10":
            self.params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"),
                                           dither_params=dither,
                                           disable_dither_gamma_correction=True)
            icsp, tcsp, trepr = pl.color_space("bt709", "bt1886"), pl.color_space("bt709", "bt1886"), ten_bit
        elif workload == "nv12_1080p_to_4k_default_preset":
            self.params = pl.render_params("default")
            icsp, tcsp, trepr = pl.color_space("bt709", "bt1886"), pl.color_space("bt709", "bt1886"), ten_bit
        elif workload == "default_preset_1080p_to_4k":
            self.params = pl.render_params("default")
            icsp, tcsp, trepr = sdr, sdr, ten_bit
        elif workload == "hdr10_4k_tonemap":
            self.params = pl.render_params(
                "default", peak_detect_params=pl.peak_detect_params(percentile=99.995))
            icsp, tcsp, trepr = hdr, bt1886, None
        elif workload == "mix_24_to_60_ewa_1080p_to_4k":
            mixer = capi.FilterConfig()
            C.memmove(C.byref(mixer), C.byref(pl.filter_config("oversample", pl.FILTER_FRAME_MIXING)),
                      C.sizeof(mixer))
            self.params = pl.render_params("fast", upscaler=pl.filter_config("ewa_lanczos"),
                                           dither_params=dither, frame_mixer=mixer,
                                           disable_dither_gamma_correction=True)
            icsp, tcsp, trepr = bt1886, bt1886, ten_bit
            self.queue = pl.Queue(self.g)
            self.pts, self.fed = 0.0, 0
        elif workload == "ewa_1080p_to_4k_hdr_tonemap":
            self.params = pl.render_params(
                "default", upscaler=pl.filter_config("ewa_lanczos"), dither_params=dither,
                peak_detect_params=pl.peak_detect_params(percentile=99.995))
            icsp, tcsp, trepr = hdr, bt1886, ten_bit
        else:
            self.params = pl.render_params(
                "high_quality", downscaler=pl.filter_config("ewa_lanczos"),
                peak_detect_params=pl.peak_detect_params(percentile=99.995))
            icsp, tcsp, trepr = hdr, bt1886, ten_bit

        self._cb = capi.RENDER_INFO_CB(self._info)
        self.params.info_callback = C.cast(self._cb, C.c_void_p)
        self.images = [pl.frame(t, components=3, color=icsp) for t in self.srcs]
        if self.nv12:
            self.images = []
            for ty, tuv in self.nv12:
                f = capi.Frame(num_planes=2)
                f.planes[0] = capi.Plane(texture=ty.ptr, components=1)
                f.planes[1] = capi.Plane(texture=tuv.ptr, components=2)
                for c in range(4):
                    f.planes[0].component_mapping[c] = [0, -1, -1, -1][c]
                    f.planes[1].component_mapping[c] = [1, 2, -1, -1][c]
                f.repr = pl.color_repr("bt709", "limited", sample_depth=8, color_depth=8)
                f.color = icsp
                pl.lib().pl_frame_set_chroma_location(C.byref(f), 1)   # PL_CHROMA_LEFT
                self.images.append(f)
        self.targets = [pl.frame(t, color=tcsp, repr_=trepr) for t in self.dsts]

    def _info(self, _priv, info):
        d = info.contents.pass_.contents
        self.pass_ns.setdefault(d.shader.contents.description.decode(), []).append(d.last)

    def step_mix(self):
        """one vsync: keep the queue two source frames ahead, ask it for the mix, render that"""
        frame, vsync = 1.0 / 24.0, 1.0 / 60.0
        q = self.queue
        while self.fed * frame <= self.pts + 2 * frame:
            q.push(self.images[self.fed % self.pool], self.fed * frame, frame)
            self.fed += 1
        st, mix = q.update(self.pts, radius=pl.frame_mix_radius(self.params), vsync_duration=vsync)
        assert st == pl.QUEUE_OK, st
        i = self.i % self.pool
        self.i += 1
        assert pl.lib().pl_render_image_mix(self.rr.rr, C.byref(mix), C.byref(self.targets[i]),
                                            C.byref(self.params)), self.g.messages[-3:]
        self.pts += vsync
        for ident in q.unmapped:
            q.frames.pop(ident, None)
        q.unmapped.clear()

    def step(self):
        if self.queue:
            return self.step_mix()
        i = self.i % self.pool
        self.i += 1
        assert self.rr.render(self.images[i], self.targets[i], self.params), self.g.messages[-3:]

    def close(self):
        self.g.finish()
        if self.queue:
            self.queue.destroy()
        self.rr.destroy()
        for t in self.srcs + self.dsts + [t for pair in (self.nv12 or []) for t in pair]:
            t.destroy()
        self.g.close()


def cpu_baseline(workload):
    """The CPU oracle (a scalar port of the reference's algorithm) timed on the host, single
    thread, on a bounded sample of the same workload."""
    import orc
    import util
    (sw, sh), (dw, dh), _, _ = WORKLOADS[workload]
    ortho = workload.startswith("lanczos") or "default_preset" in workload
    if workload.startswith("bilinear") or workload.startswith("ewa_lanczos") or \
            workload.startswith("nv12_1080p_to_4k_ewa") or ortho:
        cw, ch = sw, sh  # one whole frame: ~10 s of scalar CPU work
        src = util.chirp_rgba16(sw, sh)
        tex = orc.tex_decode(src, "rgba16")
        t0 = time.perf_counter()
        if workload.startswith("bilinear"):
            out = orc.sample_simple(tex, orc.S_BILINEAR, cw * 2, ch * 2)
        elif ortho:
            # the two passes of the separable Lanczos scaler (the colour stages of the preset
            # are not part of this sample)
            rows, n, _, _ = orc.filter_generate_ortho(orc.lanczos())
            rows = orc.ortho_lut_rows(rows, n, False)
            img = orc.op_quant_f16(orc.sample_simple(tex, orc.S_BILINEAR, cw, ch))
            tmp = orc.op_quant_f16(orc.sample_ortho(img, rows, n, 1, cw, ch * 2))
            out = orc.sample_ortho(tmp, rows, n, 0, cw * 2, ch * 2)
            orc.dither(out, util.blue_noise(pl), 10)
        else:
            img = orc.op_quant_f16(orc.sample_simple(tex, orc.S_BILINEAR, cw, ch))
            w, r, rz = orc.filter_generate_polar(orc.ewa_lanczos())
            out = orc.sample_polar(img, w, r, rz, cw * 2, ch * 2, mask=0x7)
            orc.dither(out, util.blue_noise(pl), 10)
        orc.tex_encode(out, "rgba16")
        dt = time.perf_counter() - t0
        npx, sample = cw * 2 * ch * 2, f"1 frame {cw}x{ch}->{cw * 2}x{ch * 2}"
    else:
        # colour-mapping workloads: a 1920x1080 crop through linearize -> IPT/PQ -> tone LUT
        # -> gamut 3D-LUT -> delinearize (the per-pixel part of the pass structure)
        import colormap_ref as cr
        cw, ch = 1920, 1080
        img = (synthetic_frame(workload, cw, ch).astype(np.float32) / 65535.0)
        r = cr.resolve(cr.make_csp(pl.PRIM["bt2020"], pl.TRC["pq"], max_luma=1000.0),
                       cr.make_csp(pl.PRIM["bt709"], pl.TRC["bt1886"]))
        t0 = time.perf_counter()
        cr.apply(img, r)
        dt = time.perf_counter() - t0
        npx, sample = cw * ch, f"{cw}x{ch} crop, colour-mapping stage only"
    return {
        "value": round(npx / dt / 1e6, 4),
        "unit": "Mpixels/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample} of the same workload, oracle/pl_oracle.c (scalar C, -O2), {dt:.1f} s",
    }


def companion(device, workload, steps=80, warmup=10):
    """value / ms_per_step of another workload, timed like the main one (single stream)."""
    (sw, sh), (dw, dh), _, _ = WORKLOADS[workload]
    per_frame = (sw * sh + dw * dh) * 8
    st = Stream(device, workload, max(4, -(-800_000_000 // per_frame)))
    for _ in range(warmup):
        st.step()
    st.g.finish()
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step()
    st.g.finish()
    dt = time.perf_counter() - t0
    errors = st.rr.errors()
    st.close()
    return {"value": round(steps * dw * dh / dt / 1e6, 1), "unit": "Mpixels/s",
            "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "render_errors": errors}


def baseline_metric():
    """BASELINE.json's metric name (the driver matches on it); the workload measured is named in
    config.workload."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "Mpixels/s (and frames/s) for EWA-Lanczos 1080p->4K + HDR tonemap, 1/2/4/8 GPU"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ewa_lanczos_1080p_to_4k_dither10",
                    choices=sorted(WORKLOADS))
    ap.add_argument("--pool", type=int, default=0,
                    help="rotating source/target textures per stream (0 = enough to exceed "
                         "the 256 MiB Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-companions", action="store_true",
                    help="skip the tone-mapping workloads reported next to the default one")
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

    (sw, sh), (dw, dh), alg_bytes, dominant = WORKLOADS[args.workload]
    per_frame = (sw * sh + dw * dh) * 8
    pool = args.pool or max(4, -(-800_000_000 // per_frame))
    st = Stream(local_rank, args.workload, pool)

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

    # ---- per-pass GPU time: HIP events recorded around every launch on the stream the
    # launches go to (pl_timer), reported through pl_render_params.info_callback ------------
    roofline = None
    if rank == 0:
        st.pass_ns.clear()
        for _ in range(48):
            st.step()
        st.g.finish()
        st.step()           # drains the last timers
        st.g.finish()
        passes = {k: float(np.mean(v)) for k, v in st.pass_ns.items() if v}
        if dominant:
            name = max((k for k in passes if dominant in k), key=lambda k: passes[k],
                       default=max(passes, key=passes.get))
        else:
            name = max(passes, key=passes.get)
        kern_s = passes[name] * 1e-9
        frame_s = sum(passes.values()) * 1e-9
        achieved = alg_bytes / kern_s / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            with open(prof) as f:
                traffic = json.load(f).get(args.workload)
        symbol = ("k_pass_peak" if "peak detection" in name else
                  "k_polar_pp" if name.startswith("polar") else
                  "k_ortho_fast" if name.startswith("ortho") else
                  "k_deband" if name.startswith("deband") else
                  # (bilinear + fused epilogue -> rgba16 has its own kernel, k_pass.hip)
                  "k_bilinear_fast" if args.workload == "bilinear_1080p_to_4k" else "k_pass_generic")
        roofline = {
            "bound": "hbm",
            "kernel": f"{symbol} ({name})",   # HIP kernel (rocprofv3 name) + pass description
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "kernel_us": round(kern_s * 1e6, 2),
            "algorithmic_bytes": alg_bytes,
            "passes_us": {k: round(v / 1e3, 2) for k, v in passes.items()},
            "frame_gpu_us": round(frame_s * 1e6, 2),
            "frame_frac": round(alg_bytes / frame_s / 1e9 / HBM_PEAK_GBS, 4),
        }

    if rank == 0:
        frames = args.steps * world
        out = {
            "metric": baseline_metric(),
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
                "src": f"{sw}x{sh} " + ("nv12 (r8 + rg8 planes)" if args.workload.startswith("nv12")
                                        else "rgba16"),
                "dst": f"{dw}x{dh} rgba16",
                "pool": pool,
                "api": "pl_queue_update + pl_render_image_mix" if args.workload.startswith("mix") else "pl_render_image",
                "measured": "output Mpixels/s through pl_render_image, one independent stream per GPU",
                "render_errors": st.rr.errors(),   # pl_render_error bits: no stage may be disabled
                "parallelism": f"{world} independent stream(s), one per GPU",
            },
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:    # (rank 0 at N = 1 only)
            out["cpu_baseline"] = cpu_baseline(args.workload)

    st.close()
    if rank == 0:
        # The metric's name also carries "+ HDR tonemap": the same run reports the tone-mapping
        # workloads next to the headline (shorter timed loops; not part of `value`).
        if world == 1 and not args.no_companions and args.workload == "ewa_lanczos_1080p_to_4k_dither10":
            out["companions"] = {w: companion(local_rank, w)
                                 for w in ("ewa_1080p_to_4k_hdr_tonemap", "hdr10_4k_tonemap")}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
