"""N > 1 path on CPU (gloo, world_size 2): the only data that ever crosses ranks is the 816-word
peak-detection buffer when ranks render tiles / frames of one scene: SUM on the
counters, sums and histogram, MAX on frame_max_pq. The reduced buffer must equal the buffer a
single rank measures over the whole frame."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path[:0] = [HERE, os.path.dirname(HERE)]
    import torch
    import torch.distributed as dist
    import orc
    from libplacebo_amd.dist import allreduce_peak_buffer

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(42)
    frame = rng.random((64, 96, 4)).astype(np.float32)      # whole frame, same on all ranks
    frame[..., :3] *= 0.75
    rows = 64 // world
    tile = frame[rank * rows:(rank + 1) * rows]             # this rank's rows (multiple of 16)
    luma = (0.2627, 0.6780, 0.0593)
    part = orc.detect_peak(tile, 12, 0.0, 49.26, luma, black_cutoff=1.0, use_hist=True)
    buf = torch.from_numpy(part.astype(np.int64))
    allreduce_peak_buffer(buf, dist)
    full = orc.detect_peak(frame, 12, 0.0, 49.26, luma, black_cutoff=1.0, use_hist=True)
    # slices are assigned by workgroup index within a launch, so compare slice-summed totals
    got = buf.numpy()
    def totals(b):
        return (b[0:12].sum(), b[12:24].sum(), b[24:36].sum(), b[36:48].max(),
                b[48:].reshape(12, 64).sum(axis=0))
    a, b = totals(got), totals(full.astype(np.int64))
    ok = all(np.array_equal(x, y) for x, y in zip(a, b))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, ok))


def test_peak_buffer_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_bench_rank_env_contract():
    """bench.py reads RANK / LOCAL_RANK / WORLD_SIZE and refuses a mismatching --gpus."""
    src = open(os.path.join(os.path.dirname(HERE), "bench.py")).read()
    for key in ('"RANK"', '"WORLD_SIZE"', '"LOCAL_RANK"', "MASTER_ADDR", '"nccl"',
                "dist.barrier", "ReduceOp.MAX"):
        assert key in src, key


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` started without torchrun becomes its own launcher (one process
    per rank, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them); the
    selftest mode joins a gloo group on the CPU, so the rendezvous itself is exercised here."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout      # rank 0 alone owns stdout
    out = json.loads(line[0])
    assert out["world"] == 2 and out["sum"] == 3.0 and out["local_rank"] == 0
