"""Process-group set-up and the barrier / max-over-ranks timing around a workload's steps (one process per GPU; backend nccl = RCCL, gloo only for dry runs)."""
import json
import os
import sys
import time

import numpy as np

from . import common
from .common import HBM_PEAK_GBS, ROOT, kernel_avg_ms, ramp


def dist_setup(n_gpus, backend):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    forced = world <= 1 and os.environ.get("AOMHIP_BENCH_FORCE_DIST") == "1"
    if world <= 1 and not forced:
        return None, 0, 1
    import torch
    import torch.distributed as dist
    common.BACKEND = backend
    if forced:  # tools/gpu_dist_dryrun.sh: the whole N > 1 code path (process group, RCCL communicator, exchange, reductions) with ONE rank
        torch.cuda.set_device(0)
        dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500), rank=0, world_size=1,
                                **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
        return dist, 0, 1
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1)
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:   # gloo: CPU tensors only (the launcher dry run of tests/test_bench_launcher_gloo.py runs where there is no GPU)
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend)
    return dist, rank, world


def _red_device():
    return "cuda" if common.BACKEND == "nccl" else "cpu"


def barrier(dist, dev):
    if dist is not None:
        import torch
        if common.BACKEND == "nccl":
            dist.barrier(device_ids=[dev])
            torch.cuda.synchronize()
        else:
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()


def time_steps(wl, ctx, dist, dev, steps, warmup):
    ramp(ctx, wl.step)
    for _ in range(warmup):
        wl.step()
    ctx.sync()
    barrier(dist, dev)
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    ev_ms = ctx.timer_end()  # HIP events on the launch stream, syncs
    ctx.sync()
    barrier(dist, dev)
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([wall], dtype=torch.float64, device=_red_device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, ev_ms
