"""N > 1 path on CPU: 2 gloo ranks shard a frame by tile column exactly as bench.py does on N GPUs,
each computes its column (with the oracle standing in for the device), the results are gathered, and
the outcome must be bit-identical to the single-rank run (the reference's N-worker invariance,
test/ethread_test.cc:139-201).  Also checks the tile-column rule itself (tile_common.c:76-97)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_tile_column_rule(hip):
    tcb = hip.synth.tile_column_bounds
    # 4K, 4 columns -> 15 SB = 960 px each; 8 columns -> 7 x 512 + 256
    assert tcb(3840, 4) == [(0, 960), (960, 1920), (1920, 2880), (2880, 3840)]
    c8 = tcb(3840, 8)
    assert len(c8) == 8 and [b - a for a, b in c8] == [512] * 7 + [256]
    # 1080p: 30 SB columns; 8 -> size 4: 7 full + one of 2 SBs
    assert [b - a for a, b in tcb(1920, 8)] == [256] * 7 + [128]
    for w, n in [(1920, 1), (1920, 2), (640, 3), (3840, 5)]:
        cols = tcb(w, n)
        assert cols[0][0] == 0 and cols[-1][1] == w and all(a[1] == b[0] for a, b in zip(cols, cols[1:]))


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch.distributed as dist
        import aom_av1_psy_amd as pkg
        import pyoracle as orc
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        W, H, border = 640, 360, 64
        src, ref = pkg.synth.lcg_frame(W, H, 0), pkg.synth.lcg_frame(W, H, 1)
        sb, rb = orc.extend_plane(src, border), orc.extend_plane(ref, border)
        cands, groups = pkg.synth.mode_a_worklist(W, H, 16, seed=9, search=48)
        x0, x1 = pkg.partition.column_of_rank(W, world, rank)
        mine, idx = pkg.partition.shard_by_column(groups, x0, x1)
        local = orc.sad_x4d_batch(sb, rb, border, 16, 16, mine)
        full = pkg.partition.gather_results(dist, local, idx, len(groups), "cpu")
        t = pkg.partition.reduce_scalar(dist, float(rank + 1), "MAX", "cpu")
        n = pkg.partition.reduce_scalar(dist, float(len(mine)), "SUM", "cpu")
        want = orc.sad_x4d_batch(sb, rb, border, 16, 16, groups)
        # the exchange step: every rank owns only its tile-column strip of the "reconstructed" plane
        import torch
        stride = rb.shape[1]
        mine_only = torch.zeros(rb.shape, dtype=torch.uint8)
        bounds = []
        for r in range(world):
            a, b = pkg.partition.column_of_rank(W, world, r)
            # strips in bordered coordinates; the outer strips carry the frame border with them
            bounds.append((0 if r == 0 else a + border, stride if r == world - 1 else b + border))
        a, b = bounds[rank]
        mine_only[:, a:b] = torch.from_numpy(rb[:, a:b].copy())
        pkg.partition.exchange_strips(dist, mine_only, bounds, rank)
        ok_x = bool(np.array_equal(mine_only.numpy(), rb))
        q.put((rank, bool(np.array_equal(full, want)) and ok_x, t, n, len(groups)))
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e), 0, 0, 0))


def test_two_rank_tile_column_sharding_is_bit_identical():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    ps = [ctxm.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in ps]
    for p in ps:
        p.join(60)
    for rank, ok, t, n, total in res:
        assert ok is True, (rank, ok)
        assert t == 2.0 and int(n) == total  # max over ranks; shards cover every block exactly once
