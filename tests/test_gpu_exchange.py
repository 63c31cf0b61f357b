"""aomhip_allgather_recon / aomhip_comm_* (csrc/exchange.hip): the RCCL exchange of the reconstruction between tile-column
ranks.  With one GPU: the communicator of one rank, the whole-plane call (nothing to move, borders extended) and the
loop-back transport test (pack kernel -> ncclSend / ncclRecv to self -> unpack kernel).  With two or more GPUs: one process per
GPU, every rank starts with only its own column valid and must end with the bytes of the 1-GPU plane, and the search on the
exchanged reference must equal the 1-GPU search (the reference's thread-count invariance, test/ethread_test.cc:139-201)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import aom_av1_psy_amd as pkg
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as orc
    ctx = pkg.capi.Context(0)
    comm = ctx.comm_init(pkg.capi.comm_unique_id(), 0, 1)
    yield pkg, ctx, comm, orc
    ctx.comm_destroy(comm)
    ctx.close()


@pytest.mark.parametrize("bd", [8, 10])
def test_single_rank_exchange_only_extends_the_borders(env, bd):
    pkg, ctx, comm, orc = env
    rng = np.random.default_rng(3)
    W, H, B = 200, 72, 32
    px = rng.integers(0, 1 << bd, (H, W)).astype(np.uint8 if bd == 8 else np.uint16)
    p = ctx.planes_alloc(W, H, B, bd, 2)
    ctx.planes_upload(p, 1, px)
    b, _ = pkg.capi.tile_column_bounds(W, 1)
    ctx.allgather_recon(comm, p, 1, b, -1)
    ctx.sync()
    got = ctx.planes_download(p, 1)
    want = orc.extend_plane(px, B, p.stride)
    assert np.array_equal(got[:, :W + 2 * B], want[:, :W + 2 * B])
    ctx.planes_free(p)


@pytest.mark.parametrize("bd,x0,x1,dst", [(8, 0, 64, 100), (8, 3, 70, 129), (10, 64, 128, 0), (10, 5, 6, 198), (8, 0, 0, 10), (10, 1, 200, 0)])
def test_loopback_moves_a_strip_through_rccl(env, bd, x0, x1, dst):
    pkg, ctx, comm, orc = env
    rng = np.random.default_rng(x0 + x1)
    W, H, B = 200, 40, 16
    if dst + (x1 - x0) > W:
        dst = W - (x1 - x0)
    px = rng.integers(0, 1 << bd, (H, W)).astype(np.uint8 if bd == 8 else np.uint16)
    p = ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(p, 0, px)
    pkg.capi.check(pkg.capi.lib.aomhip_exchange_loopback(ctx.h, comm, p, 0, x0, x1, dst), "loopback")
    ctx.sync()
    want = px.copy()
    want[:, dst:dst + (x1 - x0)] = px[:, x0:x1]
    got = ctx.planes_download(p, 0)
    assert np.array_equal(got[:, :W + 2 * B], orc.extend_plane(want, B, p.stride)[:, :W + 2 * B])
    ctx.planes_free(p)


def test_exchange_rejects_bad_arguments(env):
    pkg, ctx, comm, orc = env
    lib = pkg.capi.lib
    p = ctx.planes_alloc(64, 16, 16, 8, 1)
    b, _ = pkg.capi.tile_column_bounds(64, 1)
    assert lib.aomhip_allgather_recon(ctx.h, comm, p, 1, b.ctypes.data, -1) == pkg.capi.ERR_INVALID      # frame outside the ring
    assert lib.aomhip_allgather_recon(ctx.h, None, p, 0, b.ctypes.data, -1) == pkg.capi.ERR_INVALID
    assert lib.aomhip_exchange_loopback(ctx.h, comm, p, 0, 0, 65, 0) == pkg.capi.ERR_INVALID               # beyond the plane
    ctx.planes_free(p)


def _n_gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("halo", [-1, 40])
def test_ranks_end_with_the_single_gpu_plane_and_search_result(halo, tmp_path):
    n = _n_gpus()
    if n < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device); covered on one GPU by the loop-back test and on the "
                    "CPU by tests/test_exchange_plan.py")
    world = min(n, 4)
    idfile = tmp_path / "rccl_id.bin"
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "workers", "exchange_worker.py"), str(idfile), str(halo)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d:\n%s" % (r, o)
        assert "EXCHANGE-OK" in o
