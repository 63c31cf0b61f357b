"""One rank of tests/test_gpu_exchange.py's multi-GPU case (started as a child process: one process per GPU).
argv: <file the ranks pass the RCCL id through> <halo>.  No torch.distributed: the id travels through a file, as a C host
would pass it through its own launcher."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    idfile, halo = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import aom_av1_psy_amd as pkg
    import pyoracle as orc
    capi = pkg.capi
    ctx = capi.Context(int(os.environ["LOCAL_RANK"]))
    if rank == 0:
        uid = capi.comm_unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid.tobytes())
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no RCCL id" % rank)
            time.sleep(0.05)
        uid = np.fromfile(idfile, np.uint8)
    comm = ctx.comm_init(uid, rank, world)

    W, H, B, BS, bd = 64 * 3 * world - 24, 96, 96, 16, 10
    src_px, ref_px = pkg.synth.shifted_smooth_pair(W, H, 0, bd, shift=(3, -2), frac8=(0, 0))
    bounds, cols = capi.tile_column_bounds(W, world)
    x0, x1 = bounds[rank]
    src = ctx.planes_alloc(W, H, B, bd, 1)
    ref = ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(src, 0, src_px)
    mine = np.full_like(ref_px, 0x3FF)          # only this rank's column is valid before the exchange
    mine[:, x0:x1] = ref_px[:, x0:x1]
    ctx.planes_upload(ref, 0, mine)
    ctx.allgather_recon(comm, ref, 0, bounds, halo)
    ctx.sync()
    got = ctx.planes_download(ref, 0)
    lo, hi = (0, W) if halo < 0 else (max(0, x0 - halo), min(W, x1 + halo))
    want = orc.extend_plane(ref_px, B, ref.stride)
    assert np.array_equal(got[B:B + H, B + lo:B + hi], want[B:B + H, B + lo:B + hi]), "rank %d: exchanged pixels differ" % rank
    if halo < 0:
        assert np.array_equal(got[:, :W + 2 * B], want[:, :W + 2 * B]), "rank %d: borders differ" % rank

    # the search over this rank's blocks on the exchanged reference == the 1-GPU search (oracle on the full plane)
    reach = 31 if halo >= 0 else 1023   # step_param 6 -> first step 16: every candidate within 31 (+ halo 40 covers it)
    xs, ys = np.meshgrid(np.arange(x0, x1 - BS + 1, BS), np.arange(0, H - BS + 1, BS))
    b = np.zeros(xs.size, capi.search_block_dtype)
    b["bx"], b["by"] = xs.ravel(), ys.ravel()
    ext = B - 8
    b["col_min"] = np.maximum(-(b["bx"] + ext), -reach); b["col_max"] = np.minimum(W - b["bx"] - BS + ext, reach)
    b["row_min"] = np.maximum(-(b["by"] + ext), -reach); b["row_max"] = np.minimum(H - b["by"] - BS + ext, reach)
    d_b = ctx.to_device(b)
    d_mv, d_cost = ctx.malloc(b.size * 4), ctx.malloc(b.size * 4)
    ctx.fullpel_diamond_batch(src, ref, 0, BS, BS, 0, 6, capi.MV_COST_L1_HDRES, d_b, b.size, d_mv, d_cost)
    mv = ctx.from_device(d_mv, (b.size, 2), np.int16)
    sb_, rb_ = orc.extend_plane(src_px, B, src.stride), orc.extend_plane(ref_px, B, ref.stride)
    wmv, _ = orc.fullpel_diamond_batch(sb_, rb_, B, BS, BS, b, 0, 6, 3, bd, threads=2)
    assert np.array_equal(mv, wmv), "rank %d: search on the exchanged reference differs from the 1-GPU search" % rank
    ctx.comm_destroy(comm)
    ctx.close()
    print("EXCHANGE-OK rank %d columns %s" % (rank, bounds.tolist()))


if __name__ == "__main__":
    main()
