"""Tile-column sharding of the hot path across the GPUs of one node (one process per GPU).

The reference partitions a frame into uniform tile columns in superblock units
(av1/common/tile_common.c:76-97); tiles are independent units of work, so rank r owns column r and
there is no data-path collective in the SAD / transform kernels.  The helpers here are backend
agnostic (nccl == RCCL on the GPU box, gloo in the CPU tests)."""
import numpy as np

from .synth import tile_column_bounds


def column_of_rank(width, world, rank, sb=64, mode="uniform"):
    """Pixel bounds [x0, x1) of rank's tile column.  mode "uniform": av1_calculate_tile_cols' uniform spacing (tile_common.c:76-97);
    "balanced": auto_tile_size_balancing (av1/encoder/encoder.c:247-275; world must be a power of two)."""
    if mode == "balanced" and world & (world - 1) == 0:
        sb_cols = (width + sb - 1) // sb
        k = world.bit_length() - 1
        size = sb_cols >> k
        inc = world - (sb_cols - (size << k))
        cols, s = [], 0
        for i in range(world):
            if s >= sb_cols:
                break
            if i == inc:
                size += 1
            if size > 0:
                cols.append((s * sb, min((s + size) * sb, width)))
            s += size
    else:
        cols = tile_column_bounds(width, world, sb)
    return cols[rank] if rank < len(cols) else (0, 0)


def shard_by_column(items, x0, x1, key="sx"):
    """Keep the work items (structured array with an x position field) whose block starts inside [x0, x1)."""
    keep = (items[key] >= x0) & (items[key] < x1)
    return items[keep], np.nonzero(keep)[0]


def reduce_scalar(dist, value, op, device):
    """MAX / SUM of a python float over ranks; identity when not distributed."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=getattr(dist.ReduceOp, op))
    return float(t.item())


def gather_results(dist, local, index, total, device):
    """All-gather per-rank result rows (uint32) into their global positions: every rank ends with the
    full result array -- the N-GPU run must be bit-identical to the 1-GPU run (the reference's
    thread-count invariance tests, test/ethread_test.cc:139-201)."""
    if dist is None:
        return local
    import torch
    world = dist.get_world_size()
    n_loc = torch.tensor([local.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n_loc)
    mx = int(max(int(s.item()) for s in sizes))
    width = int(np.prod(local.shape[1:])) if local.ndim > 1 else 1
    pad_v = torch.zeros((mx, width), dtype=torch.int64, device=device)
    pad_i = torch.full((mx,), -1, dtype=torch.int64, device=device)
    if local.shape[0]:
        pad_v[:local.shape[0]] = torch.from_numpy(local.reshape(local.shape[0], width).astype(np.int64)).to(device)
        pad_i[:local.shape[0]] = torch.from_numpy(np.asarray(index, dtype=np.int64)).to(device)
    vs = [torch.zeros_like(pad_v) for _ in range(world)]
    is_ = [torch.zeros_like(pad_i) for _ in range(world)]
    dist.all_gather(vs, pad_v)
    dist.all_gather(is_, pad_i)
    out = np.zeros((total, width), np.uint32)
    for v, i in zip(vs, is_):
        i = i.cpu().numpy()
        ok = i >= 0
        out[i[ok]] = v.cpu().numpy()[ok].astype(np.uint32)
    return out.reshape((total,) + local.shape[1:])


def exchange_strips(dist, plane2d, bounds, rank):
    """The one real exchange step of the tile-column encoder: after frame t is reconstructed, GPU r
    holds valid pixels only in its own tile column [x0_r, x1_r) of the reconstructed plane; every GPU needs the
    whole plane as the reference of frame t+1 (MV limits are frame-relative, av1/encoder/mcomp.h:216-247).
    Realised as one broadcast per strip (north_star: "RCCL broadcast of reconstructed reference planes"); on the
    fully connected xGMI mesh each goes directly to the 7 peers.  plane2d: torch tensor [rows, stride] on this
    rank's device (nccl) or CPU (gloo); bounds: [(x0, x1)] in elements of that tensor, one per rank.  Use a uint8
    (byte) view for 10/12-bit planes: RCCL has no 16-bit integer type."""
    if dist is None:
        return plane2d
    for r, (x0, x1) in enumerate(bounds):
        if x1 <= x0:
            continue
        buf = plane2d[:, x0:x1].contiguous()
        if buf.is_cuda and dist.get_backend() == "gloo":  # single-GPU dry run only: gloo moves host memory
            host = buf.cpu()
            dist.broadcast(host, src=r)
            buf = host.to(buf.device)
        else:
            dist.broadcast(buf, src=r)
        if r != rank:
            plane2d[:, x0:x1] = buf
    return plane2d
