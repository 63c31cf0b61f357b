"""BASELINE.json configs[3]: the tile-column search pipeline with its per-frame RCCL exchange, the default (NSTEP + 8-tap) search, the mesh
search, the first pass; the N > 1 launcher dry run."""
import json
import os
import sys
import time

import numpy as np

from . import common
from .common import HBM_PEAK_GBS, ROOT, kernel_avg_ms, ramp
from .dist import _red_device, barrier, time_steps


def search_bound():
    """What bounds the search kernels.  Round 2 read the PMC figures (L1 busy 95 %, 0.81 line accesses per CU per clock) as an L1 line-rate
    bound; round 3 tested that directly -- a reference layout with 4-8 x fewer lines per candidate left the diamond kernel's time unchanged
    (profiles/r03_search.md, section 4) -- so the bound is the latency of a search's ~20 dependent rounds, and the L1 counters measure requests
    waiting for data."""
    p = os.path.join(ROOT, "profiles", "r02_search_l1_bound.json")
    try:
        d = json.load(open(p))
        return {"fullpel_diamond_kernel": {"bound": "latency of the L1 -> L2 round trip of a step's loads: ~30 dependent steps per block, 7 blocks per SIMD in flight; NOT the "
                                                    "L1 line rate and NOT VALU issue (4-8x fewer line look-ups per candidate, or 20 % fewer vector instructions: same time)",
                                           "l1_accesses_per_cu_cycle_pmc": d["fullpel_diamond"]["l1_accesses_per_cu_cycle"]},
                "subpel_bilinear_kernel": {"bound": "VALU issue (1 wave-instruction per SIMD per 2 clocks), reference footprint in LDS",
                                           "frac": d["subpel_bilinear_lds_footprint"]["valu_issue_frac"],
                                           "issue_wait_frac": d["subpel_bilinear_lds_footprint"]["SQ_WAIT_INST_ANY_over_WAVE_CYCLES"]},
                "full_pixel_search_kernel_NSTEP": {"bound": "latency, as the diamond kernel (5 blocks per SIMD at 95 VGPRs)",
                                                   "l1_accesses_per_cu_cycle_pmc": d.get("full_pixel_search_nstep", {}).get("l1_accesses_per_cu_cycle")},
                "source": "profiles/r03_search.md, profiles/r02_search_bound.md, profiles/r02_search_l1_bound.json (rocprofv3 --pmc)"}
    except Exception:
        return None


class SearchPipeline:
    """BASELINE.json configs[3]: full-pel diamond search (DIAMOND, step_param 4, MV_COST_L1_HDRES) + bilinear sub-pel
    tree (1/2, 1/4, 1/8) for every 16x16 block of 3840x2160 10-bit frame pairs, tile columns across the GPUs (STRONG
    scaling: the frame is fixed, every rank searches the blocks of its own column).  With N > 1 every step first runs the
    per-frame exchange of the real encoder, aomhip_allgather_recon (csrc/exchange.hip: pack -> one group of RCCL
    sends / receives -> unpack -> borders, on the context's stream like the kernels behind it): each rank contributes
    its column of the reference ("the reconstruction of frame t") and receives what its search can touch --
    exchange="halo": own column +- (search reach 127 + 1 + AOM_INTERP_EXTEND 4), "allgather": the whole plane."""

    W, H, BD, BORDER, BS = 3840, 2160, 10, 160, 16
    HALO = 127 + 1 + 4  # DIAMOND step_param 4: steps 64 + 32 + ... + 1 = 127; sub-pel moves < 1 more; AOM_INTERP_EXTEND

    def __init__(self, pkg, ctx, dist, rank, world, frames=4, exchange="halo"):
        self.pkg, self.ctx, self.dist, self.rank, self.world, self.F = pkg, ctx, dist, rank, world, frames
        capi, synth = pkg.capi, pkg.synth
        W, H, bd, border = self.W, self.H, self.BD, self.BORDER
        self.src = ctx.planes_alloc(W, H, border, bd, frames)
        self.ref = ctx.planes_alloc(W, H, border, bd, frames)
        self.bounds, self.n_cols = (capi.tile_column_bounds_balanced if common.TILE_COLUMNS == "balanced" and world & (world - 1) == 0 else capi.tile_column_bounds)(W, world)  # idle ranks (fewer columns than ranks): (0, 0)
        x0, x1 = (int(v) for v in self.bounds[rank])
        self.halo = -1 if exchange == "allgather" else self.HALO
        self.comm = None
        if dist is not None:
            import torch
            uid = torch.zeros(128, dtype=torch.uint8, device=_red_device())
            if rank == 0:
                uid = torch.from_numpy(capi.comm_unique_id()).to(uid.device)
            dist.broadcast(uid, src=0)
            self.comm = ctx.comm_init(uid.cpu().numpy(), rank, world)
        for f in range(frames):
            s_, r_ = synth.shifted_smooth_pair(W, H, f, bd, shift=(3 + f % 3, -2 + f % 2), frac8=(f % 8, (3 * f) % 8))
            ctx.planes_upload(self.src, f, s_)
            if dist is not None:  # a rank owns only its column of the reconstruction: the rest arrives through the exchange
                m = np.zeros_like(r_)
                m[:, x0:x1] = r_[:, x0:x1]
                r_ = m
            ctx.planes_upload(self.ref, f, r_)
        xs, ys = np.meshgrid(np.arange(x0, x1 - self.BS + 1, self.BS), np.arange(0, H - self.BS + 1, self.BS))
        n = xs.size
        b = np.zeros(n, capi.search_block_dtype)
        b["bx"], b["by"] = xs.ravel(), ys.ravel()
        ext = border - 8
        b["col_min"] = np.maximum(-(b["bx"] + ext), -1023); b["col_max"] = np.minimum(W - b["bx"] - self.BS + ext, 1023)
        b["row_min"] = np.maximum(-(b["by"] + ext), -1023); b["row_max"] = np.minimum(H - b["by"] - self.BS + ext, 1023)
        self.n = n
        self.h_blocks = b
        self.d_blocks = ctx.to_device(b) if n else None
        self.d_sub = ctx.malloc(max(16, n * 20))
        self.d_mv, self.d_cost = ctx.malloc(max(16, n * 4)), ctx.malloc(max(16, n * 4))
        self.d_smv, self.d_err, self.d_dist, self.d_sse = (ctx.malloc(max(16, n * 4)) for _ in range(4))
        self.frame = 0
        if self.comm is not None:  # make every slot's reference valid before anything reads it
            for f in range(frames):
                self.exchange(f)
            ctx.sync()

    def exchange(self, f, halo=None):
        self.ctx.allgather_recon(self.comm, self.ref, f, self.bounds, self.halo if halo is None else halo)

    def exchange_ms(self, halo, reps=10):
        """the exchange alone (HIP events on the context's stream, max over ranks is taken by the caller)."""
        k = [0]
        def once():
            self.exchange(k[0] % self.F, halo); k[0] += 1
        return kernel_avg_ms(self.ctx, once, reps)

    def free(self):
        c = self.ctx
        c.planes_free(self.src)
        c.planes_free(self.ref)
        if self.comm is not None:
            c.comm_destroy(self.comm)
        for d in (self.d_blocks, self.d_sub, self.d_mv, self.d_cost, self.d_smv, self.d_err, self.d_dist, self.d_sse):
            if d:
                c.free(d)

    def step(self):
        """one frame pair: [exchange] -> full-pel -> sub-pel (sub-pel start MVs are built on the host from the
        full-pel result of the PREVIOUS visit of this ring slot; the kernels' work is what is timed)."""
        f = self.frame % self.F
        self.frame += 1
        if self.comm is not None:
            self.exchange(f)  # same stream as the searches behind it: ordered without a host synchronisation
        if not self.n:
            return
        c, capi = self.ctx, self.pkg.capi
        c.fullpel_diamond_batch(self.src, self.ref, f, self.BS, self.BS, 0, 4, capi.MV_COST_L1_HDRES, self.d_blocks, self.n,
                                self.d_mv, self.d_cost)
        c.subpel_bilinear_batch(self.src, self.ref, f, self.BS, self.BS, capi.MV_COST_L1_HDRES, 2, 1, 0, self.d_sub_blocks(f),
                                self.n, self.d_smv, self.d_err, self.d_dist, self.d_sse)

    def d_sub_blocks(self, f):
        if not hasattr(self, "_sub"):
            self._sub = {}
        if f not in self._sub:  # built once per ring slot from a (synchronous) full-pel pass
            c, capi = self.ctx, self.pkg.capi
            c.fullpel_diamond_batch(self.src, self.ref, f, self.BS, self.BS, 0, 4, capi.MV_COST_L1_HDRES, self.d_blocks,
                                    self.n, self.d_mv, self.d_cost)
            mv = c.from_device(self.d_mv, (self.n, 2), np.int16)
            sp = self.h_blocks.copy()
            sp["start_row"], sp["start_col"] = mv[:, 0] * 8, mv[:, 1] * 8
            for k in ("row_min", "row_max", "col_min", "col_max"):
                sp[k] = np.clip(self.h_blocks[k].astype(np.int32) * 8, -16383, 16383)
            self._sub[f] = (c.to_device(sp), mv)
        return self._sub[f][0]

    def check(self, orc):
        """slot 0 against the oracle on a sample of blocks (not timed)."""
        if not self.n:
            return True
        self.d_sub_blocks(0)
        mv = self._sub[0][1]
        s_, r_ = self.pkg.synth.shifted_smooth_pair(self.W, self.H, 0, self.BD, shift=(3, -2), frac8=(0, 0))
        sb = orc.extend_plane(s_, self.BORDER, self.src.stride); rb = orc.extend_plane(r_, self.BORDER, self.ref.stride)
        idx = np.arange(0, self.n, max(1, self.n // 500))
        wmv, _ = orc.fullpel_diamond_batch(sb, rb, self.BORDER, self.BS, self.BS, self.h_blocks[idx], 0, 4, 3, self.BD, threads=8)
        return bool(np.array_equal(mv[idx], wmv))


def run_search_default(pkg, ctx, orc, steps, warmup):
    """Informational: libaom's DEFAULT search flavour on the same 4K 10-bit pair -- av1_full_pixel_search with NSTEP
    (general kernel: cost list, second-best MV) and av1_find_best_sub_pixel_tree with the 8-tap up-sampled error."""
    capi = pkg.capi
    wl = SearchPipeline(pkg, ctx, None, 0, 1)
    n = wl.n
    d_cl, d_sec = ctx.malloc(n * 20), ctx.malloc(n * 4)
    q = capi.SearchParams.make("NSTEP", 3, capi.MV_COST_L1_HDRES)
    sp = capi.SubpelParams(2, capi.MV_COST_NONE, 0, 2, 1, 0, 3)      # tree, USE_8_TAPS, no MV cost (as tf_motion_search)
    out = {}
    full = lambda f: ctx.full_pixel_search_batch(wl.src, wl.ref, f, 16, 16, q, wl.d_blocks, n, wl.d_mv, wl.d_cost, d_cl, d_sec)
    sub = lambda f: ctx.subpel_tree_batch(wl.src, wl.ref, f, 16, 16, sp, wl.d_sub_blocks(f), n, wl.d_smv, wl.d_err, wl.d_dist, wl.d_sse)
    for f in range(wl.F):
        wl.d_sub_blocks(f)
    for name, fn in (("full_pixel_search_NSTEP", full), ("subpel_tree_8tap", sub)):
        k = [0]
        def once():
            fn(k[0] % wl.F); k[0] += 1
        out[name + "_ms_per_frame"] = kernel_avg_ms(ctx, once, max(steps, 8))
    ok = None
    if orc is not None:
        full(0)
        mv = ctx.from_device(wl.d_mv, (n, 2), np.int16)
        s_, r_ = pkg.synth.shifted_smooth_pair(wl.W, wl.H, 0, wl.BD, shift=(3, -2), frac8=(0, 0))
        sb = orc.extend_plane(s_, wl.BORDER, wl.src.stride); rb = orc.extend_plane(r_, wl.BORDER, wl.ref.stride)
        idx = np.arange(0, n, max(1, n // 300))
        wmv = orc.full_pixel_search_batch(sb, rb, wl.BORDER, 16, 16, wl.h_blocks[idx], orc.search_params("NSTEP", 3, 3), bd=wl.BD, threads=8)[0]
        ok = bool(np.array_equal(mv[idx], wmv))
    tot = out["full_pixel_search_NSTEP_ms_per_frame"] + out["subpel_tree_8tap_ms_per_frame"]
    ctx.free(d_cl); ctx.free(d_sec)
    wl.free()
    out.update({"workload": "default_search_NSTEP+8tap_tree_4k_10bit", "value": n / (tot * 1e-3), "unit": "blocks/s", "blocks_per_frame": n,
                "parity_sample_slot0": ok, "config": {"frame": "3840x2160 10-bit", "block": "16x16", "full_pel": "av1_full_pixel_search, NSTEP, "
                "step_param 3, MV_COST_L1_HDRES, cost list + second-best MV", "sub_pel": "av1_find_best_sub_pixel_tree, USE_8_TAPS, 1/8 pel, iters 2"}})
    return out


def exchange_bytes_plan(pkg, width, height, elem_bytes, world, bounds, halo):
    """What aomhip_allgather_recon moves per frame, from the plan alone (aomhip_recon_exchange_plan, host only): per rank the bytes it sends
    and receives (pixel columns x visible rows x element size), for the halo exchange and for the whole-column all-gather.  The driver's
    SCALE record can be checked against these: received bytes / exchange time = the per-rank xGMI rate."""
    out = {"halo": {"send": [], "recv": []}, "allgather": {"send": [], "recv": []}}
    for mode, h in (("halo", halo), ("allgather", -1)):
        for r in range(world):
            send, recv = pkg.capi.recon_exchange_plan(world, r, bounds, width, h)
            out[mode]["send"].append(int(sum(int(b - a) for a, b in send)) * height * elem_bytes)
            out[mode]["recv"].append(int(sum(int(b - a) for a, b in recv)) * height * elem_bytes)
    return out


def run_launcher_dry_run(args, dist, rank, world):
    """--workload launcher_dry_run: everything bench.py does AROUND a measurement at N > 1 -- fresh child processes, the process group, the tile-column
    partition, the exchange plan, the reductions, the supervising parent, ONE JSON line from rank 0 -- with no device call, no oracle and nothing
    measured (value 0).  tests/test_bench_launcher_gloo.py runs it at 4 and 8 gloo ranks on the CPU and injects the two failures the real run
    must survive with a non-zero exit: a rank that dies (AOMHIP_BENCH_FAIL_RANK) and a communicator that holds fewer ranks than the job
    (AOMHIP_BENCH_FAKE_COMM_RANKS: stands for aomhip_comm_info's answer)."""
    import aom_av1_psy_amd as pkg
    W, H = SearchPipeline.W, SearchPipeline.H
    bounds, _ = (pkg.capi.tile_column_bounds_balanced if common.TILE_COLUMNS == "balanced" and world & (world - 1) == 0 else pkg.capi.tile_column_bounds)(W, world)
    x0, x1 = (int(v) for v in bounds[rank])
    blocks = ((x1 - x0) // 16) * (H // 16)
    if os.environ.get("AOMHIP_BENCH_FAIL_RANK") == str(rank):
        print("bench.py: rank %d fails on purpose (AOMHIP_BENCH_FAIL_RANK)" % rank, file=sys.stderr)
        os._exit(3)   # (the others are on their way into the barrier below: only the supervising parent can end them)
    barrier(dist, 0)
    n_comm = int(os.environ.get("AOMHIP_BENCH_FAKE_COMM_RANKS", world))
    assert n_comm == world, "RCCL communicator holds %s ranks, the job has %d" % (n_comm, world)
    red = lambda v, op: pkg.partition.reduce_scalar(dist, float(v), op, _red_device())
    total = int(red(blocks, "SUM"))
    t_max = red(1.0 + rank, "MAX")
    barrier(dist, 0)
    if rank == 0:
        widths = [int(b - a) for a, b in bounds]
        print(json.dumps({"metric": "launcher dry run (nothing measured)", "value": 0.0, "unit": "none", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 0.0, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "none",
                          "config": {"workload": "launcher_dry_run", "dist_backend": args.dist_backend},
                          "strong_scaling_search": {"blocks_per_step": total, "max_over_ranks_check": t_max, "rccl_ranks_in_communicator": n_comm,
                                                    "tile_columns_px": widths,
                                                    "exchange": {"halo_px": SearchPipeline.HALO,
                                                                 "expected_bytes_per_rank_per_frame": exchange_bytes_plan(pkg, W, H, 2, world, bounds, SearchPipeline.HALO)}}}),
              flush=True)
    dist.destroy_process_group()


def run_search(pkg, ctx, dist, dev, rank, world, orc, steps, warmup, exchange="halo"):
    wl = SearchPipeline(pkg, ctx, dist, rank, world, exchange=exchange)
    ok = wl.check(orc) if orc is not None else None   # N > 1: on the EXCHANGED reference against the oracle's whole-frame search
    for f in range(wl.F):
        if wl.n:
            wl.d_sub_blocks(f)
    wall, ev_ms = time_steps(wl, ctx, dist, dev, steps, warmup)
    total, extra = wl.n, {}
    if dist is not None:
        import torch
        red = lambda v, op: pkg.partition.reduce_scalar(dist, float(v), op, _red_device())
        total = int(red(total, "SUM"))
        ok = bool(red(1.0 if ok in (True, None) else 0.0, "MIN")) if orc is not None else None
        n_comm = ctx.comm_info(wl.comm)[1] if wl.comm else None
        assert n_comm == world, "RCCL communicator holds %s ranks, the job has %d" % (n_comm, world)  # every rank really joined
        barrier(dist, dev)
        ex_halo = red(wl.exchange_ms(wl.HALO), "MAX")
        barrier(dist, dev)
        ex_all = red(wl.exchange_ms(-1), "MAX")
        es = 2 * wl.H
        widths = [int(b - a) for a, b in wl.bounds]
        recv_all = max((wl.W - w) * es for w in widths if w) if any(widths) else 0
        recv_halo = max(min(2 * wl.HALO, wl.W - w) * es for w in widths if w) if any(widths) else 0
        # xGMI is point to point: a rank's received bytes arrive over (world - 1) links at once in the all-gather, over <= 2 in halo mode
        extra = {"rccl_ranks_in_communicator": n_comm,
                 "exchange": {"mode": exchange, "halo_px": wl.HALO, "halo_ms_per_frame": ex_halo, "allgather_ms_per_frame": ex_all,
                              "allgather_bytes_received_max_rank": recv_all, "halo_bytes_received_max_rank": recv_halo,
                              "expected_bytes_per_rank_per_frame": exchange_bytes_plan(pkg, wl.W, wl.H, 2, world, wl.bounds, wl.HALO),
                              "allgather_GBs_per_rank": recv_all / (ex_all * 1e-3) / 1e9 if ex_all > 0 else None,
                              "allgather_GBs_per_link": recv_all / (ex_all * 1e-3) / 1e9 / max(world - 1, 1) if ex_all > 0 else None,
                              "halo_GBs_per_link": recv_halo / (ex_halo * 1e-3) / 1e9 / max(min(2, world - 1), 1) if ex_halo > 0 else None,
                              "transport": "aomhip_allgather_recon: pack kernels -> one ncclGroup of per-peer ncclSend / ncclRecv (uint8) -> "
                                           "unpack kernels -> border extension, all on the context's stream"},
                 "tile_columns_px": widths, "blocks_max_rank_over_mean": max(widths) / (sum(widths) / world) if sum(widths) else None,
                 # the same frame under the other rule (the widest column is what the slowest rank searches)
                 "tile_columns_px_uniform": [int(b_ - a_) for a_, b_ in pkg.capi.tile_column_bounds(wl.W, world)[0]],
                 "tile_columns_px_balanced": ([int(b_ - a_) for a_, b_ in pkg.capi.tile_column_bounds_balanced(wl.W, world)[0]]
                                              if world & (world - 1) == 0 else None)}
    return dict({"workload": "fullpel_diamond+subpel_bilinear_4k_10bit", "value": total * steps / wall, "unit": "blocks/s",
                 "frames_per_s": steps / wall, "ms_per_step": wall / steps * 1e3, "blocks_per_step": total,
                 "parity_sample_slot0": ok, "bound": search_bound(),
                 "config": {"frame": "3840x2160 10-bit", "block": "16x16", "search": "DIAMOND step_param 4, MV_COST_L1_HDRES; "
                            "sub-pel tree pruned_more, bilinear, 1/8 pel, iters 2",
                            "partition": ("balanced tile columns (auto_tile_size_balancing, encoder.c:247-275)" if common.TILE_COLUMNS == "balanced" else
                                          "uniform tile columns (tile_common.c:76-110)") + ", one per GPU",
                            "exchange": ("per frame, aomhip_allgather_recon (RCCL), " + exchange) if dist is not None else "none (1 GPU)"}}, **extra)


def run_mesh(pkg, ctx, orc, steps, warmup):
    """SURVEY 8(d) Mode B (informational) through the reference's own exhaustive search: full_pixel_exhaustive
    (mcomp.c:1547-1617) for every 16x16 block of a 4K 10-bit frame pair, (a) one dense pass range 16 / interval 1
    (33 rows x 32 columns + the start position = 1057 SADs per block: the reference's four-at-a-time column rule
    leaves column +16 out) and (b) the speed-0 good-quality pattern {64,8},{28,4},{15,1},{7,1}."""
    sp = SearchPipeline(pkg, ctx, None, 0, 1, frames=2)
    n = sp.n
    out = {"workload": "mesh_search_4k_10bit", "blocks_per_frame": n}
    for name, pat, cands in (("dense_range16", [(16, 1), (16, 1), (0, 0), (0, 0)], 1057),
                             ("good_quality_speed0", [(64, 8), (28, 4), (15, 1), (7, 1)], 17 * 17 + 15 * 15 + 31 * 28 + 15 * 12 + 4)):
        def frame(f=0):
            ctx.mesh_search_batch(sp.src, sp.ref, f, 16, 16, pkg.capi.MV_COST_L1_HDRES, pat, 0, sp.d_blocks, n, sp.d_mv, sp.d_cost)
        for _ in range(warmup):
            frame()
        ctx.sync()
        ctx.timer_begin()
        for k in range(steps):
            frame(k % sp.F)
        ms = ctx.timer_end() / steps
        out[name] = {"ms_per_frame": ms, "frames_per_s": 1e3 / ms, "sad_candidates_per_s": n * cands / ms * 1e3,
                     "candidates_per_block": cands}
    # exact check of a sample of the last launch against the oracle
    f = (steps - 1) % sp.F
    idx = np.arange(0, n, max(1, n // 200))
    mv = ctx.from_device(sp.d_mv, (n, 2), np.int16)[idx]
    s_, r_ = pkg.synth.shifted_smooth_pair(sp.W, sp.H, f, sp.BD, shift=(3 + f % 3, -2 + f % 2), frac8=(f % 8, (3 * f) % 8))
    sb, rb = orc.extend_plane(s_, sp.BORDER, sp.src.stride), orc.extend_plane(r_, sp.BORDER, sp.ref.stride)
    wmv, _ = orc.mesh_search_batch(sb, rb, sp.BORDER, 16, 16, sp.h_blocks[idx], [(64, 8), (28, 4), (15, 1), (7, 1)], 0, 3, sp.BD, threads=8)
    out["parity_sample"] = bool(np.array_equal(mv, wmv))
    out["value"], out["unit"] = out["dense_range16"]["sad_candidates_per_s"], "candidates/s"
    sp.free()
    return out


def run_first_pass(pkg, ctx, orc, steps, warmup):
    """The inter half of the first pass for whole 4K 10-bit frames in one call each (aomhip_first_pass_inter_frame): 240 x 135 blocks of
    16x16, NSTEP on the first-pass site table with entropy MV costs, last + golden reference, the best_ref_mv chain of every block row kept
    on the device (one column of 135 searches at a time).  Beside it: the chain-free part alone (both zero-MV legs of every block through
    aomhip_first_pass_motion_search_batch), i.e. what the frame would cost if the raster dependency did not exist."""
    capi = pkg.capi
    sp = SearchPipeline(pkg, ctx, None, 0, 1, frames=3)
    cols, rows = sp.W // sp.BS, sp.H // sp.BS
    n = sp.n
    assert n == rows * cols
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    tj, t0, t1 = np.array([200, 650, 640, 1050], np.int32), (150 + bits * 310).astype(np.int32), (170 + bits * 290 + (v & 7) * 3).astype(np.int32)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    q = capi.SearchParams.make("NSTEP_FPF", 1, capi.MV_COST_ENTROPY, sad_per_bit=24, error_per_bit=70)
    rng = np.random.default_rng(3)
    intra = rng.integers(0, 1 << 16, n).astype(np.int32)       # around the inter errors of this content: the chain is both carried and reset
    d_i = ctx.to_device(intra)
    fp = capi.FirstPassParams(rows, cols, 0, 0)
    outs = [ctx.malloc(n * 4) for _ in range(5)]
    def frame(f=0):   # source f; last = ref f, golden = ref f+1, last source = ref f+2 (slots of one ring)
        ctx.first_pass_inter_frame(sp.src, f, sp.ref, f, sp.ref, (f + 1) % sp.F, sp.ref, (f + 2) % sp.F, sp.BS, sp.BS, q, fp, sp.d_blocks, d_i, outs[0], outs[2],
                                   outs[1], outs[3], outs[4], d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    for _ in range(warmup):
        frame()
    ctx.sync()
    ctx.timer_begin()
    for k in range(steps):
        frame(k % sp.F)
    ms = ctx.timer_end() / steps
    t0w = time.perf_counter()
    frame(); ctx.sync()
    wall_ms = (time.perf_counter() - t0w) * 1e3
    def legs(f=0):
        ctx.first_pass_motion_search_batch(sp.src, sp.ref, f, sp.BS, sp.BS, q, sp.d_blocks, n, sp.d_mv, sp.d_cost, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    legs(); ctx.sync()
    ctx.timer_begin()
    for k in range(steps):
        legs(k % sp.F); legs(k % sp.F)
    ms_legs = ctx.timer_end() / steps
    # parity of the last launch on a sample of block rows (rows are independent chains)
    f = (steps - 1) % sp.F
    frame(f); ctx.sync()
    got = [ctx.from_device(outs[0], (n, 2), np.int16), ctx.from_device(outs[1], (n, 2), np.int16), ctx.from_device(outs[2], (n,), np.int32),
           ctx.from_device(outs[3], (n,), np.int32), ctx.from_device(outs[4], (n,), np.int32)]
    parity, moved = None, None
    if orc is not None:
        def plane(ring, slot):
            return ctx.planes_download(ring, slot)
        sb, lb, gb, lsb = plane(sp.src, f), plane(sp.ref, f), plane(sp.ref, (f + 1) % sp.F), plane(sp.ref, (f + 2) % sp.F)
        oq = orc.search_params("NSTEP_FPF", 1, 0, sad_per_bit=24, error_per_bit=70, no_cost_list=1)
        pick = np.array([0, rows // 2, rows - 1])
        idx = (pick[:, None] * cols + np.arange(cols)[None, :]).ravel()
        want = orc.first_pass_inter_frame(sb, lb, gb, lsb, sp.BORDER, sp.BS, sp.h_blocks[idx], len(pick), cols, oq, intra[idx], 0, 0, tj, t0, t1, bd=sp.BD)
        parity = bool(all(np.array_equal(g[idx], w) for g, w in zip(got, want)))
    best = got[0].reshape(rows, cols, 2)
    moved = float((best[:, :-1] != 0).any(2).mean())
    out = {"workload": "first_pass_4k_10bit", "blocks_per_frame": n, "ms_per_frame": ms, "frames_per_s": 1e3 / ms, "wall_ms_one_frame": wall_ms,
           "block_columns": cols, "ms_zero_mv_legs_only": ms_legs, "share_of_blocks_with_nonzero_best_ref_mv": moved, "parity_sample_rows": parity,
           "value": n / ms * 1e3, "unit": "blocks/s"}
    for d in [d_j, d_c0, d_c1, d_i] + outs:
        ctx.free(d)
    sp.free()
    return out
