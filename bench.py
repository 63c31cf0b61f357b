#!/usr/bin/env python3
"""bench.py -- throughput of the aom_dsp hot path on MI355X (contract: see the task prompt).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

N > 1 is launched by the driver with torch.distributed.run (one rank per GPU).  The frame is
cut into N uniform tile columns (av1/common/tile_common.c:76-97); rank r owns column r and
processes only that column's blocks.  The ring of frame pairs grows with N (F frames per
rank), so per-GPU work is fixed: "scaling": "weak".  There is no data-path collective in the
SAD search itself (the reference planes are inputs); the only torch.distributed traffic is
the barrier / max-reduce of the timing.

A "step" is one pass of the hot path over the whole ring -- SURVEY.md 8(d) "Mode A", 5 candidates per 16x16
block: the mv (0,0) candidate plus one x4d group of four uniformly random positions in [-64,64]^2.  The step is
ONE `aomhip_sad_sb_batch` launch (superblock-bucketed lists, reference window staged in LDS by persistent
workgroups); the direct kernels (`aomhip_sad_batch` + `aomhip_sad_x4d_batch`, arbitrary lists) are timed beside it
and reported under "kernels".  Inputs are resident in HBM before the timed region starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The CPU-baseline leg pins its OpenMP threads, one per physical core (set before liboracle / libgomp load).
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

# the workloads live in benchlib/, one module each; this file keeps the driver contract: arguments, rank start-up, which workloads a run holds,
# the ONE JSON line
from benchlib import common  # noqa: E402
from benchlib.dist import barrier, dist_setup  # noqa: E402
from benchlib.encoder import run_cdef_search, run_compound_search, run_int_pro, run_tf, run_warp_error, run_wiener_stats  # noqa: E402
from benchlib.filters import run_filters_ring  # noqa: E402
from benchlib.inner_loop import run_inner_loop  # noqa: E402
from benchlib.line import build_lines, emit_lines  # noqa: E402
from benchlib.sad import WORKLOADS, run_sad_diamond_lists, run_workload  # noqa: E402
from benchlib.search import run_first_pass, run_launcher_dry_run, run_mesh, run_search, run_search_default  # noqa: E402
from benchlib.txq import TXQ_WORKLOADS, run_txq  # noqa: E402
from benchlib.variance import VAR_WORKLOADS, run_variance  # noqa: E402


def spawn_ranks(n):
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # Supervise ALL ranks: when one exits non-zero (comm_init failure, assert, out of memory) or the wall-clock limit passes, the others
    # would sit in ncclSend / ncclRecv or the barrier for ever -- end them and exit non-zero.  Only fresh children are ever started;
    # this parent never touches the GPU.
    limit = float(os.environ.get("AOMHIP_BENCH_RANKS_TIMEOUT_S", "1800"))
    t0, rc, alive = time.time(), 0, list(procs)
    while alive:
        for p_ in list(alive):
            r_ = p_.poll()
            if r_ is not None:
                alive.remove(p_)
                rc = max(rc, abs(r_))
        if alive and (rc != 0 or time.time() - t0 > limit):
            if rc == 0:
                rc = 124
                print("bench.py: ranks still running after %.0f s: terminating them" % limit, file=sys.stderr)
            else:
                print("bench.py: a rank exited with status %d: terminating the others" % rc, file=sys.stderr)
            for p_ in alive:
                p_.terminate()
            t1 = time.time()
            while any(p_.poll() is None for p_ in alive) and time.time() - t1 < 10:
                time.sleep(0.1)
            for p_ in alive:
                if p_.poll() is None:
                    p_.kill()
            for p_ in alive:
                p_.wait()
            break
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None,
                    help="default: sad16x16_modeA_1080p_8bit (BASELINE.json's metric) at every N; with N > 1 the line also carries the "
                         "strong-scaling search pipeline with its per-frame RCCL exchange as `strong_scaling_search`",
                    choices=sorted(WORKLOADS) + ["txq_1080p_8bit", "txq_4k_10bit", "search_4k_10bit", "inner_loop_4k_10bit", "default_search_4k_10bit", "cdef_search_4k_10bit",
                                                "wiener_stats_4k", "warp_error_4k", "int_pro_4k_8bit", "tf_motion_search_4k_10bit", "sad_diamond_lists_4k_8bit", "first_pass_4k_10bit", "compound_search_4k_10bit",
                                                "filters_ring_4k_10bit", "launcher_dry_run"] + sorted(VAR_WORKLOADS))
    ap.add_argument("--others", default="auto", help="comma list of extra workloads reported under 'others' (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bit-depth", type=int, default=0, choices=[0, 8, 10],
                    help="audit aid: run the *_4k_10bit search / filter workloads on planes of this depth instead (0 = the workload's own); the line says so")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real multi-GPU path); gloo only to dry-run the N > 1 code on one GPU")
    ap.add_argument("--frames-per-gpu", type=int, default=0, help="override the ring size per GPU (0 = workload default)")
    ap.add_argument("--tile-columns", default="uniform", choices=["uniform", "balanced"],
                    help="N > 1: how the frame is cut into one tile column per GPU -- the reference's uniform spacing (tile_common.c:76-97) or its "
                         "auto_tile_size_balancing (encoder.c:247-275: widths within one superblock of each other)")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"],
                    help="N > 1 search pipeline: what aomhip_allgather_recon moves per frame (both are timed; this one is in `value`)")
    args = ap.parse_args()
    if args.bit_depth:   # (audit aid, tools/bd_audit.sh: the same kernels' other instantiation on the same content)
        from benchlib import search as _search
        common.BD_OVERRIDE = _search.SearchPipeline.BD = args.bit_depth
        print("note: planes of %d bits instead of the workload's own depth (--bit-depth)" % args.bit_depth, file=sys.stderr)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, BEFORE this process touches the
        # GPU (it never does: it only waits).  Under torch.distributed.run WORLD_SIZE is set and this branch is not taken.
        sys.exit(spawn_ranks(args.gpus))

    common.FRAMES_OVERRIDE = args.frames_per_gpu
    common.TILE_COLUMNS = args.tile_columns
    dist, rank, world = dist_setup(args.gpus, args.dist_backend)
    if args.workload == "launcher_dry_run":   # (no device, no oracle: the launcher / process group / line around a measurement)
        assert dist is not None, "launcher_dry_run is an N > 1 check: --gpus N"
        run_launcher_dry_run(args, dist, rank, world)
        return
    default_multi = args.workload is None and dist is not None
    if args.workload is None:
        # ONE metric at every N: BASELINE.json's SAD-candidates/s on the 1080p 8-bit configuration, whole-job aggregate (tile columns are
        # independent: weak scaling, no data-path collective).  With N > 1 the line additionally carries the strong-scaling search
        # pipeline with its per-frame RCCL exchange as `strong_scaling_search`.
        args.workload = "sad16x16_modeA_1080p_8bit"
    dev = 0
    if world > 1:
        import torch
        dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    import aom_av1_psy_amd as pkg  # raises if libaomhip.so is missing: no fallback
    # one stream, the context's own: the kernels AND the RCCL exchange (aomhip_allgather_recon) are enqueued on it, so a step is
    # ordered without host synchronisation; torch.distributed only carries the barrier / reductions around the timed region
    ctx = pkg.capi.Context(dev, None)
    orc = None
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle as orc  # checker + cpu_baseline only
    except Exception as e:  # pragma: no cover
        print("warning: oracle unavailable (%s): no parity spot check / cpu_baseline" % e, file=sys.stderr)

    def search_block(with_single):
        """configs[3]: the search pipeline with the per-frame strip exchange over all ranks (strong scaling) + the same box's 1-GPU figure."""
        r = run_search(pkg, ctx, dist, dev, rank, world, orc, args.steps, args.warmup, args.exchange)
        blk = {"metric": "search blocks/s", "value": r["value"], "unit": "blocks/s", "scaling": "strong", "ms_per_step": r["ms_per_step"],
               "frames_per_s": r["frames_per_s"], "parity_sample_slot0": r["parity_sample_slot0"], "config": dict(r["config"], workload=r["workload"]),
               "rccl_ranks_in_communicator": r.get("rccl_ranks_in_communicator")}
        for k in ("exchange", "tile_columns_px", "blocks_max_rank_over_mean", "tile_columns_px_uniform", "tile_columns_px_balanced"):
            blk[k] = r.get(k)
        if with_single:
            # the same box's 1-GPU figure of THIS metric (rank 0 alone, whole frame, no exchange), so the speed-up can be read off one line
            if rank == 0:
                one = run_search(pkg, ctx, None, dev, 0, 1, None, args.steps, args.warmup)
                blk["single_gpu_same_box"] = {"value": one["value"], "ms_per_step": one["ms_per_step"]}
                blk["speedup_over_single_gpu"] = r["value"] / one["value"]
            barrier(dist, dev)
        return r, blk

    if args.workload == "search_4k_10bit":  # configs[3] as the headline (any N): profiling / exchange studies
        r, blk = search_block(False)
        line = {"metric": "search blocks/s", "value": r["value"], "unit": "blocks/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u16",
                "data": "synthetic", "config": dict(r["config"], workload=r["workload"]),
                "frames_per_s": r["frames_per_s"], "parity_sample_slot0": r["parity_sample_slot0"]}
        for k in ("exchange", "tile_columns_px", "blocks_max_rank_over_mean", "rccl_ranks_in_communicator", "tile_columns_px_uniform", "tile_columns_px_balanced"):
            if k in r:
                line[k] = r[k]
        ctx.close()
        if rank == 0:
            print(json.dumps(line))
        if dist is not None:
            dist.destroy_process_group()
        return
    if args.workload == "inner_loop_4k_10bit":  # profiling convenience: the configs[4] chain with per-stage timings (single GPU)
        r = run_inner_loop(pkg, ctx, orc, args.steps, args.warmup)
        ctx.close()
        print(json.dumps({"metric": "encode inner loop frames/s", "value": r["value"], "unit": "frames/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_frame"], "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
                          "config": dict(r["config"], workload=r["workload"]), "stages": r["stages"], "valu_issue_rates": r.get("valu_issue_rates"), "deblock_in_frame": r["deblock_in_frame"],
                          "launch": r["launch"], "without_graph": r["without_graph"], "recon_psnr_db_last_frame": r["recon_psnr_db_last_frame"]}))
        return
    if args.workload == "default_search_4k_10bit":  # informational: NSTEP full-pel + 8-tap sub-pel tree (single GPU)
        r = run_search_default(pkg, ctx, orc, args.steps, args.warmup)
        ctx.close()
        print(json.dumps(dict(r, metric="search blocks/s", n_gpus=1, steps=args.steps, warmup=args.warmup, higher_is_better=True,
                              scaling="weak", vs_baseline=None, dtype="u16", data="synthetic",
                              ms_per_step=r["full_pixel_search_NSTEP_ms_per_frame"] + r["subpel_tree_8tap_ms_per_frame"])))
        return
    if args.workload == "sad_diamond_lists_4k_8bit":  # non-Mode-A lists through the bucketed kernel (single GPU)
        r = run_sad_diamond_lists(pkg, ctx, orc, args.steps, args.warmup)
        ctx.close()
        print(json.dumps(dict(r, metric="SAD-candidates/s", n_gpus=1, steps=args.steps, warmup=args.warmup, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype="u8", data="synthetic", ms_per_step=r["sad_strip_kernel_ms"])))
        return
    if args.workload == "first_pass_4k_10bit":  # the first pass's inter half, one call per frame (single GPU)
        r = run_first_pass(pkg, ctx, orc, args.steps, args.warmup)
        ctx.close()
        print(json.dumps(dict(r, metric="first-pass blocks/s", n_gpus=1, steps=args.steps, warmup=args.warmup, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype="u16", data="synthetic", ms_per_step=r["ms_per_frame"], config={"workload": r["workload"]})))
        return
    if args.workload == "compound_search_4k_10bit":  # the RD path's compound / OBMC searches (single GPU)
        only = int(os.environ.get("AOMHIP_BENCH_COMPOUND_BS", "0"))   # profiling aid: this block size alone (profiles/r05e_compound_pmc.json)
        r = run_compound_search(pkg, ctx, orc if not only else None, args.steps, args.warmup, bd=args.bit_depth or 10, bs=only or 16)
        # the same five calls over the frame cut into 8x8, 32x32 and 64x64 blocks (timing only; the tests cover the sizes' parity)
        r["by_block_size"] = {"%dx%d" % (only or 16, only or 16): {k: v["ms_per_frame"] for k, v in r.items() if isinstance(v, dict) and "ms_per_frame" in v}}
        for bs_ in () if only else (8, 32, 64):
            r2 = run_compound_search(pkg, ctx, None, max(3, args.steps // 2), 1, bd=args.bit_depth or 10, bs=bs_)
            r["by_block_size"]["%dx%d" % (bs_, bs_)] = dict({k: v["ms_per_frame"] for k, v in r2.items() if isinstance(v, dict) and "ms_per_frame" in v},
                                                            blocks_per_frame=r2["blocks_per_frame"])
        ctx.close()
        print(json.dumps(dict(r, metric="compound blocks/s", n_gpus=1, steps=args.steps, warmup=args.warmup, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype="u16", data="synthetic", ms_per_step=r["ms_per_frame"], config={"workload": r["workload"]})))
        return
    if args.workload == "tf_motion_search_4k_10bit":  # SURVEY 8(f) row 1 (single GPU)
        r = run_tf(pkg, ctx, orc, args.steps, args.warmup)
        r8 = run_tf(pkg, ctx, None, args.steps, args.warmup, bd=8)   # the same pass on an 8-bit window (timing only)
        r["same_pass_8bit"] = {k: r8[k] for k in ("q30_mesh_pruned_when_close", "q12_mesh_always", "value")}
        ctx.close()
        print(json.dumps(dict(r, metric="tf block searches/s", n_gpus=1, steps=args.steps, warmup=args.warmup, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype="u16", data="synthetic", ms_per_step=r["q30_mesh_pruned_when_close"]["ms_per_filtered_frame"])))
        return
    if args.workload in ("cdef_search_4k_10bit", "wiener_stats_4k", "warp_error_4k", "int_pro_4k_8bit"):  # informational encoder-side searches (single GPU)
        r = {"cdef_search_4k_10bit": run_cdef_search, "wiener_stats_4k": run_wiener_stats, "warp_error_4k": run_warp_error,
             "int_pro_4k_8bit": run_int_pro}[args.workload](
            pkg, ctx, orc, args.steps, args.warmup)
        ctx.close()
        first = r["full_search_64"] if "full_search_64" in r else (r["8bit_units64"] if "8bit_units64" in r else (r["64x64"] if "64x64" in r else {"ms_per_frame": r["10bit"]["ms_per_call"]}))
        print(json.dumps(dict(r, metric=r["unit"], n_gpus=1, steps=args.steps, warmup=args.warmup, higher_is_better=True, scaling="weak",
                              vs_baseline=None, dtype="u16" if "cdef" in args.workload else "u8", data="synthetic",
                              ms_per_step=first["ms_per_frame"], config={"workload": r["workload"]})))
        return
    if args.workload in VAR_WORKLOADS or args.workload == "filters_ring_4k_10bit":  # profiling convenience: one HBM-ring workload (single GPU)
        r = run_variance(pkg, ctx, orc, args.steps, args.warmup, args.workload) if args.workload in VAR_WORKLOADS else run_filters_ring(pkg, ctx, orc, args.steps, args.warmup)
        ctx.close()
        print(json.dumps(dict(r, metric=r["unit"], n_gpus=1, steps=args.steps, warmup=args.warmup, higher_is_better=True, scaling="weak", vs_baseline=None,
                              dtype="u8" if "8bit" in args.workload else "u16", data="synthetic", config=dict(r["config"], workload=r["workload"]))))
        return
    if args.workload in TXQ_WORKLOADS:  # profiling convenience: transform+quantise only (single GPU)
        r = run_txq(pkg, ctx, orc, args.steps, args.warmup, not args.no_cpu_baseline, args.workload)
        ctx.close()
        print(json.dumps({"metric": "fwd_txfm+quant blocks/s", "value": r["value"], "unit": "blocks/s", "n_gpus": 1,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
                          "data": "synthetic", "config": dict(r["config"], workload=r["workload"]),
                          "roofline": r["roofline"], "cpu_baseline": r.get("cpu_baseline"), "per_size": r["per_size"]}))
        return
    main_res = run_workload(pkg, ctx, dist, dev, rank, world, args.workload, args.steps, args.warmup,
                            not args.no_cpu_baseline and world == 1, orc)
    others, strong = [], None
    if world == 1:
        names = ([n for n in ("sad16x16_modeA_4k_8bit", "sad16x16_modeA_4k_10bit", "sad16x16_modeA_1080p_8bit_range32",
                               "sad16x16_modeA_4k_8bit_range32", "sad16x16_modeA_4k_10bit_range32") if n != args.workload]
                 if args.others == "auto" else [n for n in args.others.split(",") if n])
        for n in names:
            if n in TXQ_WORKLOADS:
                others.append(run_txq(pkg, ctx, orc, args.steps, args.warmup, not args.no_cpu_baseline, n))
            else:
                others.append(run_workload(pkg, ctx, dist, dev, rank, world, n, args.steps, args.warmup, False, orc))
        if args.others == "auto" and orc is not None:
            others.append(run_txq(pkg, ctx, orc, args.steps, args.warmup, not args.no_cpu_baseline))
            others.append(run_txq(pkg, ctx, orc, args.steps, args.warmup, not args.no_cpu_baseline, "txq_4k_10bit"))
            for vn in VAR_WORKLOADS:   # the variance half of "SAD / variance"
                others.append(run_variance(pkg, ctx, orc, max(5, args.steps // 2), 1, vn))
            others.append(run_filters_ring(pkg, ctx, orc, max(3, args.steps // 5), 1))   # deblock / CDEF on a 1.3 GB ring
            others.append(run_search(pkg, ctx, None, dev, 0, 1, orc, max(4, args.steps // 2), 1))
            others.append(run_inner_loop(pkg, ctx, orc, max(40, 2 * args.steps), 2))   # (two ring slots of different cost: enough frames for a stable mean)
            others.append(run_mesh(pkg, ctx, orc, max(4, args.steps // 4), 1))
            others.append(run_search_default(pkg, ctx, orc, max(4, args.steps // 2), 1))
            others.append(run_cdef_search(pkg, ctx, orc, max(4, args.steps // 4), 1))
            others.append(run_wiener_stats(pkg, ctx, orc, max(3, args.steps // 6), 1))
            others.append(run_tf(pkg, ctx, orc, max(4, args.steps // 4), 1))
            others.append(run_first_pass(pkg, ctx, orc, max(3, args.steps // 6), 1))
            others.append(run_compound_search(pkg, ctx, orc, max(4, args.steps // 3), 1))
            others.append(run_sad_diamond_lists(pkg, ctx, orc, max(6, args.steps // 2), 1))
    if dist is not None and default_multi:
        # mandatory companion of the N > 1 line (the forced one-rank dry run emits the same schema).  The headline (SAD, no collective)
        # has been measured by now: a failure that every rank sees alike (communicator set-up, allocation) is reported inside the block
        # instead of taking the line with it.  (A rank that dies alone is the launcher's business: spawn_ranks / torchrun end the others.)
        try:
            _, strong = search_block(True)
        except Exception as e:  # noqa: BLE001
            strong = {"metric": "search blocks/s", "error": "%s: %s" % (type(e).__name__, e)}
    ctx.close()

    if rank == 0:
        full, line = build_lines(args, world, main_res, others, strong)
        emit_lines(full, line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
