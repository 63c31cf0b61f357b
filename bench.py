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

from benchlib.common import HBM_PEAK_GBS, kernel_avg_ms, ramp  # noqa: E402
from benchlib.filters import run_filters_ring  # noqa: E402
from benchlib.variance import VAR_WORKLOADS, run_variance  # noqa: E402

FRAMES_OVERRIDE = 0
TILE_COLUMNS = "uniform"   # --tile-columns
SAD16_BYTES_8BIT = 516  # src block + ref block + 4 B result

WORKLOADS = {
    # BASELINE.json configs[1]
    "sad16x16_modeA_1080p_8bit": dict(width=1920, height=1080, bit_depth=8, frames=64),
    # the north-star target size
    "sad16x16_modeA_4k_8bit": dict(width=3840, height=2160, bit_depth=8, frames=64),
    "sad16x16_modeA_4k_10bit": dict(width=3840, height=2160, bit_depth=10, frames=32),
}
# The same three rings under a +-32 search-range contract (lists uniform in [-32, 32]^2, aomhip_sad_sb_batch's `range` = 32): the LDS window's
# halo halves, so a step of the strip walk holds 30 blocks instead of 20 on 16-bit planes and the 8-bit cells get wider (profiles/r04_sad_strip.md).
# Reported NEXT TO the +-64 figures (roofline.*_range32), never instead of them.
for _k in list(WORKLOADS):
    WORKLOADS[_k + "_range32"] = dict(WORKLOADS[_k], search_range=32)


_BACKEND = "nccl"  # RCCL; "gloo" only for the single-GPU dry run of the N > 1 code path (tools/gpu_dist_dryrun.sh)


def dist_setup(n_gpus, backend):
    global _BACKEND
    world = int(os.environ.get("WORLD_SIZE", "1"))
    forced = world <= 1 and os.environ.get("AOMHIP_BENCH_FORCE_DIST") == "1"
    if world <= 1 and not forced:
        return None, 0, 1
    import torch
    import torch.distributed as dist
    _BACKEND = backend
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
    return "cuda" if _BACKEND == "nccl" else "cpu"


def barrier(dist, dev):
    if dist is not None:
        import torch
        if _BACKEND == "nccl":
            dist.barrier(device_ids=[dev])
            torch.cuda.synchronize()
        else:
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()


class SadModeA:
    """HBM-resident ring of frame pairs + the Mode-A work list of one tile column."""

    def __init__(self, pkg, ctx, name, rank, world, frames_per_rank=None, seed=1):
        cfg = WORKLOADS[name]
        self.name, self.cfg, self.ctx, self.pkg = name, cfg, ctx, pkg
        W, H, bd = cfg["width"], cfg["height"], cfg["bit_depth"]
        self.F = (frames_per_rank or cfg["frames"])
        self.ring = self.F * world
        self.border = 160
        synth, capi = pkg.synth, pkg.capi
        self.src = ctx.planes_alloc(W, H, self.border, bd, self.ring)
        self.ref = ctx.planes_alloc(W, H, self.border, bd, self.ring)
        self.host_pair0 = None
        self.host_frames = []  # every base frame pair (the CPU baseline walks the same ring as the GPU)
        for f in range(self.F):  # ring slots beyond F re-use the F base frames' pixels
            s = synth.lcg_frame(W, H, 2 * f, 0, bd)
            r = synth.lcg_frame(W, H, 2 * f + 1, 0, bd)
            if f == 0:
                self.host_pair0 = (s, r)
            if f == self.F - 1:
                self.host_pair_last = (s, r)  # pixels of the LAST ring slot (slot ring - 1 re-uses base frame F - 1)
            if rank == 0:
                self.host_frames.append((s, r))
            for k in range(world):
                ctx.planes_upload(self.src, f + k * self.F, s)
                ctx.planes_upload(self.ref, f + k * self.F, r)
        x0, x1 = pkg.partition.column_of_rank(W, world, rank, mode=TILE_COLUMNS)
        self.range = SR = int(cfg.get("search_range", 64))
        cands, groups = synth.mode_a_worklist(W, H, 16, seed=seed, search=SR)
        keep = (cands["sx"] >= x0) & (cands["sx"] < x1)
        self.blocks_per_frame = int(keep.sum())
        base_c, base_g = cands[keep], groups[keep]
        n = self.blocks_per_frame
        # distinct random positions per frame (same block grid)
        rng = np.random.default_rng(seed + 977 * rank)
        allg = np.tile(base_g, (self.ring, 1))
        allg["rx"] = allg["sx"][..., None] + rng.integers(-SR, SR + 1, (self.ring, n, 4), dtype=np.int16)
        allg["ry"] = allg["sy"][..., None] + rng.integers(-SR, SR + 1, (self.ring, n, 4), dtype=np.int16)
        self.h_cands, self.h_groups0 = base_c, allg[0].copy()
        self.h_groups_last = allg[self.ring - 1].copy()
        self.h_groups_all = allg[:self.F] if rank == 0 else None
        self.d_cands = ctx.to_device(base_c) if n else None
        self.d_groups = ctx.to_device(allg) if n else None
        # Superblock-bucketed copy of the same lists (aomhip_sad_sb_batch), range 64.  The kernel walks STRIPS (columns of
        # cells) with the reference window in an LDS ring, so a cell is one step of that walk.  It is the path the step uses;
        # AOMHIP_SAD_PATH=direct|sb overrides.
        self.path = os.environ.get("AOMHIP_SAD_PATH", "sb")
        # Cells are anchored at x = 0, tile columns start at multiples of their width: a cell width that divides the
        # column width keeps every strip inside one rank's column.  Tuned width when it divides, else the largest
        # divisor below it.
        # r02 sweeps (profiles/r02_sad_strip.md): 60 blocks per step keep all eight evaluating wavefronts busy, two per SIMD, and the strips
        # per frame x 64 frames must be a whole number of items per CU: 8-bit 240 x 64 at 1080p (8 strips), 320 x 48 at 4K (12 strips, ~2 %
        # ahead of 240 x 64 there); 16-bit 160 x 32 (LDS)
        col_w = pkg.partition.column_of_rank(W, world, 0, mode=TILE_COLUMNS)[1] - pkg.partition.column_of_rank(W, world, 0, mode=TILE_COLUMNS)[0]
        # (a rank's items = strips of its column x ring frames; the kernel's persistent grid is 256 workgroups: prefer the cell whose
        # item count is a multiple of that -- 320 x 48 on a whole 4K frame, 240 x 64 on a 1080p frame or a 1920 / 960-wide tile column)
        options = [(320, 48), (240, 64)] if bd == 8 else [(160, 32)]
        if SR <= 32:  # r04 sweep (profiles/r04_sad_strip.md): 480 x 32 (8-bit), 160 x 48 (16-bit: 24 strips x 32 frames = 3 items per workgroup)
            options = [(480, 32), (320, 48), (240, 64)] if bd == 8 else [(160, 48), (256, 32), (160, 32)]
        fits = [c for c in options if col_w % c[0] == 0]
        whole = [c for c in fits if ((col_w // c[0]) * self.ring) % 256 == 0]
        tuned, cell_h = (whole or fits or options[-1:])[0]
        cw = tuned if col_w % tuned == 0 else max([d for d in range(16, tuned + 1, 16) if col_w % d == 0] or [tuned])
        self.cell = (cw, cell_h)
        self.d_sb = None
        if n and self.path == "sb":
            perm, off = synth.bucket_order(base_c["sx"], base_c["sy"], W, H, *self.cell)
            self.perm, self.n_buckets = perm, len(off) - 1
            self.d_sb = (ctx.to_device(np.ascontiguousarray(allg[:, perm])), ctx.to_device(base_c[perm]), ctx.to_device(off))
            self.d_sb_out4 = ctx.malloc(max(16, self.ring * n * 16))
            self.d_sb_out1 = ctx.malloc(max(16, self.ring * n * 4))
        self.d_out1 = ctx.malloc(max(16, self.ring * n * 4))
        self.d_out4 = ctx.malloc(max(16, self.ring * n * 16))
        self.cands_per_step = 5 * n * self.ring
        self.tile = (x0, x1)

    def launch_single(self):
        if self.blocks_per_frame:
            self.ctx.sad_batch(self.src, self.ref, 0, self.ring, 16, 16, 0, self.d_cands, self.blocks_per_frame, 0,
                               self.d_out1)

    def launch_x4d(self):
        if self.blocks_per_frame:
            self.ctx.sad_x4d_batch(self.src, self.ref, 0, self.ring, 16, 16, 0, self.d_groups, self.blocks_per_frame,
                                   self.blocks_per_frame, self.d_out4)

    def launch_sb(self):
        if self.d_sb:
            n = self.blocks_per_frame
            self.ctx.sad_sb_batch(self.src, self.ref, 0, self.ring, 16, 16, 0, self.cell[0], self.cell[1], self.range, self.n_buckets,
                                  self.d_sb[0], self.d_sb[2], n, n, self.d_sb_out4, self.d_sb[1], self.d_sb[2], n, 0,
                                  self.d_sb_out1)

    def launch_probe(self):
        """the transport of launch_sb alone (aomhip_strip_read_probe): same ring, same cells, same range, nothing evaluated."""
        if self.d_sb:
            self.probe_bytes = self.ctx.strip_read_probe(self.src, self.ref, 0, self.ring, self.tile[0], self.tile[1], self.cell[0], self.cell[1], self.range)

    def step(self):
        if self.path == "sb":
            self.launch_sb()
        else:
            self.launch_single()
            self.launch_x4d()

    def bytes_per_cand(self):
        return SAD16_BYTES_8BIT if self.cfg["bit_depth"] == 8 else 1028

    def check_frame0(self, orc):
        """Exact check of ring slot 0 AND of the last ring slot against the oracle (not timed): the last slot has its own list of
        reference positions and sits at the far end of every per-frame stride the launch uses."""
        n = self.blocks_per_frame
        if not n:
            return True
        bd = self.cfg["bit_depth"]
        ok = True
        for slot, (s, r), groups in ((0, self.host_pair0, self.h_groups0), (self.ring - 1, self.host_pair_last, self.h_groups_last)):
            sb = orc.extend_plane(s, self.border, self.src.stride)
            rb = orc.extend_plane(r, self.border, self.ref.stride)
            if self.path == "sb":  # un-permute the bucket order
                got1, got4 = np.empty((n,), np.uint32), np.empty((n, 4), np.uint32)
                got1[self.perm] = self.ctx.from_device(self.d_sb_out1 + slot * n * 4, (n,), np.uint32)
                got4[self.perm] = self.ctx.from_device(self.d_sb_out4 + slot * n * 16, (n, 4), np.uint32)
            else:
                got1 = self.ctx.from_device(self.d_out1 + slot * n * 4, (n,), np.uint32)
                got4 = self.ctx.from_device(self.d_out4 + slot * n * 16, (n, 4), np.uint32)
            ok &= np.array_equal(got1, orc.sad_batch(sb, rb, self.border, 16, 16, self.h_cands, bd=bd, threads=4))
            ok &= np.array_equal(got4, orc.sad_x4d_batch(sb, rb, self.border, 16, 16, groups, bd=bd, threads=4))
        return bool(ok)

    def cpu_baseline(self, orc, seconds=None):
        """The same Mode-A ring on the host cores (oracle/aomref_bench.c, kind "port"): static partition of the candidate
        list over the threads, thread-private results, every base frame pair of the ring; scalar C and AVX2-intrinsics
        kernels, one thread and all physical cores (pinned: OMP_PROC_BIND=close OMP_PLACES=cores)."""
        seconds = float(os.environ.get("AOMHIP_BENCH_CPU_SECONDS", "5.0")) if seconds is None else seconds
        bd = self.cfg["bit_depth"]
        sp = [orc.extend_plane(s, self.border, self.src.stride) for s, _ in self.host_frames]
        rp = [orc.extend_plane(r, self.border, self.ref.stride) for _, r in self.host_frames]
        groups = np.ascontiguousarray(self.h_groups_all).reshape(-1)
        host_phys, logical, model = orc.physical_cores()
        usable, quota = orc.usable_cpus()
        phys = max(1, min(host_phys, usable))  # one thread per core this process may really use
        legs = {}
        for name, threads, avx2, secs in (("scalar_1_thread", 1, 0, seconds * 0.6), ("avx2_1_thread", 1, 1, seconds * 0.6),
                                          ("scalar_all_usable_cores", phys, 0, seconds), ("avx2_all_usable_cores", phys, 1, seconds)):
            rate, done, el = orc.bench_sad_mode_a(sp, rp, self.border, self.h_cands, groups, bd, threads, avx2, secs)
            legs[name] = {"candidates_per_s": rate, "threads": threads, "seconds": el, "candidates": done}
        best = legs["avx2_all_usable_cores"]
        return {"value": best["candidates_per_s"], "unit": "candidates/s", "cores": phys, "kind": "port",
                "cpu_model": model, "logical_cpus": logical, "host_physical_cores": host_phys, "cgroup_cpu_quota": quota, "legs": legs,
                "per_core": best["candidates_per_s"] / phys,
                "sample": "%d candidates = whole passes over the Mode-A lists of all %d base frame pairs of the ring (%.1f s), "
                          "oracle/aomref_bench.c AVX2-intrinsics 16x16 SAD (gcc -O3 -mavx2), static partition over %d pinned "
                          "threads = the cores this process may use (host: %d physical cores, cgroup CPU quota %s); `legs` has the "
                          "scalar-C and 1-thread figures"
                          % (best["candidates"], len(sp), best["seconds"], phys, host_phys, quota),
                "sample_short": "%.1f s of AVX2 16x16 SAD over the ring's Mode-A lists (%d candidates), %d pinned threads" % (best["seconds"], best["candidates"], phys)}

    def free(self):
        c = self.ctx
        for p in (self.src, self.ref):
            c.planes_free(p)
        for d in (self.d_cands, self.d_groups, self.d_out1, self.d_out4) + (tuple(self.d_sb) + (self.d_sb_out4, self.d_sb_out1)
                                                                          if self.d_sb else ()):
            if d:
                c.free(d)


TXQ_SIZES = [(0, 4), (1, 8), (2, 16), (3, 32)]  # (TX_SIZE, n) : TX_4X4, TX_8X8, TX_16X16, TX_32X32


TXQ_WORKLOADS = {
    # BASELINE.json configs[2]: 1920x1088 residual planes of 8-bit video (9-bit signed samples), aom_quantize_b
    "txq_1080p_8bit": dict(width=1920, height=1088, bit_depth=8, frames=32),
    # the metric's other size ("1080p & 4K"): 3840x2176 residual planes of 10-bit video (11-bit signed samples), aom_highbd_quantize_b
    "txq_4k_10bit": dict(width=3840, height=2176, bit_depth=10, frames=12),
}


class TxqGrid:
    """BASELINE.json configs[2]: av1_fwd_txfm2d_{4x4..32x32} + aom_[highbd_]quantize_b over every transform block of
    F residual planes (int16; (bit_depth + 1)-bit signed samples), DCT_DCT, qindex 100.
    One launch per transform size over the whole ring (grid mode: the ring is one tall plane)."""

    def __init__(self, pkg, ctx, orc, name="txq_1080p_8bit", qindex=100, seed=5):
        cfg = TXQ_WORKLOADS[name]
        self.name, self.W, self.H, self.bd = name, cfg["width"], cfg["height"], cfg["bit_depth"]
        frames = cfg["frames"]
        self.hbd = self.bd > 8
        self.ctx, self.pkg, self.orc, self.F = ctx, pkg, orc, frames
        rng = np.random.default_rng(seed)
        m, half = (2 << self.bd) - 1, 1 << self.bd  # 8-bit video: (x & 511) - 256; 10-bit: (x & 2047) - 1024
        mk = lambda: ((rng.integers(0, 1 << 16, (self.H, self.W)) & m) - half).astype(np.int16)
        self.h_res0 = mk()
        self.d_res = ctx.malloc(frames * self.H * self.W * 2)
        self.h_planes = []
        keep = 8 if self.bd == 8 else 2  # the CPU baseline walks >= 33 MB of residual (past any core's private caches)
        for f in range(frames):
            plane = self.h_res0 if f == 0 else mk()
            if f < keep:
                self.h_planes.append(plane)
            pkg.capi.check(pkg.capi.lib.aomhip_memcpy_h2d(ctx.h, self.d_res + f * self.H * self.W * 2,
                                                          plane.ctypes.data, plane.nbytes), "h2d")
        self.samples = frames * self.H * self.W
        self.d_q, self.d_dq = ctx.malloc(self.samples * 4), ctx.malloc(self.samples * 4)
        self.d_eob = ctx.malloc(2 * self.samples // 16)
        self.qt = orc.build_quantizer_y(self.bd, qindex) if orc is not None else None
        self.qp = pkg.capi.QuantParams.from_tables(self.qt) if self.qt else None
        self.blocks = {n: (self.W // n) * (self.H // n) * frames for _, n in TXQ_SIZES}
        self.blocks_per_step = sum(self.blocks.values())

    def launch(self, tx_size, n, tx_type=0):
        self.ctx.xform_quant_batch(self.d_res, self.W, tx_size, None, self.blocks[n], self.W // n, tx_type, self.qp, self.hbd, None,
                                   self.d_q, self.d_dq, self.d_eob)

    def step(self):
        for ts, n in TXQ_SIZES:
            self.launch(ts, n)

    def check(self):
        """Exact check of frame 0, 16x16, against the oracle (not timed)."""
        n = (self.W // 16) * (self.H // 16)
        self.launch(2, 16)
        gq = self.ctx.from_device(self.d_q, (n * 256,), np.int32)
        ge = self.ctx.from_device(self.d_eob, (n,), np.uint16)
        _, wq, _, we = self.orc.xform_quant_batch(self.h_res0, 2, None, n, self.W // 16, 0, self.qt, self.hbd, n * 256,
                                                  False, threads=8)
        return bool(np.array_equal(gq, wq) and np.array_equal(ge, we))

    def cpu_baseline(self, seconds=None):
        """fwd_txfm2d + quantize_b over every 4x4 / 8x8 / 16x16 / 32x32 block of the residual planes on the host cores
        (oracle/aomref_bench.c): blocks partitioned statically over pinned threads, thread-private outputs; scalar C, and
        scalar transform + AVX2 quantiser; one thread and all physical cores."""
        seconds = float(os.environ.get("AOMHIP_BENCH_CPU_SECONDS", "4.0")) if seconds is None else seconds
        host_phys, logical, model = self.orc.physical_cores()
        usable, quota = self.orc.usable_cpus()
        phys = max(1, min(host_phys, usable))
        planes = self.h_planes
        legs = {}
        for name, threads, avx2, secs in (("scalar_1_thread", 1, 0, seconds * 0.6), ("scalar_all_usable_cores", phys, 0, seconds),
                                          ("scalar_txfm+avx2_quant_all_usable_cores", phys, 1, seconds)):
            rate, done, el = self.orc.bench_txq(planes, self.qt, threads, avx2, secs, bd=self.bd)
            legs[name] = {"blocks_per_s": rate, "threads": threads, "seconds": el, "blocks": done}
        best = legs["scalar_txfm+avx2_quant_all_usable_cores"]
        return {"value": best["blocks_per_s"], "unit": "blocks/s", "cores": phys, "kind": "port", "cpu_model": model,
                "logical_cpus": logical, "host_physical_cores": host_phys, "cgroup_cpu_quota": quota, "legs": legs,
                "sample": "%d blocks = whole passes over all 4x4/8x8/16x16/32x32 blocks of %d residual planes (%.1f s), oracle C "
                          "forward transform (scalar, gcc -O3 -mavx2 auto-vectorised) + %s, static "
                          "partition over %d pinned threads" % (best["blocks"], len(planes), best["seconds"],
                                                                "scalar-C aom_highbd_quantize_b (the port has no SIMD form of it)" if self.hbd
                                                                else "AVX2-intrinsics quantize_b", phys)}

    def free(self):
        for d in (self.d_res, self.d_q, self.d_dq, self.d_eob):
            self.ctx.free(d)


def pmc_calibration_ops(ctx):
    """tools/gpu_pmc_txq.sh (AOMHIP_PMC_CALIB=1): two launches with KNOWN HBM byte counts inside the profiled process, so that the
    FETCH_SIZE / WRITE_SIZE counters of the kernels of interest can be scaled by factors measured in the same run: a 256 MiB fill
    (writes only) and aomhip_plane_sse over two 3840x2160 16-bit planes (reads every visible byte of both once, writes 8 bytes)."""
    d = ctx.malloc(256 << 20)
    a, b = ctx.planes_alloc(3840, 2160, 32, 10, 1), ctx.planes_alloc(3840, 2160, 32, 10, 1)
    d_sse = ctx.malloc(8)
    for _ in range(3):
        ctx.memset(d, 1, 256 << 20)
        ctx.plane_sse(a, 0, b, 0, d_sse)
    ctx.sync()
    ctx.free(d); ctx.free(d_sse); ctx.planes_free(a); ctx.planes_free(b)


def run_txq(pkg, ctx, orc, steps, warmup, want_cpu, name="txq_1080p_8bit"):
    if os.environ.get("AOMHIP_PMC_CALIB") == "1":
        pmc_calibration_ops(ctx)
    wl = TxqGrid(pkg, ctx, orc, name)
    ok = wl.check()
    ramp(ctx, wl.step)
    for _ in range(warmup):
        wl.step()
    ctx.sync()
    t0 = time.perf_counter()
    ctx.timer_begin()
    for _ in range(steps):
        wl.step()
    ev_ms = ctx.timer_end()
    wall = time.perf_counter() - t0
    per = {}
    for ts, n in TXQ_SIZES:
        ms = kernel_avg_ms(ctx, lambda: wl.launch(ts, n), max(steps, 10))
        nbytes = wl.blocks[n] * (10 * n * n + 2)  # 2 B in + 4 + 4 B out per sample + eob
        per["%dx%d" % (n, n)] = {"avg_launch_ms": ms, "blocks_per_launch": wl.blocks[n],
                                 "blocks_per_s": wl.blocks[n] / (ms * 1e-3),
                                 "achieved_GBs": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    dom = max(per, key=lambda k: per[k]["avg_launch_ms"])
    res = {"workload": "fwd_txfm2d+quantize_b_%s" % name[4:], "value": wl.blocks_per_step * steps / wall, "unit": "blocks/s",
           "ms_per_step": wall / steps * 1e3, "event_ms_per_step": ev_ms / steps, "blocks_per_step": wl.blocks_per_step,
           "parity_frame0_16x16": ok, "config": {"plane": "%dx%d int16 residual, %d-bit signed" % (wl.W, wl.H, wl.bd + 1), "ring_planes": wl.F,
                                                 "quantiser": "aom_highbd_quantize_b" if wl.hbd else "aom_quantize_b",
                                                 "tx_type": "DCT_DCT", "qindex": 100, "sizes": "4x4,8x8,16x16,32x32 (all blocks of each)"},
           "roofline": {"bound": "hbm",   # measured fabric traffic = algorithmic bytes (profiles/*_pmc_txq*.json); a pure copy kernel runs at 0.63-0.79 here
                        "kernel": "xform_quant_kernel<%s>" % dom, "achieved": per[dom]["achieved_GBs"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": per[dom]["frac"],
                        "traffic": load_traffic(("txq_" if name == "txq_1080p_8bit" else name + "_") + dom),
                        "avg_launch_ms": per[dom]["avg_launch_ms"],
                        "note": "algorithmic bytes = (10*N + 2) per block of N samples (int16 in, qcoeff + dqcoeff out, eob)"},
           "per_size": per}
    if res["roofline"]["traffic"]:   # profiles/traffic.json, tools/gpu_pmc_txq.sh: FETCH_SIZE x 2 (guide) + WRITE_SIZE calibrated on a fill
        t, ms = res["roofline"]["traffic"], per[dom]["avg_launch_ms"]
        res["roofline"]["traffic_GBs"] = t / (ms * 1e-3) / 1e9
        res["roofline"]["traffic_over_algorithmic"] = t / (wl.blocks[int(dom.split("x")[0])] * (10 * int(dom.split("x")[0]) ** 2 + 2))
    if orc is not None and os.environ.get("AOMHIP_BENCH_TXQ_SWEEPS", "1") != "0":
        # the quantiser at qindex 20 / 200 next to the default 100 (the quantiser's dead zone decides how many
        # coefficients survive, the bytes moved do not change) and the 16 transform types of the <= 16x16 sizes, timed on the 16x16 launch
        b16 = wl.blocks[16] * (10 * 256 + 2)
        frac16 = lambda fn: b16 / (kernel_avg_ms(ctx, fn, max(steps, 10)) * 1e-3) / 1e9 / HBM_PEAK_GBS
        qp100 = wl.qp
        res["qindex_sweep_16x16_frac"] = {"100": per["16x16"]["frac"]}
        for qi in (20, 200):
            wl.qp = pkg.capi.QuantParams.from_tables(orc.build_quantizer_y(wl.bd, qi))
            res["qindex_sweep_16x16_frac"][str(qi)] = frac16(lambda: wl.launch(2, 16))
        wl.qp = qp100
        by_type = [frac16(lambda t=t: wl.launch(2, 16, t)) for t in range(16)]
        res["tx_type_sweep_16x16_frac"] = {"min": min(by_type), "max": max(by_type), "by_tx_type": by_type}
    if want_cpu and orc is not None:
        res["cpu_baseline"] = wl.cpu_baseline()
    wl.free()
    return res


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
        self.bounds, self.n_cols = (capi.tile_column_bounds_balanced if TILE_COLUMNS == "balanced" and world & (world - 1) == 0 else capi.tile_column_bounds)(W, world)  # idle ranks (fewer columns than ranks): (0, 0)
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
    bounds, _ = (pkg.capi.tile_column_bounds_balanced if TILE_COLUMNS == "balanced" and world & (world - 1) == 0 else pkg.capi.tile_column_bounds)(W, world)
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
                            "partition": ("balanced tile columns (auto_tile_size_balancing, encoder.c:247-275)" if TILE_COLUMNS == "balanced" else
                                          "uniform tile columns (tile_common.c:76-110)") + ", one per GPU",
                            "exchange": ("per frame, aomhip_allgather_recon (RCCL), " + exchange) if dist is not None else "none (1 GPU)"}}, **extra)


def run_inner_loop(pkg, ctx, orc, steps, warmup):
    """BASELINE.json configs[4] on one GPU: the whole 4K 10-bit encode inner loop per frame, every stage on the
    device and chained through HBM: full-pel diamond search + bilinear sub-pel refinement (16x16 blocks) ->
    motion-compensated prediction at the sub-pel MV (8-tap interpolation) -> subtract + fwd_txfm2d 16x16 + quantize_b (qindex 100) ->
    inverse transform + reconstruction -> deblocking (every 8x8 edge, level 32) -> CDEF (pri 4, sec 2, damping 6)."""
    sp = SearchPipeline(pkg, ctx, None, 0, 1, frames=2)
    W, H, bd, border, F = sp.W, sp.H, sp.BD, sp.BORDER, sp.F
    capi = pkg.capi
    pred = ctx.planes_alloc(W, H, border, bd, F)  # slot f: prediction, then reconstruction, of ring frame f
    out = ctx.planes_alloc(W, H, border, bd, 1)
    dbk = ctx.planes_alloc(W, H, border, bd, 1)
    fused_middle = os.environ.get("AOMHIP_BENCH_MIDDLE", "fused") != "three_calls"   # (three_calls: the separate predictor / transform / inverse launches)
    fused_deblock = os.environ.get("AOMHIP_BENCH_DEBLOCK", "two_pass") == "fused"   # (round 4: the two in-place passes with four lines per lane are the faster form)
    n = sp.n
    nc = 256
    d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    qp = capi.QuantParams.from_tables(orc.build_quantizer_y(bd, 100))
    params = np.zeros((H // 4, W // 4, 4), np.uint8)
    params[:, 2::2, 0] = 8; params[:, 2::2, 1] = 32; params[2::2, :, 2] = 8; params[2::2, :, 3] = 32
    d_params = ctx.to_device(params)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    d_pri, d_sec = ctx.to_device(np.full((fbh, fbw), 4, np.uint8)), ctx.to_device(np.full((fbh, fbw), 2, np.uint8))
    d_skip = ctx.to_device(np.zeros((H // 8, W // 8), np.uint8))
    for f in range(F):
        sp.d_sub_blocks(f)
    state = {"f": 0}

    def frame(f=None):
        if f is None:
            f = state["f"] % F
            state["f"] += 1
        ctx.fullpel_diamond_batch(sp.src, sp.ref, f, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, sp.d_blocks, n, sp.d_mv, sp.d_cost)
        ctx.subpel_bilinear_batch(sp.src, sp.ref, f, 16, 16, capi.MV_COST_L1_HDRES, 2, 1, 0, sp.d_sub_blocks(f), n,
                                  sp.d_smv, sp.d_err, sp.d_dist, sp.d_sse)
        if fused_middle:
            # prediction -> residual -> transform + quantise -> inverse + add in one kernel (csrc/encode_block.hip); EIGHTTAP_REGULAR both ways
            ctx.encode_inter_blocks_batch(sp.src, f, sp.ref, f, pred, f, 16, sp.d_blocks, sp.d_smv, n, qp, d_q, d_dq, d_e, 0, 0, 0)
        else:
            ctx.build_inter_pred_batch(sp.ref, f, pred, f, 16, 16, sp.d_blocks, sp.d_smv, n, 0, 0)
            # grid mode: block i of the plane == block i of the raster list used above
            ctx.subtract_xform_quant_batch(sp.src, pred, f, 2, None, n, W // 16, 0, qp, None, d_q, d_dq, d_e)
            ctx.inv_txfm_add_batch(d_dq, 2, None, n, W // 16, 0, d_e, pred, f)
        # both deblocking passes in one launch, out of place into `dbk` (CDEF reads a second buffer anyway); AOMHIP_BENCH_DEBLOCK=two_pass
        # keeps the in-place vertical + horizontal launches
        if fused_deblock:
            ctx.deblock_plane_fused(pred, f, dbk, 0, d_params, W // 4, 0)
            ctx.cdef_luma_plane(dbk, 0, out, 0, d_pri, d_sec, fbw, d_skip, 6)
        else:
            ctx.deblock_plane(pred, f, d_params, W // 4, 0, 3)
            ctx.cdef_luma_plane(pred, f, out, 0, d_pri, d_sec, fbw, d_skip, 6)

    for _ in range(max(warmup, F)):
        frame()
    ctx.sync()
    # The frame's chain replayed as one hipGraph per ring slot (aomhip_graph_*): the eight launches then follow each other without the
    # queue's per-launch dispatch latency (AOMHIP_BENCH_GRAPH=0: enqueue them one by one; both figures are reported)
    use_graph = os.environ.get("AOMHIP_BENCH_GRAPH", "1") != "0"
    def timed(step_fn):
        step_fn(); ctx.sync()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(steps):
            step_fn()
        ev = ctx.timer_end()
        return time.perf_counter() - t0, ev
    wall_plain, ev_plain = timed(frame)
    wall, ev_ms, graph_note = wall_plain, ev_plain, None
    if use_graph:
        graphs = [ctx.capture(lambda f=f: frame(f)) for f in range(F)]
        def replay():
            f = state["f"] % F
            state["f"] += 1
            ctx.graph_launch(graphs[f])
        wall, ev_ms = timed(replay)
        graph_note = {"frames_per_s_launches_one_by_one": steps / wall_plain, "ms_per_frame_launches_one_by_one": wall_plain / steps * 1e3}
        # Between two graph launches the queue idles ~9 us (profiles/r05_inner_loop_timeline.md): the ring's F frames as ONE graph halve that
        # share per frame (an encoder submits its frames back to back; AOMHIP_BENCH_GRAPH=frame keeps one graph per frame).  Exactly `steps`
        # frames are run: steps // F ring graphs, then the remainder frame by frame.
        if os.environ.get("AOMHIP_BENCH_GRAPH", "ring") != "frame" and F > 1 and steps >= F:
            state["f"] = 0
            ring = ctx.capture(lambda: [frame(f) for f in range(F)])
            def timed_ring():
                ctx.graph_launch(ring); ctx.sync()
                t0 = time.perf_counter()
                ctx.timer_begin()
                for _ in range(steps // F):
                    ctx.graph_launch(ring)
                for f in range(steps % F):
                    ctx.graph_launch(graphs[f])
                ev = ctx.timer_end()
                return time.perf_counter() - t0, ev
            wall_frame, ev_frame = wall, ev_ms
            wall, ev_ms = timed_ring()
            state["f"] = steps % F if steps % F else F
            graph_note.update({"frames_per_s_one_graph_per_frame": steps / wall_frame, "ms_per_frame_one_graph_per_frame": wall_frame / steps * 1e3,
                               "frames_per_graph": F})
            ctx.sync()
            ctx.graph_destroy(ring)
        ctx.sync()
        for g in graphs:
            ctx.graph_destroy(g)
    # ---- the same frame with its two 4:2:0 chroma planes (8x8 chroma blocks at the luma block's MV, TX_8X8, deblock at level 32 on the 8x8
    # chroma grid's 4-sample units, CDEF chroma with the luma directions): luma chain + two chroma chains as ONE graph per ring slot
    # (tests/test_gpu_full_size.py::test_config4... checks this chain bit for bit against the oracle)
    yuv = None
    if use_graph and os.environ.get("AOMHIP_BENCH_420", "1") != "0":
        CW, CH, cbd = W // 2, H // 2, border // 2
        cs = [ctx.planes_alloc(CW, CH, cbd, bd, F) for _ in range(2)]
        cr = [ctx.planes_alloc(CW, CH, cbd, bd, F) for _ in range(2)]
        cp = [ctx.planes_alloc(CW, CH, cbd, bd, F) for _ in range(2)]
        co = ctx.planes_alloc(CW, CH, cbd, bd, 1)
        for f in range(F):
            ys, yr = ctx.planes_download(sp.src, f)[border:border + H, border:border + W], ctx.planes_download(sp.ref, f)[border:border + H, border:border + W]
            for pl, off in enumerate((200, 330)):
                ctx.planes_upload(cs[pl], f, np.clip(ys[::2, ::2].astype(np.int32) // 2 + off, 0, 1023).astype(np.uint16))
                ctx.planes_upload(cr[pl], f, np.clip(yr[::2, ::2].astype(np.int32) // 2 + off, 0, 1023).astype(np.uint16))
        cblocks = sp.h_blocks.copy()
        cblocks["bx"] //= 2; cblocks["by"] //= 2
        d_cb = ctx.to_device(cblocks)
        cparams = np.zeros((CH // 4, CW // 4, 4), np.uint8)
        cparams[:, 2::2, 0] = 6; cparams[:, 2::2, 1] = 32; cparams[2::2, :, 2] = 6; cparams[2::2, :, 3] = 32
        d_cparams = ctx.to_device(cparams)
        d_cq, d_cdq, d_ce = ctx.malloc(n * 64 * 4), ctx.malloc(n * 64 * 4), ctx.malloc(2 * n)
        d_dir, d_var = ctx.malloc((H // 8) * (W // 8)), ctx.malloc((H // 8) * (W // 8) * 4)

        def frame_420(f):
            ctx.fullpel_diamond_batch(sp.src, sp.ref, f, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, sp.d_blocks, n, sp.d_mv, sp.d_cost)
            ctx.subpel_bilinear_batch(sp.src, sp.ref, f, 16, 16, capi.MV_COST_L1_HDRES, 2, 1, 0, sp.d_sub_blocks(f), n, sp.d_smv, sp.d_err, sp.d_dist, sp.d_sse)
            ctx.encode_inter_blocks_batch(sp.src, f, sp.ref, f, pred, f, 16, sp.d_blocks, sp.d_smv, n, qp, d_q, d_dq, d_e, 0, 0, 0)
            ctx.deblock_plane(pred, f, d_params, W // 4, 0, 3)
            ctx.cdef_luma_plane(pred, f, out, 0, d_pri, d_sec, fbw, d_skip, 6, d_dir, d_var)
            for pl in range(2):
                ctx.build_inter_pred_batch(cr[pl], f, cp[pl], f, 8, 8, d_cb, sp.d_smv, n, 0, 0, 1, 1)
                ctx.subtract_xform_quant_batch(cs[pl], cp[pl], f, 1, None, n, CW // 8, 0, qp, None, d_cq, d_cdq, d_ce)
                ctx.inv_txfm_add_batch(d_cdq, 1, None, n, CW // 8, 0, d_ce, cp[pl], f)
                ctx.deblock_plane(cp[pl], f, d_cparams, CW // 4, 0, 3)
                ctx.cdef_chroma_plane(cp[pl], f, co, 0, 1, 1, d_dir, d_pri, d_sec, fbw, d_skip, 6)
        for f in range(F):
            frame_420(f)
        ctx.sync()
        ring420 = ctx.capture(lambda: [frame_420(f) for f in range(F)])
        reps = max(2, steps // F)
        ctx.graph_launch(ring420); ctx.sync()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(reps):
            ctx.graph_launch(ring420)
        ev420 = ctx.timer_end()
        wall420 = time.perf_counter() - t0
        ctx.graph_destroy(ring420)
        yuv = {"ms_per_frame": wall420 / (reps * F) * 1e3, "event_ms_per_frame": ev420 / (reps * F), "frames_per_s": reps * F / wall420,
               "chain": "the luma chain + per chroma plane: 8x8 prediction at the luma MV (ss 1, 1), subtract + fwd_txfm2d_8x8 + quantize_b, inverse + add, deblock, CDEF chroma"}
        for pl in range(2):
            for x in (cs[pl], cr[pl], cp[pl]):
                ctx.planes_free(x)
        ctx.planes_free(co)
        for d in (d_cb, d_cparams, d_cq, d_cdq, d_ce, d_dir, d_var):
            ctx.free(d)
    # sanity: the reconstruction of the last frame is close to its source (fine quantiser, converged search)
    f_last = (state["f"] - 1) % F
    rec = ctx.planes_download(out, 0)[border:border + H, border:border + W].astype(np.int32)
    srcf = ctx.planes_download(sp.src, f_last)[border:border + H, border:border + W].astype(np.int32)
    psnr = 10 * np.log10(1023.0 ** 2 / max(np.mean((rec - srcf) ** 2), 1e-9))
    # per-stage launch times (each stage alone, same inputs) and the algorithmic rate of the memory-bound ones
    # (deblock / CDEF read + write each pixel once per pass; transform stages as the txq workload)
    px_bytes = W * H * 2

    def stage_fns_of(f0):
        return {
            "fullpel_diamond": lambda: ctx.fullpel_diamond_batch(sp.src, sp.ref, f0, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, sp.d_blocks, n, sp.d_mv, sp.d_cost),
            "subpel_bilinear": lambda: ctx.subpel_bilinear_batch(sp.src, sp.ref, f0, 16, 16, capi.MV_COST_L1_HDRES, 2, 1, 0, sp.d_sub_blocks(f0), n, sp.d_smv, sp.d_err, sp.d_dist, sp.d_sse),
            "inter_pred_8tap": lambda: ctx.build_inter_pred_batch(sp.ref, f0, pred, f0, 16, 16, sp.d_blocks, sp.d_smv, n, 0, 0),
            "subtract_xform_quant_16x16": lambda: ctx.subtract_xform_quant_batch(sp.src, pred, f0, 2, None, n, W // 16, 0, qp, None, d_q, d_dq, d_e),
            "inv_txfm_add_16x16": lambda: ctx.inv_txfm_add_batch(d_dq, 2, None, n, W // 16, 0, d_e, pred, f0),
            "encode_inter_blocks_16x16": lambda: ctx.encode_inter_blocks_batch(sp.src, f0, sp.ref, f0, pred, f0, 16, sp.d_blocks, sp.d_smv, n, qp, d_q, d_dq, d_e, 0, 0, 0),
            "deblock_vert+horz": lambda: ctx.deblock_plane(pred, f0, d_params, W // 4, 0, 3),
            "deblock_fused": lambda: ctx.deblock_plane_fused(pred, f0, dbk, 0, d_params, W // 4, 0),
            "cdef_luma": lambda: ctx.cdef_luma_plane(pred, f0, out, 0, d_pri, d_sec, fbw, d_skip, 6),
        }
    stage_bytes = {"encode_inter_blocks_16x16": 3 * px_bytes + n * (256 * 8 + 2), "inter_pred_8tap": 2 * px_bytes, "subtract_xform_quant_16x16": 2 * px_bytes + n * (256 * 8 + 2),
                   "inv_txfm_add_16x16": n * 256 * 4 + 2 * px_bytes, "deblock_vert+horz": 2 * 2 * px_bytes, "deblock_fused": 2 * px_bytes, "cdef_luma": 2 * px_bytes}
    # The stages are data dependent (the search by the motion, the inverse transform by the share of blocks with coefficients) and the ring's
    # frames differ (profiles/r05_inner_loop_timeline.md: 408 vs 323 us per frame): every stage is timed on every ring slot, each slot prepared
    # by running the chain up to that stage on it, and the mean over the slots is reported (`ms_by_slot` has them all).
    stages = {}
    order = ["fullpel_diamond", "subpel_bilinear", "inter_pred_8tap", "subtract_xform_quant_16x16", "inv_txfm_add_16x16", "encode_inter_blocks_16x16",
             "deblock_vert+horz", "deblock_fused", "cdef_luma"]
    for f0 in range(F):
        fns = stage_fns_of(f0)
        for name in order:
            if name in ("deblock_fused", "encode_inter_blocks_16x16"):   # out of place / idempotent: re-running them leaves the chain's state as it is
                ms = kernel_avg_ms(ctx, fns[name], max(steps, 8))
            else:
                fns[name](); ctx.sync()     # (the chain's state for the next stage; deblock is in place: its re-runs filter an already filtered plane, same work)
                ms = kernel_avg_ms(ctx, fns[name], max(steps, 8))
                if name in ("inv_txfm_add_16x16", "deblock_vert+horz"):   # in-place stages: restore the chain before the next stage is timed
                    for nm in order[2:order.index(name) + 1]:
                        fns[nm]()
                    ctx.sync()
            stages.setdefault(name, {"ms_by_slot": []})["ms_by_slot"].append(ms)
    for name in order:
        stages[name]["ms"] = sum(stages[name]["ms_by_slot"]) / F
    eob_share = []
    for f0 in range(F):
        fns = stage_fns_of(f0)
        for nm in order[:4]:
            fns[nm]()
        ctx.sync()
        eob_share.append(float((ctx.from_device(d_e, (n,), np.uint16) > 0).mean()))
    for name in order:
        ms = stages[name]["ms"]
        if name in stage_bytes:
            # NOT an HBM figure: the whole luma chain of a 4K frame (~100 MB) lives in the 256 MiB Infinity Cache between the
            # dependent stages, so this is the rate at which the stage moves its algorithmic bytes through the cache hierarchy
            stages[name]["cache_resident_GBs"] = stage_bytes[name] / (ms * 1e-3) / 1e9
            stages[name]["cache_resident_rate_over_8TBs"] = stages[name]["cache_resident_GBs"] / HBM_PEAK_GBS
    # Each stage's own roofline: these kernels are bound by instruction issue, not by bytes.  VALU wave-instructions per launch come from the
    # committed PMC passes (profiles/r0N_inner_loop_pmc.json, tools/gpu_pmc_stages.sh: SQ_INSTS_VALU / SQ_WAVES of the same kernel x the
    # launch's wavefronts).  The issue rate is MEASURED in this run (aomhip_valu_issue_probe, csrc/probe.hip; profiles/r05_valu_issue.md):
    # a SIMD of gfx950 retires one wave64 instruction per ~2 clocks for a small "fast" class (v_add/sub_u32, v_mov, v_and/or/xor,
    # v_lshrrev, v_ashrrev, fp32 add / mul / fma) and one per ~4 clocks for every other integer / packed / dot / SAD / DPP / 64-bit opcode
    # the kernels issue; the kernel's class shares are its static opcode mix (profiles/r05_isa_mix.json, tools/isa_mix.py).
    # floor = insts x sum(share_c / rate_c) / (CUs x 4 SIMDs); valu_frac = floor / the launch time measured HERE.
    pmc_map = {"fullpel_diamond": "fullpel_diamond_kernel", "subpel_bilinear": "subpel_bilinear_kernel", "inter_pred_8tap": "inter_pred_kernel",
               "subtract_xform_quant_16x16": "xform_quant_staged_kernel", "inv_txfm_add_16x16": "inv_txfm_add_kernel", "encode_inter_blocks_16x16": "encode_inter_block_kernel",
               "deblock_vert+horz": ("deblock_vert", "deblock_horz"), "deblock_fused": "deblock_fused_kernel", "cdef_luma": "cdef_luma_kernel"}
    pmc = latest_profile_json("_inner_loop_pmc.json")
    rates = valu_class_rates(ctx)
    mix = (latest_profile_json("_isa_mix.json") or {}).get("kernels", {})
    for name, kn in pmc_map.items():
        kns = kn if isinstance(kn, tuple) else (kn,)
        ents = [next(((k_, e) for k_, e in pmc.items() if k_.startswith(x)), None) for x in kns]
        if name not in stages or any(e is None for e in ents):
            continue
        insts = sum(e["SQ_INSTS_VALU_per_wavefront"] * e["wavefronts_per_launch"] for _, e in ents)
        floor_s = sum(e["SQ_INSTS_VALU_per_wavefront"] * e["wavefronts_per_launch"] * valu_seconds_per_inst(mix.get(k_), rates) for k_, e in ents)
        st = stages[name]
        st["valu_wave_insts_per_launch"] = insts
        st["valu_floor_ms"] = floor_s / (rates["compute_units"] * 4) * 1e3
        st["valu_frac"] = st["valu_floor_ms"] / st["ms"] if st["ms"] > 0 else None
        st["valu_fast_share_static"] = [round((mix.get(k_) or {}).get("share", {}).get("fast", 0.0), 3) for k_, _ in ents]
        st["insts_per_wavefront"] = {k_.replace("SQ_INSTS_", "").replace("_per_wavefront", "").lower(): round(sum(e.get(k_, 0.0) for _, e in ents), 1)
                                     for k_ in ("SQ_INSTS_VALU_per_wavefront", "SQ_INSTS_SALU_per_wavefront", "SQ_INSTS_LDS_per_wavefront",
                                                "SQ_INSTS_VMEM_RD_per_wavefront", "SQ_INSTS_VMEM_WR_per_wavefront")}
    # blocks whose quantised coefficients are all zero skip the inverse transform (and cost the forward stage its coefficient writes only)
    for nm in ("inv_txfm_add_16x16", "subtract_xform_quant_16x16"):
        stages[nm]["eob_nonzero_share_by_slot"] = eob_share
    # BASELINE.json configs[4] asks "fps + HBM-roofline fraction": the frame's algorithmic bytes by SURVEY 8(d)'s units -- 16x16 transform blocks at
    # 10 N + 2 B, one deblocked and one CDEF-filtered pixel at 4 B each (the search has no byte unit there) -- over the frame time.  The chain is
    # bound by the search kernels' instruction issue, not by bytes: the fraction says how far from an HBM limit the frame is, nothing more.
    algo_luma = n * (10 * 256 + 2) + 2 * (4 * W * H)
    algo_420 = algo_luma + 2 * (n * (10 * 64 + 2) + 2 * (4 * (W // 2) * (H // 2)))
    roof = {"bound": "issue/latency (search kernels 2/3 of the frame)", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_frame": algo_luma,
            "achieved": algo_luma / (wall / steps) / 1e9, "frac": algo_luma / (wall / steps) / 1e9 / HBM_PEAK_GBS}
    if yuv:
        yuv["algorithmic_bytes_per_frame"] = algo_420
        yuv["roofline_frac"] = algo_420 / (yuv["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {"workload": "encode_inner_loop_4k_10bit", "value": steps / wall, "unit": "frames/s", "roofline": roof, "roofline_frac": roof["frac"], "yuv420": yuv,
            "ms_per_frame": wall / steps * 1e3, "event_ms_per_frame": ev_ms / steps, "blocks_per_frame": n,
            "recon_psnr_db_last_frame": float(psnr), "stages": stages, "valu_issue_rates": rates, "deblock_in_frame": "fused" if fused_deblock else "two_pass",
            "launch": ("one hipGraph per ring of %d frames (aomhip_graph_launch)" % graph_note["frames_per_graph"] if graph_note and "frames_per_graph" in graph_note
                       else "one hipGraph per frame (aomhip_graph_launch)" if graph_note else "eight launches per frame"), "without_graph": graph_note,
            "middle_of_frame": "one kernel (aomhip_encode_inter_blocks_batch)" if fused_middle else "three launches",
            "config": {"frame": "3840x2160 10-bit luma", "stages": "fullpel diamond + subpel bilinear (16x16) -> inter prediction at the "
                       "sub-pel MV (8-tap regular, av1_highbd_convolve_2d_sr) -> subtract+fwd_txfm2d_16x16+quantize_b q100 -> inv_txfm_add -> deblock (8x8 edges, level 32) -> "
                       "CDEF (pri 4, sec 2, damping 6)", "gpus": 1}}


def latest_profile_json(suffix):
    """The newest profiles/r0N*<suffix> (rounds sort by name); {} when there is none."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*" + suffix)))
    try:
        return json.load(open(fs[-1])) if fs else {}
    except Exception:  # noqa: BLE001
        return {}


VALU_CLASS_OPS = {"fast": ("v_add_u32", "v_mov_b32", "v_and_b32", "v_ashrrev_i32"),
                  "slow": ("v_mad_i32_i24", "v_add3_u32", "v_sad_u16", "v_dot2_i32_i16", "v_perm_b32", "v_lshl_add_u64", "v_lshlrev_b32"),
                  "trans": ("v_exp_f32",), "trans64": ("v_rcp_f64",)}
_VALU_RATES = {}


def valu_class_rates(ctx):
    """Wave-instructions per second per SIMD of each issue class, measured on this box in this run (8 wavefronts per SIMD, 8 independent
    chains each, ~4 ms per opcode after a ramp launch of the same kernel): the harmonic mean over the class's probe opcodes."""
    if _VALU_RATES:
        return _VALU_RATES
    import aom_av1_psy_amd as pkg
    names = pkg.capi.valu_issue_probe_names()
    per_op, cus, hz = {}, 256, []
    for cls, ops in VALU_CLASS_OPS.items():
        inv = []
        for op in ops:
            r = ctx.valu_issue_probe(names.index(op), 8, 300)
            iters = max(200, int(4e-3 * r["wave_insts_per_s_per_simd"] / 8 / 128))
            r = ctx.valu_issue_probe(names.index(op), 8, iters)
            per_op[op] = r["wave_insts_per_s_per_simd"]
            inv.append(1.0 / r["wave_insts_per_s_per_simd"])
            cus = r["compute_units"]
            hz.append(r["memtime_hz"])
        _VALU_RATES[cls] = len(inv) / sum(inv)
    _VALU_RATES["per_op"] = per_op
    _VALU_RATES["compute_units"] = cus
    _VALU_RATES["clock_hz_median"] = sorted(hz)[len(hz) // 2]
    _VALU_RATES["clocks_per_wave_inst"] = {c: _VALU_RATES["clock_hz_median"] / _VALU_RATES[c] for c in VALU_CLASS_OPS}
    return _VALU_RATES


def valu_seconds_per_inst(mix_entry, rates):
    """Seconds of one SIMD per wave-instruction of a kernel with this static class mix (no mix known: everything at the 4-clock rate)."""
    share = (mix_entry or {}).get("share") or {"slow": 1.0}
    return sum(v / rates[c] for c, v in share.items())



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


def run_cdef_search(pkg, ctx, orc, steps, warmup):
    """The distortion table of av1_cdef_search (pickcdef.c:401-615) for a 4K 10-bit luma plane, CDEF_FULL_SEARCH (64 strength
    pairs per 64x64 filter block), one launch; also the 16-pair list of CDEF_FAST_SEARCH_LVL1-sized searches.  Informational."""
    W, H, bd, border = 3840, 2160, 10, 64
    recon = pkg.synth.lcg_frame(W, H, 2, 0, bd)
    rng = np.random.default_rng(9)
    source = np.clip(recon.astype(np.int64) + rng.integers(-20, 21, recon.shape), 0, 1023).astype(recon.dtype)
    pr, ps = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pr, 0, recon); ctx.planes_upload(ps, 0, source)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    skip = np.zeros((H // 8, W // 8), np.uint8)
    d_skip = ctx.to_device(skip)
    full = np.array([(gi // 4, (gi % 4) + (gi % 4 == 3)) for gi in range(64)], np.uint8)
    d_st, d_sse = ctx.to_device(full), ctx.malloc(8 * 64 * fbh * fbw)
    out = {"workload": "cdef_search_luma_4k_10bit", "filter_blocks": fbh * fbw}
    for name, n in (("full_search_64", 64), ("fast_search_16", 16)):
        ms = kernel_avg_ms(ctx, lambda n=n: ctx.cdef_search_sse_luma(pr, 0, ps, 0, d_st, n, d_skip, 5, fbw, d_sse), max(steps, 4))
        out[name] = {"ms_per_frame": ms, "strength_evaluations_per_s": fbh * fbw * n / ms * 1e3,
                     "filtered_pixels_per_s": float(W) * H * n / ms * 1e3}
    # exact check of one filter-block row against the oracle (4 strengths)
    sub = slice(0, 64)
    want = orc.cdef_search_sse_luma(recon[sub, :256], source[sub, :256], [tuple(int(v) for v in full[i]) for i in (0, 5, 30, 63)], skip[:8, :32], 5, bd)
    p2, s2 = ctx.planes_alloc(256, 64, border, bd, 1), ctx.planes_alloc(256, 64, border, bd, 1)
    ctx.planes_upload(p2, 0, np.ascontiguousarray(recon[sub, :256])); ctx.planes_upload(s2, 0, np.ascontiguousarray(source[sub, :256]))
    d_s4, d_o4, d_k4 = ctx.to_device(np.ascontiguousarray(full[[0, 5, 30, 63]])), ctx.malloc(8 * 4 * 4), ctx.to_device(np.zeros((8, 32), np.uint8))
    ctx.cdef_search_sse_luma(p2, 0, s2, 0, d_s4, 4, d_k4, 5, 4, d_o4)
    out["parity_sample"] = bool(np.array_equal(ctx.from_device(d_o4, (4, 1, 4), np.uint64), want))
    out["value"], out["unit"] = out["full_search_64"]["strength_evaluations_per_s"], "filter-block strength evaluations/s"
    for d in (d_skip, d_st, d_sse, d_s4, d_o4, d_k4):
        ctx.free(d)
    for p in (pr, ps, p2, s2):
        ctx.planes_free(p)
    return out


def run_wiener_stats(pkg, ctx, orc, steps, warmup):
    """av1_compute_stats for every restoration unit of a 4K luma plane (7x7 window): 8-bit with 64x64 and 256x256 units, and
    10-bit 64x64.  Informational; 1274 multiply-adds per pixel (1225 H entries + 49 M entries)."""
    import ctypes as C
    W, H, border = 3840, 2160, 16
    out = {"workload": "wiener_stats_luma_4k"}
    for name, bd, unit in (("8bit_units64", 8, 64), ("8bit_units256", 8, 256), ("10bit_units64", 10, 64)):
        dgd = pkg.synth.lcg_frame(W, H, 3, 0, bd)
        src = pkg.synth.lcg_frame(W, H, 3, 1, bd)
        pd, ps = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(pd, 0, dgd); ctx.planes_upload(ps, 0, src)
        rects = [(x, min(x + unit, W), y, min(y + unit, H)) for y in range(0, H, unit) for x in range(0, W, unit)]
        units = np.zeros(len(rects), pkg.capi.rect_dtype)
        for i, r in enumerate(rects):
            units[i] = r
        d_u, d_M, d_H = ctx.to_device(units), ctx.malloc(8 * 49 * len(rects)), ctx.malloc(8 * 2401 * len(rects))
        ms = kernel_avg_ms(ctx, lambda: ctx.compute_stats_batch(pd, 0, ps, 0, 7, d_u, None, len(rects), 0, d_M, d_H), max(steps, 3))
        out[name] = {"ms_per_frame": ms, "units": len(rects), "mac_per_s": float(W) * H * 1274 / ms * 1e3}
        if name == "8bit_units64":          # exact check of two units against the oracle
            Hm = ctx.from_device(d_H, (len(rects), 2401), np.int64)
            db, sb = orc.extend_plane(dgd, border), orc.extend_plane(src, border)
            ok = True
            f = orc.lib.orc_compute_stats
            f.restype = None
            for i in (0, len(rects) - 1):
                wm, wh = np.zeros(49, np.int64), np.zeros(2401, np.int64)
                hs, he, vs, ve = rects[i]
                f(7, C.c_void_p(orc._addr(db, border, border)), C.c_void_p(orc._addr(sb, border, border)), hs, he, vs, ve, db.shape[1], sb.shape[1], 0, 8, 0,
                  C.c_void_p(wm.ctypes.data), C.c_void_p(wh.ctypes.data))
                ok = ok and bool(np.array_equal(Hm[i], wh))
            out["parity_sample"] = ok
        for d in (d_u, d_M, d_H):
            ctx.free(d)
        ctx.planes_free(pd); ctx.planes_free(ps)
    out["value"], out["unit"] = out["8bit_units64"]["mac_per_s"], "window multiply-adds/s"
    return out


def run_warp_error(pkg, ctx, orc, steps, warmup):
    """The global-motion search's inner loop (av1_warp_error, av1/encoder/global_motion.c:128-224) on a 4K luma plane: 14 candidate models per call
    (the +step / -step pair of one parameter for 7 references' worth of candidates), every 32 x 32 tile active, 10 and 8 bits; and the baseline
    av1_segmented_frame_error.  Informational.  Algorithmic bytes per model: the reference and the current frame once each."""
    import ctypes as C
    capi = pkg.capi
    W, H, border, n_models = 3840, 2160, 32, 14
    out = {"workload": "global_motion_warp_error_luma_4k", "models_per_call": n_models}
    rng = np.random.default_rng(5)
    for name, bd in (("10bit", 10), ("8bit", 8)):
        ref = pkg.synth.lcg_frame(W, H, 3, 0, bd)
        cur = pkg.synth.lcg_frame(W, H, 3, 1, bd)
        pr, pc = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(pr, 0, ref); ctx.planes_upload(pc, 0, cur)
        models = np.zeros(n_models, capi.warp_model_dtype)
        for i in range(n_models):
            while True:
                models["mat"][i] = [rng.integers(-8 << 16, 8 << 16), rng.integers(-8 << 16, 8 << 16), (1 << 16) + rng.integers(-(1 << 10), 1 << 10),
                                    rng.integers(-(1 << 10), 1 << 10), rng.integers(-(1 << 10), 1 << 10), (1 << 16) + rng.integers(-(1 << 10), 1 << 10)]
                if capi.get_shear_params(models[i:i + 1])[0]:
                    break
        sw, sh = (W + 31) // 32, (H + 31) // 32
        seg = np.ones((sh, sw), np.uint8)
        d_m, d_s, d_e = ctx.to_device(models), ctx.to_device(seg), ctx.malloc(8 * n_models)
        once = lambda: ctx.warp_error_batch(pr, 0, pc, 0, 0, 0, d_m, n_models, 0, 0, W, H, d_s, sw, d_e)
        for _ in range(warmup):
            once()
        ms = kernel_avg_ms(ctx, once, max(steps, 3))
        es = 2 if bd > 8 else 1
        out[name] = {"ms_per_call": ms, "ms_per_model": ms / n_models, "model_pixels_per_s": float(W) * H * n_models / ms * 1e3,
                     "algorithmic_GBps": 2.0 * W * H * es * n_models / ms / 1e6}
        ms_f = kernel_avg_ms(ctx, lambda: ctx.segmented_frame_error(pr, 0, pc, 0, W, H, d_s, sw, d_e), max(steps, 3))
        out[name]["segmented_frame_error_ms"] = ms_f
        if name == "10bit":   # exact check of one model over the whole frame against the oracle
            once()
            got = ctx.from_device(d_e, (n_models,), np.int64)
            f = orc.lib.orc_warp_error
            f.restype = C.c_int64
            m = np.ascontiguousarray(models["mat"][3], np.int32)
            sh4 = np.array([models[k][3] for k in ("alpha", "beta", "gamma", "delta")], np.int16)
            rc, cc = np.ascontiguousarray(ref), np.ascontiguousarray(cur)
            t0 = time.perf_counter()
            want = f(C.c_void_p(m.ctypes.data), C.c_void_p(sh4.ctypes.data), C.c_void_p(rc.ctypes.data), 1, W, H, W, C.c_void_p(cc.ctypes.data), 0, 0, W, H, W, 0, 0, bd,
                     C.c_int64((1 << 63) - 1), C.c_void_p(seg.ctypes.data), sw)
            out["cpu_port_ms_per_model"] = (time.perf_counter() - t0) * 1e3     # the C restatement, one host core, the same frame
            out["parity_sample"] = bool(int(got[3]) == int(want))
        for d in (d_m, d_s, d_e):
            ctx.free(d)
        ctx.planes_free(pr); ctx.planes_free(pc)
    out["value"], out["unit"] = out["10bit"]["model_pixels_per_s"], "model pixels/s"
    return out


def run_int_pro(pkg, ctx, orc, steps, warmup):
    """av1_int_pro_motion_estimation (av1/encoder/mcomp.c:1897-2105) for every block of a 4K 8-bit luma plane: the 64 x 64 superblocks (the vector
    variance partitioning starts from) and all 16 x 16 / 32 x 32 blocks.  Informational."""
    import ctypes as C
    capi = pkg.capi
    W, H, border = 3840, 2160, 160
    base, _ = pkg.synth.shifted_smooth_pair(W + 64, H + 64, 0, 8)
    rng = np.random.default_rng(3)
    src = np.clip(base[32:32 + H, 32:32 + W].astype(np.int32) + rng.integers(-3, 4, (H, W)), 0, 255).astype(np.uint8)
    ref = np.clip(base[29:29 + H, 37:37 + W].astype(np.int32) + rng.integers(-3, 4, (H, W)), 0, 255).astype(np.uint8)
    ps, pr = ctx.planes_alloc(W, H, border, 8, 1), ctx.planes_alloc(W, H, border, 8, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    out = {"workload": "int_pro_motion_estimation_luma_4k_8bit"}
    for bs in (64, 32, 16):
        pos = [(x, y) for y in range(0, H - bs + 1, bs) for x in range(0, W - bs + 1, bs)]
        blocks = np.zeros(len(pos), capi.search_block_dtype)
        blocks["bx"], blocks["by"] = [p[0] for p in pos], [p[1] for p in pos]
        blocks["row_min"], blocks["row_max"], blocks["col_min"], blocks["col_max"] = -1023, 1023, -1023, 1023
        n = len(pos)
        d_b, d_mv, d_sad = ctx.to_device(blocks), ctx.malloc(4 * n), ctx.malloc(4 * n)
        once = lambda: ctx.int_pro_motion_estimation_batch(ps, 0, pr, 0, bs, bs, d_b, n, d_mv, d_sad)
        for _ in range(warmup):
            once()
        ms = kernel_avg_ms(ctx, once, max(steps, 3))
        out["%dx%d" % (bs, bs)] = {"ms_per_frame": ms, "blocks": n, "blocks_per_s": n / ms * 1e3}
        if bs == 64:   # a sample of blocks against the oracle, and the CPU restatement's rate on them
            mv, sad = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_sad, (n,), np.uint32)
            sb, rb = np.pad(src, border, mode="edge"), np.pad(ref, border, mode="edge")
            f = orc.lib.orc_int_pro_motion_estimation
            f.restype = C.c_uint
            lim, rm, o = np.array([-1023, 1023, -1023, 1023], np.int32), np.zeros(2, np.int16), np.zeros(2, np.int16)
            ok, t0, sample = True, time.perf_counter(), range(0, n, 17)
            for i in sample:
                off = (border + pos[i][1]) * sb.shape[1] + border + pos[i][0]
                w = f(C.c_void_p(sb.ctypes.data + off), sb.shape[1], C.c_void_p(rb.ctypes.data + off), rb.shape[1], bs, bs, 8, C.c_void_p(lim.ctypes.data),
                      C.c_void_p(rm.ctypes.data), C.c_void_p(o.ctypes.data))
                ok = ok and int(w) == int(sad[i]) and o.tolist() == mv[i].tolist()
            out["cpu_port_blocks_per_s_64x64"] = len(sample) / (time.perf_counter() - t0)
            out["parity_sample"] = bool(ok)
            out["vectors_found"] = int(len({tuple(v) for v in mv.tolist()}))
        for d in (d_b, d_mv, d_sad):
            ctx.free(d)
    # the variance tree's leaves on the same pair of planes (what follows the vector in av1_choose_var_based_partitioning)
    n8x, n8y = W // 8, (H + 7) // 8
    d_s8, d_mm, d_s4 = ctx.malloc(2 * n8x * n8y), ctx.malloc(4 * (W // 16) * ((H + 15) // 16)), ctx.malloc(2 * (W // 4) * (H // 4))
    ms8 = kernel_avg_ms(ctx, lambda: ctx.vbp_8x8_stats_plane(ps, 0, pr, 0, W, H, d_s8, n8x, d_mm, W // 16), max(steps, 3))
    ms4 = kernel_avg_ms(ctx, lambda: ctx.vbp_4x4_avg_plane(ps, 0, W, H, 0, d_s4, W // 4), max(steps, 3))
    out["vbp_leaves"] = {"ms_8x8_stats": ms8, "GBps_8x8_stats": 2.0 * W * H / ms8 / 1e6, "ms_4x4_avg": ms4, "GBps_4x4_avg": 1.0 * W * H / ms4 / 1e6}
    for d in (d_s8, d_mm, d_s4):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
    out["value"], out["unit"] = out["64x64"]["blocks_per_s"], "64x64 blocks/s"
    return out


def run_tf(pkg, ctx, orc, steps, warmup, width=3840, height=2160, bd=10, n_frames=5):
    """the temporal filter's motion search (tf_motion_search, temporal_filter.c:87-253) for every 32x32 block of a
    4K 10-bit frame against the 4 other frames of a 5-frame window, one aomhip_tf_motion_search_frames call per filtered frame: per
    reference frame the 32x32 NSTEP + mesh full-pel search, the 8-tap sub-pel tree, the same pair for the four 16x16 sub-blocks, the
    partition decision and the ref_mv hand-over, all in device memory."""
    capi, synth = pkg.capi, pkg.synth
    border, filt = 160, n_frames // 2
    mesh = [(64, 8), (28, 4), (15, 1), (7, 1)]   # good_quality_mesh_patterns[0] (speed_features.c:25-33)
    planes = ctx.planes_alloc(width, height, border, bd, n_frames)
    base, _ = synth.shifted_smooth_pair(width + 64, height + 64, 0, bd)
    rng = np.random.default_rng(11)
    host = []
    for f in range(n_frames):
        d = f - filt
        img = base[32 + d:32 + d + height, 32 - 2 * d:32 - 2 * d + width].astype(np.int32) + rng.integers(-3, 4, (height, width))
        host.append(np.clip(img, 0, (1 << bd) - 1).astype(np.uint16 if bd > 8 else np.uint8))
        ctx.planes_upload(planes, f, host[-1])
    blocks = capi.tf_block_list(width, height, border)
    n = len(blocks)
    d_b = ctx.to_device(blocks)
    d_mv, d_mse, d_ref = ctx.malloc(n_frames * n * 16), ctx.malloc(n_frames * n * 16), ctx.malloc(n * 4)
    out = {}
    for name, q in (("q30_mesh_pruned_when_close", 30), ("q12_mesh_always", 12)):
        tp = capi.TfParams.default(width, height, bd, q, 1, mesh)
        once = lambda: ctx.tf_motion_search_frames(planes, filt, tp, d_b, n, d_mv, d_mse, d_ref)
        for _ in range(warmup):
            once()
        out[name] = {"ms_per_filtered_frame": kernel_avg_ms(ctx, once, max(3, steps // 4))}
    # what follows the search in av1_tf_do_filtering_row, on the MVs / errors still in HBM: predictors (12-tap), pixel weights, accumulation and
    # normalisation of the whole frame in one launch (aomhip_tf_apply_frames) -- luma + 4:2:0 chroma planes of the same window
    cw, ch = (width + 1) >> 1, (height + 1) >> 1
    chroma = [ctx.planes_alloc(cw, ch, border, bd, n_frames) for _ in range(2)]
    for f in range(n_frames):
        for c in chroma:
            ctx.planes_upload(c, f, host[f][::2, ::2][:ch, :cw])
    outs = [ctx.planes_alloc(width, height, border, bd, 1)] + [ctx.planes_alloc(cw, ch, border, bd, 1) for _ in range(2)]
    ap3 = capi.TfApplyParams.make([2.0, 1.5, 1.5], 30, 5, 3, 1, 1)
    ap1 = capi.TfApplyParams.make([2.0, 0, 0], 30, 5, 1, 0, 0)
    d_diff = ctx.malloc(16)
    apply3 = lambda: ctx.tf_apply_frames([planes] + chroma, filt, ap3, n, d_mv, d_mse, outs, 0, d_diff=d_diff)
    apply1 = lambda: ctx.tf_apply_frames([planes], filt, ap1, n, d_mv, d_mse, outs[:1], 0)
    vis = width * height * (2 if bd > 8 else 1)
    for nm, fn, planes_n in (("apply_yuv420", apply3, 1.5), ("apply_luma", apply1, 1.0)):
        ms_a = kernel_avg_ms(ctx, fn, max(3, steps // 4))
        moved = vis * planes_n * (n_frames + 1)   # every window frame read once + the filtered frame written
        out[nm] = {"ms_per_filtered_frame": ms_a, "GBs_window_plus_output": moved / (ms_a * 1e-3) / 1e9, "frac_of_8TBs": moved / (ms_a * 1e-3) / 1e9 / HBM_PEAK_GBS}
    apply_ok = None
    if orc is not None:   # the luma launch against the oracle on every 61st block (blocks are independent)
        apply1(); ctx.sync()
        mvs_a = ctx.from_device(d_mv, (n_frames, n, 4, 2), np.int16)
        mses_a = ctx.from_device(d_mse, (n_frames, n, 4), np.int32)
        mb_cols = (width + 31) // 32
        fb = [orc.extend_plane(h, border, planes.stride) for h in host]
        want = orc.tf_apply_frames([fb], border, width, height, filt, mvs_a, mses_a, [2.0, 0, 0], 30, 5, bd=bd, block_first=0, block_step=61)[0]
        got = ctx.planes_download(outs[0], 0)
        okb = []
        for bi in range(0, n, 61):
            r0, c0 = border + 32 * (bi // mb_cols), border + 32 * (bi % mb_cols)
            okb.append(np.array_equal(got[r0:r0 + 32, c0:c0 + 32], want[r0:r0 + 32, c0:c0 + 32]))
        apply_ok = {"blocks_checked": len(okb), "identical": bool(all(okb))}
    out["apply_parity_sample"] = apply_ok
    for pl in chroma + outs:
        ctx.planes_free(pl)
    ctx.free(d_diff)
    ok = None
    if orc is not None:   # the last call (q 12) against the oracle on every 97th block (blocks are independent)
        mvs = ctx.from_device(d_mv, (n_frames, n, 4, 2), np.int16)
        mses = ctx.from_device(d_mse, (n_frames, n, 4), np.int32)
        idx = np.arange(0, n, 97)
        fb = [orc.extend_plane(h, border, planes.stride) for h in host]
        wmv, wmse, _ = orc.tf_motion_search_frames(fb, filt, border, orc.tf_block_list(width, height, border)[idx], orc.tf_params(width, height, bd, 12, 1, mesh),
                                                   threads=8)
        ok = bool(np.array_equal(mvs[:, idx], wmv) and np.array_equal(mses[:, idx], wmse))
    for d in (d_b, d_mv, d_mse, d_ref):
        ctx.free(d)
    ctx.planes_free(planes)
    ms = out["q30_mesh_pruned_when_close"]["ms_per_filtered_frame"]
    return dict(out, workload="tf_motion_search_4k_10bit", value=n * (n_frames - 1) / (ms * 1e-3), unit="block searches/s (32x32 block x reference frame)",
                blocks_per_frame=n, reference_frames=n_frames - 1, parity_sample=ok,
                config={"frame": "%dx%d %d-bit" % (width, height, bd), "window": n_frames, "search": "NSTEP + mesh (run_mesh_search 1, prune LVL_1), L1_HDRES; "
                        "av1_find_best_sub_pixel_tree USE_8_TAPS; 32x32 + four 16x16 per block and frame"})


def run_compound_search(pkg, ctx, orc, steps, warmup, width=3840, height=2160, bd=10, bs=16):
    """SURVEY 8(f) row 1, the RD path's compound searches of handle_newmv on every 16x16 block of a 4K 10-bit frame against two references:
    av1_joint_motion_search on both branches (8-neighbour refinement: speed >= 1; av1_full_pixel_search on the compound prediction with the second
    sub-pel start: speed 0), av1_compound_single_motion_search_interinter (masked), and the OBMC pair (av1_obmc_full_pixel_search +
    av1_find_best_obmc_sub_pixel_tree_up).  One call per frame each; ms per frame.  A sample of blocks is checked against the oracle."""
    capi, synth = pkg.capi, pkg.synth
    border = 160
    src, ref0 = synth.shifted_smooth_pair(width, height, 61, bd, shift=(2, -3), frac8=(3, 0))
    _, ref1 = synth.shifted_smooth_pair(width, height, 61, bd, shift=(-3, 2), frac8=(0, 5))
    ps, p0, p1 = (ctx.planes_alloc(width, height, border, bd, 1) for _ in range(3))
    for p_, a in ((ps, src), (p0, ref0), (p1, ref1)):
        ctx.planes_upload(p_, 0, a)
    gc, gr = width // bs, height // bs
    n = gc * gr
    rng = np.random.default_rng(5)
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    ext = border - 8 - 16
    blocks["col_min"], blocks["col_max"] = np.maximum(-(blocks["bx"] + ext), -1000), np.minimum(width - blocks["bx"] - bs + ext, 1000)
    blocks["row_min"], blocks["row_max"] = np.maximum(-(blocks["by"] + ext), -1000), np.minimum(height - blocks["by"] - bs + ext, 1000)
    ref_mv = rng.integers(-24, 25, (n, 2, 2)).astype(np.int16)
    cur = np.zeros((n, 2, 2), np.int16)
    cur[:, 0] = np.array([-3 * 8, 2 * 8]) + rng.integers(-20, 21, (n, 2))      # the single-reference results: a few pixels off the true motion
    cur[:, 1] = np.array([2 * 8, -3 * 8]) + rng.integers(-20, 21, (n, 2))
    mask = np.clip((np.arange(bs)[None, None, :] * 64 // bs + rng.integers(-6, 7, (n, bs, bs))), 0, 64).astype(np.uint8)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    tabs = (d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    d_b, d_r, d_m = ctx.to_device(blocks), ctx.to_device(ref_mv), ctx.to_device(mask)
    d_cur = ctx.malloc(n * 8)
    d_rate, d_err = ctx.malloc(n * 4), ctx.malloc(n * 4)
    sub8 = capi.SubpelParams(2, 0, 61, 2, 1, 0, 3)      # SUBPEL_TREE, USE_8_TAPS (speed 0)
    sub4 = capi.SubpelParams(2, 0, 61, 2, 1, 0, 2)      # SUBPEL_TREE, USE_4_TAPS (speed 1 - 2)
    full = capi.SearchParams.make("NSTEP", 5, 0, 22, 61, mesh_diff_thr=4, mesh=[(64, 8), (28, 4), (15, 1), (7, 1)])
    # every call starts from the single-reference results again: a 261 KB host copy on the stream, inside the timed region (~1 % of the shortest call)
    reset = lambda: ctx.memcpy_h2d(d_cur, cur)
    out = {}

    def timed(name, fn, note):
        def once():
            reset()
            fn()
        for _ in range(max(1, warmup)):
            once()
        ms = kernel_avg_ms(ctx, once, max(3, steps // 2))
        out[name] = {"ms_per_frame": ms, "blocks_per_s": n / (ms * 1e-3), "what": note}
    timed("joint_refining_4tap", lambda: ctx.joint_motion_search_batch(ps, p0, p1, 0, bs, bs, 0, 22, sub4, 0, d_b, d_r, d_cur, None, n, d_rate, d_err, *tabs),
          "av1_joint_motion_search, disable_extensive_joint_motion_search (speed >= 1): 4 iterations of {predictor, av1_refining_search_8p_c, compound sub-pel tree USE_4_TAPS}")
    timed("joint_extensive_8tap", lambda: ctx.joint_motion_search_extensive_batch(ps, p0, p1, 0, bs, bs, full, sub8, 1, 0, d_b, d_r, d_cur, None, n, d_rate, d_err, *tabs),
          "av1_joint_motion_search, speed 0: 4 iterations of {predictor, av1_full_pixel_search(.., 5, ..) on the compound, compound sub-pel tree USE_8_TAPS twice (second MV)}")
    want = None
    if orc is not None:   # the extensive call against the oracle's composition on every 211th block
        got_mv = ctx.from_device(d_cur, (n, 2, 2), np.int16)
        got_rate, got_err = ctx.from_device(d_rate, (n,), np.int32), ctx.from_device(d_err, (n,), np.int32)
        idx = np.arange(0, n, 211)
        sb, r0b, r1b = (orc.extend_plane(a, border, ps.stride) for a in (src, ref0, ref1))
        oq = orc.search_params("NSTEP", 5, 0, 22, 61, 0, 0, 0, 4, 2147483647, 0, [(64, 8), (28, 4), (15, 1), (7, 1)], no_cost_list=1)
        w_mv, w_rate, w_err, _ = orc.joint_motion_search_batch(sb, r0b, r1b, border, width, height, bs, bs, blocks[idx], ref_mv[idx], cur[idx], None, cost_type=0,
                                                               sad_per_bit=22, sub=dict(tree=2, subpel_search_type=3, error_per_bit=61, iters_per_step=2, allow_hp=1),
                                                               mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8, full=oq, allow_second_mv=1)
        want = bool(np.array_equal(got_mv[idx], w_mv) and np.array_equal(got_rate[idx], w_rate) and np.array_equal(got_err[idx], w_err))
    d_this, d_other = ctx.to_device(np.ascontiguousarray(cur[:, 0])), ctx.to_device(np.ascontiguousarray(cur[:, 1]))
    d_this_w, d_ref0 = ctx.malloc(n * 4), ctx.to_device(np.ascontiguousarray(ref_mv[:, 0]))
    this0 = np.ascontiguousarray(cur[:, 0])
    reset = lambda: ctx.memcpy_h2d(d_this_w, this0)
    timed("compound_single_masked_4tap", lambda: ctx.compound_single_motion_search_batch(ps, p0, p1, 0, bs, bs, full, sub4, 0, d_b, d_ref0, d_this_w, d_other, 0, 0, None, d_m, 0,
                                                                                         n, d_rate, d_err, *tabs),
          "av1_compound_single_motion_search_interinter with a mask: predictor of the other side, av1_full_pixel_search(.., 5, ..) on the masked compound, sub-pel tree USE_4_TAPS")
    # OBMC: weighted source / mask of calc_target_weighted_pred (synthetic: top / left neighbours overlap half a block)
    om = np.full((bs, bs), 4096, np.int64)
    om[:bs // 2, :] = (np.linspace(36, 64, bs // 2).astype(np.int64)[:, None]) * 64
    om[:, :bs // 2] = np.minimum(om[:, :bs // 2], (np.linspace(34, 64, bs // 2).astype(np.int64)[None, :]) * 64)
    sblk = src[:gr * bs, :gc * bs].reshape(gr, bs, gc, bs).transpose(0, 2, 1, 3).reshape(n, bs, bs).astype(np.int64)
    nb = np.clip(sblk + rng.integers(-(10 << (bd - 8)), (10 << (bd - 8)) + 1, sblk.shape), 0, (1 << bd) - 1)
    ws = (sblk * 4096 - nb * (4096 - om[None])).astype(np.int32)
    d_ws, d_om = ctx.to_device(ws), ctx.to_device(np.broadcast_to(om.astype(np.int32), (n, bs, bs)).copy())
    ob = blocks.copy()
    ob["ref_row"], ob["ref_col"] = ref_mv[:, 0, 0], ref_mv[:, 0, 1]
    ob["start_row"], ob["start_col"] = cur[:, 0, 0] >> 3, cur[:, 0, 1] >> 3
    ob["row_min"], ob["row_max"] = np.maximum(ob["row_min"], -64), np.minimum(ob["row_max"], 64)
    ob["col_min"], ob["col_max"] = np.maximum(ob["col_min"], -64), np.minimum(ob["col_max"], 64)
    sbl = ob.copy()
    for k_ in ("start_row", "start_col", "row_min", "row_max", "col_min", "col_max"):
        sbl[k_] = ob[k_] * 8
    d_ob, d_sbl = ctx.to_device(ob), ctx.to_device(sbl)
    d_mv, d_dist, d_sse = ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 4)
    reset = lambda: None
    timed("obmc_full_pixel_nstep", lambda: ctx.obmc_full_pixel_search_batch(p0, 0, bs, bs, "NSTEP", 4, 0, 0, 22, 61, d_ob, n, d_ws, d_om, d_mv, d_err, *tabs),
          "av1_obmc_full_pixel_search: obmc_full_pixel_diamond, NSTEP from step_param 4")
    timed("obmc_subpel_tree_4tap", lambda: ctx.obmc_subpel_tree_batch(p0, 0, bs, bs, sub4, d_sbl, n, d_ws, d_om, d_mv, d_err, d_dist, d_sse, *tabs),
          "av1_find_best_obmc_sub_pixel_tree_up, USE_4_TAPS, from the full-pel start")
    for d in (d_j, d_c0, d_c1, d_b, d_r, d_m, d_cur, d_rate, d_err, d_this, d_other, d_this_w, d_ref0, d_ws, d_om, d_ob, d_sbl, d_mv, d_dist, d_sse):
        ctx.free(d)
    for p_ in (ps, p0, p1):
        ctx.planes_free(p_)
    ms = out["joint_refining_4tap"]["ms_per_frame"]
    return dict(out, workload="compound_search_4k_10bit", value=n / (ms * 1e-3), unit="compound blocks/s (av1_joint_motion_search, refining branch)", ms_per_frame=ms,
                blocks_per_frame=n, parity_sample_extensive=want)


def run_sad_diamond_lists(pkg, ctx, orc, steps, warmup, width=3840, height=2160, bd=8, frames=16):
    """lists that are NOT Mode-A shaped through aomhip_sad_sb_batch -- one diamond step per 16x16 block as the
    encoder issues it (mcomp.c:1299-1416): 8 sites = two x4d groups at (+-r, 0), (0, +-r), (+-r, +-r) around a per-block centre within
    +-40 of the block, r in {1, 2, 4, 8, 16}, no single candidates.  These take the kernel's general per-entry path (source rows re-read
    per group, no fused group + candidate block), still out of the LDS ring; compared with the direct x4d kernel on the same lists."""
    capi, synth = pkg.capi, pkg.synth
    border = 160
    src, ref = ctx.planes_alloc(width, height, border, bd, frames), ctx.planes_alloc(width, height, border, bd, frames)
    for f in range(frames):
        ctx.planes_upload(src, f, synth.lcg_frame(width, height, 2 * f, 0, bd))
        ctx.planes_upload(ref, f, synth.lcg_frame(width, height, 2 * f + 1, 0, bd))
    _, g0 = synth.mode_a_worklist(width, height, 16, seed=3)
    nb = len(g0)
    rng = np.random.default_rng(9)
    cx = g0["sx"].astype(np.int32) + rng.integers(-40, 41, nb)
    cy = g0["sy"].astype(np.int32) + rng.integers(-40, 41, nb)
    r = (1 << rng.integers(0, 5, nb)).astype(np.int32)
    groups = np.zeros(2 * nb, capi.sad_x4d_dtype)
    groups["sx"] = np.repeat(g0["sx"], 2); groups["sy"] = np.repeat(g0["sy"], 2)
    dr = np.array([[-1, 1, 0, 0], [-1, 1, -1, 1]]); dc = np.array([[0, 0, -1, 1], [-1, 1, 1, -1]])   # site order of mcomp.c:366-370
    for k in range(2):
        groups["ry"][k::2] = cy[:, None] + dr[k][None, :] * r[:, None]
        groups["rx"][k::2] = cx[:, None] + dc[k][None, :] * r[:, None]
    cell = (384, 32) if bd == 8 else (160, 32)
    perm, off = synth.bucket_order(groups["sx"], groups["sy"], width, height, *cell)
    gs = np.ascontiguousarray(groups[perm])
    d_gs, d_off, d_g = ctx.to_device(gs), ctx.to_device(off), ctx.to_device(groups)
    n = len(groups)
    d_o_sb, d_o_dir = ctx.malloc(frames * n * 16), ctx.malloc(frames * n * 16)
    sb = lambda: ctx.sad_sb_batch(src, ref, 0, frames, 16, 16, 0, cell[0], cell[1], 64, len(off) - 1, d_gs, d_off, n, 0, d_o_sb)
    direct = lambda: ctx.sad_x4d_batch(src, ref, 0, frames, 16, 16, 0, d_g, n, 0, d_o_dir)
    for _ in range(warmup):
        sb(); direct()
    ms_sb, ms_dir = kernel_avg_ms(ctx, sb, max(5, steps // 2)), kernel_avg_ms(ctx, direct, max(5, steps // 2))
    a = ctx.from_device(d_o_sb, (frames, n, 4), np.uint32)
    b = ctx.from_device(d_o_dir, (frames, n, 4), np.uint32)
    same = bool(np.array_equal(a, b[:, perm]))
    ok = None
    if orc is not None:
        s0, r0 = synth.lcg_frame(width, height, 0, 0, bd), synth.lcg_frame(width, height, 1, 0, bd)
        idx = np.arange(0, n, 53)
        want = orc.sad_x4d_batch(orc.extend_plane(s0, border, src.stride), orc.extend_plane(r0, border, ref.stride), border, 16, 16, groups[idx], bd=bd, threads=8)
        ok = bool(np.array_equal(b[0][idx], want))
    for d in (d_gs, d_off, d_g, d_o_sb, d_o_dir):
        ctx.free(d)
    ctx.planes_free(src); ctx.planes_free(ref)
    cands = 4 * n * frames
    return {"workload": "sad16x16_diamond_step_lists_4k_%dbit" % bd, "value": cands / (ms_sb * 1e-3), "unit": "candidates/s",
            "sad_strip_kernel_ms": ms_sb, "sad_x4d_kernel_ms": ms_dir, "direct_candidates_per_s": cands / (ms_dir * 1e-3),
            "strip_equals_direct": same, "parity_sample_frame0": ok, "candidates_per_launch": cands,
            "config": {"frame": "%dx%d %d-bit x %d pairs" % (width, height, bd, frames), "list": "8 diamond sites (2 x4d groups) per 16x16 block, "
                       "centre within +-40, radius 1..16; no single candidates (not Mode-A shaped)", "cell": list(cell)}}


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


TRAFFIC_SOURCES = {"sb": ("sad_sb.hip",), "sad": ("sad.hip",), "txq": ("xform_quant.hip", "txfm_device.h", "quant_device.h")}


def traffic_kind(name):
    return "sb" if name.endswith(":sb") else "txq" if name.startswith("txq") else "sad"


def kernel_source_sha(kind):
    """sha256[:16] of the kernel source a traffic figure describes (tools/pmc_traffic*.py store it beside the figure)."""
    import hashlib
    h = hashlib.sha256()
    for f in TRAFFIC_SOURCES[kind]:
        with open(os.path.join(ROOT, "aom-av1-psy_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_traffic(name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json,
    produced by tools/pmc_traffic*.py from separate rocprofv3 --pmc runs); None when not measured OR when the
    figure was measured on another version of the kernel source than the one in this tree (a stale counter is not evidence)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(p))
        if (t.get("_measured_on") or {}).get(name) != kernel_source_sha(traffic_kind(name)):
            return None
        return t.get(name)
    except Exception:
        return None


def run_workload(pkg, ctx, dist, dev, rank, world, name, steps, warmup, want_cpu, orc):
    wl = SadModeA(pkg, ctx, name, rank, world, frames_per_rank=FRAMES_OVERRIDE or None)
    wl.step()
    ctx.sync()
    ok = wl.check_frame0(orc) if orc is not None else None
    wall, ev_ms = time_steps(wl, ctx, dist, dev, steps, warmup)
    total = wl.cands_per_step
    if dist is not None:
        import torch
        t = torch.tensor([total], dtype=torch.float64, device=_red_device())
        dist.all_reduce(t)
        total = int(t.item())
    kx_ms = kernel_avg_ms(ctx, wl.launch_x4d, max(steps, 10))
    k1_ms = kernel_avg_ms(ctx, wl.launch_single, max(steps, 10))
    if wl.path == "sb":  # dominant (only) kernel of the step: all five candidates of every block in one launch
        k_ms = kernel_avg_ms(ctx, wl.launch_sb, max(steps, 10))
        x4d_bytes = 5 * wl.blocks_per_frame * wl.ring * wl.bytes_per_cand()
        kname, traffic = "sad_strip_kernel<16x16>", load_traffic(name + ":sb")
    else:
        k_ms = kx_ms
        x4d_bytes = 4 * wl.blocks_per_frame * wl.ring * wl.bytes_per_cand()
        kname, traffic = "sad_x4d_kernel<16x16>", load_traffic(name)
    # Roofline of the dominant kernel, as an HBM figure: COMPULSORY bytes = every visible source and reference byte of
    # the ring once + the work-list entries read + the results written (a launch cannot move less), over the launch
    # time, against the 8 TB/s spec peak.  The SURVEY 8(d) per-candidate figure (516 / 1028 B) counts overlapping
    # reference bytes once per candidate -- they are served by LDS, so that rate (`achieved_algorithmic`) is not an
    # HBM rate and is never divided by the HBM peak.  `traffic` = fabric bytes per launch from the PMC passes.
    cfg = wl.cfg
    es = 1 if cfg["bit_depth"] == 8 else 2
    col_px = wl.tile[1] - wl.tile[0]
    n_blk = wl.blocks_per_frame
    compulsory = wl.ring * (2 * col_px * cfg["height"] * es + n_blk * (5 * 4 + 20 + 8)) if wl.path == "sb" else x4d_bytes
    ach = compulsory / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    alg = x4d_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    res = {
        "workload": name, "value": total * steps / wall, "unit": "candidates/s", "ms_per_step": wall / steps * 1e3,
        "event_ms_per_step": ev_ms / steps, "candidates_per_step": total, "parity_frame0": ok,
        # `bound`: what the counters say limits the kernel (profiles/r02_sad_strip.md, r03_sad_strip.md): measured fabric traffic is
        # 1.05-1.09 x the compulsory bytes and the transport alone runs at 0.70 of the peak, but no unit is saturated (VALU 47 %, LDS
        # 50 % busy) -- the launch time is the evaluating wavefronts' serial instruction chain, one iteration per step at two
        # wavefronts per SIMD.  `frac` stays what the north star asks for: compulsory HBM bytes / time / HBM peak.
        # (`bound` names the ROOFLINE the fraction is priced against -- the contract's "hbm" | "mfma" --; `limited_by` what actually limits the kernel)
        "roofline": {"bound": "hbm", "limited_by": "issue/latency" if wl.path == "sb" else "L1 fill path (TA)", "frac_is": "compulsory HBM bytes / launch time / 8 TB/s",
                     "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                     "avg_launch_ms": k_ms, "compulsory_bytes_per_launch": compulsory,
                     "achieved_algorithmic": alg, "algorithmic_bytes_per_launch": x4d_bytes,
                     "note": "achieved / frac = COMPULSORY bytes (each visible src + ref byte of this rank's tile column once, "
                             "+ lists + results) / launch time; achieved_algorithmic = 516 B (1028 B 10-bit) per candidate / "
                             "launch time, an LDS-side rate that is NOT an HBM rate"},
        "kernels": {"path": wl.path, "cell": list(wl.cell), "sad_strip_kernel_avg_ms": k_ms if wl.path == "sb" else None,
                    "sad_x4d_kernel_avg_ms": kx_ms, "sad_cand_kernel_avg_ms": k1_ms},
        "ring_frames": wl.ring, "blocks_per_frame_this_rank": wl.blocks_per_frame, "tile_column_px": list(wl.tile),
    }
    if wl.path == "sb" and wl.d_sb and k_ms > 0:
        # the ceiling of THIS walk on THIS box in THIS run: the kernel's transport with everything else removed (csrc/probe.hip), timed like
        # the kernel; compulsory bytes over its launch time is what a kernel whose evaluation hid completely behind the transport would reach
        p_ms = kernel_avg_ms(ctx, wl.launch_probe, max(steps, 10))
        if p_ms > 0:
            res["roofline"]["ceiling_GBs"] = compulsory / (p_ms * 1e-3) / 1e9
            res["roofline"]["frac_of_ceiling"] = ach / res["roofline"]["ceiling_GBs"]
            res["roofline"]["ceiling_launch_ms"] = p_ms
            res["roofline"]["ceiling_requested_GBs"] = wl.probe_bytes / (p_ms * 1e-3) / 1e9
            res["roofline"]["ceiling_is"] = ("aomhip_strip_read_probe: the same strips / cells / range read into registers and discarded, "
                                             "timed in this run; ceiling_GBs counts the same compulsory bytes as `achieved`")
    if traffic and k_ms > 0:  # the mandatory companion figure
        res["roofline"]["traffic_GBs"] = traffic / (k_ms * 1e-3) / 1e9
        res["roofline"]["traffic_frac_of_peak"] = res["roofline"]["traffic_GBs"] / HBM_PEAK_GBS
        res["roofline"]["traffic_over_compulsory"] = traffic / compulsory
    if want_cpu and rank == 0 and orc is not None:
        res["cpu_baseline"] = wl.cpu_baseline(orc)
    wl.free()
    return res


def _sig(x, n=5):
    """Floats to n significant digits, recursively (the printed line is read by a parser with a size limit; the side file keeps full precision)."""
    if isinstance(x, float):
        return float("%.*g" % (n, x))
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


ROOFLINE_KEYS = ("bound", "limited_by", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "compulsory_bytes_per_launch",
                 "algorithmic_bytes_per_launch", "ceiling_GBs", "frac_of_ceiling", "traffic_over_compulsory", "traffic_measured_on")
LINE_LIMIT = 6000  # bytes of the final stdout line; the driver's record keeps an 8 KB tail (round 4's 31 KB line was not parsed)


def _other_summary(o):
    """One or two scalars per informational workload for the printed line; the whole entry goes to the side file / stderr."""
    out = {}
    for k in ("value", "ms_per_frame", "ms_per_step", "frames_per_s", "roofline_frac"):   # (units: the full record)
        if isinstance(o.get(k), (int, float, str)):
            out[k] = o[k]
    if isinstance(o.get("yuv420"), dict):   # the inner loop's 4:2:0 leg
        out["yuv420_ms_per_frame"], out["yuv420_roofline_frac"] = o["yuv420"]["ms_per_frame"], o["yuv420"].get("roofline_frac")
    for k, v in o.items():  # nested legs that carry a per-frame time (TF q30 / q12, joint search branches, NSTEP / 8-tap ...)
        if isinstance(v, dict) and isinstance(v.get("ms_per_frame", v.get("ms_per_filtered_frame")), (int, float)):
            out[k + "_ms"] = v.get("ms_per_frame", v.get("ms_per_filtered_frame"))
        elif k.endswith("_ms_per_frame") and isinstance(v, (int, float)):
            out[k] = v
    for k in o:
        if k.startswith("parity") and o[k] is not None:
            out["parity"] = bool(out.get("parity", True)) and bool(o[k])
    return out


def build_lines(args, world, main_res, others, strong):
    """(full record, printed line).  The printed line carries the contract's keys, the roofline as flat scalars (the three north-star sizes
    side by side), the cpu baseline, the transform half of the metric and one summary scalar set per informational workload -- and stays
    under LINE_LIMIT bytes.  Everything else (per-size tables, stage timings, notes, cpu legs) is in the full record."""
    cfg = WORKLOADS[args.workload]
    sad_all = [main_res] + [o for o in others if str(o.get("workload", "")).startswith("sad16x16_modeA")]
    roof_full = dict(main_res["roofline"])
    roof_full["sizes"] = {r_["workload"]: dict(r_["roofline"], candidates_per_s=r_["value"]) for r_ in sad_all}
    roof = {k: main_res["roofline"][k] for k in ROOFLINE_KEYS if main_res["roofline"].get(k) is not None}
    roof.setdefault("traffic", None)
    for r_ in sad_all[1:]:
        tag = r_["workload"].replace("sad16x16_modeA_", "")   # 4k_8bit / 4k_10bit / *_range32
        for k_ in ("frac", "avg_launch_ms", "frac_of_ceiling", "traffic_over_compulsory"):
            if r_["roofline"].get(k_) is not None:
                roof["%s_%s" % (k_, tag)] = r_["roofline"][k_]
        roof["candidates_per_s_%s" % tag] = r_["value"]
    txqs = [o for o in others if str(o.get("workload", "")).startswith("fwd_txfm2d+quantize_b")]
    vars_ = [o for o in others if o.get("workload") in VAR_WORKLOADS]
    filt = next((o for o in others if o.get("workload") == "filters_ring_4k_10bit"), None)
    rest = [o for o in others if o not in txqs and o not in sad_all and o not in vars_ and o is not filt]
    cpu = main_res.get("cpu_baseline")
    head = {
        "metric": "SAD-candidates/s", "value": main_res["value"], "unit": "candidates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8" if cfg["bit_depth"] == 8 else "u16", "data": "synthetic",
        "config": {"workload": args.workload, "frame": "%dx%d" % (cfg["width"], cfg["height"]),
                   "bit_depth": cfg["bit_depth"], "block": "16x16",
                   "mode": "A: 1 sad16x16 @mv(0,0) + 1 sad16x16x4d (uniform in [-%d,%d]^2) per block" % (cfg.get("search_range", 64), cfg.get("search_range", 64)),
                   "ring_frame_pairs_per_gpu": FRAMES_OVERRIDE or cfg["frames"], "candidates_per_step": main_res["candidates_per_step"],
                   "partition": ("balanced tile columns (encoder.c:247-275)" if TILE_COLUMNS == "balanced" else "uniform tile columns (tile_common.c:76-97)") +
                                ", one per GPU; no data-path collective",
                   "clock_ramp_s": float(os.environ.get("AOMHIP_BENCH_RAMP_S", "0.25"))},
    }
    full = dict(head, roofline=roof_full, cpu_baseline=cpu,
                txq={t["workload"]: t for t in txqs} or None, strong_scaling_search=strong,
                parity_frame0_and_last_slot=main_res["parity_frame0"], kernels=main_res["kernels"], others=others)
    line = dict(head, roofline=roof,
                cpu_baseline=None if cpu is None else dict({k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "cpu_model")},
                                                           sample=cpu.get("sample_short", "")),
                # the other half of BASELINE.json's metric: fwd_txfm+quant blocks/s at 1080p (8-bit) and 4K (10-bit)
                txq={t["workload"]: {"value": t["value"], "unit": "blocks/s",
                                     "roofline": {k: t["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms")},
                                     "cpu_baseline": {k: (t.get("cpu_baseline") or {}).get(k) for k in ("value", "cores", "kind")},
                                     "per_size_frac": {k: v["frac"] for k, v in t["per_size"].items()},
                                     "qindex_frac_16x16": t.get("qindex_sweep_16x16_frac"),
                                     "tx_type_frac_16x16": None if not t.get("tx_type_sweep_16x16_frac") else
                                     {k: t["tx_type_sweep_16x16_frac"][k] for k in ("min", "max")}} for t in txqs} or None,
                # SURVEY 8(d) rows A6-A8 / E / F on HBM-resident rings: frac = algorithmic bytes, c = compulsory bytes (every ring byte once),
                # t = counter traffic (null until measured on this kernel source), all / launch time / 8 TB/s
                variance={o["workload"].replace("variance16x16_modeA_", "var_").replace("sub_pixel_var_", "subpel_var_"):
                          dict({"frac": o["roofline"]["frac"], "c": o["roofline"]["frac_compulsory"], "t": o["roofline"]["frac_traffic"],
                                "ms": o["roofline"]["avg_launch_ms"], "parity": o["parity_sample_slot0_and_last"]},
                               # (full-pel lists through the strip walk, aomhip_variance_sb_batch: the same evaluations, bit-identical)
                               **({"sb_ms": o["strip_walk"]["avg_launch_ms"], "sb_c": o["strip_walk"]["frac_compulsory"], "sb_t": o["strip_walk"]["frac_traffic"],
                                   "sb_same": o["strip_walk"]["identical_to_direct_slot0_and_last"]} if o.get("strip_walk") else {})) for o in vars_} or None,
                filters_ring=None if filt is None else dict(
                    {k: {"us": filt[k]["ms_per_plane"] * 1e3, "frac": filt[k]["frac"], "c": filt[k]["frac_compulsory"], "t": filt[k]["frac_traffic"]}
                     for k in ("deblock_vert+horz", "cdef_luma")}, parity=filt["parity_slot0_and_last"], ring_GB=1.32),
                strong_scaling_search=strong,
                parity_frame0_and_last_slot=main_res["parity_frame0"],
                parity_all=all(bool(v) for o in [main_res] + others for k, v in o.items() if k.startswith("parity") and v is not None),
                others={str(o.get("workload")): _other_summary(o) for o in rest} or None)
    il = next((o for o in rest if o.get("workload") == "encode_inner_loop_4k_10bit"), None)
    if il and il.get("valu_issue_rates"):
        vr = il["valu_issue_rates"]
        # the measured denominator of every valu_frac (aomhip_valu_issue_probe, this run) and the stage fractions re-based on it
        line["valu_issue"] = {"unit": "G wave-instr/s/SIMD", "fast": vr["fast"] / 1e9, "slow": vr["slow"] / 1e9, "trans": vr["trans"] / 1e9,
                              "clocks_per_wave_inst": vr["clocks_per_wave_inst"], "clock_GHz": vr["clock_hz_median"] / 1e9,
                              "stage_valu_frac": {k: v.get("valu_frac") for k, v in il["stages"].items() if v.get("valu_frac") is not None},
                              "stage_ms": {k: v["ms"] for k, v in il["stages"].items()}}
    line = _sig(line)
    # never let the line outgrow the record that reads it: shed the least important keys first (they stay in the full record)
    for drop in ("valu_issue.stage_ms", "valu_issue.stage_valu_frac", "strong_scaling_search.tile_columns_px_balanced", "strong_scaling_search.tile_columns_px_uniform",
                 "others.wiener_stats_luma_4k", "others.cdef_search_luma_4k_10bit", "others.mesh_search_4k_10bit", "others", "txq"):
        if len(json.dumps(line, separators=(",", ":"))) <= LINE_LIMIT:
            break
        if "." in drop:
            a, b = drop.split(".")
            if isinstance(line.get(a), dict):
                line[a].pop(b, None)
        else:
            line[drop] = None
    return full, line


def emit_lines(full, line):
    """Full record -> bench_full.json (gpurun_out/ when it exists, else beside this script; AOMHIP_BENCH_FULL overrides) and, one JSON object
    per workload, to stderr; then the ONE stdout line."""
    path = os.environ.get("AOMHIP_BENCH_FULL")
    if not path:
        d = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(d if os.path.isdir(d) else ROOT, "bench_full.json")
    try:
        with open(path, "w") as f:
            json.dump(full, f)
        line["full_record"] = os.path.relpath(path, ROOT)
    except OSError as e:
        print("bench.py: could not write %s (%s)" % (path, e), file=sys.stderr)
    for o in full.get("others") or []:
        print(json.dumps(_sig(o, 6)), file=sys.stderr)
    sys.stderr.flush()
    out = json.dumps(line, separators=(",", ":"))
    assert len(out) <= LINE_LIMIT + 200, "bench line grew to %d bytes" % len(out)
    print(out, flush=True)


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
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the real multi-GPU path); gloo only to dry-run the N > 1 code on one GPU")
    ap.add_argument("--frames-per-gpu", type=int, default=0, help="override the ring size per GPU (0 = workload default)")
    ap.add_argument("--tile-columns", default="uniform", choices=["uniform", "balanced"],
                    help="N > 1: how the frame is cut into one tile column per GPU -- the reference's uniform spacing (tile_common.c:76-97) or its "
                         "auto_tile_size_balancing (encoder.c:247-275: widths within one superblock of each other)")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"],
                    help="N > 1 search pipeline: what aomhip_allgather_recon moves per frame (both are timed; this one is in `value`)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, BEFORE this process touches the
        # GPU (it never does: it only waits).  Under torch.distributed.run WORLD_SIZE is set and this branch is not taken.
        sys.exit(spawn_ranks(args.gpus))

    global FRAMES_OVERRIDE, TILE_COLUMNS
    FRAMES_OVERRIDE = args.frames_per_gpu
    TILE_COLUMNS = args.tile_columns
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
        r = run_compound_search(pkg, ctx, orc if not only else None, args.steps, args.warmup, bs=only or 16)
        # the same five calls over the frame cut into 8x8, 32x32 and 64x64 blocks (timing only; the tests cover the sizes' parity)
        r["by_block_size"] = {"%dx%d" % (only or 16, only or 16): {k: v["ms_per_frame"] for k, v in r.items() if isinstance(v, dict) and "ms_per_frame" in v}}
        for bs_ in () if only else (8, 32, 64):
            r2 = run_compound_search(pkg, ctx, None, max(3, args.steps // 2), 1, bs=bs_)
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
