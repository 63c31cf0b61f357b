"""BASELINE.json configs[1] and the north-star sizes: the Mode-A SAD rings (aomhip_sad_sb_batch + the direct kernels), the headline workload."""
import json
import os
import sys
import time

import numpy as np

from . import common
from .common import HBM_PEAK_GBS, ROOT, kernel_avg_ms, ramp
from .common import load_traffic
from .dist import _red_device, barrier, time_steps


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
        x0, x1 = pkg.partition.column_of_rank(W, world, rank, mode=common.TILE_COLUMNS)
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
        col_w = pkg.partition.column_of_rank(W, world, 0, mode=common.TILE_COLUMNS)[1] - pkg.partition.column_of_rank(W, world, 0, mode=common.TILE_COLUMNS)[0]
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


def run_workload(pkg, ctx, dist, dev, rank, world, name, steps, warmup, want_cpu, orc):
    wl = SadModeA(pkg, ctx, name, rank, world, frames_per_rank=common.FRAMES_OVERRIDE or None)
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
