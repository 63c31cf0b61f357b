"""BASELINE.json configs[2]: av1_fwd_txfm2d + aom_[highbd_]quantize_b over whole residual planes."""
import json
import os
import sys
import time

import numpy as np

from . import common
from .common import HBM_PEAK_GBS, ROOT, kernel_avg_ms, ramp
from .common import load_traffic


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
