"""SURVEY 8(d) row "1 deblocked or CDEF-filtered pixel" on a working set that is NOT cache resident: the two in-loop filters of configs[4] over a
ring of 32 distinct 4K 10-bit luma planes (0.66 GB per bordered ring; the deblocking filter works in place, CDEF reads that ring and writes a
second one: 1.3 GB in all against 256 MiB of Infinity Cache).  Every step filters the whole ring, one plane after the other, with the
inner loop's parameters (every 8x8 edge at level 32; CDEF pri 4 / sec 2 / damping 6, no skipped blocks).

  algorithmic bytes per pixel     4 (16-bit planes: read + write once per filter, vertical + horizontal counted once -- SURVEY 8(d))
  compulsory bytes per pixel      deblock as launched here (two in-place passes): 8; CDEF: 4
Content: low-pass noise with a random offset per 8x8 block (a reconstruction with blocking: the edge filters have real work, CDEF finds
directions); the deblocked ring is restored from the host before every timed pass, so that every pass filters unfiltered planes.
"""
import os
import time

import numpy as np

from .common import HBM_PEAK_GBS, load_traffic_entry, source_sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blocky_frames(synth, W, H, bd, n_base, n_frames):
    mx = (1 << bd) - 1
    bases = [synth.shifted_smooth_pair(W, H, 100 + k, bd)[0].astype(np.int32) for k in range(n_base)]
    out = []
    for f in range(n_frames):
        rng = np.random.default_rng(777 + f)
        off = rng.integers(-6 << (bd - 8), (6 << (bd - 8)) + 1, (H // 8, W // 8))
        img = bases[f % n_base] + np.kron(off, np.ones((8, 8), np.int32))
        out.append(np.clip(img, 0, mx).astype(np.uint16 if bd > 8 else np.uint8))
    return out


def run_filters_ring(pkg, ctx, orc, steps, warmup, width=3840, height=2160, bd=10, frames=32):
    W, H, F = width, height, frames
    border = 160
    synth = pkg.synth
    host = blocky_frames(synth, W, H, bd, 4, F)
    p, out = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    params = np.zeros((H // 4, W // 4, 4), np.uint8)
    params[:, 2::2, 0] = 8; params[:, 2::2, 1] = 32; params[2::2, :, 2] = 8; params[2::2, :, 3] = 32
    d_params = ctx.to_device(params)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    pri, sec, skip = np.full((fbh, fbw), 4, np.uint8), np.full((fbh, fbw), 2, np.uint8), np.zeros((H // 8, W // 8), np.uint8)
    d_pri, d_sec, d_skip = ctx.to_device(pri), ctx.to_device(sec), ctx.to_device(skip)

    def restore():
        for f in range(F):
            ctx.planes_upload(p, f, host[f])

    def deblock_ring():
        for f in range(F):
            ctx.deblock_plane(p, f, d_params, W // 4, 0, 3)

    def cdef_ring():
        for f in range(F):
            ctx.cdef_luma_plane(p, f, out, f, d_pri, d_sec, fbw, d_skip, 6)

    # parity first (ring slot 0 and the last slot against the oracle), on the same planes the timed passes filter
    restore(); deblock_ring(); cdef_ring(); ctx.sync()
    ok = None
    if orc is not None:
        ok = True
        for f in (0, F - 1):
            want_d = orc.deblock_plane(host[f], params, sharpness=0, bd=bd)
            got_d = ctx.planes_download(p, f)[border:border + H, border:border + W]
            ok &= bool(np.array_equal(got_d, want_d))
            want_c = orc.cdef_plane_luma(want_d, pri, sec, skip, 6, bd=bd)
            want_c = want_c[0] if isinstance(want_c, tuple) else want_c
            got_c = ctx.planes_download(out, f)[border:border + H, border:border + W]
            ok &= bool(np.array_equal(got_c, want_c))
    for _ in range(max(1, warmup)):
        restore(); deblock_ring(); cdef_ring()
    ctx.sync()
    dbk_ms = cdef_ms = 0.0
    t0 = time.perf_counter()
    for _ in range(steps):
        restore(); ctx.sync()
        ctx.timer_begin(); deblock_ring(); dbk_ms += ctx.timer_end()
        ctx.timer_begin(); cdef_ring(); cdef_ms += ctx.timer_end()
    wall = time.perf_counter() - t0
    px = W * H
    es = 2 if bd > 8 else 1
    algo = 2 * es * px   # per plane and filter

    def leg(ms_total, compulsory_factor, kernels, sources):
        ms = ms_total / (steps * F)
        key = "filters_ring_4k_10bit:" + kernels[0]
        traffic = load_traffic_entry(ROOT, key, source_sha(ROOT, sources))
        return {"ms_per_plane": ms, "kernels": kernels, "achieved": algo / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "bound": "hbm",
                "frac": algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_plane": algo, "compulsory_bytes_per_plane": compulsory_factor * algo,
                "frac_compulsory": compulsory_factor * algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "frac_traffic": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None}

    # one trial of the loop-filter level search (search_filter_level -> try_filter_frame, picklpf.c:49-193) = aomhip_lpf_search_sse: copy the frame,
    # deblock the copy at the trial's levels, SSE against the source -- eight trials over ring slots 0 (reconstruction), 1 (scratch), 2 (source)
    n_trials = 8
    tp = np.zeros((n_trials,) + params.shape, np.uint8)
    for t in range(n_trials):
        tp[t] = params
        tp[t, :, 2::2, 1] = 4 * (t + 1); tp[t, 2::2, :, 3] = 4 * (t + 1)
    d_tp, d_tsse = ctx.to_device(tp), ctx.malloc(8 * n_trials)
    restore()
    trial = lambda: ctx.lpf_search_sse(p, 0, p, 1, p, 2, d_tp, params.size, n_trials, W // 4, 0, 3, d_tsse)
    trial(); ctx.sync()
    lpf_ok = None
    if orc is not None:   # the first trial's SSE against the oracle's deblocked plane
        want_d = orc.deblock_plane(host[0], tp[0], sharpness=0, bd=bd)
        lpf_ok = int(ctx.from_device(d_tsse, (n_trials,), np.uint64)[0]) == int(((want_d.astype(np.int64) - host[2].astype(np.int64)) ** 2).sum())
    ctx.timer_begin()
    for _ in range(max(steps, 3)):
        trial()
    lpf_us = ctx.timer_end() / (max(steps, 3) * n_trials) * 1e3
    ctx.free(d_tp); ctx.free(d_tsse)
    res = {"workload": "filters_ring_4k_10bit", "value": steps * F / ((dbk_ms + cdef_ms) * 1e-3), "unit": "planes/s (deblock + CDEF)",
           "lpf_search_trial": {"us_per_trial": lpf_us, "trial": "frame copy + deblock at the trial's levels + plane SSE against the source"},
           "parity_lpf_first_trial": lpf_ok,
           "ms_per_step": (dbk_ms + cdef_ms) / steps, "wall_ms_per_step_incl_restore": wall / steps * 1e3, "parity_slot0_and_last": ok,
           "deblock_vert+horz": leg(dbk_ms, 2, ("deblock_vert", "deblock_horz"), ("deblock.hip",)),
           "cdef_luma": leg(cdef_ms, 1, ("cdef_luma_kernel",), ("cdef.hip",)),
           "config": {"frame": "%dx%d %d-bit luma, ring of %d planes in place + %d CDEF outputs (%.2f GB)" % (W, H, bd, F, F, 2 * F * (W + 2 * border) * (H + 2 * border) * es / 1e9),
                      "filters": "deblock: every 8x8 edge, level 32 (two in-place passes); CDEF: pri 4, sec 2, damping 6",
                      "content": "low-pass noise + a random offset per 8x8 block; the ring is restored from the host before every timed deblock pass"}}
    for d in (d_params, d_pri, d_sec, d_skip):
        ctx.free(d)
    ctx.planes_free(p); ctx.planes_free(out)
    return res
