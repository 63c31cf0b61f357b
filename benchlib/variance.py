"""SURVEY 8(d) row "1 variance / sub-pel variance evaluation": aom_variance16x16 / aom_sub_pixel_variance16x16 (aom_dsp/variance.c:56-163) over
the Mode-A rings -- per 16x16 block five evaluations: the zero-MV position and four positions uniform in [-64, 64]^2 (sub-pel: each with its own
1/8-pel offset, never (0, 0)) -- HBM-resident frame pairs, lists distinct per frame, one aomhip_[sub_pixel_]variance_batch launch per step.

  algorithmic bytes per evaluation   8-bit 520 (2 x 256 + var + sse) / 553 (256 + 17 x 17 + 8); 16-bit 1 032 / 1 098      
  compulsory bytes per launch        every visible source and reference byte of the ring once + 12 B of list + 8 B of results per evaluation
"""
import os

import numpy as np

from .common import HBM_PEAK_GBS, kernel_avg_ms, load_traffic_entry, ramp, source_sha

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VAR_WORKLOADS = {
    "variance16x16_modeA_1080p_8bit": dict(width=1920, height=1080, bit_depth=8, frames=64, subpel=False),
    "sub_pixel_variance16x16_modeA_1080p_8bit": dict(width=1920, height=1080, bit_depth=8, frames=64, subpel=True),
    "variance16x16_modeA_4k_10bit": dict(width=3840, height=2160, bit_depth=10, frames=32, subpel=False),
    "sub_pixel_variance16x16_modeA_4k_10bit": dict(width=3840, height=2160, bit_depth=10, frames=32, subpel=True),
}
VAR_SOURCES = ("variance.hip", "variance_device.h")


def bytes_per_eval(bd, subpel):
    es = 1 if bd == 8 else 2
    return (256 + 17 * 17 if subpel else 512) * es + 8


def run_variance(pkg, ctx, orc, steps, warmup, name):
    cfg = VAR_WORKLOADS[name]
    W, H, bd, F, subpel = cfg["width"], cfg["height"], cfg["bit_depth"], cfg["frames"], cfg["subpel"]
    capi, synth = pkg.capi, pkg.synth
    border = 160
    src, ref = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    pairs = {}
    for f in range(F):
        s, r = synth.lcg_frame(W, H, 2 * f, 0, bd), synth.lcg_frame(W, H, 2 * f + 1, 0, bd)
        if f in (0, F - 1):
            pairs[f] = (s, r)
        ctx.planes_upload(src, f, s); ctx.planes_upload(ref, f, r)
    base_c, base_g = synth.mode_a_worklist(W, H, 16, seed=1, search=64)
    nb = len(base_c)
    n = 5 * nb
    rng = np.random.default_rng(4242)
    cands = np.zeros((F, nb, 5), capi.var_cand_dtype)
    cands["sx"], cands["sy"] = base_c["sx"][None, :, None], base_c["sy"][None, :, None]
    cands["rx"], cands["ry"] = cands["sx"], cands["sy"]
    cands["rx"][:, :, 1:] += rng.integers(-64, 65, (F, nb, 4), dtype=np.int16)
    cands["ry"][:, :, 1:] += rng.integers(-64, 65, (F, nb, 4), dtype=np.int16)
    if subpel:
        off = rng.integers(1, 64, (F, nb, 5))    # (xoff, yoff) != (0, 0)
        cands["xoff"], cands["yoff"] = off & 7, off >> 3
    cands = cands.reshape(F, n)
    d_c = ctx.to_device(np.ascontiguousarray(cands))
    d_var, d_sse = ctx.malloc(F * n * 4), ctx.malloc(F * n * 4)

    def step():
        ctx.variance_batch(src, ref, 0, F, 16, 16, d_c, n, n, d_var, d_sse, subpel=subpel)

    ramp(ctx, step)
    for _ in range(warmup):
        step()
    ctx.sync()
    import time
    ctx.timer_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ev_ms = ctx.timer_end()
    wall = time.perf_counter() - t0
    avg_ms = kernel_avg_ms(ctx, step, max(steps, 10))
    # parity: a sample of ring slot 0 and of the last slot against the oracle (the oracle's per-candidate Python loop bounds the sample)
    ok = None
    if orc is not None:
        ok = True
        pick = np.random.default_rng(9).choice(n, 2500, replace=False)
        for f in (0, F - 1):
            sb, rb = orc.extend_plane(pairs[f][0], border, src.stride), orc.extend_plane(pairs[f][1], border, ref.stride)
            want = orc.variance_cands(sb, rb, border, 16, 16, cands[f][pick], subpel=subpel, bd=bd)
            got_v = ctx.from_device(d_var + f * n * 4, (n,), np.uint32)[pick]
            got_s = ctx.from_device(d_sse + f * n * 4, (n,), np.uint32)[pick]
            ok &= bool(np.array_equal(got_v, want[:, 0]) and np.array_equal(got_s, want[:, 1]))
    # ---- the same evaluations through the strip walk (aomhip_variance_sb_batch, full-pel only): lists bucketed by cell as for the SAD kernel
    strip = None
    if not subpel:
        cell = (240, 64) if bd == 8 and W == 1920 else (320, 48) if bd == 8 else (160, 32)
        perm, off = synth.bucket_order(base_c["sx"], base_c["sy"], W, H, *cell)
        c5 = cands.reshape(F, nb, 5)
        g = np.zeros((F, nb), capi.sad_x4d_dtype)
        g["sx"], g["sy"] = c5["sx"][:, :, 0], c5["sy"][:, :, 0]
        g["rx"], g["ry"] = c5["rx"][:, :, 1:], c5["ry"][:, :, 1:]
        d_g, d_c1, d_off = ctx.to_device(np.ascontiguousarray(g[:, perm])), ctx.to_device(base_c[perm]), ctx.to_device(off)
        d_v4, d_s4, d_v1, d_s1 = ctx.malloc(F * nb * 16), ctx.malloc(F * nb * 16), ctx.malloc(F * nb * 4), ctx.malloc(F * nb * 4)

        def step_sb():
            ctx.variance_sb_batch(src, ref, 0, F, 16, 16, cell[0], cell[1], 64, len(off) - 1, d_g, d_off, nb, nb, d_v4, d_s4, d_c1, d_off, nb, 0, d_v1, d_s1)

        sb_ms = kernel_avg_ms(ctx, step_sb, max(steps, 10))
        # parity: every evaluation of ring slot 0 and of the last slot against the direct kernel's results (which the oracle sample above checks)
        same = True
        for f in (0, F - 1):
            dv = ctx.from_device(d_var + f * n * 4, (nb, 5), np.uint32)[perm]
            dq = ctx.from_device(d_sse + f * n * 4, (nb, 5), np.uint32)[perm]
            same &= bool(np.array_equal(ctx.from_device(d_v4 + f * nb * 16, (nb, 4), np.uint32), dv[:, 1:]) and
                         np.array_equal(ctx.from_device(d_s4 + f * nb * 16, (nb, 4), np.uint32), dq[:, 1:]) and
                         np.array_equal(ctx.from_device(d_v1 + f * nb * 4, (nb,), np.uint32), dv[:, 0]) and
                         np.array_equal(ctx.from_device(d_s1 + f * nb * 4, (nb,), np.uint32), dq[:, 0]))
        es_ = 1 if bd == 8 else 2
        comp_sb = F * (2 * W * H * es_ + nb * (20 + 8 / F + 40))   # planes + one group record, the shared single list, 10 results per block
        strip = {"kernel": "sad_strip_kernel<VAR>", "cell": list(cell), "avg_launch_ms": sb_ms, "identical_to_direct_slot0_and_last": same,
                 "frac": F * n * bytes_per_eval(bd, False) / (sb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "frac_compulsory": comp_sb / (sb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "speedup_over_direct": avg_ms / sb_ms,
                 "traffic": load_traffic_entry(ROOT, name + ":sb", source_sha(ROOT, ("sad_sb.hip",)))}
        strip["frac_traffic"] = (strip["traffic"] / (sb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if strip["traffic"] else None
        for d in (d_g, d_c1, d_off, d_v4, d_s4, d_v1, d_s1):
            ctx.free(d)
    es = 1 if bd == 8 else 2
    algo = F * n * bytes_per_eval(bd, subpel)
    compulsory = F * (2 * W * H * es + n * (12 + 8))
    key = name
    traffic = load_traffic_entry(ROOT, key, source_sha(ROOT, VAR_SOURCES))
    res = {"workload": name, "value": F * n * steps / wall, "unit": "evaluations/s", "ms_per_step": wall / steps * 1e3, "event_ms_per_step": ev_ms / steps,
           "evaluations_per_step": F * n, "parity_sample_slot0_and_last": ok, "strip_walk": strip,
           "parity_strip_walk": None if strip is None else strip["identical_to_direct_slot0_and_last"],
           "roofline": {"bound": "hbm", "kernel": "variance_kernel", "avg_launch_ms": avg_ms, "achieved": algo / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": algo / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo,
                        "bytes_per_evaluation": bytes_per_eval(bd, subpel), "compulsory_bytes_per_launch": compulsory,
                        "frac_compulsory": compulsory / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                        "frac_traffic": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                        "note": "five evaluations of a block share its source rows and the +-64 windows overlap: `frac` above 1 means that reuse is served "
                                "by L1 / L2, not HBM; `frac_compulsory` (every ring byte once) is the HBM-side figure, `frac_traffic` the counters'"},
           "config": {"frame": "%dx%d %d-bit, ring of %d pairs (%.2f GB)" % (W, H, bd, F, 2 * F * (W + 2 * border) * (H + 2 * border) * es / 1e9),
                      "list": "Mode A: zero-MV + 4 positions in [-64, 64]^2 per 16x16 block, %s" % ("1/8-pel offsets != (0, 0)" if subpel else "full-pel")}}
    for d in (d_c, d_var, d_sse):
        ctx.free(d)
    ctx.planes_free(src); ctx.planes_free(ref)
    return res
