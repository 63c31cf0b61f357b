"""aomhip_plane_sse and a filter-level trial of aomhip_lpf_search_sse (copy + deblock + SSE against the source) on a 4K plane, 10 and 8 bits:
    python tools/plane_sse_time.py            (AOMHIP_LIB selects the library)"""
import importlib, sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("aom-av1-psy_amd")
from benchlib import common
ctx = pkg.capi.Context(0)
for bd in (10, 8):
    W, H, border = 3840, 2160, 64
    a = pkg.synth.lcg_frame(W, H, 2, 0, bd); b = pkg.synth.lcg_frame(W, H, 2, 1, bd)
    pa, pb, pscr = (ctx.planes_alloc(W, H, border, bd, 1) for _ in range(3))
    ctx.planes_upload(pa, 0, a); ctx.planes_upload(pb, 0, b)
    d = ctx.malloc(8 * 8)
    fn = lambda: ctx.plane_sse(pa, 0, pb, 0, d)
    common.ramp(ctx, fn, 0.1)
    ms = common.kernel_avg_ms(ctx, fn, 50)
    got = int(ctx.from_device(d, (1,), np.uint64)[0]); want = int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
    # eight trials: 8x8 transform edges everywhere, levels 4 .. 32
    n_trials = 8
    params = np.zeros((n_trials, H // 4, W // 4, 4), np.uint8)
    for t in range(n_trials):
        params[t, :, 2::2, 0] = 8; params[t, :, 2::2, 1] = 4 * (t + 1)
        params[t, 2::2, :, 2] = 8; params[t, 2::2, :, 3] = 4 * (t + 1)
    dp = ctx.to_device(params)
    trial = lambda: ctx.lpf_search_sse(pa, 0, pscr, 0, pb, 0, dp, params[0].size, n_trials, W // 4, 0, 3, d)
    common.ramp(ctx, trial, 0.1)
    tms = common.kernel_avg_ms(ctx, trial, 10) / n_trials
    print("%d-bit: plane_sse %.1f us (%s), filter-level trial %.1f us" % (bd, ms * 1000, "exact" if got == want else "WRONG", tms * 1000))
