"""The small RD helpers over every block of a 4K frame (timing only): aomhip_sse_batch, aomhip_sum_sse_2d_i16_batch, aomhip_hadamard_batch.
    python tools/rd_helpers_time.py            (AOMHIP_LIB selects the library)
Prints microseconds per frame and the rate against the bytes each call has to read."""
import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("aom-av1-psy_amd")
from benchlib import common
capi = pkg.capi
ctx = capi.Context(0)
W, H, border = 3840, 2160, 64
for bd in (10, 8):
    a = pkg.synth.lcg_frame(W, H, 2, 0, bd); b = pkg.synth.lcg_frame(W, H, 2, 1, bd)
    pa, pb = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pa, 0, a); ctx.planes_upload(pb, 0, b)
    es = 2 if bd > 8 else 1
    for bs in (8, 16, 32, 64):
        xs, ys = np.meshgrid(np.arange(0, W - bs + 1, bs), np.arange(0, H - bs + 1, bs))
        n = xs.size
        c = np.zeros(n, capi.sad_cand_dtype)
        c["sx"], c["sy"], c["rx"], c["ry"] = xs.ravel(), ys.ravel(), xs.ravel(), ys.ravel()
        dc, do = ctx.to_device(c), ctx.malloc(8 * n)
        fn = lambda: ctx.sse_batch(pa, pb, 0, bs, bs, dc, n, do)
        common.ramp(ctx, fn, 0.05)
        ms = common.kernel_avg_ms(ctx, fn, 20)
        got = ctx.from_device(do, (n,), np.int64)
        k = n // 2
        want = int(((a[ys.ravel()[k]:ys.ravel()[k] + bs, xs.ravel()[k]:xs.ravel()[k] + bs].astype(np.int64) - b[ys.ravel()[k]:ys.ravel()[k] + bs, xs.ravel()[k]:xs.ravel()[k] + bs].astype(np.int64)) ** 2).sum())
        print("%2d-bit sse %3dx%-3d: %7.1f us per frame, %5.2f TB/s of its %d MB%s" % (bd, bs, bs, ms * 1e3, 2.0 * n * bs * bs * es / ms / 1e9, 2 * n * bs * bs * es >> 20, "" if int(got[k]) == want else "  WRONG"))
        ctx.free(dc); ctx.free(do)
    ctx.planes_free(pa); ctx.planes_free(pb)
# int16 residual plane: sum / sse per transform block
res = np.random.default_rng(1).integers(-1023, 1024, (H, W)).astype(np.int16)
dres = ctx.to_device(res)
for bs in (8, 16, 32, 64):
    xs, ys = np.meshgrid(np.arange(0, W - bs + 1, bs), np.arange(0, H - bs + 1, bs))
    n = xs.size
    t = np.zeros(n, capi.txb_dtype)
    t["x"], t["y"] = xs.ravel(), ys.ravel()
    dt, dsse, dsum = ctx.to_device(t), ctx.malloc(8 * n), ctx.malloc(4 * n)
    fn = lambda: ctx.sum_sse_2d_i16_batch(dres, W, bs, bs, dt, n, dsse, dsum)
    common.ramp(ctx, fn, 0.05)
    ms = common.kernel_avg_ms(ctx, fn, 20)
    print("sum_sse_2d_i16 %3dx%-3d: %7.1f us per frame, %5.2f TB/s" % (bs, bs, ms * 1e3, 2.0 * n * bs * bs / ms / 1e9))
    for d in (dt, dsse, dsum):
        ctx.free(d)
