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
# frame-level statistics of the variance-based partition and the global-motion segment error, and the Hadamard / SATD helper
for bd in (10, 8):
    a = pkg.synth.lcg_frame(W, H, 2, 0, bd); b = pkg.synth.lcg_frame(W, H, 2, 1, bd)
    pa, pb = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pa, 0, a); ctx.planes_upload(pb, 0, b)
    n16x, n16y = (W + 15) // 16, (H + 15) // 16
    d_sum8 = ctx.malloc(8 * 4 * (2 * n16x) * (2 * n16y) + 4096)
    f8 = lambda: ctx.vbp_8x8_stats_plane(pa, 0, pb, 0, W, H, d_sum8, 2 * n16x)
    common.ramp(ctx, f8, 0.05)
    m8 = common.kernel_avg_ms(ctx, f8, 20)
    d_sum4 = ctx.malloc(8 * 4 * (W // 4 + 8) * (H // 4 + 8))
    f4 = lambda: ctx.vbp_4x4_avg_plane(pa, 0, W, H, 0, d_sum4, W // 4)
    common.ramp(ctx, f4, 0.05)
    m4 = common.kernel_avg_ms(ctx, f4, 20)
    seg = np.ones((H // 32 + 1, W // 32 + 1), np.uint8)
    d_seg, d_err = ctx.to_device(seg), ctx.malloc(8)
    fe = lambda: ctx.segmented_frame_error(pa, 0, pb, 0, W, H, d_seg, seg.shape[1], d_err)
    common.ramp(ctx, fe, 0.05)
    me = common.kernel_avg_ms(ctx, fe, 20)
    print("%2d-bit 4K: vbp 8x8 statistics %.1f us, vbp 4x4 averages %.1f us, segmented frame error %.1f us" % (bd, m8 * 1e3, m4 * 1e3, me * 1e3))
    ctx.planes_free(pa); ctx.planes_free(pb)
for nn, fl in ((8, 0), (16, 0), (32, 0), (8, 2), (16, 2), (32, 2)):
    xs, ys = np.meshgrid(np.arange(0, W - nn + 1, nn), np.arange(0, H - nn + 1, nn))
    n = xs.size
    t = np.zeros(n, capi.txb_dtype)
    t["x"], t["y"] = xs.ravel(), ys.ravel()
    t["out_offset"] = np.arange(n, dtype=np.uint32) * (nn * nn)
    dt, dsatd = ctx.to_device(t), ctx.malloc(4 * n)
    fh = lambda: ctx.hadamard_batch(dres, W, nn, fl, dt, n, None, dsatd)
    try:
        common.ramp(ctx, fh, 0.05)
        ms = common.kernel_avg_ms(ctx, fh, 20)
        print("hadamard + satd %2dx%-2d flavour %d: %7.1f us per frame, %5.2f TB/s" % (nn, nn, fl, ms * 1e3, 2.0 * n * nn * nn / ms / 1e9))
    except Exception as e:
        print("hadamard", nn, fl, type(e).__name__, e)
    ctx.free(dt); ctx.free(dsatd)
