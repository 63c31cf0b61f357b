"""Loop-restoration filters over every 64x64 unit of a 4K plane (timing only): aomhip_wiener_convolve_add_src_batch, aomhip_apply_selfguided_restoration_batch.
    python tools/lr_apply_time.py       (AOMHIP_LIB selects the library)"""
import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("aom-av1-psy_amd")
from benchlib import common
capi = pkg.capi
ctx = capi.Context(0)
W, H, B = 3840, 2160, 16
for bd in (10, 8):
    for U in (64, 256):
        dat = pkg.synth.lcg_frame(W, H, 3, 0, bd)
        p, q = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(p, 0, dat); ctx.planes_upload(q, 0, np.zeros_like(dat))
        units = [(x, min(x + U, W), y, min(y + U, H)) for y in range(0, H, U) for x in range(0, W, U)]
        n = len(units)
        rec = np.zeros(n, capi.rect_dtype)
        for i, (x0, x1, y0, y1) in enumerate(units):
            rec["h_start"][i], rec["h_end"][i], rec["v_start"][i], rec["v_end"][i] = x0, x1, y0, y1
        d_u = ctx.to_device(rec)
        filt = np.tile(np.array([3, -7, 15, -22, 15, -7, 3, 0] * 2, np.int16), (n, 1))
        d_f = ctx.to_device(filt)
        fw = lambda: ctx.wiener_convolve_add_src_batch(p, 0, q, 0, d_u, rec, n, d_f, U, U)
        common.ramp(ctx, fw, 0.05)
        wms = common.kernel_avg_ms(ctx, fw, 10)
        idx = np.array([(5 * i + 2) % 16 for i in range(n)], np.int32)
        xqd = np.tile(np.array([-30, 40], np.int32), (n, 1))
        d_i, d_x = ctx.to_device(idx), ctx.to_device(xqd)
        d_f0, d_f1 = ctx.malloc(4 * n * U * U), ctx.malloc(4 * n * U * U)
        fs = lambda: ctx.apply_selfguided_restoration_batch(p, 0, q, 0, d_u, rec, n, d_i, d_x, U, U, d_f0, d_f1, U, U * U)
        common.ramp(ctx, fs, 0.05)
        sms = common.kernel_avg_ms(ctx, fs, 10)
        es = 2 if bd > 8 else 1
        print("%2d-bit, %3dx%-3d units (%d): wiener %.1f us (%.2f TB/s read+write), self-guided %.1f us" % (bd, U, U, n, wms * 1e3, 2.0 * W * H * es / wms / 1e9, sms * 1e3))
        for d in (d_u, d_f, d_i, d_x, d_f0, d_f1):
            ctx.free(d)
        ctx.planes_free(p); ctx.planes_free(q)
