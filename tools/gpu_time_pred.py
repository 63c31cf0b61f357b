#!/usr/bin/env python3
"""Times the prediction kernels on a 4K 10-bit luma plane, 32 400 16x16 blocks with random 1/8-pel MVs (HIP events, 20 launches
each): single reference, compound average, distance-weighted, masked and diff-weighted.  gpurun -- 'python tools/gpu_time_pred.py'"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aom_av1_psy_amd as pkg  # noqa: E402
capi = pkg.capi
ctx = capi.Context(0)
W, H, bd, border = 3840, 2160, 10, 64
r0, r1, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
ctx.planes_upload(r0, 0, pkg.synth.lcg_frame(W, H, 1, 0, bd)); ctx.planes_upload(r1, 0, pkg.synth.lcg_frame(W, H, 2, 1, bd))
rng = np.random.default_rng(1)
xs, ys = np.meshgrid(np.arange(0, W, 16), np.arange(0, H, 16))
n = xs.size
blocks = np.zeros(n, capi.search_block_dtype)
blocks["bx"], blocks["by"] = xs.ravel(), ys.ravel()
mv0, mv1 = rng.integers(-256, 257, (n, 2)).astype(np.int16), rng.integers(-256, 257, (n, 2)).astype(np.int16)
mask = rng.integers(0, 65, (n, 16, 16)).astype(np.uint8)
moff = (np.arange(n) * 256).astype(np.uint32)
d_b, d_0, d_1, d_m, d_o, d_mo = (ctx.to_device(a) for a in (blocks, mv0, mv1, mask, moff, np.zeros(n * 256, np.uint8)))


def t(fn, reps=20):
    fn(); ctx.sync(); ctx.timer_begin()
    for _ in range(reps):
        fn()
    return ctx.timer_end() / reps


rows = [("single reference (inter_pred_kernel)", lambda: ctx.build_inter_pred_batch(r0, 0, pp, 0, 16, 16, d_b, d_0, n, 0, 0)),
        ("compound average", lambda: ctx.build_compound_pred_batch(r0, 0, r1, 0, pp, 0, 16, 16, d_b, d_0, d_1, n, 0, 0)),
        ("compound distance-weighted 9/7", lambda: ctx.build_compound_pred_batch(r0, 0, r1, 0, pp, 0, 16, 16, d_b, d_0, d_1, n, 0, 0, 9, 7)),
        ("masked compound (mask per block)", lambda: ctx.build_masked_compound_pred_batch(r0, 0, r1, 0, pp, 0, 16, 16, d_b, d_0, d_1, n, 0, 0, d_m, d_o, 16)),
        ("diff-weighted compound (+ mask out)", lambda: ctx.build_diffwtd_compound_pred_batch(r0, 0, r1, 0, pp, 0, 16, 16, d_b, d_0, d_1, n, 0, 0, 0, d_mo))]
for name, fn in rows:
    print("%-40s %.4f ms" % (name, t(fn)))
