#!/usr/bin/env python3
"""round 5: the coefficient-rate kernels on a 4K frame's worth of 16x16 transform blocks (32 400): av1_txb_init_levels + av1_get_nz_map_contexts and
av1_cost_coeffs_txb; HIP-event averages, one JSON line (gpurun -- python3 tools/r05_l.py)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import aom_av1_psy_amd as pkg  # noqa: E402

ctx = pkg.capi.Context(0, None)
rng = np.random.default_rng(1)
out = {}
for tx_size, w in ((2, 16), (1, 8), (3, 32)):
    n = w * w
    nb = (3840 // w) * (2160 // w)
    # 512 template blocks with coefficients below a per-block bound in raster order (plausible sparse blocks), tiled over the frame
    tmpl = np.zeros((512, n), np.int32)
    for i in range(512):
        e = int(rng.integers(1, n // 3))
        tmpl[i, :e] = rng.choice([0, 1, 1, 2, 3, 8, 20], e) * rng.choice([-1, 1], e)
    coeff = np.ascontiguousarray(tmpl[np.arange(nb) % 512])
    d_q, d_e = ctx.to_device(coeff), ctx.to_device(np.full(nb, n // 4, np.uint16))
    d_x, d_t, d_o = ctx.to_device(np.zeros(2 * nb, np.uint8)), ctx.to_device(rng.integers(1, 5000, 966).astype(np.int32)), ctx.malloc(4 * nb)
    lp = (w + 4) * (w + 4) + 16
    d_l, d_c = ctx.malloc(lp * nb), ctx.malloc(n * nb)
    ms_cost = bench.kernel_avg_ms(ctx, lambda: ctx.cost_coeffs_txb_batch(d_q, tx_size, None, nb, 0, d_e, d_x, d_t, d_o), 20)
    ms_lv = bench.kernel_avg_ms(ctx, lambda: ctx.txb_init_levels_batch(d_q, w, w, None, nb, d_l, lp), 20)
    ms_nz = bench.kernel_avg_ms(ctx, lambda: ctx.get_nz_map_contexts_batch(d_l, lp, tx_size, None, nb, 0, d_e, d_c, n), 20)
    out["%dx%d" % (w, w)] = {"blocks": nb, "cost_coeffs_txb_ms": ms_cost, "blocks_per_s": nb / ms_cost * 1e3, "coeff_GBps": 4.0 * n * nb / ms_cost / 1e6,
                             "txb_init_levels_ms": ms_lv, "get_nz_map_contexts_ms": ms_nz}
    for d in (d_q, d_e, d_x, d_t, d_o, d_l, d_c):
        ctx.free(d)
ctx.close()
print(json.dumps({"workload": "coefficient_rate_4k_frame", **out}))
