"""aomhip_estimate_txfm_yrd_batch (csrc/txfm_yrd.hip) against av1_estimate_txfm_yrd interpreted with its callees (tests/golden/ref_eval_yrd.npz: every
case, one call each -- the cost tables are per case) and against the oracle's restatement on batches of random blocks with one table set."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import pyoracle as orc
from test_golden_yrd import load, oracle_yrd

pytestmark = pytest.mark.gpu


def planes_for(ctx, residuals, positions, W, H, border, bd):
    """src / pred planes whose difference at positions[i] is residuals[i] (src - pred = residual exactly, both inside the pixel range)."""
    mid = 1 << (bd - 1)
    dt = np.uint8 if bd == 8 else np.uint16
    src, pred = np.full((H, W), mid, np.int32), np.full((H, W), mid, np.int32)
    for r, (x, y) in zip(residuals, positions):
        h, w = r.shape
        p = mid - (r.astype(np.int32) >> 1)
        pred[y:y + h, x:x + w] = p
        src[y:y + h, x:x + w] = p + r
    assert src.min() >= 0 and src.max() < (1 << bd) and pred.min() >= 0 and pred.max() < (1 << bd)
    ps, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src.astype(dt)); ctx.planes_upload(pp, 0, pred.astype(dt))
    return ps, pp


def test_every_interpreted_case(hip, ctx):
    capi = hip.capi
    z, cases = load()
    for c in cases:
        k, bw, bh, bd = c["k"], c["bw"], c["bh"], c["bd"]
        res = z["res%d" % k]
        ps, pp = planes_for(ctx, [res], [(32, 16)], 192, 160, 32, bd)
        b = np.zeros(1, capi.txfm_yrd_block_dtype)
        b["bx"], b["by"] = 32, 16
        b["tx_size_rate"], b["no_skip_txfm_rate"], b["skip_txfm_rate"] = c["tx_size_rate"], c["no_skip_txfm_rate"], c["skip_txfm_rate"]
        b["above_ctx"][0, :bw // 4] = z["above%d" % k]; b["left_ctx"][0, :bh // 4] = z["left%d" % k]
        d_b, d_c, d_s = ctx.to_device(b), ctx.to_device(np.ascontiguousarray(z["costs%d" % k], np.int32)), ctx.malloc(32)
        qp = capi.QuantParams.from_tables(orc.build_quantizer_y(bd, c["qindex"]))
        ctx.estimate_txfm_yrd_batch(ps, pp, 0, bw, bh, qp, d_c, c["tx_type_rate"], c["rdmult"], 0, d_b, 1, d_s)
        s = ctx.from_device(d_s, (1,), capi.txfm_yrd_stats_dtype)[0]
        assert (int(s["rd"]), int(s["rate"]), int(s["skip_txfm"]), int(s["dist"]), int(s["sse"])) == \
               (int(c["rd"]), c["rate"], c["skip_txfm"], int(c["dist"]), int(c["sse"])), c
        for d in (d_b, d_c, d_s):
            ctx.free(d)
        ctx.planes_free(ps); ctx.planes_free(pp)


@pytest.mark.parametrize("bw,bh,bd", [(16, 16, 8), (8, 32, 10), (64, 64, 10), (128, 128, 8), (64, 128, 12), (4, 4, 8), (32, 16, 12)])
def test_batches_against_the_oracle(hip, ctx, bw, bh, bd):
    capi = hip.capi
    rng = np.random.default_rng(bw * 7 + bh + bd)
    n = 24 if bw * bh <= 4096 else 6
    W, H, border = 1024, 512, 32
    cols = W // max(bw, 16)
    positions = [((i % cols) * max(bw, 16), (i // cols) * max(bh, 16)) for i in range(n)]
    amp = 40 << (bd - 8)
    residuals = []
    for i in range(n):
        a = [amp, amp // 6, 2 << (bd - 8)][i % 3]
        r = rng.integers(-a, a + 1, (bh, bw)) + (np.add.outer(np.arange(bh), np.arange(bw)) % 5) * (a // 4)
        residuals.append(np.clip(r, -(1 << bd) + 1, (1 << bd) - 1).astype(np.int16))
    ps, pp = planes_for(ctx, residuals, positions, W, H, border, bd)
    qindex = int(rng.choice([30, 110, 200]))
    q = orc.build_quantizer_y(bd, qindex)
    costs = rng.integers(10, 3000, 966).astype(np.int32)
    b = np.zeros(n, capi.txfm_yrd_block_dtype)
    b["bx"], b["by"] = [p[0] for p in positions], [p[1] for p in positions]
    b["tx_size_rate"], b["no_skip_txfm_rate"], b["skip_txfm_rate"] = rng.integers(0, 2000, n), rng.integers(20, 2000, n), rng.integers(20, 2000, n)
    b["above_ctx"] = rng.integers(0, 7, (n, 32)) | (rng.integers(0, 3, (n, 32)) << 3)
    b["left_ctx"] = rng.integers(0, 7, (n, 32)) | (rng.integers(0, 3, (n, 32)) << 3)
    tx_type_rate = 0 if max(bw, bh) > 32 else 333
    rdmult = int(rng.choice([60, 700, 30000]))
    for lossless in (0, 1):
        d_b, d_c, d_s = ctx.to_device(b), ctx.to_device(costs), ctx.malloc(32 * n)
        ctx.estimate_txfm_yrd_batch(ps, pp, 0, bw, bh, capi.QuantParams.from_tables(q), d_c, tx_type_rate, rdmult, lossless, d_b, n, d_s)
        got = ctx.from_device(d_s, (n,), capi.txfm_yrd_stats_dtype)
        if lossless and bw == 4:   # (the Walsh-Hadamard transform of lossless 4x4 blocks: the oracle form below is the DCT one)
            continue
        tabs = np.ascontiguousarray(np.stack([np.asarray(q[k_], np.int16)[:2] for k_ in ("zbin", "round", "quant", "quant_shift", "dequant")]))
        f = orc.lib.orc_estimate_txfm_yrd
        f.restype = C.c_int64
        skipped = forced = 0
        for i in range(n):
            res = np.ascontiguousarray(residuals[i])
            out = np.zeros(4, np.int64)
            ab, lf = np.ascontiguousarray(b["above_ctx"][i]), np.ascontiguousarray(b["left_ctx"][i])
            rd = f(C.c_void_p(res.ctypes.data), bw, bw, bh, bd, int(bd > 8), C.c_void_p(tabs.ctypes.data), C.c_void_p(ab.ctypes.data), C.c_void_p(lf.ctypes.data),
                   C.c_void_p(costs.ctypes.data), tx_type_rate, int(b["tx_size_rate"][i]), int(b["no_skip_txfm_rate"][i]), int(b["skip_txfm_rate"][i]), rdmult, lossless,
                   C.c_void_p(out.ctypes.data))
            g = got[i]
            assert (int(g["rd"]), int(g["rate"]), int(g["skip_txfm"]), int(g["dist"]), int(g["sse"])) == (int(rd), int(out[0]), int(out[1]), int(out[2]), int(out[3])), (i, lossless)
            skipped += int(out[1])
        for d in (d_b, d_c, d_s):
            ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pp)


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p8, p10 = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 10, 1)
    d = ctx.malloc(4096)
    qp = capi.QuantParams.from_tables(orc.build_quantizer_y(8, 100))
    with pytest.raises(capi.AomHipError):
        ctx.estimate_txfm_yrd_batch(p8, p10, 0, 16, 16, qp, d, 0, 100, 0, d, 1, d)      # bit depths differ
    with pytest.raises(capi.AomHipError):
        ctx.estimate_txfm_yrd_batch(p8, p8, 1, 16, 16, qp, d, 0, 100, 0, d, 1, d)       # no such frame
    with pytest.raises(capi.AomHipError):
        ctx.estimate_txfm_yrd_batch(p8, p8, 0, 16, 16, qp, None, 0, 100, 0, d, 1, d)    # no cost tables
    ctx.free(d)
    ctx.planes_free(p8); ctx.planes_free(p10)
