"""aomhip_compute_stats_batch (av1_compute_stats / av1_compute_stats_highbd, av1/encoder/pickrst.c:948-1083) through the
C ABI: against the interpreted reference's vectors and against the oracle for lists of restoration units -- 7x7 and 5x5
windows, 8 / 10 / 12-bit, the down-sampled mode, units at the frame edges (the window reads the extended border), odd
sizes and the 256-wide maximum."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(hip, ctx, dgd, src, bd, win, rects, downsample, border=16):
    H, W = dgd.shape
    pd, ps = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pd, 0, dgd); ctx.planes_upload(ps, 0, src)
    units = np.zeros(len(rects), hip.capi.rect_dtype)
    for i, r in enumerate(rects):
        units[i] = tuple(r)
    win2 = win * win
    d_u, d_M, d_H = ctx.to_device(units), ctx.malloc(8 * win2 * len(rects)), ctx.malloc(8 * win2 * win2 * len(rects))
    ctx.compute_stats_batch(pd, 0, ps, 0, win, d_u, units, len(rects), downsample, d_M, d_H)
    M, Hm = ctx.from_device(d_M, (len(rects), win2), np.int64), ctx.from_device(d_H, (len(rects), win2 * win2), np.int64)
    for d in (d_u, d_M, d_H):
        ctx.free(d)
    ctx.planes_free(pd); ctx.planes_free(ps)
    return M, Hm


def test_wiener_stats_goldens(hip, ctx):
    z = np.load(os.path.join(GOLD, "ref_eval_lrstats.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    assert len(cases) == 8
    for c in cases:
        bd = c["bd"]
        dt = np.uint8 if bd == 8 else np.uint16
        dgd, src = np.ascontiguousarray(z["dgd%d" % bd], dt), np.ascontiguousarray(z["src%d" % bd], dt)
        M, Hm = _run(hip, ctx, dgd, src, bd, c["win"], [c["rect"]], c["downsample"])
        assert np.array_equal(M[0], z["M%d" % c["k"]]) and np.array_equal(Hm[0], z["H%d" % c["k"]]), c


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_wiener_stats_vs_oracle(hip, oracle, ctx, bd):
    rng = np.random.default_rng(bd)
    W, H, border = 384, 160, 16
    dgd = hip.synth.lcg_frame(W, H, 6, 0, bd)
    src = np.clip(dgd.astype(np.int64) + rng.integers(-(8 << (bd - 8)), (8 << (bd - 8)) + 1, (H, W)), 0, (1 << bd) - 1).astype(dgd.dtype)
    dgd[:40, :70] = np.where(rng.integers(0, 2, (40, 70)) > 0, (1 << bd) - 1, 0).astype(dgd.dtype)      # extreme content: the largest sums
    rects = [(0, 64, 0, 64), (64, 320, 0, 64), (320, 384, 0, 37), (0, 96, 64, 160), (96, 352, 64, 160), (352, 384, 64, 160), (5, 18, 7, 12),
             (100, 101, 50, 51), (128, 384, 96, 160), (0, 250, 3, 36)]
    db, sb = oracle.extend_plane(dgd, border), oracle.extend_plane(src, border)
    f = oracle.lib.orc_compute_stats
    f.restype = None
    for win in (7, 5):
        for ds in ((0, 1) if bd == 8 else (0,)):
            M, Hm = _run(hip, ctx, dgd, src, bd, win, rects, ds, border)
            wm, wh = np.zeros(win * win, np.int64), np.zeros(win ** 4, np.int64)
            for i, (hs, he, vs, ve) in enumerate(rects):
                f(win, C.c_void_p(oracle._addr(db, border, border)), C.c_void_p(oracle._addr(sb, border, border)), hs, he, vs, ve, db.shape[1], sb.shape[1],
                  int(bd > 8), bd, ds, C.c_void_p(wm.ctypes.data), C.c_void_p(wh.ctypes.data))
                assert np.array_equal(M[i], wm) and np.array_equal(Hm[i], wh), (bd, win, ds, i)
            assert np.array_equal(Hm.reshape(len(rects), win * win, win * win), Hm.reshape(len(rects), win * win, win * win).transpose(0, 2, 1))


def test_wiener_stats_rejects_bad_arguments(hip, ctx):
    K = hip.capi
    p8, p10 = ctx.planes_alloc(64, 64, 16, 8, 1), ctx.planes_alloc(64, 64, 16, 10, 1)
    thin = ctx.planes_alloc(64, 64, 2, 8, 1)
    d = ctx.malloc(8 * 2401)
    ok = np.zeros(1, K.rect_dtype); ok[0] = (0, 64, 0, 64)
    out = np.zeros(1, K.rect_dtype); out[0] = (0, 65, 0, 64)
    d_u = ctx.to_device(ok)
    with pytest.raises(K.AomHipError):
        ctx.compute_stats_batch(p8, 0, p8, 0, 3, d_u, ok, 1, 0, d, d)        # window
    with pytest.raises(K.AomHipError):
        ctx.compute_stats_batch(p10, 0, p10, 0, 7, d_u, ok, 1, 1, d, d)      # down-sampled mode is 8-bit only
    with pytest.raises(K.AomHipError):
        ctx.compute_stats_batch(thin, 0, thin, 0, 7, d_u, ok, 1, 0, d, d)    # border too small for the window
    with pytest.raises(K.AomHipError):
        ctx.compute_stats_batch(p8, 0, p8, 0, 7, d_u, out, 1, 0, d, d)       # unit leaves the plane
    with pytest.raises(K.AomHipError):
        ctx.compute_stats_batch(p8, 0, p10, 0, 7, d_u, ok, 1, 0, d, d)       # bit depths differ
    ctx.compute_stats_batch(p8, 0, p8, 0, 7, None, None, 0, 0, d, d)
    ctx.free(d); ctx.free(d_u)
    for p in (p8, p10, thin):
        ctx.planes_free(p)
