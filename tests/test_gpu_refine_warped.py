"""aomhip_refine_warped_mv_batch (csrc/warp_refine.hip) against av1_refine_warped_mv interpreted as written (tests/golden/ref_eval_refine_warped.npz:
one call per case) and against the oracle's composition on dense batches of blocks with random neighbourhoods (8 / 10-bit, four block sizes, every MV
cost type, both precisions, tight limits, single-sample blocks)."""
import numpy as np
import pytest

from test_golden_refine_warped import COST, block_of, load

pytestmark = pytest.mark.gpu


def tables(ctx, j, c0, c1):
    d_j, d_c0, d_c1 = ctx.to_device(np.ascontiguousarray(j, np.int32)), ctx.to_device(np.ascontiguousarray(c0, np.int32)), ctx.to_device(np.ascontiguousarray(c1, np.int32))
    return (d_j, d_c0, d_c1), (d_j, d_c0 + (c0.size // 2) * 4, d_c1 + (c1.size // 2) * 4)


def test_device_equals_the_interpreted_function(hip, ctx):
    capi = hip.capi
    z, meta = load()
    B, W, H = meta["border"], meta["width"], meta["height"]
    own, tabs = tables(ctx, z["mvjcost"], z["mvcost0"], z["mvcost1"])
    planes = {}
    for bd in (8, 10):
        ps, pr, pp = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 4)
        ctx.planes_upload(ps, 0, np.ascontiguousarray(z["src%d" % bd][B:B + H, B:B + W])); ctx.planes_upload(pr, 0, np.ascontiguousarray(z["ref%d" % bd][B:B + H, B:B + W]))
        planes[bd] = (ps, pr, pp)
    for c in meta["cases"]:
        ps, pr, pp = planes[c["bd"]]
        d_b, d_r = ctx.to_device(block_of(c, capi.warp_refine_block_dtype)), ctx.malloc(64)
        ctx.refine_warped_mv_batch(ps, pr, 0, pp, c["w"], c["h"], c["allow_hp"], COST[c["cost_type"]], c["error_per_bit"], d_b, 1, d_r, *tabs)
        r = ctx.from_device(d_r, (1,), capi.warp_refine_result_dtype)[0]
        got = ([int(r["mv_row"]), int(r["mv_col"])], int(r["bestmse"]), int(r["num_proj_ref"]), r["model"]["mat"].tolist(),
               [int(r["model"][f]) for f in ("alpha", "beta", "gamma", "delta")])
        assert got == (c["best_mv"], c["bestmse"], c["best_num_proj_ref"], c["best_model"]["mat"], c["best_model"]["shear"]), (c["k"], got)
        ctx.free(d_b); ctx.free(d_r)
    for d in own:
        ctx.free(d)
    for t in planes.values():
        for p in t:
            ctx.planes_free(p)


@pytest.mark.parametrize("bd,bw,bh,cost,allow_hp", [(8, 16, 16, "ENTROPY", 1), (10, 8, 8, "L1_HDRES", 1), (8, 32, 16, "NONE", 0), (10, 64, 64, "L1_LOWRES", 1),
                                                    (10, 16, 32, "ENTROPY", 0), (8, 128, 128, "L1_MIDRES", 1)])
def test_dense_batches_equal_the_oracle_composition(hip, oracle, ctx, bd, bw, bh, cost, allow_hp):
    capi = hip.capi
    W, H, B = 512, 384, 64
    rng = np.random.default_rng(bd * 100 + bw + bh + allow_hp)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 7, bd, shift=(2, -1), frac8=(3, 6))
    k = 6 << (bd - 8)
    src = np.clip(src.astype(np.int32) + rng.integers(-k, k + 1, src.shape), 0, (1 << bd) - 1).astype(src.dtype)
    ps, pr, pp = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 4)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bw, H // bh
    recs = []
    for i in range(gc * gr):
        bx, by = (i % gc) * bw, (i // gc) * bh
        mv = [int(rng.integers(-30, 31)), int(rng.integers(-30, 31))]
        n = 1 if i % 6 == 0 else int(rng.integers(2, 9))
        pts = np.zeros((n, 2), np.int64)
        pts[:, 0] = rng.integers(-8 * 48, 8 * (bw + 24), n); pts[:, 1] = rng.integers(-8 * 48, 8 * (bh + 24), n)
        a = np.array([[1.0 + rng.normal(0, 0.02), rng.normal(0, 0.02)], [rng.normal(0, 0.02), 1.0 + rng.normal(0, 0.02)]])
        ctr = np.array([bw * 4.0, bh * 4.0])
        pin = np.rint((pts - ctr) @ a.T + ctr + np.array([mv[1], mv[0]]) + rng.normal(0, 2.0, (n, 2))).astype(np.int64)
        sp, sq = np.ascontiguousarray(pts.ravel(), np.int32), np.ascontiguousarray(pin.ravel(), np.int32)
        np0 = capi.select_samples(mv, sp, sq, n, bw, bh) if n > 1 else 1     # (host entry points: what the caller prepares, as motion_mode_rd does)
        m = np.zeros(1, capi.warp_model_dtype)
        m["mat"][0] = [0, 0, 1 << 16, 0, 0, 1 << 16]
        if not capi.find_projection(np0, sp, sq, bw, bh, mv, m, by // 4, bx // 4):
            continue
        b = np.zeros(1, capi.warp_refine_block_dtype)
        b["bx"], b["by"], b["mv_row"], b["mv_col"] = bx, by, mv[0], mv[1]
        b["ref_row"], b["ref_col"] = mv[0] + int(rng.integers(-9, 10)), mv[1] + int(rng.integers(-9, 10))
        lim = (3, 3, 3, 3) if i % 5 == 0 else (64, 64, 64, 64)
        b["col_min"], b["col_max"], b["row_min"], b["row_max"] = mv[1] - int(rng.integers(0, lim[0] + 1)), mv[1] + int(rng.integers(0, lim[1] + 1)), \
            mv[0] - int(rng.integers(0, lim[2] + 1)), mv[0] + int(rng.integers(0, lim[3] + 1))
        b["total_samples"], b["num_proj_ref"] = n, np0
        b["pts"][0, :2 * n], b["pts_inref"][0, :2 * n] = pts.ravel(), pin.ravel()
        b["model"] = m
        recs.append(b)
    blocks = np.concatenate(recs)
    n = len(blocks)
    assert n >= min(12, gc * gr // 2)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (150 + bits * 310 + (v & 7) * 11).astype(np.int32), (170 + bits * 290 + (v & 7) * 7).astype(np.int32)
    tj = np.array([200, 650, 640, 1050], np.int32)
    own, tabs = tables(ctx, tj, t0, t1)
    d_b, d_r = ctx.to_device(blocks), ctx.malloc(n * 44)
    ctx.refine_warped_mv_batch(ps, pr, 0, pp, bw, bh, allow_hp, COST[cost], 77, d_b, n, d_r, *tabs)
    got = ctx.from_device(d_r, (n,), capi.warp_refine_result_dtype)
    moved = 0
    for i in range(n):
        w = oracle.refine_warped_mv(src, ref, bd, bw, bh, blocks[i], allow_hp, COST[cost], 77, tj, t0, t1)
        g = got[i]
        assert ([int(g["mv_row"]), int(g["mv_col"])], int(g["bestmse"]), int(g["num_proj_ref"]), g["model"]["mat"].tolist(),
                [int(g["model"][f]) for f in ("alpha", "beta", "gamma", "delta")]) == (w["mv"], w["bestmse"], w["num_proj_ref"], w["mat"], w["shear"]), (i, g, w)
        moved += w["mv"] != [int(blocks["mv_row"][i]), int(blocks["mv_col"][i])]
    assert 0 < moved
    for d in list(own) + [d_b, d_r]:
        ctx.free(d)
    for p in (ps, pr, pp):
        ctx.planes_free(p)


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p1, p4, q4 = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 8, 4), ctx.planes_alloc(64, 64, 32, 10, 4)
    d = ctx.malloc(4096)
    for args in ((p1, p1, 0, p1, 16, 16), (p1, p1, 0, q4, 16, 16), (p1, p1, 0, p4, 4, 16), (p1, p1, 1, p4, 16, 16)):     # ring of one frame; other depth; 4 wide; no such frame
        with pytest.raises(capi.AomHipError):
            ctx.refine_warped_mv_batch(*args, 1, capi.MV_COST_NONE, 0, d, 1, d)
    with pytest.raises(capi.AomHipError):
        ctx.refine_warped_mv_batch(p1, p1, 0, p4, 16, 16, 1, capi.MV_COST_ENTROPY, 10, d, 1, d)      # entropy costs without tables
    ctx.free(d)
    for p in (p1, p4, q4):
        ctx.planes_free(p)
