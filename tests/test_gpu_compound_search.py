"""aomhip_refining_search_8p_batch / aomhip_obmc_full_pixel_search_batch (csrc/mcomp_compound.hip): the compound-reference refinement
(av1_refining_search_8p_c + av1_get_mvpred_compound_var, av1/encoder/mcomp.c:1621-1691, :3679-3693) and the OBMC full-pel search
(av1_obmc_full_pixel_search, :2110-2285) -- straight against the values obtained by interpreting the reference
(tests/golden/ref_eval_compound_search.npz), and against the oracle on whole batches of blocks, 8 / 10-bit, five block sizes."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_compound_search.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _tables(ctx, z):
    j, c0, c1 = z["mvjcost"].astype(np.int32), z["mvcost0"].astype(np.int32), z["mvcost1"].astype(np.int32)
    mv_max = c0.size // 2
    d_j, d_c0, d_c1 = ctx.to_device(j), ctx.to_device(c0), ctx.to_device(c1)
    return (d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4), (j, c0, c1)


def _planes(ctx, z, meta, bd):
    B, W, H = meta["border"], meta["width"], meta["height"]
    s, r = z["src%d" % bd], z["ref%d" % bd]
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, np.ascontiguousarray(s[B:B + H, B:B + W])); ctx.planes_upload(pr, 0, np.ascontiguousarray(r[B:B + H, B:B + W]))
    return ps, pr


def _blocks(capi, rows):
    b = np.zeros(len(rows), capi.search_block_dtype)
    for i, blk in enumerate(rows):
        for n, v in zip(("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max"), blk):
            b[n][i] = v
    return b


def test_both_searches_match_the_interpreted_reference(hip, ctx):
    capi = hip.capi
    z, meta = _load()
    (d_j, d_c0, d_c1), _ = _tables(ctx, z)
    planes = {bd: _planes(ctx, z, meta, bd) for bd in (8, 10)}
    n8 = nob = 0
    for c in meta["cases"]:
        ps, pr = planes[c["bd"]]
        dt = np.uint8 if c["bd"] == 8 else np.uint16
        d_b = ctx.to_device(_blocks(capi, [c["block"]]))
        d_mv, d_a, d_v = ctx.malloc(16), ctx.malloc(16), ctx.malloc(16)
        k = c["k"]
        if c["kind"] == "refine8p":
            d_sp = ctx.to_device(np.ascontiguousarray(z["sp%d" % k].astype(dt)))
            d_m = ctx.to_device(np.ascontiguousarray(z["mask%d" % k])) if c["masked"] else None
            ctx.refining_search_8p_batch(ps, pr, 0, c["w"], c["h"], c["cost_type"], c["sad_per_bit"], c["error_per_bit"], d_b, 1, d_sp, d_m, c["inv"], d_mv, d_a,
                                         d_v, d_j, d_c0, d_c1)
            got = (ctx.from_device(d_mv, (2,), np.int16).tolist(), int(ctx.from_device(d_a, (1,), np.int32)[0]), int(ctx.from_device(d_v, (1,), np.int32)[0]))
            assert got == (c["mv"], c["sad"], c["var"]), c
            n8 += 1
            ctx.free(d_sp)
            if d_m:
                ctx.free(d_m)
        else:
            d_ws, d_om = ctx.to_device(np.ascontiguousarray(z["ws%d" % k])), ctx.to_device(np.ascontiguousarray(z["om%d" % k]))
            ctx.obmc_full_pixel_search_batch(pr, 0, c["w"], c["h"], c["method"], c["step_param"], c["fast"], c["cost_type"], c["sad_per_bit"], c["error_per_bit"],
                                             d_b, 1, d_ws, d_om, d_mv, d_a, d_j, d_c0, d_c1)
            got = (ctx.from_device(d_mv, (2,), np.int16).tolist(), int(ctx.from_device(d_a, (1,), np.int32)[0]))
            assert got == (c["mv"], c["cost"]), c
            nob += 1
            ctx.free(d_ws); ctx.free(d_om)
        for d in (d_b, d_mv, d_a, d_v):
            ctx.free(d)
    assert n8 >= 40 and nob >= 32


@pytest.mark.parametrize("bd", [8, 10, 12])   # (12 bits with a mask: the 32-bit blend)
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (4, 4), (16, 8), (32, 16), (16, 32), (32, 32), (16, 64), (64, 64), (128, 128)])   # 4 / 2 candidates per wavefront, <= 512 px, <= 1024 px, streamed
def test_batches_match_the_oracle(hip, oracle, ctx, bd, bw, bh):
    capi = hip.capi
    W, H, B = 320, 192, 96
    rng = np.random.default_rng(1000 * bd + bw + bh)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 5, bd, shift=(2, -3), frac8=(3, 5))
    mx = (1 << bd) - 1
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3 << (bd - 8), (3 << (bd - 8)) + 1, ref.shape), 0, mx).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bw, H // bh
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bh
    blocks["start_row"], blocks["start_col"] = rng.integers(-6, 7, n), rng.integers(-6, 7, n)
    blocks["ref_row"], blocks["ref_col"] = rng.integers(-60, 61, n), rng.integers(-60, 61, n)
    ext = B - 8 - 16
    blocks["col_min"], blocks["col_max"] = np.maximum(-(blocks["bx"] + ext), -40), np.minimum(W - blocks["bx"] - bw + ext, 40)
    blocks["row_min"], blocks["row_max"] = np.maximum(-(blocks["by"] + ext), -40), np.minimum(H - blocks["by"] - bh + ext, 40)
    blocks["row_max"][::7] = 2; blocks["col_min"][::5] = -1          # tight limits: neighbours out of range, clamped starts
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    d_b = ctx.to_device(blocks)
    d_mv, d_a, d_v = ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 4)
    # second predictors: the reference block near the start MV, perturbed; masks: a ramp per block
    sp = np.zeros((n, bh, bw), src.dtype)
    for i in range(n):
        y, x = B + blocks["by"][i] + int(np.clip(blocks["start_row"][i], -8, 8)), B + blocks["bx"][i] + int(np.clip(blocks["start_col"][i], -8, 8))
        sp[i] = np.clip(rb[y:y + bh, x:x + bw].astype(np.int32) + rng.integers(-4 << (bd - 8), (4 << (bd - 8)) + 1, (bh, bw)), 0, mx)
    mask = np.clip((np.arange(bw)[None, None, :] * 64 // bw + rng.integers(-6, 7, (n, bh, bw))), 0, 64).astype(np.uint8)
    d_sp, d_m = ctx.to_device(sp), ctx.to_device(mask)
    for (m, inv, ct) in ((None, 0, capi.MV_COST_ENTROPY), (mask, 0, capi.MV_COST_L1_HDRES), (mask, 1, capi.MV_COST_NONE)):
        ctx.refining_search_8p_batch(ps, pr, 0, bw, bh, ct, 23, 71, d_b, n, d_sp, None if m is None else d_m, inv, d_mv, d_a, d_v, d_j, d_c0 + mv_max * 4,
                                     d_c1 + mv_max * 4)
        got = (ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_a, (n,), np.int32), ctx.from_device(d_v, (n,), np.int32))
        want = oracle.refining_search_8p_batch(sb, rb, B, bw, bh, blocks, sp, m, inv, cost_type=ct, sad_per_bit=23, error_per_bit=71, mvjcost=tj, mvcost0=t0,
                                               mvcost1=t1, bd=bd, threads=8)
        for g_, w_ in zip(got, want):
            assert np.array_equal(g_, w_)
    # OBMC: weighted source / mask as calc_target_weighted_pred builds them (mask = own weight x 64, 4096 where nothing overlaps)
    om = np.full((n, bh, bw), 4096, np.int64)
    om[:, :bh // 2, :] = (np.linspace(36, 64, bh // 2).astype(np.int64) * 64)[None, :, None]
    nb = np.clip(sp.astype(np.int64) + rng.integers(-8 << (bd - 8), (8 << (bd - 8)) + 1, sp.shape), 0, mx)
    sblk = np.stack([sb[B + b["by"]:B + b["by"] + bh, B + b["bx"]:B + b["bx"] + bw] for b in blocks]).astype(np.int64)
    ws = (sblk * 4096 - nb * (4096 - om)).astype(np.int32)
    om = om.astype(np.int32)
    d_ws, d_om = ctx.to_device(ws), ctx.to_device(om)
    for (method, sp_, fast, ct) in (("NSTEP", 4, 0, capi.MV_COST_ENTROPY), ("DIAMOND", 6, 0, capi.MV_COST_L1_HDRES), ("NSTEP", 0, 1, capi.MV_COST_L1_LOWRES)):
        ctx.obmc_full_pixel_search_batch(pr, 0, bw, bh, method, sp_, fast, ct, 19, 55, d_b, n, d_ws, d_om, d_mv, d_a, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
        got = (ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_a, (n,), np.int32))
        want = oracle.obmc_full_pixel_search_batch(rb, B, bw, bh, blocks, ws, om, method, sp_, fast, cost_type=ct, sad_per_bit=19, error_per_bit=55, mvjcost=tj,
                                                   mvcost0=t0, mvcost1=t1, bd=bd, threads=8)
        for g_, w_ in zip(got, want):
            assert np.array_equal(g_, w_)
    for d in (d_j, d_c0, d_c1, d_b, d_mv, d_a, d_v, d_sp, d_m, d_ws, d_om):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_bad_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p = ctx.planes_alloc(64, 64, 32, 8, 1)
    d = ctx.malloc(1024)
    with pytest.raises(capi.AomHipError):
        ctx.refining_search_8p_batch(p, p, 0, 16, 16, capi.MV_COST_ENTROPY, 1, 1, d, 1, d, None, 0, d, d, d)   # entropy costs without tables
    with pytest.raises(capi.AomHipError):
        ctx.obmc_full_pixel_search_batch(p, 0, 16, 16, "NSTEP", 99, 0, capi.MV_COST_NONE, 0, 0, d, 1, d, d, d, d)   # step_param past the table
    with pytest.raises(capi.AomHipError):
        ctx.obmc_full_pixel_search_batch(p, 0, 16, 12, "NSTEP", 0, 0, capi.MV_COST_NONE, 0, 0, d, 1, d, d, d, d)   # not a block size
    ctx.free(d); ctx.planes_free(p)
