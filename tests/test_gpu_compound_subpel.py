"""aomhip_compound_subpel_tree_batch: the sub-pel trees on a compound prediction (ms_buffers.second_pred [/ mask / inv_mask]; svaf / msvf or the
comp_avg / comp_mask up-sampled error) -- straight against the values obtained by interpreting the reference (tests/golden/ref_eval_compound_subpel.npz),
and against the oracle on whole batches, 8 / 10-bit, three trees, both error forms."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _tables(ctx, j, c0, c1):
    mv_max = c0.size // 2
    return ctx.to_device(j.astype(np.int32)), ctx.to_device(c0.astype(np.int32)), ctx.to_device(c1.astype(np.int32)), mv_max


@pytest.mark.parametrize("fixture,least", [("ref_eval_compound_subpel.npz", 60), ("ref_eval_subpel_taps.npz", 32)])
def test_matches_the_interpreted_reference(hip, ctx, fixture, least):
    """(ref_eval_subpel_taps.npz: the tree with USE_4_TAPS / USE_2_TAPS, single-reference cases through aomhip_subpel_tree_batch)"""
    capi = hip.capi
    z = np.load(os.path.join(HERE, "golden", fixture))
    meta = json.loads(bytes(z["meta"]).decode())
    B, W, H = meta["border"], meta["width"], meta["height"]
    d_j, d_c0, d_c1, mv_max = _tables(ctx, z["mvjcost"], z["mvcost0"], z["mvcost1"])
    planes = {}
    for bd in (8, 10):
        ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(ps, 0, np.ascontiguousarray(z["src%d" % bd][B:B + H, B:B + W])); ctx.planes_upload(pr, 0, np.ascontiguousarray(z["ref%d" % bd][B:B + H, B:B + W]))
        planes[bd] = (ps, pr)
    n = 0
    for c in meta["cases"]:
        blk, lim, k = c["block"], c["subpel_limits"], c["k"]
        b = np.zeros(1, capi.search_block_dtype)
        for name, v in zip(("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max"),
                           (blk[0], blk[1], blk[2] * 8, blk[3] * 8, blk[4], blk[5], lim[0], lim[1], lim[2], lim[3])):
            b[name] = v
        p = capi.SubpelParams(c["tree"], c["cost_type"], c["error_per_bit"], c["iters"], c["allow_hp"], c["forced_stop"], c["subpel_search_type"])
        dt = np.uint8 if c["bd"] == 8 else np.uint16
        d_b, d_sp = ctx.to_device(b), ctx.to_device(np.ascontiguousarray(z["sp%d" % k].astype(dt)))
        d_m = ctx.to_device(np.ascontiguousarray(z["mask%d" % k])) if c["masked"] else None
        outs = [ctx.malloc(16) for _ in range(4)]
        ps, pr = planes[c["bd"]]
        if c.get("compound", 1):
            ctx.compound_subpel_tree_batch(ps, pr, 0, c["w"], c["h"], p, d_b, 1, d_sp, d_m, c["inv"], outs[0], outs[1], outs[2], outs[3], d_j, d_c0 + mv_max * 4,
                                           d_c1 + mv_max * 4)
        else:
            ctx.subpel_tree_batch(ps, pr, 0, c["w"], c["h"], p, d_b, 1, outs[0], outs[1], outs[2], outs[3], None, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
        got = (ctx.from_device(outs[0], (2,), np.int16).tolist(), int(ctx.from_device(outs[1], (1,), np.uint32)[0]), int(ctx.from_device(outs[2], (1,), np.int32)[0]),
               int(ctx.from_device(outs[3], (1,), np.uint32)[0]))
        assert got == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        n += 1
        for d in [d_b, d_sp] + ([d_m] if d_m is not None else []) + outs:
            ctx.free(d)
    assert n >= least
    for d in (d_j, d_c0, d_c1):
        ctx.free(d)
    for ps, pr in planes.values():
        ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (32, 16), (64, 64)])
def test_batches_match_the_oracle(hip, oracle, ctx, bd, bw, bh):
    capi = hip.capi
    W, H, B = 320, 192, 96
    rng = np.random.default_rng(91 * bd + bw + 5 * bh)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 13, bd, shift=(2, -1), frac8=(5, 2))
    mx = (1 << bd) - 1
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bw, H // bh
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bh
    blocks["start_row"], blocks["start_col"] = rng.integers(-4, 5, n) * 8, rng.integers(-4, 5, n) * 8
    blocks["ref_row"], blocks["ref_col"] = rng.integers(-60, 61, n), rng.integers(-60, 61, n)
    blocks["row_min"], blocks["row_max"] = blocks["start_row"] - rng.integers(3, 40, n), blocks["start_row"] + rng.integers(3, 40, n)
    blocks["col_min"], blocks["col_max"] = blocks["start_col"] - rng.integers(3, 40, n), blocks["start_col"] + rng.integers(3, 40, n)
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    d_j, d_c0, d_c1, _ = _tables(ctx, tj, t0, t1)
    sp = np.zeros((n, bh, bw), src.dtype)
    for i in range(n):
        y, x = B + blocks["by"][i] + blocks["start_row"][i] // 8, B + blocks["bx"][i] + blocks["start_col"][i] // 8
        sp[i] = np.clip(rb[y:y + bh, x:x + bw].astype(np.int32) + rng.integers(-4 << (bd - 8), (4 << (bd - 8)) + 1, (bh, bw)), 0, mx)
    mask = np.clip((np.arange(bw)[None, None, :] * 64 // bw + rng.integers(-6, 7, (n, bh, bw))), 0, 64).astype(np.uint8)
    d_b, d_sp, d_m = ctx.to_device(blocks), ctx.to_device(sp), ctx.to_device(mask)
    outs = [ctx.malloc(n * 4) for _ in range(4)]
    for (tree, sst, m, inv, ct, iters, hp, fs) in ((0, 0, None, 0, capi.MV_COST_ENTROPY, 2, 1, 0), (1, 0, mask, 1, capi.MV_COST_L1_HDRES, 2, 0, 0),
                                                   (2, 0, mask, 0, capi.MV_COST_NONE, 2, 1, 1), (2, 3, None, 0, capi.MV_COST_ENTROPY, 2, 1, 0),
                                                   (2, 3, mask, 1, capi.MV_COST_L1_HDRES, 1, 0, 0), (2, 2, None, 0, capi.MV_COST_ENTROPY, 2, 1, 0),
                                                   (2, 2, mask, 1, capi.MV_COST_L1_HDRES, 2, 0, 0), (2, 1, mask, 0, capi.MV_COST_NONE, 1, 1, 0)):
        p = capi.SubpelParams(tree, ct, 63, iters, hp, fs, sst)
        ctx.compound_subpel_tree_batch(ps, pr, 0, bw, bh, p, d_b, n, d_sp, None if m is None else d_m, inv, outs[0], outs[1], outs[2], outs[3], d_j,
                                       d_c0 + mv_max * 4, d_c1 + mv_max * 4)
        got = (ctx.from_device(outs[0], (n, 2), np.int16), ctx.from_device(outs[1], (n,), np.uint32), ctx.from_device(outs[2], (n,), np.int32),
               ctx.from_device(outs[3], (n,), np.uint32))
        want = oracle.compound_subpel_tree_batch(sb, rb, B, bw, bh, blocks, sp, m, inv, tree=tree, subpel_search_type=sst, cost_type=ct, error_per_bit=63, mvjcost=tj,
                                                 mvcost0=t0, mvcost1=t1, iters_per_step=iters, allow_hp=hp, forced_stop=fs, bd=bd, threads=8)
        for name, g_, w_ in zip(("mv", "err", "distortion", "sse"), got, want):
            assert np.array_equal(g_, w_), (tree, sst, name, np.flatnonzero((g_ != w_).reshape(n, -1).any(1))[:6])
    with pytest.raises(capi.AomHipError):
        ctx.compound_subpel_tree_batch(ps, pr, 0, bw, bh, capi.SubpelParams(2, capi.MV_COST_NONE, 0, 2, 1, 0, 0), d_b, n, None, None, 0, outs[0], outs[1], outs[2], outs[3])
    for d in [d_j, d_c0, d_c1, d_b, d_sp, d_m] + outs:
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
