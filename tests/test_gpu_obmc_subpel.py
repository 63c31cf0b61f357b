"""aomhip_obmc_subpel_tree_batch (csrc/mcomp_compound.hip): av1_find_best_obmc_sub_pixel_tree_up (av1/encoder/mcomp.c:3588-3633), both error forms
(USE_2_TAPS_ORIG: osvf + estimate_obmc_mvcost, centre at ref->buf; USE_8_TAPS: up-sampled prediction + ovf + mv_err_cost_) -- straight against the
values obtained by interpreting the reference (tests/golden/ref_eval_obmc_subpel.npz), and against the oracle on whole batches, 8 / 10-bit."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _tables(ctx, j, c0, c1):
    mv_max = c0.size // 2
    d_j, d_c0, d_c1 = ctx.to_device(j.astype(np.int32)), ctx.to_device(c0.astype(np.int32)), ctx.to_device(c1.astype(np.int32))
    return d_j, d_c0, d_c1, mv_max


@pytest.mark.parametrize("fixture,least", [("ref_eval_obmc_subpel.npz", 50), ("ref_eval_obmc_subpel_taps.npz", 24)])
def test_matches_the_interpreted_reference(hip, ctx, fixture, least):
    """(ref_eval_obmc_subpel_taps.npz: USE_4_TAPS / USE_2_TAPS)"""
    capi = hip.capi
    z = np.load(os.path.join(HERE, "golden", fixture))
    meta = json.loads(bytes(z["meta"]).decode())
    B, W, H = meta["border"], meta["width"], meta["height"]
    d_j, d_c0, d_c1, mv_max = _tables(ctx, z["mvjcost"], z["mvcost0"], z["mvcost1"])
    planes = {}
    for bd in (8, 10):
        planes[bd] = ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(planes[bd], 0, np.ascontiguousarray(z["ref%d" % bd][B:B + H, B:B + W]))
    n = 0
    for c in meta["cases"]:
        blk, lim, k = c["block"], c["subpel_limits"], c["k"]
        b = np.zeros(1, capi.search_block_dtype)
        for name, v in zip(("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max"),
                           (blk[0], blk[1], blk[2] * 8, blk[3] * 8, blk[4], blk[5], lim[0], lim[1], lim[2], lim[3])):
            b[name] = v
        p = capi.SubpelParams(2, c["cost_type"], c["error_per_bit"], c["iters"], c["allow_hp"], c["forced_stop"], c["subpel_search_type"])
        d_b, d_ws, d_om = ctx.to_device(b), ctx.to_device(np.ascontiguousarray(z["ws%d" % k])), ctx.to_device(np.ascontiguousarray(z["om%d" % k]))
        outs = [ctx.malloc(16) for _ in range(4)]
        ctx.obmc_subpel_tree_batch(planes[c["bd"]], 0, c["w"], c["h"], p, d_b, 1, d_ws, d_om, outs[0], outs[1], outs[2], outs[3], d_j, d_c0 + mv_max * 4,
                                   d_c1 + mv_max * 4)
        got = (ctx.from_device(outs[0], (2,), np.int16).tolist(), int(ctx.from_device(outs[1], (1,), np.uint32)[0]), int(ctx.from_device(outs[2], (1,), np.int32)[0]),
               int(ctx.from_device(outs[3], (1,), np.uint32)[0]))
        assert got == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        n += 1
        for d in [d_b, d_ws, d_om] + outs:
            ctx.free(d)
    assert n >= least
    for d in (d_j, d_c0, d_c1):
        ctx.free(d)
    for p_ in planes.values():
        ctx.planes_free(p_)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (4, 8), (32, 16), (8, 32), (64, 16), (32, 64), (64, 64), (128, 128)])   # (the up-sampled form's LDS strips: one, two, twelve per block)
def test_batches_match_the_oracle(hip, oracle, ctx, bd, bw, bh):
    capi = hip.capi
    W, H, B = 320, 192, 96
    rng = np.random.default_rng(77 * bd + bw + 3 * bh)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 9, bd, shift=(1, -2), frac8=(3, 6))
    mx = (1 << bd) - 1
    pr = ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bw, H // bh
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bh
    blocks["start_row"], blocks["start_col"] = rng.integers(-4, 5, n) * 8, rng.integers(-4, 5, n) * 8      # full-pel start, 1/8 pel units
    blocks["ref_row"], blocks["ref_col"] = rng.integers(-60, 61, n), rng.integers(-60, 61, n)
    blocks["row_min"], blocks["row_max"] = blocks["start_row"] - rng.integers(3, 40, n), blocks["start_row"] + rng.integers(3, 40, n)   # some tighter than the tree's reach
    blocks["col_min"], blocks["col_max"] = blocks["start_col"] - rng.integers(3, 40, n), blocks["start_col"] + rng.integers(3, 40, n)
    rb = oracle.extend_plane(ref, B, pr.stride)
    sb = oracle.extend_plane(src, B, pr.stride)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    d_j, d_c0, d_c1, _ = _tables(ctx, tj, t0, t1)
    om = np.full((n, bh, bw), 4096, np.int64)
    om[:, :bh // 2, :] = (np.linspace(36, 64, bh // 2).astype(np.int64) * 64)[None, :, None]
    om[:, :, :bw // 2] = np.minimum(om[:, :, :bw // 2], (np.linspace(34, 64, bw // 2).astype(np.int64) * 64)[None, None, :])
    sblk = np.stack([sb[B + b["by"]:B + b["by"] + bh, B + b["bx"]:B + b["bx"] + bw] for b in blocks]).astype(np.int64)
    nb = np.clip(sblk + rng.integers(-8 << (bd - 8), (8 << (bd - 8)) + 1, sblk.shape), 0, mx)
    ws = (sblk * 4096 - nb * (4096 - om)).astype(np.int32)
    om = om.astype(np.int32)
    d_b, d_ws, d_om = ctx.to_device(blocks), ctx.to_device(ws), ctx.to_device(om)
    outs = [ctx.malloc(n * 4) for _ in range(4)]
    moved = 0
    for (sst, ct, iters, hp, fs) in ((0, capi.MV_COST_ENTROPY, 2, 1, 0), (0, capi.MV_COST_NONE, 1, 0, 0), (3, capi.MV_COST_ENTROPY, 2, 1, 0),
                                     (3, capi.MV_COST_L1_HDRES, 2, 0, 1), (2, capi.MV_COST_ENTROPY, 2, 1, 0), (1, capi.MV_COST_L1_HDRES, 1, 1, 0)):
        p = capi.SubpelParams(2, ct, 63, iters, hp, fs, sst)
        ctx.obmc_subpel_tree_batch(pr, 0, bw, bh, p, d_b, n, d_ws, d_om, outs[0], outs[1], outs[2], outs[3], d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
        got = (ctx.from_device(outs[0], (n, 2), np.int16), ctx.from_device(outs[1], (n,), np.uint32), ctx.from_device(outs[2], (n,), np.int32),
               ctx.from_device(outs[3], (n,), np.uint32))
        want = oracle.obmc_subpel_tree_batch(rb, B, bw, bh, blocks, ws, om, cost_type=ct, error_per_bit=63, mvjcost=tj, mvcost0=t0, mvcost1=t1, iters_per_step=iters,
                                             allow_hp=hp, forced_stop=fs, subpel_search_type=sst, bd=bd, threads=8)
        for name, g_, w_ in zip(("mv", "err", "distortion", "sse"), got, want):
            assert np.array_equal(g_, w_), (sst, ct, name, np.flatnonzero((g_ != w_).reshape(n, -1).any(1))[:6])
        moved += int((want[0] != np.stack([blocks["start_row"], blocks["start_col"]], 1)).any(1).sum())
    assert moved > n
    # optional outputs, bad arguments
    ctx.obmc_subpel_tree_batch(pr, 0, bw, bh, capi.SubpelParams(2, capi.MV_COST_NONE, 0, 2, 1, 0, 0), d_b, n, d_ws, d_om, outs[0], outs[1])
    with pytest.raises(capi.AomHipError):
        ctx.obmc_subpel_tree_batch(pr, 0, bw, bh, capi.SubpelParams(2, capi.MV_COST_NONE, 0, 2, 1, 0, 4), d_b, n, d_ws, d_om, outs[0], outs[1])   # not a SUBPEL_SEARCH_TYPE
    with pytest.raises(capi.AomHipError):
        ctx.obmc_subpel_tree_batch(pr, 0, bw, bh, capi.SubpelParams(2, capi.MV_COST_ENTROPY, 1, 2, 1, 0, 0), d_b, n, d_ws, d_om, outs[0], outs[1])   # no tables
    for d in [d_j, d_c0, d_c1, d_b, d_ws, d_om] + outs:
        ctx.free(d)
    ctx.planes_free(pr)
