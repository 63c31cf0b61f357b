"""aomhip_subpel_tree_list_batch against the interpreted reference (tests/golden/ref_eval_single.npz: the sub-pel trees on a
last_mv_search_list, where they stop and what they leave) and aomhip_single_motion_search_batch (csrc/tf_search.hip: two start candidates,
second-MV refinement, rate) against the oracle's composition of av1_single_motion_search's SIMPLE_TRANSLATION core."""
import ctypes as C

import numpy as np
import pytest

from test_golden_single import TREE, fixture, subpel_block
from test_oracle_fp import block_of

pytestmark = pytest.mark.gpu
TREE_ID = {"pruned_more": 0, "pruned": 1, "tree": 2}


def centre_ptr(ctx, table):
    t = np.ascontiguousarray(table, np.int32)
    d = ctx.to_device(t)
    return d, d + (t.size // 2) * 4


def test_device_trees_on_a_search_list_equal_the_interpreted_reference(hip, ctx):
    capi = hip.capi
    z, meta = fixture()
    B, W, H = meta["border"], meta["W"], meta["H"]
    d_j = ctx.to_device(np.ascontiguousarray(z["mvjcost"], np.int32))
    d_c0, c0 = centre_ptr(ctx, z["mvcost0"])
    d_c1, c1 = centre_ptr(ctx, z["mvcost1"])
    planes = {}
    for bd in (8, 10):
        ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(ps, 0, z["src%d" % bd][B:B + H, B:B + W]); ctx.planes_upload(pr, 0, z["ref%d" % bd][B:B + H, B:B + W])
        planes[bd] = (ps, pr)
    d_mv, d_err, d_dist, d_sse, d_cl, d_l = (ctx.malloc(64) for _ in range(6))
    n_stopped = 0
    for c in meta["cases"]:
        ps, pr = planes[c["bd"]]
        sp = capi.SubpelParams(TREE_ID[TREE[c["fn"]]], c["cost_type"], c["error_per_bit"], c["iters"], c["allow_hp"], c["forced_stop"], 0)
        ctx.memcpy_h2d(d_cl, np.asarray(c["cost_list"], np.int32))
        for call in c["calls"]:
            ctx.memcpy_h2d(d_l, np.asarray(call["list_before"], np.int16))
            d_b = ctx.to_device(subpel_block(c, call["start"], capi.search_block_dtype))
            ctx.subpel_tree_batch(ps, pr, 0, c["w"], c["h"], sp, d_b, 1, d_mv, d_err, d_dist, d_sse, d_cl if c["use_cost_list"] else None, d_j, c0, c1,
                                  d_mv_lists=d_l)
            got = (int(ctx.from_device(d_err, (1,), np.int32)[0]), ctx.from_device(d_mv, (2,), np.int16).tolist(), int(ctx.from_device(d_dist, (1,), np.int32)[0]),
                   int(ctx.from_device(d_sse, (1,), np.uint32)[0]), ctx.from_device(d_l, (3, 2), np.int16).tolist())
            assert got == (call["err"], call["mv"], call["distortion"], call["sse"], call["list_after"]), (c["fn"], c["bd"], call, got)
            n_stopped += call["err"] == 2147483647
            ctx.free(d_b)
    assert n_stopped >= 50
    for d in (d_j, d_c0, d_c1, d_mv, d_err, d_dist, d_sse, d_cl, d_l):
        ctx.free(d)
    for ps, pr in planes.values():
        ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd,tree,method,use_cl,second,force_int,bs", [(8, "pruned_more", "NSTEP", 1, 1, 0, 16), (10, "tree", "DIAMOND", 0, 1, 0, 16),
                                                                      (8, "pruned", "BIGDIA", 1, 0, 0, 16), (10, "pruned_more", "NSTEP", 0, 0, 1, 16),
                                                                      (8, "tree", "HEX", 0, 1, 0, 16), (10, "pruned_more", "NSTEP", 1, 1, 0, 32),
                                                                      (8, "pruned", "NSTEP", 0, 1, 0, 8)])
def test_single_motion_search_core_equals_the_oracle_composition(hip, oracle, ctx, bd, tree, method, use_cl, second, force_int, bs):
    capi = hip.capi
    W, H, B = 352, 288, 64
    rng = np.random.default_rng(bd * 31 + len(tree) + second)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 5, bd, shift=(4, -6), frac8=(3, 5))
    k = 60 << (bd - 8)   # heavy noise on both frames: the full-pel winner and its runner-up are close, so the second sub-pel search wins sometimes
    ref = np.clip(ref.astype(np.int32) + rng.integers(-k, k + 1, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    src = np.clip(src.astype(np.int32) + rng.integers(-k, k + 1, src.shape), 0, (1 << bd) - 1).astype(src.dtype)
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    blocks["ref_row"], blocks["ref_col"] = rng.integers(-40, 41, n), rng.integers(-40, 41, n)
    blocks["start_row"] = (blocks["ref_row"].astype(np.int32) + 3 + (blocks["ref_row"] >= 0)) >> 3    # start_mv = get_fullmv_from_mv(&ref_mv)
    blocks["start_col"] = (blocks["ref_col"].astype(np.int32) + 3 + (blocks["ref_col"] >= 0)) >> 3
    ext = B - 8
    blocks["col_min"] = np.maximum(-(blocks["bx"] + ext), -1023); blocks["col_max"] = np.minimum(W - blocks["bx"] - bs + ext, 1023)
    blocks["row_min"] = np.maximum(-(blocks["by"] + ext), -1023); blocks["row_max"] = np.minimum(H - blocks["by"] - bs + ext, 1023)
    # cand[1]: a TPL candidate for two thirds of the blocks; some first candidates skipped by the start-MV stack rule; a few blocks with neither
    start2 = np.stack([rng.integers(-10, 11, n), rng.integers(-10, 11, n)], 1).astype(np.int16)
    start2[rng.random(n) < 0.33] = -32768
    skip0 = rng.random(n) < 0.15
    blocks["start_row"][skip0] = -32768; blocks["start_col"][skip0] = -32768
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (150 + bits * 310).astype(np.int32), (170 + bits * 290 + (v & 7) * 3).astype(np.int32)
    tj = np.array([200, 650, 640, 1050], np.int32)
    q = capi.SearchParams.make(method, 2, capi.MV_COST_ENTROPY, sad_per_bit=24, error_per_bit=70)
    oq = oracle.search_params(method, 2, 0, sad_per_bit=24, error_per_bit=70)
    sp = capi.SubpelParams(TREE_ID[tree], capi.MV_COST_ENTROPY, 70, 2, 1, 0, 0)
    d_b, d_s2, d_j, d_c0, d_c1 = ctx.to_device(blocks), ctx.to_device(start2), ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    outs = [ctx.malloc(n * 4) for _ in range(6)]
    ctx.single_motion_search_batch(ps, pr, 0, bs, bs, q, sp, d_b, n, outs[0], outs[1], outs[2], d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4, d_start2=d_s2,
                                   use_cost_list=use_cl, try_second_mv=second, force_integer_mv=force_int, d_pred_sse=outs[3], d_full_mv=outs[4],
                                   d_second_best=outs[5])
    got = dict(best_mv=ctx.from_device(outs[0], (n, 2), np.int16), bestsme=ctx.from_device(outs[1], (n,), np.int32),
               rate_mv=ctx.from_device(outs[2], (n,), np.int32), pred_sse=ctx.from_device(outs[3], (n,), np.uint32),
               full_mv=ctx.from_device(outs[4], (n, 2), np.int16), second_best=ctx.from_device(outs[5], (n, 2), np.int16))
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    want = oracle.single_motion_search_batch(sb, rb, B, bs, bs, blocks, oq, dict(tree=tree, cost_type=0, error_per_bit=70, iters=2, allow_hp=1, forced_stop=0),
                                             start2=start2, use_cost_list=use_cl, try_second_mv=second, force_integer_mv=force_int, mvjcost=tj, mvcost0=t0,
                                             mvcost1=t1, bd=bd, threads=8)
    for k in want:
        assert np.array_equal(got[k], want[k]), (k, np.flatnonzero((got[k] != want[k]).reshape(n, -1).any(1))[:8])
    dead = (want["best_mv"][:, 0] == -32768)
    assert 0 < dead.sum() < n // 4 and (want["rate_mv"][~dead] > 0).any()
    if second and method in ("NSTEP", "DIAMOND") and bs == 16:   # (the pattern searches leave second_best_mv invalid) the second start won somewhere: exercised, not just executed
        first = oracle.single_motion_search_batch(sb, rb, B, bs, bs, blocks, oq, dict(tree=tree, cost_type=0, error_per_bit=70, iters=2, allow_hp=1, forced_stop=0),
                                                  start2=start2, use_cost_list=use_cl, try_second_mv=0, mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8)
        assert (first["best_mv"] != want["best_mv"]).any()
    # candidate 1 won for some blocks and lost for others
    only0 = oracle.single_motion_search_batch(sb, rb, B, bs, bs, blocks, oq, dict(tree=tree, cost_type=0, error_per_bit=70, iters=2, allow_hp=1, forced_stop=0),
                                              start2=None, use_cost_list=use_cl, try_second_mv=second, force_integer_mv=force_int, mvjcost=tj, mvcost0=t0,
                                              mvcost1=t1, bd=bd, threads=8)
    assert (only0["full_mv"] != want["full_mv"]).any() and (only0["full_mv"] == want["full_mv"]).all(1).any()
    for d in [d_b, d_s2, d_j, d_c0, d_c1] + outs:
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd,tree,method,bs,use_cl", [(8, "pruned_more", "NSTEP", 16, 1), (10, "tree", "DIAMOND", 16, 0), (8, "pruned", "NSTEP", 32, 0),
                                                      (10, "pruned_more", "NSTEP", 8, 1), (8, "tree", "NSTEP", 64, 0)])
def test_rd_form_of_the_second_mv_decision_equals_the_oracle_composition(hip, oracle, ctx, bd, tree, method, bs, use_cl):
    """aomhip_single_motion_search_rd_batch: disable_second_mv == 0 (motion_search_facade.c:378-418) -- both candidates' predictors through
    av1_estimate_txfm_yrd's composite, the second one kept when its RDCOST is smaller."""
    capi = hip.capi
    W, H, B = 352, 288, 64
    rng = np.random.default_rng(bd * 17 + len(tree) + bs)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 5, bd, shift=(4, -6), frac8=(3, 5))
    k = 60 << (bd - 8)
    ref = np.clip(ref.astype(np.int32) + rng.integers(-k, k + 1, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    src = np.clip(src.astype(np.int32) + rng.integers(-k, k + 1, src.shape), 0, (1 << bd) - 1).astype(src.dtype)
    ps, pr, pp = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    blocks["ref_row"], blocks["ref_col"] = rng.integers(-40, 41, n), rng.integers(-40, 41, n)
    blocks["start_row"] = (blocks["ref_row"].astype(np.int32) + 3 + (blocks["ref_row"] >= 0)) >> 3
    blocks["start_col"] = (blocks["ref_col"].astype(np.int32) + 3 + (blocks["ref_col"] >= 0)) >> 3
    ext = B - 8
    blocks["col_min"] = np.maximum(-(blocks["bx"] + ext), -1023); blocks["col_max"] = np.minimum(W - blocks["bx"] - bs + ext, 1023)
    blocks["row_min"] = np.maximum(-(blocks["by"] + ext), -1023); blocks["row_max"] = np.minimum(H - blocks["by"] - bs + ext, 1023)
    start2 = np.stack([rng.integers(-10, 11, n), rng.integers(-10, 11, n)], 1).astype(np.int16)
    start2[rng.random(n) < 0.33] = -32768
    skip0 = rng.random(n) < 0.1
    blocks["start_row"][skip0] = -32768; blocks["start_col"][skip0] = -32768
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (150 + bits * 310).astype(np.int32), (170 + bits * 290 + (v & 7) * 3).astype(np.int32)
    tj = np.array([200, 650, 640, 1050], np.int32)
    q = capi.SearchParams.make(method, 2, capi.MV_COST_ENTROPY, sad_per_bit=24, error_per_bit=70)
    oq = oracle.search_params(method, 2, 0, sad_per_bit=24, error_per_bit=70)
    sp = capi.SubpelParams(TREE_ID[tree], capi.MV_COST_ENTROPY, 70, 2, 1, 0, 0)
    sub = dict(tree=tree, cost_type=0, error_per_bit=70, iters=2, allow_hp=1, forced_stop=0)
    # the RD side: quantiser, cost tables, per-block header rates and contexts
    qt = oracle.build_quantizer_y(bd, 120)
    costs = rng.integers(10, 3000, 966).astype(np.int32)
    yb = np.zeros(n, capi.txfm_yrd_block_dtype)
    yb["bx"], yb["by"] = blocks["bx"], blocks["by"]
    yb["tx_size_rate"], yb["no_skip_txfm_rate"], yb["skip_txfm_rate"] = rng.integers(0, 2000, n), rng.integers(20, 2000, n), rng.integers(20, 2000, n)
    yb["above_ctx"] = rng.integers(0, 7, (n, 32)) | (rng.integers(0, 3, (n, 32)) << 3)
    yb["left_ctx"] = rng.integers(0, 7, (n, 32)) | (rng.integers(0, 3, (n, 32)) << 3)
    rdmult, tx_type_rate = 900, (0 if bs > 32 else 250)
    d_b, d_s2, d_j, d_c0, d_c1 = ctx.to_device(blocks), ctx.to_device(start2), ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    d_costs, d_yb, d_sa, d_sb, d_cm = ctx.to_device(costs), ctx.to_device(yb), ctx.malloc(32 * n), ctx.malloc(32 * n), ctx.malloc(8 * n)
    qp = capi.QuantParams.from_tables(qt)
    rd = capi.SingleRdParams()
    rd.pred, rd.filter_x, rd.filter_y, rd.qparams, rd.d_costs = C.pointer(pp), 0, 0, C.pointer(qp), d_costs
    rd.tx_type_rate, rd.rdmult, rd.lossless, rd.d_yrd_blocks, rd.d_stats_first, rd.d_stats_second = tx_type_rate, rdmult, 0, d_yb, d_sa, d_sb
    rd.d_candidate_mvs = d_cm
    outs = [ctx.malloc(n * 4) for _ in range(6)]
    ctx.single_motion_search_rd_batch(ps, pr, 0, bs, bs, q, sp, d_b, n, rd, outs[0], outs[1], outs[2], d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4, d_start2=d_s2,
                                      use_cost_list=use_cl, d_pred_sse=outs[3], d_full_mv=outs[4], d_second_best=outs[5])
    got = dict(best_mv=ctx.from_device(outs[0], (n, 2), np.int16), bestsme=ctx.from_device(outs[1], (n,), np.int32),
               rate_mv=ctx.from_device(outs[2], (n,), np.int32), pred_sse=ctx.from_device(outs[3], (n,), np.uint32),
               full_mv=ctx.from_device(outs[4], (n, 2), np.int16), second_best=ctx.from_device(outs[5], (n, 2), np.int16))
    sa, sb_ = ctx.from_device(d_sa, (n,), capi.txfm_yrd_stats_dtype), ctx.from_device(d_sb, (n,), capi.txfm_yrd_stats_dtype)
    sbuf, rbuf = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    common = dict(start2=start2, use_cost_list=use_cl, mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8)
    want = oracle.single_motion_search_batch(sbuf, rbuf, B, bs, bs, blocks, oq, sub, try_second_mv=1,
                                             rd=dict(filter_x=0, filter_y=0, q=qt, costs=costs, tx_type_rate=tx_type_rate, rdmult=rdmult, lossless=0, yrd_blocks=yb), **common)
    for k_ in ("best_mv", "bestsme", "rate_mv", "pred_sse", "full_mv", "second_best"):
        assert np.array_equal(got[k_], want[k_]), (k_, np.flatnonzero((got[k_] != want[k_]).reshape(n, -1).any(1))[:8])
    cm = ctx.from_device(d_cm, (n, 2, 2), np.int16)
    two = want["cand_mvs"][:, 1, 0] != -32768
    assert np.array_equal(cm[two], want["cand_mvs"][two]) and (cm[~two, 1] == -32768).all()
    dead = want["best_mv"][:, 0] == -32768
    assert np.array_equal(cm[~dead, 0], first_pass_mvs(oracle, sbuf, rbuf, B, bs, blocks, oq, sub, common)[~dead]) and (cm[dead] == -32768).all()
    tried = 0
    for i in range(n):
        for g, w_ in ((sa[i], want["stats_first"][i]), (sb_[i], want["stats_second"][i])):
            if w_ is not None:
                assert (int(g["rd"]), int(g["rate"]), int(g["skip_txfm"]), int(g["dist"]), int(g["sse"])) == (w_["rd"], w_["rate"], w_["skip_txfm"], w_["dist"], w_["sse"]), i
        tried += want["stats_second"][i] is not None
    # exercised: second searches ran, the RD rule took the second candidate for some blocks and kept the first for others, and it disagrees with the variance rule somewhere
    by_var = oracle.single_motion_search_batch(sbuf, rbuf, B, bs, bs, blocks, oq, sub, try_second_mv=1, **common)
    first = oracle.single_motion_search_batch(sbuf, rbuf, B, bs, bs, blocks, oq, sub, try_second_mv=0, **common)
    took = (want["best_mv"] != first["best_mv"]).any(1).sum()
    assert tried >= 4 and (bs >= 32 or (0 < took < tried)), (tried, took)
    if bs == 16:
        assert (want["best_mv"] != by_var["best_mv"]).any()
    for d in [d_b, d_s2, d_j, d_c0, d_c1, d_costs, d_yb, d_sa, d_sb, d_cm] + outs:
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr); ctx.planes_free(pp)


def test_rd_form_refuses_incomplete_parameters(hip, ctx):
    capi = hip.capi
    p8, p10 = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 10, 1)
    d = ctx.malloc(1 << 18)
    q = capi.SearchParams.make("NSTEP", 2, capi.MV_COST_ENTROPY, sad_per_bit=24, error_per_bit=70)
    sp = capi.SubpelParams(0, capi.MV_COST_ENTROPY, 70, 2, 1, 0, 0)
    qp = capi.QuantParams.from_tables(orc_tables(8))
    mid = d + (1 << 17)
    for pred, costs in ((p10, d), (p8, None)):     # predictor ring of another bit depth; no cost tables
        rd = capi.SingleRdParams()
        rd.pred, rd.qparams, rd.d_costs, rd.d_yrd_blocks, rd.rdmult = C.pointer(pred), C.pointer(qp), costs, d, 100
        with pytest.raises(capi.AomHipError):
            ctx.single_motion_search_rd_batch(p8, p8, 0, 16, 16, q, sp, d, 1, rd, d, d, d, d, mid, mid)
    with pytest.raises(capi.AomHipError):
        ctx.single_motion_search_rd_batch(p8, p8, 0, 16, 16, q, sp, d, 1, None, d, d, d, d, mid, mid)
    ctx.free(d)
    ctx.planes_free(p8); ctx.planes_free(p10)


def first_pass_mvs(oracle, sbuf, rbuf, B, bs, blocks, oq, sub, common):
    """the first sub-pel search's MV of every block (the composition without the second search)"""
    return oracle.single_motion_search_batch(sbuf, rbuf, B, bs, bs, blocks, oq, sub, try_second_mv=0, **common)["best_mv"]


def orc_tables(bd):
    import pyoracle
    return pyoracle.build_quantizer_y(bd, 100)
