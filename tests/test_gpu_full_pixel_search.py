"""av1_full_pixel_search on the device (aomhip_full_pixel_search_batch): bit-exact against
  * the golden vectors obtained by interpreting the reference's mcomp.c (tests/golden/ref_eval_mcomp.npz): all 11
    search methods, entropy / L1 / no MV cost, cost lists, second-best MVs, mesh follow-ups, the downsampled-SAD
    re-check, 8- and 10-bit;
  * the oracle on larger random batches (several block sizes, 8/10/12-bit, tight limits, noise content)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INT_MAX = 2147483647


def _run(hip, ctx, ps, pr, frame, bw, bh, q, blocks, tables=None, want_cl=True, want_second=True):
    n = len(blocks)
    d_b = ctx.to_device(blocks)
    d_mv, d_c, d_cl, d_s = ctx.malloc(max(16, n * 4)), ctx.malloc(max(16, n * 4)), ctx.malloc(max(32, n * 20)), ctx.malloc(max(16, n * 4))
    extra, keep = {}, []
    if tables is not None:
        j, c0, c1 = (np.ascontiguousarray(t, np.int32) for t in tables)
        dj, d0, d1 = ctx.to_device(j), ctx.to_device(c0), ctx.to_device(c1)
        keep = [dj, d0, d1]
        extra = dict(d_mvjcost=dj, d_mvcost_row=d0 + 4 * (c0.size // 2), d_mvcost_col=d1 + 4 * (c1.size // 2))
    ctx.full_pixel_search_batch(ps, pr, frame, bw, bh, q, d_b, n, d_mv, d_c, d_cl if want_cl else None, d_s if want_second else None, **extra)
    out = (ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_c, (n,), np.int32), ctx.from_device(d_cl, (n, 5), np.int32),
           ctx.from_device(d_s, (n, 2), np.int16))
    for d in [d_b, d_mv, d_c, d_cl, d_s] + keep:
        ctx.free(d)
    return out


def test_site_tables_match_reference_evaluation(hip):
    meta = json.loads(bytes(np.load(os.path.join(GOLD, "ref_eval_mcomp.npz"))["cases"]).decode())
    for method, ref in meta["sites"].items():
        ns, per, rad, mv = hip.capi.search_sites(method)
        assert ns == ref["num_search_steps"]
        first = ref["first_stage"]
        lo = 1 if method in ("DIAMOND", "CLAMPED_DIAMOND", "NSTEP", "NSTEP_8PT", "NSTEP_FPF") else 0
        for i in range(ns):
            st = i + first
            assert per[st] == ref["searches_per_step"][i] and rad[st] == ref["radius"][i]
            assert mv[st, lo:lo + per[st]].tolist() == ref["mv"][i]


def test_full_pixel_search_matches_reference_goldens(hip, ctx):
    z = np.load(os.path.join(GOLD, "ref_eval_mcomp.npz"))
    meta = json.loads(bytes(z["cases"]).decode())
    W, H, border = meta["W"], meta["H"], meta["border"]
    planes = {}
    for bd in (8, 10):
        ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(ps, 0, np.ascontiguousarray(z["src%d" % bd][border:border + H, border:border + W]))
        ctx.planes_upload(pr, 0, np.ascontiguousarray(z["ref%d" % bd][border:border + H, border:border + W]))
        planes[bd] = (ps, pr)
    tables = (z["mvjcost"], z["mvcost0"], z["mvcost1"])
    n = 0
    for c in meta["cases"]:
        if c["kind"] not in ("search", "diamond"):
            continue
        q = hip.capi.SearchParams.make(c["method"], c["step_param"], c["cost_type"], c.get("sad_per_bit", 20), c.get("error_per_bit", 60),
                                       c.get("skip_sad", False), c.get("run_mesh", 0), c.get("prune_mesh", 0), c.get("mesh_diff_thr", 0),
                                       c.get("force_mesh_thresh", INT_MAX), c.get("fine_interval", 0), c.get("mesh"))
        blk = np.zeros(1, hip.capi.search_block_dtype)
        for name, v in zip(blk.dtype.names, c["block"]):
            blk[name] = v
        ps, pr = planes[c["bd"]]
        mv, cost, cl, sec = _run(hip, ctx, ps, pr, 0, c["w"], c["h"], q, blk, tables)
        got = (mv[0].tolist(), int(cost[0]), cl[0].tolist())
        assert got == (c["mv"], c["cost"], c["cost_list"]), (c, got)
        if c.get("second_best") is not None and c["kind"] == "search":
            assert sec[0].tolist() == c["second_best"], (c, sec[0].tolist())
        n += 1
    assert n >= 180
    for ps, pr in planes.values():
        ctx.planes_free(ps); ctx.planes_free(pr)


def _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, n, start_range=0, ref_range=0):
    b = np.zeros(n, hip.capi.search_block_dtype)
    b["bx"] = rng.integers(0, W - bw + 1, n); b["by"] = rng.integers(0, H - bh + 1, n)
    b["start_row"] = rng.integers(-start_range, start_range + 1, n); b["start_col"] = rng.integers(-start_range, start_range + 1, n)
    b["ref_row"] = rng.integers(-ref_range, ref_range + 1, n); b["ref_col"] = rng.integers(-ref_range, ref_range + 1, n)
    for i in range(n):
        lim = oracle.mv_limits_for_block(int(b["bx"][i]), int(b["by"][i]), bw, bh, W, H, border, int(b["ref_row"][i]), int(b["ref_col"][i]))
        b["row_min"][i], b["row_max"][i], b["col_min"][i], b["col_max"][i] = lim
    return b


def _cost_tables(rng):
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    c0 = (100 + bits * 310 + rng.integers(0, 64, v.size)).astype(np.int32)
    c1 = (140 + bits * 290 + rng.integers(0, 64, v.size)).astype(np.int32)
    return np.asarray([150, 650, 700, 1200], np.int32), c0, c1


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (32, 32), (4, 4), (64, 32), (16, 64), (128, 128)])
def test_full_pixel_search_matches_oracle(hip, oracle, ctx, bw, bh, bd):
    rng = np.random.default_rng(bw * 5 + bh * 3 + bd)
    W, H, border = 384, 256, 160
    src, ref = hip.synth.shifted_smooth_pair(W, H, bw + bd, bd, shift=(int(rng.integers(-9, 10)), int(rng.integers(-9, 10))))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3 << (bd - 8), (3 << (bd - 8)) + 1, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    n = 96 if bw * bh >= 64 * 64 else 193
    blocks = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, n, start_range=6, ref_range=40)
    blocks["col_max"][::7] = np.minimum(blocks["col_max"][::7], 3)   # some tight limits
    blocks["row_min"][::5] = np.maximum(blocks["row_min"][::5], -2)
    tables = _cost_tables(rng)
    mesh = [(16, 4), (8, 2), (4, 1), (3, 1)]
    configs = [("NSTEP", 2, 3, {}), ("NSTEP_8PT", 0, 0, {}), ("HEX", 3, 4, {}), ("BIGDIA", 1, 3, {}), ("SQUARE", 2, 0, {}),
               ("FAST_HEX", 0, 1, {}), ("FAST_DIAMOND", 4, 2, {}), ("FAST_BIGDIA", 0, 3, {}), ("VFAST_DIAMOND", 0, 0, {}),
               ("DIAMOND", 4, 3, {}), ("CLAMPED_DIAMOND", 1, 0, {}),
               ("NSTEP", 3, 3, dict(force_mesh_thresh=1 << 12, mesh=mesh)), ("HEX", 2, 3, dict(run_mesh=1, prune_mesh=1, mesh_diff_thr=3, mesh=mesh)),
               ("NSTEP", 1, 0, dict(skip_sad=1)), ("BIGDIA", 0, 3, dict(skip_sad=1, run_mesh=1, mesh=mesh, fine_interval=1))]
    if bw * bh >= 64 * 64:
        configs = configs[::3]
    for method, step_param, cost_type, kw in configs:
        if kw.get("skip_sad") and bh < 8:
            continue
        qd = hip.capi.SearchParams.make(method, step_param, cost_type, 23, 71, **kw)
        qo = oracle.search_params(method, step_param, cost_type, 23, 71, **kw)
        got = _run(hip, ctx, ps, pr, 0, bw, bh, qd, blocks, tables)
        want = oracle.full_pixel_search_batch(sb, rb, border, bw, bh, blocks, qo, *tables, bd=bd)
        for gname, a, w in zip(("mv", "cost", "cost_list", "second"), got, want):
            assert np.array_equal(a, w), (method, step_param, cost_type, kw, gname, np.flatnonzero((np.asarray(a) != np.asarray(w)).reshape(n, -1).any(1))[:5])
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd,bw,bh", [(8, 16, 16), (10, 16, 16), (10, 32, 32), (8, 8, 8)])
def test_entropy_tables_with_negative_entries_take_the_literal_comparison(hip, oracle, ctx, bd, bw, bh):
    """libaom's MV cost tables are bit counts (>= 0) and a step of the diamond / n-step searches then ends in one key reduction; the API takes
    any int32 tables, and with a negative sad cost `sad < best` no longer follows from `sad + cost < best`: those steps replay the reference's
    in-order double comparison.  Tables whose small differences cost LESS than nothing pull both forms into every search."""
    rng = np.random.default_rng(17 * bd + bw)
    W, H, border = 320, 192, 160
    src, ref = hip.synth.shifted_smooth_pair(W, H, 3 + bd, bd, shift=(4, -5))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3 << (bd - 8), (3 << (bd - 8)) + 1, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    blocks = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, 150, start_range=6, ref_range=40)
    tj, c0, c1 = _cost_tables(rng)
    mid = c0.size // 2
    c0 = c0.copy(); c1 = c1.copy()
    c0[mid - 40:mid + 40] -= 2600      # differences below 5 pixels: a large negative row cost (sad cost = bits * sad_per_bit >> 9 < 0)
    c1[mid - 24:mid + 24:3] -= 1900    # and every third small column difference
    tables = (tj, c0, c1)
    for method, step_param in (("NSTEP", 2), ("DIAMOND", 4), ("NSTEP_8PT", 1)):
        qd = hip.capi.SearchParams.make(method, step_param, 0, 210, 71)
        qo = oracle.search_params(method, step_param, 0, 210, 71)
        got = _run(hip, ctx, ps, pr, 0, bw, bh, qd, blocks, tables)
        want = oracle.full_pixel_search_batch(sb, rb, border, bw, bh, blocks, qo, *tables, bd=bd)
        for gname, a, w in zip(("mv", "cost", "cost_list", "second"), got, want):
            assert np.array_equal(a, w), (method, gname, np.flatnonzero((np.asarray(a) != np.asarray(w)).reshape(len(blocks), -1).any(1))[:5])
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_full_pixel_search_without_cost_list_and_noise(hip, oracle, ctx):
    """cost_list == NULL changes pattern_search's last scale for the 4-candidate tables (mcomp.c:1077); pure noise makes
    every descent data-dependent."""
    rng = np.random.default_rng(5)
    W, H, border, bw, bh = 256, 192, 96, 16, 16
    src = rng.integers(0, 256, (H, W)).astype(np.uint8); ref = rng.integers(0, 256, (H, W)).astype(np.uint8)
    ps, pr = ctx.planes_alloc(W, H, border, 8, 1), ctx.planes_alloc(W, H, border, 8, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    blocks = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, 300, start_range=10, ref_range=30)
    for method in ("BIGDIA", "FAST_DIAMOND", "HEX", "NSTEP", "SQUARE"):
        for no_cl in (0, 1):
            qd = hip.capi.SearchParams.make(method, 1, 3)
            qo = oracle.search_params(method, 1, 3, no_cost_list=no_cl)
            mv, cost, _, sec = _run(hip, ctx, ps, pr, 0, bw, bh, qd, blocks, None, want_cl=not no_cl)
            wmv, wcost, _, wsec = oracle.full_pixel_search_batch(sb, rb, border, bw, bh, blocks, qo)
            assert np.array_equal(mv, wmv) and np.array_equal(cost, wcost) and np.array_equal(sec, wsec), (method, no_cl)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_full_pixel_search_rejects_bad_arguments(hip, ctx):
    ps, pr = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 8, 1)
    blk = np.zeros(1, hip.capi.search_block_dtype)
    d_b, d_mv, d_c = ctx.to_device(blk), ctx.malloc(16), ctx.malloc(16)
    with pytest.raises(Exception):      # entropy cost without tables
        ctx.full_pixel_search_batch(ps, pr, 0, 16, 16, hip.capi.SearchParams.make("NSTEP", 0, 0), d_b, 1, d_mv, d_c)
    with pytest.raises(Exception):      # step_param beyond the table (== num_search_steps is the diamonds' "start position only")
        ctx.full_pixel_search_batch(ps, pr, 0, 16, 16, hip.capi.SearchParams.make("DIAMOND", 12, 3), d_b, 1, d_mv, d_c)
    with pytest.raises(Exception):      # ... which the pattern searches do not have
        ctx.full_pixel_search_batch(ps, pr, 0, 16, 16, hip.capi.SearchParams.make("HEX", 11, 3), d_b, 1, d_mv, d_c)
    with pytest.raises(Exception):      # unknown method
        ctx.full_pixel_search_batch(ps, pr, 0, 16, 16, hip.capi.SearchParams.make(12, 0, 3), d_b, 1, d_mv, d_c)
    for d in (d_b, d_mv, d_c):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


# ---- the bilinear sub-pel trees (aomhip_subpel_tree_batch)

def _run_subpel(hip, ctx, ps, pr, bw, bh, blocks, tree, cost_type, error_per_bit, iters, allow_hp, forced_stop, cost_lists=None,
                tables=None, sst=0):
    n = len(blocks)
    d_b = ctx.to_device(blocks)
    d_mv, d_e, d_d, d_s = (ctx.malloc(max(16, n * 4)) for _ in range(4))
    keep, extra = [], {}
    if cost_lists is not None:
        d_cl = ctx.to_device(np.ascontiguousarray(cost_lists, np.int32)); keep.append(d_cl); extra["d_cost_list"] = d_cl
    if tables is not None:
        j, c0, c1 = (np.ascontiguousarray(t, np.int32) for t in tables)
        dj, d0, d1 = ctx.to_device(j), ctx.to_device(c0), ctx.to_device(c1)
        keep += [dj, d0, d1]
        extra.update(d_mvjcost=dj, d_mvcost_row=d0 + 4 * (c0.size // 2), d_mvcost_col=d1 + 4 * (c1.size // 2))
    p = hip.capi.SubpelParams(hip.capi.SUBPEL_TREES.get(tree, tree), cost_type, error_per_bit, iters, allow_hp, forced_stop, sst)
    ctx.subpel_tree_batch(ps, pr, 0, bw, bh, p, d_b, n, d_mv, d_e, d_d, d_s, **extra)
    out = (ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_e, (n,), np.uint32), ctx.from_device(d_d, (n,), np.int32),
           ctx.from_device(d_s, (n,), np.uint32))
    for d in [d_b, d_mv, d_e, d_d, d_s] + keep:
        ctx.free(d)
    return out


def test_subpel_trees_match_reference_goldens(hip, ctx):
    z = np.load(os.path.join(GOLD, "ref_eval_mcomp.npz"))
    meta = json.loads(bytes(z["cases"]).decode())
    W, H, border = meta["W"], meta["H"], meta["border"]
    planes = {}
    for bd in (8, 10):
        ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(ps, 0, np.ascontiguousarray(z["src%d" % bd][border:border + H, border:border + W]))
        ctx.planes_upload(pr, 0, np.ascontiguousarray(z["ref%d" % bd][border:border + H, border:border + W]))
        planes[bd] = (ps, pr)
    tables = (z["mvjcost"], z["mvcost0"], z["mvcost1"])
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "subpel":
            continue
        blk = np.zeros(1, hip.capi.search_block_dtype)
        vals = list(c["block"])
        vals[2], vals[3] = c["fullpel_mv"][0] * 8, c["fullpel_mv"][1] * 8
        vals[6:10] = c["subpel_limits"]
        for name, v in zip(blk.dtype.names, vals):
            blk[name] = v
        tree = {"av1_find_best_sub_pixel_tree_pruned_more": "pruned_more", "av1_find_best_sub_pixel_tree_pruned": "pruned",
                "av1_find_best_sub_pixel_tree": "tree"}[c["fn"]]
        ps, pr = planes[c["bd"]]
        mv, err, dist, sse = _run_subpel(hip, ctx, ps, pr, c["w"], c["h"], blk, tree, c["cost_type"], c["error_per_bit"], c["iters"],
                                         c["allow_hp"], c["forced_stop"], [c["cost_list"]] if "cost_list" in c else None, tables,
                                         c.get("subpel_search_type", 0))
        assert (mv[0].tolist(), int(err[0]), int(dist[0]), int(sse[0])) == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        n += 1
    assert n >= 58
    for ps, pr in planes.values():
        ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (32, 16), (4, 8), (64, 64), (128, 128), (4, 4)])
def test_subpel_trees_match_oracle(hip, oracle, ctx, bw, bh, bd):
    rng = np.random.default_rng(bw * 7 + bh + bd)
    W, H, border = 320, 192, 96
    src, ref = hip.synth.shifted_smooth_pair(W, H, bw + bd, bd, shift=(int(rng.integers(-5, 6)), int(rng.integers(-5, 6))))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-2 << (bd - 8), (2 << (bd - 8)) + 1, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    n = 151
    fb = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, n, start_range=4, ref_range=30)
    tables = _cost_tables(rng)
    # full-pel stage (NSTEP) gives the start MVs and the cost lists, as in the encoder
    q = oracle.search_params("NSTEP", 3, 3, 20, 60)
    fmv, _, cl, _ = oracle.full_pixel_search_batch(sb, rb, border, bw, bh, fb, q, *tables, bd=bd)
    blocks = fb.copy()
    blocks["start_row"], blocks["start_col"] = fmv[:, 0] * 8, fmv[:, 1] * 8
    # av1_set_subpel_mv_search_range (mcomp.h:345-368)
    for i in range(n):
        rr, rc = int(blocks["ref_row"][i]), int(blocks["ref_col"][i])
        blocks["col_min"][i] = max(int(fb["col_min"][i]) * 8, rc - 8184, -16383 + 0)
        blocks["col_max"][i] = min(int(fb["col_max"][i]) * 8, rc + 8184, 16383)
        blocks["row_min"][i] = max(int(fb["row_min"][i]) * 8, rr - 8184, -16383)
        blocks["row_max"][i] = min(int(fb["row_max"][i]) * 8, rr + 8184, 16383)
    cl[::9, 2] = 2147483647          # some unusable lists
    for tree in ("pruned_more", "pruned", "tree"):
        for cost_type, use_cl, iters, allow_hp, forced_stop in ((3, True, 2, 1, 0), (0, True, 1, 0, 1), (4, False, 2, 1, 0), (1, True, 2, 1, 2),
                                                                (0, False, 2, 0, 0), (3, False, 1, 1, 3)):
            got = _run_subpel(hip, ctx, ps, pr, bw, bh, blocks, tree, cost_type, 77, iters, allow_hp, forced_stop, cl if use_cl else None, tables)
            want = oracle.subpel_tree_batch(sb, rb, border, bw, bh, blocks, tree=tree, cost_type=cost_type, error_per_bit=77, mvjcost=tables[0],
                                            mvcost0=tables[1], mvcost1=tables[2], iters=iters, allow_hp=allow_hp, forced_stop=forced_stop,
                                            cost_lists=cl if use_cl else None, bd=bd)
            for name, a, w_ in zip(("mv", "err", "dist", "sse"), got, want):
                assert np.array_equal(a, w_), (tree, cost_type, use_cl, iters, allow_hp, forced_stop, name)
    # the tree with the up-sampled (8-tap) prediction error
    for cost_type, iters, allow_hp, forced_stop in ((3, 2, 1, 0), (0, 1, 0, 1), (4, 2, 0, 0)):
        got = _run_subpel(hip, ctx, ps, pr, bw, bh, blocks, "tree", cost_type, 77, iters, allow_hp, forced_stop, None, tables, sst=3)
        want = oracle.subpel_tree_batch(sb, rb, border, bw, bh, blocks, tree="tree", cost_type=cost_type, error_per_bit=77, mvjcost=tables[0],
                                        mvcost0=tables[1], mvcost1=tables[2], iters=iters, allow_hp=allow_hp, forced_stop=forced_stop, bd=bd,
                                        subpel_search_type=3)
        for name, a, w_ in zip(("mv", "err", "dist", "sse"), got, want):
            assert np.array_equal(a, w_), ("8tap", cost_type, iters, allow_hp, forced_stop, name)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_edge_cases_empty_ragged_degenerate_limits(hip, oracle, ctx):
    """n = 0, n not a multiple of the 4 blocks per workgroup, a single legal MV, start MVs outside the limits (clamped)."""
    rng = np.random.default_rng(11)
    W, H, border, bw, bh = 128, 96, 64, 8, 8
    src = rng.integers(0, 256, (H, W)).astype(np.uint8); ref = rng.integers(0, 256, (H, W)).astype(np.uint8)
    ps, pr = ctx.planes_alloc(W, H, border, 8, 1), ctx.planes_alloc(W, H, border, 8, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    q = hip.capi.SearchParams.make("NSTEP", 0, 3)
    d_mv, d_c = ctx.malloc(64), ctx.malloc(64)
    ctx.full_pixel_search_batch(ps, pr, 0, bw, bh, q, None, 0, d_mv, d_c)          # empty batch: a no-op
    ctx.free(d_mv); ctx.free(d_c)
    for n in (1, 2, 3, 5, 7):
        blocks = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, n, start_range=40, ref_range=10)   # starts beyond the limits
        blocks["row_min"][0] = blocks["row_max"][0] = 2; blocks["col_min"][0] = blocks["col_max"][0] = -3   # one legal position
        if n > 2:
            blocks["row_min"][2], blocks["row_max"][2] = -1, 0                                            # a 2 x N strip
        for method in ("NSTEP", "HEX", "BIGDIA", "DIAMOND"):
            got = _run(hip, ctx, ps, pr, 0, bw, bh, hip.capi.SearchParams.make(method, 1, 3, run_mesh=1, mesh=[(8, 2), (4, 1), (2, 1), (1, 1)]), blocks)
            want = oracle.full_pixel_search_batch(sb, rb, border, bw, bh, blocks,
                                                  oracle.search_params(method, 1, 3, run_mesh=1, mesh=[(8, 2), (4, 1), (2, 1), (1, 1)]))
            for name, a, w_ in zip(("mv", "cost", "cost_list", "second"), got, want):
                assert np.array_equal(a, w_), (n, method, name)
            assert got[0][0].tolist() == [2, -3]
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("bs", [16, 32, 64])
def test_mesh_passes_over_a_grid_of_neighbours(hip, oracle, ctx, bs, bd):
    """NSTEP + the good-quality mesh pattern on EVERY block of a frame cut into a grid -- the layout the cells are made for: horizontal neighbours
    share a window, and a mesh batch whose candidates all lie inside it reads the window instead of the plane (round 6).  Ranges 64 / 28 / 15 / 7
    straddle the window's reach (32 at 16x16, 8 above) in both directions; the frame pair moves by a different vector in each quadrant so that
    neighbouring blocks end their n-step search on different centres.  Also with the pruning rule on (mesh only where the n-step result is far)."""
    rng = np.random.default_rng(bs + bd)
    W, H, border = 256, 192, 160
    src, ref = hip.synth.shifted_smooth_pair(W, H, bs + bd, bd, shift=(5, -7))
    _, ref2 = hip.synth.shifted_smooth_pair(W, H, bs + bd, bd, shift=(-11, 3))
    ref[H // 2:, :] = ref2[H // 2:, :]
    ref[:, W // 2:] = np.roll(ref[:, W // 2:], 9, axis=1)
    ref = np.clip(ref.astype(np.int32) + rng.integers(-2 << (bd - 8), (2 << (bd - 8)) + 1, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    xs, ys = np.meshgrid(np.arange(0, W - bs + 1, bs), np.arange(0, H - bs + 1, bs))
    b = np.zeros(xs.size, hip.capi.search_block_dtype)
    b["bx"], b["by"] = xs.ravel(), ys.ravel()
    for i in range(b.size):
        b["row_min"][i], b["row_max"][i], b["col_min"][i], b["col_max"][i] = oracle.mv_limits_for_block(int(b["bx"][i]), int(b["by"][i]), bs, bs, W, H, border, 0, 0)
    mesh = [(64, 8), (28, 4), (15, 1), (7, 1)]
    for kw in (dict(run_mesh=1), dict(run_mesh=1, prune_mesh=1, mesh_diff_thr=4), dict(force_mesh_thresh=1 << 10)):
        got = _run(hip, ctx, ps, pr, 0, bs, bs, hip.capi.SearchParams.make("NSTEP", 3, 3, mesh=mesh, **kw), b)
        want = oracle.full_pixel_search_batch(sb, rb, border, bs, bs, b, oracle.search_params("NSTEP", 3, 3, mesh=mesh, **kw), bd=bd)
        for name, a, w_ in zip(("mv", "cost", "cost_list", "second"), got, want):
            assert np.array_equal(a, w_), (bs, bd, kw, name, int(np.flatnonzero((a != w_).reshape(len(a), -1).any(1))[0]))
        assert len(np.unique(got[0], axis=0)) > 2   # the quadrants did end on different vectors
    ctx.planes_free(ps); ctx.planes_free(pr)
