"""aomhip_joint_motion_search_batch / _extensive_batch: av1_joint_motion_search (av1/encoder/motion_search_facade.c:496-702) on the refining-search
branch and on the extensive one (av1_full_pixel_search on the compound prediction, second sub-pel start), as one call per batch of compound blocks, against the composition of the pinned pieces (oracle.joint_motion_search_batch: predictor of the other reference ->
av1_refining_search_8p_c -> compound sub-pel tree -> the update / early-out rules, four alternating iterations) -- 8 / 10-bit, averaged and masked
compounds, entropy and L1 MV costs, force_integer_mv; and aomhip_build_inter_pred_contiguous_batch against the plane form."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables():
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    return mv_max, np.array([190, 660, 655, 1040], np.int32), (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)


@pytest.mark.parametrize("bd,bw,bh,masked,ct,tree,sst,force_int", [(8, 16, 16, 0, 0, 2, 0, 0), (10, 16, 16, 1, 3, 0, 0, 0), (8, 32, 16, 1, 0, 1, 0, 0),
                                                                   (10, 8, 8, 0, 0, 2, 3, 0), (8, 16, 16, 0, 3, 2, 0, 1)])
def test_joint_search_equals_the_composition(hip, oracle, ctx, bd, bw, bh, masked, ct, tree, sst, force_int):
    capi = hip.capi
    W, H, B = 256, 160, 96
    rng = np.random.default_rng(5 * bd + bw + 7 * masked + tree)
    # the source lies between two references that moved in opposite directions: the compound of both predicts it, each alone does not
    src, ref0 = hip.synth.shifted_smooth_pair(W, H, 21, bd, shift=(2, -3), frac8=(3, 0))
    _, ref1 = hip.synth.shifted_smooth_pair(W, H, 21, bd, shift=(-3, 2), frac8=(0, 5))
    mx = (1 << bd) - 1
    noisy = lambda a, k: np.clip(a.astype(np.int32) + rng.integers(-k, k + 1, a.shape), 0, mx).astype(a.dtype)
    ref0, ref1 = noisy(ref0, 2 << (bd - 8)), noisy(ref1, 2 << (bd - 8))
    ps, p0, p1 = (ctx.planes_alloc(W, H, B, bd, 1) for _ in range(3))
    for p_, a in ((ps, src), (p0, ref0), (p1, ref1)):
        ctx.planes_upload(p_, 0, a)
    gc, gr = W // bw, H // bh
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bh
    ext = B - 8 - 16
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext), W - blocks["bx"] - bw + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bh + ext
    ref_mv = rng.integers(-40, 41, (n, 2, 2)).astype(np.int16)
    cur = np.zeros((n, 2, 2), np.int16)
    cur[:, 0] = np.array([-3 * 8 + 1, 2 * 8 - 3]) + rng.integers(-10, 11, (n, 2))      # near the true motion of each reference, a few eighths off
    cur[:, 1] = np.array([2 * 8 - 2, -3 * 8 + 4]) + rng.integers(-10, 11, (n, 2))
    cur[::5] = 0                                                                             # some start from zero
    mask = np.clip((np.arange(bw)[None, None, :] * 64 // bw + rng.integers(-6, 7, (n, bh, bw))), 0, 64).astype(np.uint8) if masked else None
    mv_max, tj, t0, t1 = _tables()
    sb, r0b, r1b = (oracle.extend_plane(a, B, ps.stride) for a in (src, ref0, ref1))
    sub_kw = dict(tree=tree, subpel_search_type=sst, error_per_bit=61, iters_per_step=2, allow_hp=1)
    want_mv, want_rate, want_err, iters = oracle.joint_motion_search_batch(sb, r0b, r1b, B, W, H, bw, bh, blocks, ref_mv, cur, mask, cost_type=ct, sad_per_bit=22,
                                                                           sub=sub_kw, force_integer_mv=force_int, mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    d_b, d_r, d_cur = ctx.to_device(blocks), ctx.to_device(ref_mv), ctx.to_device(cur)
    d_m = ctx.to_device(mask) if masked else None
    d_rate, d_err = ctx.malloc(n * 4), ctx.malloc(n * 4)
    sub = capi.SubpelParams(tree, ct, 61, 2, 1, 3, sst)        # forced_stop is overridden (EIGHTH_PEL) as the reference does
    ctx.joint_motion_search_batch(ps, p0, p1, 0, bw, bh, ct, 22, sub, force_int, d_b, d_r, d_cur, d_m, n, d_rate, d_err, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    got_mv = ctx.from_device(d_cur, (n, 2, 2), np.int16)
    assert np.array_equal(got_mv, want_mv), np.flatnonzero((got_mv != want_mv).reshape(n, -1).any(1))[:8]
    assert np.array_equal(ctx.from_device(d_rate, (n,), np.int32), want_rate)
    assert np.array_equal(ctx.from_device(d_err, (n,), np.int32), want_err)
    # the loop is exercised: blocks that stop after 1, 2, 3 and 4 iterations all occur across the cases; most MVs move
    assert (want_mv != cur).any(axis=(1, 2)).mean() > 0.5 and len(set(iters.tolist())) >= 2
    for d in [d_j, d_c0, d_c1, d_b, d_r, d_cur, d_rate, d_err] + ([d_m] if masked else []):
        ctx.free(d)
    for p_ in (ps, p0, p1):
        ctx.planes_free(p_)


@pytest.mark.parametrize("bd,bw,bh,masked,ct,tree,sst,second,mesh_thr", [(8, 16, 16, 0, 0, 2, 0, 1, None), (10, 16, 16, 1, 3, 2, 3, 1, 0), (8, 32, 16, 0, 0, 1, 0, 0, None),
                                                                         (10, 8, 8, 1, 0, 2, 0, 1, 3000)])
def test_extensive_joint_search_equals_the_composition(hip, oracle, ctx, bd, bw, bh, masked, ct, tree, sst, second, mesh_thr):
    """disable_extensive_joint_motion_search == 0 (speed 0): av1_full_pixel_search(.., 5, ..) on the compound prediction per iteration, the second
    sub-pel start from second_best_mv (allow_second_mv), NSTEP's mesh follow-up on the plain SAD."""
    capi = hip.capi
    W, H, B = 256, 160, 96
    rng = np.random.default_rng(11 * bd + bw + 7 * masked + tree)
    src, ref0 = hip.synth.shifted_smooth_pair(W, H, 33, bd, shift=(3, -4), frac8=(2, 0))
    _, ref1 = hip.synth.shifted_smooth_pair(W, H, 33, bd, shift=(-4, 3), frac8=(0, 6))
    mx = (1 << bd) - 1
    noisy = lambda a, k: np.clip(a.astype(np.int32) + rng.integers(-k, k + 1, a.shape), 0, mx).astype(a.dtype)
    ref0, ref1 = noisy(ref0, 2 << (bd - 8)), noisy(ref1, 2 << (bd - 8))
    ps, p0, p1 = (ctx.planes_alloc(W, H, B, bd, 1) for _ in range(3))
    for p_, a in ((ps, src), (p0, ref0), (p1, ref1)):
        ctx.planes_upload(p_, 0, a)
    gc, gr = W // bw, H // bh
    n = min(gc * gr, 120)
    pick = rng.permutation(gc * gr)[:n]
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (pick % gc) * bw, (pick // gc) * bh
    ext = B - 8 - 16
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext), W - blocks["bx"] - bw + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bh + ext
    ref_mv = rng.integers(-40, 41, (n, 2, 2)).astype(np.int16)
    cur = np.zeros((n, 2, 2), np.int16)
    cur[:, 0] = np.array([-4 * 8, 3 * 8]) + rng.integers(-60, 61, (n, 2))        # up to 7 PIXELS off the true motion: out of the 8-neighbour refinement's reach
    cur[:, 1] = np.array([3 * 8, -4 * 8]) + rng.integers(-60, 61, (n, 2))
    cur[::5] = 0
    mask = np.clip((np.arange(bw)[None, None, :] * 64 // bw + rng.integers(-6, 7, (n, bh, bw))), 0, 64).astype(np.uint8) if masked else None
    mv_max, tj, t0, t1 = _tables()
    sb, r0b, r1b = (oracle.extend_plane(a, B, ps.stride) for a in (src, ref0, ref1))
    sub_kw = dict(tree=tree, subpel_search_type=sst, error_per_bit=61, iters_per_step=2, allow_hp=1)
    mesh = [(12, 4), (6, 2), (4, 1), (3, 1)]
    kw = {} if mesh_thr is None else dict(force_mesh_thresh=mesh_thr, mesh=mesh)
    oq = oracle.search_params("NSTEP", 5, ct, 22, 61, no_cost_list=1, **kw)
    want_mv, want_rate, want_err, iters = oracle.joint_motion_search_batch(sb, r0b, r1b, B, W, H, bw, bh, blocks, ref_mv, cur, mask, cost_type=ct, sad_per_bit=22,
                                                                           sub=sub_kw, mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8, full=oq,
                                                                           allow_second_mv=second)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    d_b, d_r, d_cur = ctx.to_device(blocks), ctx.to_device(ref_mv), ctx.to_device(cur)
    d_m = ctx.to_device(mask) if masked else None
    d_rate, d_err = ctx.malloc(n * 4), ctx.malloc(n * 4)
    sub = capi.SubpelParams(tree, ct, 61, 2, 1, 3, sst)
    full = capi.SearchParams.make("NSTEP", 5, ct, 22, 61, **kw)
    ctx.joint_motion_search_extensive_batch(ps, p0, p1, 0, bw, bh, full, sub, second, 0, d_b, d_r, d_cur, d_m, n, d_rate, d_err, d_j, d_c0 + mv_max * 4,
                                            d_c1 + mv_max * 4)
    got_mv = ctx.from_device(d_cur, (n, 2, 2), np.int16)
    assert np.array_equal(got_mv, want_mv), np.flatnonzero((got_mv != want_mv).reshape(n, -1).any(1))[:8]
    assert np.array_equal(ctx.from_device(d_rate, (n,), np.int32), want_rate)
    assert np.array_equal(ctx.from_device(d_err, (n,), np.int32), want_err)
    assert (want_mv != cur).any(axis=(1, 2)).mean() > 0.5 and len(set(iters.tolist())) >= 2
    # ... and it is a different search from the refining branch on these inputs
    ref_mv_, _, ref_err, _ = oracle.joint_motion_search_batch(sb, r0b, r1b, B, W, H, bw, bh, blocks, ref_mv, cur, mask, cost_type=ct, sad_per_bit=22, sub=sub_kw,
                                                              mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8)
    assert (ref_mv_ != want_mv).any()
    if second:   # the second start does win somewhere (else the branch is not exercised)
        no2, _, _, _ = oracle.joint_motion_search_batch(sb, r0b, r1b, B, W, H, bw, bh, blocks, ref_mv, cur, mask, cost_type=ct, sad_per_bit=22, sub=sub_kw,
                                                        mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8, full=oq, allow_second_mv=0)
        test_extensive_joint_search_equals_the_composition.second_won = getattr(test_extensive_joint_search_equals_the_composition, "second_won", 0) + \
            int((no2 != want_mv).any())
    for d in [d_j, d_c0, d_c1, d_b, d_r, d_cur, d_rate, d_err] + ([d_m] if masked else []):
        ctx.free(d)
    for p_ in (ps, p0, p1):
        ctx.planes_free(p_)


def test_the_second_subpel_start_won_somewhere():
    assert getattr(test_extensive_joint_search_equals_the_composition, "second_won", 0) >= 1


@pytest.mark.parametrize("bd,bw,bh,masked,ref_idx,ct,tree,sst,given_pred,force_int", [(8, 16, 16, 1, 0, 0, 2, 0, 0, 0), (10, 16, 16, 1, 1, 3, 2, 3, 0, 0),
                                                                                     (8, 32, 32, 0, 1, 0, 1, 0, 1, 0), (10, 8, 16, 1, 0, 0, 0, 0, 1, 0),
                                                                                     (8, 16, 8, 1, 1, 3, 2, 0, 0, 1)])
def test_compound_single_search_equals_the_composition(hip, oracle, ctx, bd, bw, bh, masked, ref_idx, ct, tree, sst, given_pred, force_int):
    """av1_compound_single_motion_search[_interinter] (motion_search_facade.c:703-853): one side of a (masked) compound against the fixed predictor
    of the other -- the interinter form builds that predictor from the other reference with the block's own filters, the plain form is handed it."""
    capi = hip.capi
    W, H, B = 256, 160, 96
    rng = np.random.default_rng(13 * bd + bw + 5 * bh + ref_idx)
    src, ref0 = hip.synth.shifted_smooth_pair(W, H, 41, bd, shift=(3, -4), frac8=(2, 0))
    _, ref1 = hip.synth.shifted_smooth_pair(W, H, 41, bd, shift=(-4, 3), frac8=(0, 6))
    mx = (1 << bd) - 1
    noisy = lambda a, k: np.clip(a.astype(np.int32) + rng.integers(-k, k + 1, a.shape), 0, mx).astype(a.dtype)
    ref0, ref1 = noisy(ref0, 2 << (bd - 8)), noisy(ref1, 2 << (bd - 8))
    refs = (ref0, ref1)
    ps, p0, p1 = (ctx.planes_alloc(W, H, B, bd, 1) for _ in range(3))
    for p_, a in ((ps, src), (p0, ref0), (p1, ref1)):
        ctx.planes_upload(p_, 0, a)
    planes = (p0, p1)
    gc, gr = W // bw, H // bh
    n = min(gc * gr, 100)
    pick = rng.permutation(gc * gr)[:n]
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (pick % gc) * bw, (pick // gc) * bh
    ext = B - 8 - 16
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext), W - blocks["bx"] - bw + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bh + ext
    true_mv = (np.array([-4 * 8, 3 * 8]), np.array([3 * 8, -4 * 8]))
    ref_mv = rng.integers(-40, 41, (n, 2)).astype(np.int16)
    this = (true_mv[ref_idx] + rng.integers(-44, 45, (n, 2))).astype(np.int16)
    this[::6] = 0
    other = (true_mv[1 - ref_idx] + rng.integers(-6, 7, (n, 2))).astype(np.int16)
    mask = np.clip((np.arange(bw)[None, None, :] * 64 // bw + rng.integers(-6, 7, (n, bh, bw))), 0, 64).astype(np.uint8) if masked else None
    mv_max, tj, t0, t1 = _tables()
    sb = oracle.extend_plane(src, B, ps.stride)
    rb = [oracle.extend_plane(a, B, ps.stride) for a in refs]
    sub_kw = dict(tree=tree, subpel_search_type=sst, error_per_bit=61, iters_per_step=2, allow_hp=1)
    mesh = [(12, 4), (6, 2), (4, 1), (3, 1)]
    kw = dict(force_mesh_thresh=6000 if bw * bh >= 256 else 1500, mesh=mesh)
    oq = oracle.search_params("NSTEP", 5, ct, 22, 61, no_cost_list=1, **kw)
    fx, fy = 2, 1   # AOMHIP_INTERP_SHARP horizontally, _SMOOTH vertically (mbmi->interp_filters)
    sp = None
    if given_pred:   # the interintra caller's form: any predictor
        sp = np.stack([np.clip(rb[1 - ref_idx][B + b["by"] + 1:B + b["by"] + 1 + bh, B + b["bx"] - 2:B + b["bx"] - 2 + bw].astype(np.int32) +
                               rng.integers(-5, 6, (bh, bw)), 0, mx) for b in blocks]).astype(src.dtype)
    want_mv, want_rate, want_sme = oracle.compound_single_motion_search_batch(sb, rb[ref_idx], B, W, H, bw, bh, blocks, ref_mv, this, oq, sub_kw, rb[1 - ref_idx],
                                                                              other, fx, fy, sp, mask, ref_idx, force_int, tj, t0, t1, bd=bd, threads=8)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    d_b, d_r, d_this, d_o = ctx.to_device(blocks), ctx.to_device(ref_mv), ctx.to_device(this), ctx.to_device(other)
    d_m = ctx.to_device(mask) if masked else None
    d_sp = ctx.to_device(sp) if given_pred else None
    d_rate, d_sme = ctx.malloc(n * 4), ctx.malloc(n * 4)
    sub = capi.SubpelParams(tree, ct, 61, 2, 1, 3, sst)
    full = capi.SearchParams.make("NSTEP", 5, ct, 22, 61, **kw)
    ctx.compound_single_motion_search_batch(ps, planes[ref_idx], None if given_pred else planes[1 - ref_idx], 0, bw, bh, full, sub, force_int, d_b, d_r, d_this,
                                            None if given_pred else d_o, fx, fy, d_sp, d_m, ref_idx, n, d_rate, d_sme, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    got_mv = ctx.from_device(d_this, (n, 2), np.int16)
    assert np.array_equal(got_mv, want_mv), np.flatnonzero((got_mv != want_mv).any(1))[:8]
    assert np.array_equal(ctx.from_device(d_rate, (n,), np.int32), want_rate)
    assert np.array_equal(ctx.from_device(d_sme, (n,), np.int32), want_sme)
    assert (want_mv != this).any(1).mean() > 0.5
    for d in [d_j, d_c0, d_c1, d_b, d_r, d_this, d_o, d_rate, d_sme] + ([d_m] if masked else []) + ([d_sp] if given_pred else []):
        ctx.free(d)
    for p_ in (ps, p0, p1):
        ctx.planes_free(p_)


def test_device_matches_the_interpreted_callers(hip, ctx):
    """The three entry points straight against av1_joint_motion_search (both branches) / av1_compound_single_motion_search interpreted as they are
    written (tests/golden/ref_eval_joint.npz): no oracle in between."""
    import json
    import os
    from test_golden_joint import TAPS, TREES
    capi = hip.capi
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_joint.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    B, W, H = meta["border"], meta["width"], meta["height"]
    j, c0, c1 = z["mvjcost"].astype(np.int32), z["mvcost0"].astype(np.int32), z["mvcost1"].astype(np.int32)
    mv_max = c0.size // 2
    d_j, d_c0, d_c1 = ctx.to_device(j), ctx.to_device(c0), ctx.to_device(c1)
    planes = {}
    for bd in (8, 10):
        planes[bd] = [ctx.planes_alloc(W, H, B, bd, 1) for _ in range(3)]
        for p_, name in zip(planes[bd], ("src%d", "ref0_%d", "ref1_%d")):
            ctx.planes_upload(p_, 0, np.ascontiguousarray(z[name % bd][B:B + H, B:B + W]))
    n = 0
    for c in meta["cases"]:
        bd, w, h, k = c["bd"], c["w"], c["h"], c["k"]
        ps, p0, p1 = planes[bd]
        b = np.zeros(1, capi.search_block_dtype)
        b["bx"], b["by"] = c["bx"], c["by"]
        b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
        kw = {} if "mesh_thr" not in c else dict(force_mesh_thresh=c["mesh_thr"])
        full = capi.SearchParams.make("NSTEP", 5, 0, c["sadperbit"], c["errorperbit"], mesh_diff_thr=4, mesh=meta["mesh"], **kw)
        sub = capi.SubpelParams(TREES[c["tree"]], 0, c["errorperbit"], 2, 1, 3, TAPS[c["taps"]])
        d_b = ctx.to_device(b)
        d_m = ctx.to_device(np.ascontiguousarray(z["mask%d" % k])) if c["masked"] else None
        d_rate, d_err = ctx.malloc(16), ctx.malloc(16)
        fi = c.get("force_int", 0)
        extra = []
        if c["fn"] == "joint":
            d_r, d_cur = ctx.to_device(np.array(c["ref_mv"], np.int16)), ctx.to_device(np.array(c["cur_in"], np.int16))
            if c["ext"]:
                ctx.joint_motion_search_extensive_batch(ps, p0, p1, 0, w, h, full, sub, c["second"], fi, d_b, d_r, d_cur, d_m, 1, d_rate, d_err, d_j,
                                                        d_c0 + mv_max * 4, d_c1 + mv_max * 4)
            else:
                ctx.joint_motion_search_batch(ps, p0, p1, 0, w, h, 0, c["sadperbit"], sub, fi, d_b, d_r, d_cur, d_m, 1, d_rate, d_err, d_j, d_c0 + mv_max * 4,
                                              d_c1 + mv_max * 4)
            got = (ctx.from_device(d_cur, (2, 2), np.int16).tolist(), int(ctx.from_device(d_rate, (1,), np.int32)[0]), int(ctx.from_device(d_err, (1,), np.int32)[0]))
            assert got == (c["cur_out"], c["rate_mv"], c["err"]), (c, got)
        else:
            ri = c["ref_idx"]
            dt = np.uint8 if bd == 8 else np.uint16
            d_r, d_cur = ctx.to_device(np.array(c["ref_mv"][ri], np.int16)), ctx.to_device(np.array(c["cur_in"][ri], np.int16))
            d_sp = ctx.to_device(np.ascontiguousarray(z["sp%d" % k].astype(dt)))
            extra.append(d_sp)
            ctx.compound_single_motion_search_batch(ps, (p0, p1)[ri], None, 0, w, h, full, sub, fi, d_b, d_r, d_cur, None, 0, 0, d_sp, d_m, ri, 1, d_rate, d_err, d_j,
                                                    d_c0 + mv_max * 4, d_c1 + mv_max * 4)
            got = (ctx.from_device(d_cur, (2,), np.int16).tolist(), int(ctx.from_device(d_rate, (1,), np.int32)[0]), int(ctx.from_device(d_err, (1,), np.int32)[0]))
            assert got == (c["this_out"], c["rate_mv"], c["err"]), (c, got)
        n += 1
        for d in [d_b, d_rate, d_err, d_r, d_cur] + extra + ([d_m] if d_m is not None else []):
            ctx.free(d)
    assert n >= 22
    for d in (d_j, d_c0, d_c1):
        ctx.free(d)
    for ps_ in planes.values():
        for p_ in ps_:
            ctx.planes_free(p_)


def test_contiguous_predictor_equals_the_plane_form(hip, ctx):
    capi = hip.capi
    W, H, B, bw, bh, bd = 128, 96, 64, 16, 8, 10
    rng = np.random.default_rng(3)
    _, ref = hip.synth.shifted_smooth_pair(W, H, 2, bd, shift=(1, 1), frac8=(0, 0))
    pr, pp = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(pr, 0, ref)
    n = (W // bw) * (H // bh)
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % (W // bw)) * bw, (np.arange(n) // (W // bw)) * bh
    mv = rng.integers(-60, 61, (n, 2)).astype(np.int16)
    d_b, d_mv, d_out = ctx.to_device(blocks), ctx.to_device(mv), ctx.malloc(n * bw * bh * 2)
    ctx.build_inter_pred_batch(pr, 0, pp, 0, bw, bh, d_b, d_mv, n, 0, 0)
    ctx.build_inter_pred_contiguous_batch(pr, 0, d_out, bw, bh, d_b, d_mv, n, 0, 0)
    plane = ctx.planes_download(pp, 0)[B:B + H, B:B + W]
    got = ctx.from_device(d_out, (n, bh, bw), np.uint16)
    want = np.stack([plane[b["by"]:b["by"] + bh, b["bx"]:b["bx"] + bw] for b in blocks])
    assert np.array_equal(got, want)
    for d in (d_b, d_mv, d_out):
        ctx.free(d)
    ctx.planes_free(pr); ctx.planes_free(pp)
