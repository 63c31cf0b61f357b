"""aomhip_motion_estimation_batch (csrc/tf_search.hip): tpl_model.c's motion_estimation (av1/encoder/tpl_model.c:248-301) for a block list --
av1_full_pixel_search around a centre MV, then the sub-pel search, with both limit sets derived on the device -- against the oracle's
composition (whose limit derivations are pinned by the interpreted reference, tests/test_oracle_me.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd,bs,method,cost,tree,ucl", [(8, 16, "NSTEP", "ENTROPY", "tree", 0), (10, 16, "DIAMOND", "L1_HDRES", "pruned_more", 1),
                                                       (8, 32, "NSTEP_8PT", "ENTROPY", "pruned", 1), (10, 32, "BIGDIA", "NONE", "tree", 0),
                                                       (8, 8, "SQUARE", "ENTROPY", "pruned_more", 0)])
def test_device_matches_the_oracle(hip, oracle, ctx, bd, bs, method, cost, tree, ucl):
    capi = hip.capi
    W, H, B = 320, 256, 64
    rng = np.random.default_rng(bd + bs + len(method))
    src, ref = hip.synth.shifted_smooth_pair(W, H, 2, bd, shift=(3, -2), frac8=(3, 6))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-4, 5, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    blocks["ref_row"], blocks["ref_col"] = rng.integers(-70, 71, n), rng.integers(-70, 71, n)    # centre MVs in 1/8 pel, most not multiples of 8
    blocks["ref_row"][::5] = 0; blocks["ref_col"][::5] = 0
    ext = B - 8                                                                                   # raw x->mv_limits (av1_set_mv_limits)
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext), W - blocks["bx"] - bs + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bs + ext
    ct = {"ENTROPY": 0, "L1_HDRES": 3, "NONE": 4}[cost]
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    full = capi.SearchParams.make(method, 2, ct, sad_per_bit=20, error_per_bit=64)
    sub = capi.SubpelParams(capi.SUBPEL_TREES[tree], capi.MV_COST_NONE, 64, 2, 1, 0, 1)          # USE_2_TAPS (1), MV_COST_NONE (tpl_model.c:293-294)
    d_b = ctx.to_device(blocks)
    d_mv, d_err, d_dist, d_sse, d_fmv = (ctx.malloc(n * 4) for _ in range(5))
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    ctx.motion_estimation_batch(ps, pr, 0, bs, bs, full, sub, ucl, d_b, n, d_mv, d_err, d_dist, d_sse, d_fmv, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    got = (ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_err, (n,), np.uint32), ctx.from_device(d_dist, (n,), np.int32),
           ctx.from_device(d_sse, (n,), np.uint32), ctx.from_device(d_fmv, (n, 2), np.int16))
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    oq = oracle.search_params(method, 2, ct, sad_per_bit=20, error_per_bit=64, no_cost_list=int(not ucl))
    want = oracle.motion_estimation_batch(sb, rb, B, bs, bs, blocks, oq, dict(tree=tree, cost_type=4, error_per_bit=64, iters=2, allow_hp=1, forced_stop=0, subpel_search_type=1),
                                          ucl, tj, t0, t1, bd=bd, threads=8)
    for g, w_, name in zip(got, want, ("mv", "err", "distortion", "sse", "full_mv")):
        assert np.array_equal(g, w_), name
    assert (got[0] & 7).any() and got[4].any()
    for d in (d_b, d_mv, d_err, d_dist, d_sse, d_fmv, d_j, d_c0, d_c1):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
