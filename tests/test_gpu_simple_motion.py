"""aomhip_simple_motion_search_batch (csrc/tf_search.hip): av1_simple_motion_search / av1_simple_motion_sse_var
(av1/encoder/motion_search_facade.c:925-1060) for one level of the partition tree of every superblock -- full-pel search from the
parent's start MV around ref_mv = 0, the sub-pel search, the EIGHTTAP_REGULAR predictor and vf(src, pred) -- against the oracle's
composition of the pieces the interpreted reference pins (full-pel search, sub-pel trees, limits, convolve, variance)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables():
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    return (np.array([190, 660, 655, 1040], np.int32), (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32), mv_max)


@pytest.mark.parametrize("bd,bs,method,subpel,tree,ucl", [(8, 64, "NSTEP", True, "pruned", 1), (10, 32, "DIAMOND", True, "tree", 0), (8, 16, "NSTEP", False, None, 0),
                                                         (10, 128, "NSTEP", True, "pruned_more", 1), (8, 8, "BIGDIA", True, "pruned", 0)])
def test_level_of_the_tree_matches_the_oracle(hip, oracle, ctx, bd, bs, method, subpel, tree, ucl):
    capi = hip.capi
    W, H, B = 512, 384, 160
    rng = np.random.default_rng(bd + bs)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 4, bd, shift=(5, -3), frac8=(2, 5))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3, 4, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr, pp = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 2)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    blocks["start_row"], blocks["start_col"] = rng.integers(-9, 10, n), rng.integers(-9, 10, n)   # the parent level's MVs (full pel)
    blocks["start_row"][::4] = 0; blocks["start_col"][::4] = 0
    ext = B - 8 - 4 * 0
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext - 64 * 0), W - blocks["bx"] - bs + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bs + ext
    blocks["col_min"] = np.maximum(blocks["col_min"], -(blocks["bx"] + B - 16)); blocks["row_min"] = np.maximum(blocks["row_min"], -(blocks["by"] + B - 16))
    tj, t0, t1, mv_max = _tables()
    full = capi.SearchParams.make(method, 3, 0, sad_per_bit=22, error_per_bit=70)                # MV_COST_ENTROPY
    sub = capi.SubpelParams(capi.SUBPEL_TREES[tree], 0, 70, 2, 1, 0, 3) if subpel else None      # entropy cost, USE_8_TAPS, forced_stop 0
    d_b = ctx.to_device(blocks)
    d_mv, d_sse, d_var = (ctx.malloc(n * 4) for _ in range(3))
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    ctx.simple_motion_search_batch(ps, pr, 0, bs, bs, full, sub, ucl, d_b, n, pp, 1, d_mv, d_sse, d_var, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    got_mv, got_sse, got_var = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_sse, (n,), np.uint32), ctx.from_device(d_var, (n,), np.uint32)
    got_pred = ctx.planes_download(pp, 1)[B:B + H, B:B + W]
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    oq = oracle.search_params(method, 3, 0, sad_per_bit=22, error_per_bit=70, no_cost_list=int(not ucl))
    osub = dict(tree=tree, cost_type=0, error_per_bit=70, iters=2, allow_hp=1, forced_stop=0, subpel_search_type=3) if subpel else None
    want_mv, want_sse, want_var, want_pred = oracle.simple_motion_search_batch(sb, rb, B, W, H, bs, bs, blocks, oq, osub, ucl, tj, t0, t1, bd=bd, threads=8)
    assert np.array_equal(got_mv, want_mv)
    assert np.array_equal(got_pred[:gr * bs, :gc * bs], want_pred[:gr * bs, :gc * bs])
    assert np.array_equal(got_sse, want_sse) and np.array_equal(got_var, want_var)
    assert got_mv.any() and ((got_mv & 7).any() == bool(subpel))
    for d in (d_b, d_mv, d_sse, d_var, d_j, d_c0, d_c1):
        ctx.free(d)
    for p in (ps, pr, pp):
        ctx.planes_free(p)
