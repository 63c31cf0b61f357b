"""aomhip_tpl_inter_estimation_batch (csrc/tf_search.hip): the inter leg of tpl_model.c's mode_estimation (av1/encoder/tpl_model.c:620-770) for
independent blocks -- candidate pruning by SAD, motion_estimation from every remaining candidate, the EIGHTTAP_REGULAR predictor, the DCT SATD cost,
the best reference -- against the oracle's composition of the pinned pieces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd,bs,prune,n_refs", [(8, 16, 0, 2), (10, 16, 1, 3), (8, 32, 2, 2), (10, 8, 3, 1), (10, 16, 2, 3)])
def test_device_matches_the_oracle(hip, oracle, ctx, bd, bs, prune, n_refs):
    capi = hip.capi
    W, H, B = 256, 192, 64
    rng = np.random.default_rng(100 * bd + bs + prune)
    dt = np.uint8 if bd == 8 else np.uint16
    src, ref0 = hip.synth.shifted_smooth_pair(W, H, 3, bd, shift=(2, -3), frac8=(5, 2))
    refs = []
    for r in range(n_refs):
        shifted = np.roll(ref0, (r, -2 * r), (0, 1)).astype(np.int32)
        amp = np.full(W, 24)                                                   # every reference is the clean one in its own band of columns
        amp[r * W // n_refs:(r + 1) * W // n_refs] = 3
        noise = (rng.integers(-32, 33, ref0.shape) * amp[None, :]) // 32
        refs.append(np.clip(shifted + noise * (1 << (bd - 8)), 0, (1 << bd) - 1).astype(dt))
    src = src.astype(dt)
    ps = ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src)
    prs = []
    for r in range(n_refs):
        p = ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(p, 0, refs[r])
        prs.append(p)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    ext = B - 8
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext), W - blocks["bx"] - bs + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bs + ext
    centers = rng.integers(-90, 91, (n, n_refs, 4, 2)).astype(np.int16)
    centers[:, :, 0] = 0                                                      # candidate 0 is the zero MV (tpl_model.c:646-649)
    centers[::3, :, 2] = centers[::3, :, 1]                                   # equal candidates: equal SADs, the ranking must keep their order
    counts = rng.integers(1, 5, (n, n_refs)).astype(np.uint8)
    if n_refs > 1:
        counts[1::7, n_refs - 1] = 0                                          # a reference some blocks do not have
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    full = capi.SearchParams.make("NSTEP", 2, 0, sad_per_bit=20, error_per_bit=64)
    sub = capi.SubpelParams(capi.SUBPEL_TREES["pruned"], capi.MV_COST_NONE, 64, 2, 1, 0, 1)
    d_b, d_c, d_n = ctx.to_device(blocks), ctx.to_device(centers), ctx.to_device(counts)
    d_mv, d_pe, d_rf, d_bc = ctx.malloc(n * n_refs * 4), ctx.malloc(n * n_refs * 4), ctx.malloc(n), ctx.malloc(n * 4)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    ctx.tpl_inter_estimation_batch(ps, prs, 0, bs, full, sub, 1, prune, d_b, d_c, d_n, n, d_mv, d_pe, d_rf, d_bc, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    g_mv, g_pe = ctx.from_device(d_mv, (n, n_refs, 2), np.int16), ctx.from_device(d_pe, (n, n_refs), np.int32)
    g_rf, g_bc = ctx.from_device(d_rf, (n,), np.int8), ctx.from_device(d_bc, (n,), np.int32)
    sb = oracle.extend_plane(src, B, ps.stride)
    rbs = [oracle.extend_plane(refs[r], B, prs[r].stride) for r in range(n_refs)]
    oq = oracle.search_params("NSTEP", 2, 0, sad_per_bit=20, error_per_bit=64, no_cost_list=0)
    w_mv, w_pe, w_rf, w_bc = oracle.tpl_inter_estimation_batch(
        sb, rbs, B, W, H, bs, blocks, centers, counts, oq,
        dict(tree="pruned", cost_type=4, error_per_bit=64, iters=2, allow_hp=1, forced_stop=0, subpel_search_type=1), 1, prune, tj, t0, t1, bd=bd, threads=8)
    bad = np.argwhere((g_mv != w_mv).any(-1) | (g_pe != w_pe))
    assert len(bad) == 0, [(tuple(b), g_mv[tuple(b)].tolist(), w_mv[tuple(b)].tolist(), int(g_pe[tuple(b)]), int(w_pe[tuple(b)]), int(counts[tuple(b)]),
                            centers[tuple(b)].tolist()) for b in bad[:6]]
    assert np.array_equal(g_rf, w_rf) and np.array_equal(g_bc, w_bc)
    assert (g_mv[counts > 0] & 7).any() and (g_rf >= 0).all() and len(set(g_rf.tolist())) >= min(2, n_refs)
    for d in (d_b, d_c, d_n, d_mv, d_pe, d_rf, d_bc, d_j, d_c0, d_c1):
        ctx.free(d)
    ctx.planes_free(ps)
    for p in prs:
        ctx.planes_free(p)
