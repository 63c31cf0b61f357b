"""aomhip_lpf_search_sse: the trial loop of av1_pick_filter_level (try_filter_frame, av1/encoder/picklpf.c:49-86 --
deblock a copy, aom_get_sse_plane against the source) for a list of candidate levels in one call, against the oracle
(oracle deblock_plane per trial + the plane's squared error); the reconstruction must come back untouched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10])
def test_lpf_trials_vs_oracle(hip, oracle, ctx, bd):
    rng = np.random.default_rng(bd)
    W, H, border = 320, 192, 32
    recon = hip.synth.lcg_frame(W, H, 4, 0, bd)
    # blocky reconstruction: quantise 8x8 means so that the filters have real edges to work on
    blk = recon.reshape(H // 8, 8, W // 8, 8).mean(axis=(1, 3), keepdims=True)
    recon = np.clip(0.5 * recon.reshape(H // 8, 8, W // 8, 8) + 0.5 * blk, 0, (1 << bd) - 1).astype(recon.dtype).reshape(H, W)
    source = hip.synth.lcg_frame(W, H, 4, 0, bd)
    pr, ps = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pr, 0, recon); ctx.planes_upload(ps, 0, source)
    base = oracle.random_edge_params(rng, W, H)
    levels = [0, 3, 9, 17, 30, 47, 63]
    trials = []
    for lv in levels:
        p = base.copy()
        p[..., 1] = np.where(p[..., 0] > 0, lv, 0)      # every coded edge filtered at this trial's level
        p[..., 3] = np.where(p[..., 2] > 0, lv, 0)
        if lv == 30:                                      # a partial-frame trial: only the middle rows are filtered
            p[:H // 16] = 0
            p[3 * H // 16:] = 0
        trials.append(p)
    stack = np.ascontiguousarray(np.stack(trials))
    d_p, d_sse = ctx.to_device(stack), ctx.malloc(8 * len(levels))
    ctx.lpf_search_sse(pr, 0, pr, 1, ps, 0, d_p, stack[0].size, len(levels), stack.shape[2], 2, 3, d_sse)
    got = ctx.from_device(d_sse, (len(levels),), np.uint64)
    for t, p in enumerate(trials):
        filt = oracle.deblock_plane(recon, p, 2, bd)
        assert int(got[t]) == int(((filt.astype(np.int64) - source.astype(np.int64)) ** 2).sum()), (bd, levels[t])
    assert int(got[0]) == int(((recon.astype(np.int64) - source.astype(np.int64)) ** 2).sum())   # level 0 filters nothing
    assert np.array_equal(ctx.planes_download(pr, 0)[border:border + H, border:border + W], recon)  # untouched
    with pytest.raises(hip.capi.AomHipError):
        ctx.lpf_search_sse(pr, 0, pr, 0, ps, 0, d_p, stack[0].size, 1, stack.shape[2], 2, 3, d_sse)      # scratch == reconstruction
    with pytest.raises(hip.capi.AomHipError):
        ctx.lpf_search_sse(pr, 0, pr, 1, ps, 0, d_p, 16, 2, stack.shape[2], 2, 3, d_sse)                  # trials overlap
    ctx.free(d_p); ctx.free(d_sse)
    ctx.planes_free(pr); ctx.planes_free(ps)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_plane_sse(hip, ctx, bd):
    """aom_get_y_sse on whole planes (odd width: the visible area only, never the border)."""
    rng = np.random.default_rng(bd)
    W, H, border = 333, 77, 16
    a = rng.integers(0, 1 << bd, (H, W)).astype(np.uint8 if bd == 8 else np.uint16)
    b = rng.integers(0, 1 << bd, (H, W)).astype(a.dtype)
    a[:5], b[:5] = (1 << bd) - 1, 0
    pa, pb = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pa, 1, a); ctx.planes_upload(pb, 0, b)
    d = ctx.malloc(8)
    ctx.plane_sse(pa, 1, pb, 0, d)
    assert int(ctx.from_device(d, (1,), np.uint64)[0]) == int(((a.astype(np.int64) - b.astype(np.int64)) ** 2).sum())
    ctx.plane_sse(pb, 0, pb, 0, d)
    assert int(ctx.from_device(d, (1,), np.uint64)[0]) == 0
    with pytest.raises(hip.capi.AomHipError):
        ctx.plane_sse(pa, 2, pb, 0, d)
    ctx.free(d); ctx.planes_free(pa); ctx.planes_free(pb)
