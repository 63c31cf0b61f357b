"""Parity of the device-side motion search with the oracle (bit-exact MVs and costs): full-pel diamond search
(DIAMOND / CLAMPED_DIAMOND, several step_param, all supported MV cost types, tight and frame-edge limits,
8/10/12-bit, several block sizes) and the bilinear sub-pel tree (all forced_stop levels, iters 1/2, allow_hp)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, n, start_range=0, ref_range=0):
    b = np.zeros(n, hip.capi.search_block_dtype)
    b["bx"] = rng.integers(0, W - bw + 1, n); b["by"] = rng.integers(0, H - bh + 1, n)
    b["start_row"] = rng.integers(-start_range, start_range + 1, n); b["start_col"] = rng.integers(-start_range, start_range + 1, n)
    b["ref_row"] = rng.integers(-ref_range, ref_range + 1, n); b["ref_col"] = rng.integers(-ref_range, ref_range + 1, n)
    for i in range(n):
        lim = oracle.mv_limits_for_block(int(b["bx"][i]), int(b["by"][i]), bw, bh, W, H, border, int(b["ref_row"][i]), int(b["ref_col"][i]))
        b["row_min"][i], b["row_max"][i], b["col_min"][i], b["col_max"][i] = lim
    return b


def _upload(hip, ctx, src, ref, border, bd):
    H, W = src.shape
    ps, pr = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 2)
    ctx.planes_upload(ps, 1, src); ctx.planes_upload(pr, 1, ref)
    return ps, pr


def _fullpel(hip, ctx, ps, pr, bw, bh, blocks, clamped, step_param, cost_type):
    n = len(blocks)
    d_b, d_mv, d_c = ctx.to_device(blocks), ctx.malloc(max(16, n * 4)), ctx.malloc(max(16, n * 4))
    ctx.fullpel_diamond_batch(ps, pr, 1, bw, bh, clamped, step_param, cost_type, d_b, n, d_mv, d_c)
    mv, cost = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_c, (n,), np.int32)
    for d in (d_b, d_mv, d_c):
        ctx.free(d)
    return mv, cost


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (32, 32), (64, 64), (4, 4), (16, 8), (128, 128), (8, 32)])
def test_fullpel_diamond_matches_oracle(hip, oracle, ctx, bw, bh, bd):
    rng = np.random.default_rng(bw * 3 + bh + bd)
    W, H, border = 384, 256, 160
    src, ref = hip.synth.shifted_smooth_pair(W, H, bw + bd, bd, shift=(int(rng.integers(-9, 10)), int(rng.integers(-9, 10))))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3, 4, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = _upload(hip, ctx, src, ref, border, bd)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    blocks = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, 257, start_range=6, ref_range=40)
    blocks["col_max"][::7] = np.minimum(blocks["col_max"][::7], 3)   # some tight limits
    blocks["row_min"][::5] = np.maximum(blocks["row_min"][::5], -2)
    for clamped, step_param, cost_type in [(0, 4, 3), (0, 0, 4), (1, 2, 1), (0, 7, 2), (0, 10, 3)]:
        mv, cost = _fullpel(hip, ctx, ps, pr, bw, bh, blocks, clamped, step_param, cost_type)
        wmv, wcost = oracle.fullpel_diamond_batch(sb, rb, border, bw, bh, blocks, clamped, step_param, cost_type, bd)
        assert np.array_equal(mv, wmv), (bw, bh, bd, clamped, step_param, cost_type)
        assert np.array_equal(cost, wcost)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_fullpel_noise_content_and_entropy_refusal(hip, oracle, ctx):
    """Pure noise: the greedy descent takes data-dependent paths everywhere (ties, num00 restarts)."""
    rng = np.random.default_rng(1)
    W, H, border = 256, 192, 160
    src, ref = hip.synth.lcg_frame(W, H, 4), hip.synth.lcg_frame(W, H, 5)
    ps, pr = _upload(hip, ctx, src, ref, border, 8)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    blocks = _mk_blocks(hip, oracle, rng, W, H, 16, 16, border, 600, start_range=20, ref_range=100)
    for step_param in (0, 3, 6, 9):
        mv, cost = _fullpel(hip, ctx, ps, pr, 16, 16, blocks, 0, step_param, 3)
        wmv, wcost = oracle.fullpel_diamond_batch(sb, rb, border, 16, 16, blocks, 0, step_param, 3, 8)
        assert np.array_equal(mv, wmv) and np.array_equal(cost, wcost), step_param
    with pytest.raises(hip.capi.AomHipError):
        _fullpel(hip, ctx, ps, pr, 16, 16, blocks, 0, 4, 0)  # MV_COST_ENTROPY needs cost tables: refused, not faked
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (32, 16), (64, 64), (4, 8)])
def test_subpel_bilinear_matches_oracle(hip, oracle, ctx, bw, bh, bd):
    rng = np.random.default_rng(bw + bh * 5 + bd)
    W, H, border = 320, 192, 160
    src, ref = hip.synth.shifted_smooth_pair(W, H, 11 + bd, bd, shift=(3, -2), frac8=(int(rng.integers(0, 8)), int(rng.integers(0, 8))))
    ps, pr = _upload(hip, ctx, src, ref, border, bd)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    blocks = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, 193, start_range=0, ref_range=30)
    fmv, _ = oracle.fullpel_diamond_batch(sb, rb, border, bw, bh, blocks, 0, 4, 3, bd)
    sp = blocks.copy()
    sp["start_row"], sp["start_col"] = fmv[:, 0] * 8, fmv[:, 1] * 8
    for k in ("row_min", "row_max", "col_min", "col_max"):
        sp[k] = np.clip(blocks[k].astype(np.int32) * 8, -16383, 16383)
    sp["col_max"][::6] = sp["start_col"][::6] + 1   # sub-pel limits that cut the pattern
    n = len(sp)
    d_b = ctx.to_device(sp)
    d_mv, d_e, d_d, d_s = ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 4)
    for cost_type, iters, allow_hp, forced_stop in [(3, 2, 1, 0), (4, 1, 1, 0), (1, 2, 0, 0), (3, 2, 1, 1), (2, 2, 1, 2), (3, 1, 1, 3)]:
        ctx.subpel_bilinear_batch(ps, pr, 1, bw, bh, cost_type, iters, allow_hp, forced_stop, d_b, n, d_mv, d_e, d_d, d_s)
        got = (ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_e, (n,), np.uint32),
               ctx.from_device(d_d, (n,), np.int32), ctx.from_device(d_s, (n,), np.uint32))
        want = oracle.subpel_bilinear_batch(sb, rb, border, bw, bh, sp, cost_type, iters, allow_hp, forced_stop, bd)
        for g, w_, name in zip(got, want, ("mv", "err", "distortion", "sse")):
            assert np.array_equal(g, w_), (bw, bh, bd, cost_type, iters, allow_hp, forced_stop, name)
    for d in (d_b, d_mv, d_e, d_d, d_s):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (32, 32), (64, 64), (4, 4), (16, 8), (8, 32), (128, 128)])
def test_mesh_search_matches_oracle(hip, oracle, ctx, bw, bh, bd):
    """full_pixel_exhaustive: every row of good_quality_mesh_patterns + an intrabc-style dense pattern, tight limits
    (column spans of every length mod 4 exercise the reference's four-at-a-time column rule), start MVs that grow the
    first range, fine_search_interval, all cost types."""
    rng = np.random.default_rng(bw * 5 + bh + bd)
    W, H, border = 320, 256, 160
    src, ref = hip.synth.shifted_smooth_pair(W, H, bw + bd, bd, shift=(int(rng.integers(-12, 13)), int(rng.integers(-12, 13))))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3, 4, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = _upload(hip, ctx, src, ref, border, bd)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    n = 40 if bw * bh <= 1024 else 12
    blocks = _mk_blocks(hip, oracle, rng, W, H, bw, bh, border, n, start_range=20, ref_range=40)
    k = np.arange(n)
    blocks["col_max"][::3] = np.minimum(blocks["col_max"][::3], blocks["start_col"][::3] + (k[::3] % 7))  # spans mod 4
    blocks["col_min"][1::4] = np.maximum(blocks["col_min"][1::4], blocks["start_col"][1::4] - (k[1::4] % 5))
    blocks["row_min"][::5] = np.maximum(blocks["row_min"][::5], -3)
    blocks["start_row"][2::6] = np.clip(blocks["start_row"][2::6] * 4, blocks["row_min"][2::6], blocks["row_max"][2::6])  # larger |mv|
    pats = list(oracle.GOOD_QUALITY_MESH_PATTERNS[::2]) + [[(16, 1), (16, 1), (0, 0), (0, 0)], [(64, 4), (16, 1), (0, 0), (0, 0)]]
    d_b, d_mv, d_c = ctx.to_device(blocks), ctx.malloc(n * 4), ctx.malloc(n * 4)
    for pi, pat in enumerate(pats):
        for fine, cost_type in ((0, 3), (1, 1), (0, 4)) if pi == 0 else ((0, 3),):
            ctx.mesh_search_batch(ps, pr, 1, bw, bh, cost_type, pat, fine, d_b, n, d_mv, d_c)
            mv, cost = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_c, (n,), np.int32)
            wmv, wcost = oracle.mesh_search_batch(sb, rb, border, bw, bh, blocks, pat, fine, cost_type, bd)
            assert np.array_equal(mv, wmv), (bw, bh, bd, pi, fine, cost_type, np.nonzero((mv != wmv).any(1))[0][:5])
            assert np.array_equal(cost, wcost)
    # illegal pattern -> INT_MAX, MV = start (mcomp.c:1567-1570)
    ctx.mesh_search_batch(ps, pr, 1, bw, bh, 3, [(4, 1), (0, 0), (0, 0), (0, 0)], 0, d_b, n, d_mv, d_c)
    assert (ctx.from_device(d_c, (n,), np.int32) == np.iinfo(np.int32).max).all()
    assert np.array_equal(ctx.from_device(d_mv, (n, 2), np.int16), np.stack([blocks["start_row"], blocks["start_col"]], 1))
    for d in (d_b, d_mv, d_c):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_mesh_search_4k_10bit_tile(hip, oracle, ctx):
    """One 4K 10-bit superblock row of 16x16 blocks, speed-0 pattern: device == oracle; the mesh finds the global shift."""
    rng = np.random.default_rng(21)
    W, H, border, bd = 3840, 128, 160, 10
    src, ref = hip.synth.shifted_smooth_pair(W, H, 4, bd, shift=(-6, 11))
    ps, pr = _upload(hip, ctx, src, ref, border, bd)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    bx, by = np.meshgrid(np.arange(0, W, 16), np.arange(0, H, 16))
    n = bx.size
    blocks = np.zeros(n, hip.capi.search_block_dtype)
    blocks["bx"], blocks["by"] = bx.ravel(), by.ravel()
    for i in range(n):
        lim = oracle.mv_limits_for_block(int(blocks["bx"][i]), int(blocks["by"][i]), 16, 16, W, H, border)
        blocks["row_min"][i], blocks["row_max"][i], blocks["col_min"][i], blocks["col_max"][i] = lim
    d_b, d_mv, d_c = ctx.to_device(blocks), ctx.malloc(n * 4), ctx.malloc(n * 4)
    ctx.mesh_search_batch(ps, pr, 1, 16, 16, 3, oracle.GOOD_QUALITY_MESH_PATTERNS[0], 0, d_b, n, d_mv, d_c)
    mv, cost = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_c, (n,), np.int32)
    wmv, wcost = oracle.mesh_search_batch(sb, rb, border, 16, 16, blocks, oracle.GOOD_QUALITY_MESH_PATTERNS[0], 0, 3, bd, threads=8)
    assert np.array_equal(mv, wmv) and np.array_equal(cost, wcost)
    assert ((mv == np.array([11, -6])).all(1)).mean() > 0.3  # (coarse first pass: not every block lands on it)
    for d in (d_b, d_mv, d_c):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
