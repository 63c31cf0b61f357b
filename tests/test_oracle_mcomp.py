"""Motion-search oracle sanity (no reference unit test drives mcomp.c): on smooth content
with a known global shift the diamond search converges to it, its returned cost equals variance + MV cost
at the winner, limits are respected, and the bilinear sub-pel refinement never increases the error."""
import numpy as np
import pytest


def _blocks(hip, oracle, W, H, bs, border, start=(0, 0), ref=(0, 0), subpel_from=None):
    xs, ys = np.meshgrid(np.arange(32, W - 32 - bs, bs * 2), np.arange(32, H - 32 - bs, bs * 2))
    n = xs.size
    b = np.zeros(n, hip.capi.search_block_dtype)
    b["bx"], b["by"] = xs.ravel(), ys.ravel()
    b["start_row"], b["start_col"] = start
    b["ref_row"], b["ref_col"] = ref
    for i in range(n):
        lim = oracle.mv_limits_for_block(int(b["bx"][i]), int(b["by"][i]), bs, bs, W, H, border, *ref)
        b["row_min"][i], b["row_max"][i], b["col_min"][i], b["col_max"][i] = lim
    return b


@pytest.mark.parametrize("bd", [8, 10])
def test_diamond_recovers_global_shift(hip, oracle, bd):
    W, H, border = 320, 192, 96
    dx, dy = 5, -3
    src, ref = hip.synth.shifted_smooth_pair(W, H, 7, bd, shift=(dx, dy))
    sb, rb = oracle.extend_plane(src, border), oracle.extend_plane(ref, border)
    blocks = _blocks(hip, oracle, W, H, 16, border)
    mv, cost = oracle.fullpel_diamond_batch(sb, rb, border, 16, 16, blocks, 0, 4, 3, bd)
    # ref(x, y) = src(x - dx, y - dy): the best match of a source block lies at mv (row, col) = (dy, dx)
    hit = (mv[:, 0] == dy) & (mv[:, 1] == dx)
    assert hit.mean() > 0.8
    for i in np.nonzero(hit)[0][:10]:
        b = blocks[i]
        v, sse, _ = oracle.variance(sb, border + b["by"], border + b["bx"], rb, border + b["by"] + dy, border + b["bx"] + dx, 16, 16, bd)
        assert cost[i] == v + ((1 * (abs(8 * dy) + abs(8 * dx))) >> 3)  # MV_COST_L1_HDRES, SSE lambda 1, ref_mv 0
    # tight limits clamp the result
    blocks2 = blocks.copy(); blocks2["col_max"] = 2; blocks2["row_min"] = -1
    mv2, _ = oracle.fullpel_diamond_batch(sb, rb, border, 16, 16, blocks2, 0, 4, 3, bd)
    assert (mv2[:, 1] <= 2).all() and (mv2[:, 0] >= -1).all()


def test_subpel_refinement_monotone(hip, oracle):
    W, H, border = 256, 128, 64
    src, ref = hip.synth.shifted_smooth_pair(W, H, 3, 8, shift=(2, 1), frac8=(4, 2))  # true motion (2.5, 1.25) px
    sb, rb = oracle.extend_plane(src, border), oracle.extend_plane(ref, border)
    blocks = _blocks(hip, oracle, W, H, 16, border)
    mv, _ = oracle.fullpel_diamond_batch(sb, rb, border, 16, 16, blocks, 0, 4, 4, 8)
    sp = blocks.copy()
    sp["start_row"], sp["start_col"] = mv[:, 0] * 8, mv[:, 1] * 8
    for k in ("row_min", "row_max", "col_min", "col_max"):
        sp[k] = np.clip(blocks[k].astype(np.int32) * 8, -16383, 16383)
    m0, e0, d0, s0 = oracle.subpel_bilinear_batch(sb, rb, border, 16, 16, sp, 4, 2, 1, 3)  # FULL_PEL: centre only
    m1, e1, d1, s1 = oracle.subpel_bilinear_batch(sb, rb, border, 16, 16, sp, 4, 2, 1, 0)
    assert np.array_equal(m0, np.stack([sp["start_row"], sp["start_col"]], 1))
    assert (e1 <= e0).all() and (e1 < e0).any()
    assert (np.abs(m1 - m0) <= 7).all()  # 4 + 2 + 1 eighth-pel at most per axis


def test_mesh_search_is_argmin_with_reference_column_rule(hip, oracle):
    """exhaustive_mesh_search against an independent numpy restatement: arg-min of sad + cost over exactly the
    positions the reference visits (four columns at a time at interval 1, the tail group skipping column end_col,
    mcomp.c:1512-1537), first in raster order on ties, the start position winning ties."""
    rng = np.random.default_rng(5)
    W, H, border, bd, bw = 160, 128, 64, 8, 16
    src = rng.integers(0, 256, (H, W)).astype(np.uint8); ref = rng.integers(0, 256, (H, W)).astype(np.uint8)
    sb, rb = oracle.extend_plane(src, border), oracle.extend_plane(ref, border)
    n = 24
    b = np.zeros(n, hip.capi.search_block_dtype)
    b["bx"], b["by"] = rng.integers(0, W - bw, n), rng.integers(0, H - bw, n)
    for i in range(n):
        lim = oracle.mv_limits_for_block(int(b["bx"][i]), int(b["by"][i]), bw, bw, W, H, border)
        b["row_min"][i], b["row_max"][i], b["col_min"][i], b["col_max"][i] = lim
    b["col_max"] = np.minimum(b["col_max"], b["start_col"] + np.arange(n) % 9)  # spans of every length mod 4
    pat = [(7, 1), (0, 0), (0, 0), (0, 0)]  # single dense pass (interval 1: no progressive passes)
    mv, cost = oracle.mesh_search_batch(sb, rb, border, bw, bw, b, pat, 0, 4, bd)  # MV_COST_NONE
    for i in range(n):
        x, y = int(b["bx"][i]) + border, int(b["by"][i]) + border
        blk = sb[y:y + bw, x:x + bw].astype(np.int32)
        sr = min(max(0, b["row_min"][i]), b["row_max"][i]); sc = min(max(0, b["col_min"][i]), b["col_max"][i])
        r0, r1 = max(-7, b["row_min"][i] - sr), min(7, b["row_max"][i] - sr)
        c0, c1 = max(-7, b["col_min"][i] - sc), min(7, b["col_max"][i] - sc)
        span = c1 - c0 + 1
        ncols = 4 * (span // 4) + max(span % 4 - 1, 0) if span > 0 else 0
        best, best_mv = int(np.abs(blk - rb[y + sr:y + sr + bw, x + sc:x + sc + bw]).sum()), (sr, sc)
        for r in range(r0, r1 + 1):
            for c in range(c0, c0 + ncols):
                s = int(np.abs(blk - rb[y + sr + r:y + sr + r + bw, x + sc + c:x + sc + c + bw]).sum())
                if s < best:
                    best, best_mv = s, (sr + r, sc + c)
        assert tuple(mv[i]) == best_mv, i
