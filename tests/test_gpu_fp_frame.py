"""aomhip_first_pass_inter_frame (csrc/tf_search.hip): the inter half of a first-pass frame in one call, the best_ref_mv chain along each
block row kept on the device, against the oracle's scalar raster walk of firstpass_inter_prediction (av1/encoder/firstpass.c:690-815)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables():
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    return mv_max, np.array([200, 650, 640, 1050], np.int32), (150 + bits * 310).astype(np.int32), (170 + bits * 290 + (v & 7) * 3).astype(np.int32)


@pytest.mark.parametrize("bd,golden,thr,skip_zero,cost,bs", [(8, True, 0, 0, "ENTROPY", 16), (10, True, 300, 0, "ENTROPY", 16), (8, False, 0, 1, "L1_HDRES", 16),
                                                            (10, False, 0, 0, "NONE", 16),
                                                            (8, True, 0, 0, "ENTROPY", 8)])   # fp_block_size BLOCK_8X8: frames of at most 352x288 (firstpass.c get_fp_block_size)
# the row kernel (csrc/fp_row.hip) as shipped -- 16 wavefronts per row speculating along the chain, a batch's window of reach 64 in LDS, golden leg on
# the side stream --, with 8 wavefronts, with one wavefront per row on one stream (no speculation), without a window, with a window too small for
# most steps, and the column-at-a-time fallback: the intra errors below reset the chain at a fifth of the blocks and perturb the rest, so batches
# are cut short at every width
@pytest.mark.parametrize("form", ["rows", "rows_1_serial", "rows_16", "rows_16_no_window", "rows_16_reach_8", "columns"])
def test_frame_call_equals_the_scalar_raster_walk(hip, oracle, ctx, bd, golden, thr, skip_zero, cost, bs, form, monkeypatch):
    monkeypatch.setenv("AOMHIP_FP_COLUMNS", "1" if form == "columns" else "0")
    monkeypatch.setenv("AOMHIP_FP_ROW_WAVES", {"rows_1_serial": "1", "rows": "8"}.get(form, "16"))
    monkeypatch.setenv("AOMHIP_FP_ROW_R", {"rows_16_no_window": "-1", "rows_16_reach_8": "8"}.get(form, "64"))
    monkeypatch.setenv("AOMHIP_FP_SERIAL", "1" if form == "rows_1_serial" else "0")
    capi = hip.capi
    W, H, B = (352, 288, 64) if bs == 16 else (176, 144, 64)
    rng = np.random.default_rng(7 * bd + thr + skip_zero)
    # the frame moved by (5, -7) against the last reconstruction and by (-2, 3) against the golden frame; noise so that errors differ per block
    src, last = hip.synth.shifted_smooth_pair(W, H, 11, bd, shift=(5, -7), frac8=(0, 0))
    _, gold = hip.synth.shifted_smooth_pair(W, H, 11, bd, shift=(-2, 3), frac8=(0, 0))
    hi = (1 << bd) - 1
    noisy = lambda a, k: np.clip(a.astype(np.int32) + rng.integers(-k, k + 1, a.shape), 0, hi).astype(a.dtype)
    last, gold, lsrc = noisy(last, 3), noisy(gold, 5), noisy(last, 6)
    lsrc[:4 * bs, :] = src[:4 * bs, :]                      # four rows of blocks whose raw_motion_error is 0: the search is skipped at any threshold
    rings = [ctx.planes_alloc(W, H, B, bd, 2) for _ in range(4)]
    ps, pl, pg, pls = rings
    frames = {0: (ps, 1, src), 1: (pl, 0, last), 2: (pg, 1, gold), 3: (pls, 1, lsrc)}
    for ring, f, img in frames.values():
        ctx.planes_upload(ring, f, img)
        ctx.planes_upload(ring, 1 - f, np.full_like(img, 7))   # the other slot must never be read
    rows, cols = H // bs, W // bs
    n = rows * cols
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % cols) * bs, (np.arange(n) // cols) * bs
    blocks["ref_row"], blocks["start_col"] = 99, -99           # ignored by the call
    ext = B - 8
    blocks["col_min"] = np.maximum(-(blocks["bx"] + ext), -1023); blocks["col_max"] = np.minimum(W - blocks["bx"] - bs + ext, 1023)
    blocks["row_min"] = np.maximum(-(blocks["by"] + ext), -1023); blocks["row_max"] = np.minimum(H - blocks["by"] - bs + ext, 1023)
    ct = {"ENTROPY": 0, "L1_HDRES": 3, "NONE": 4}[cost]
    mv_max, tj, t0, t1 = _tables()
    q = capi.SearchParams.make("NSTEP_FPF", 2, ct, sad_per_bit=24, error_per_bit=70)
    oq = oracle.search_params("NSTEP_FPF", 2, ct, sad_per_bit=24, error_per_bit=70, no_cost_list=1)
    sb, lb, gb, lsb = (oracle.extend_plane(a, B, ps.stride) for a in (src, last, gold, lsrc))
    # intra errors around the inter errors, so that the chain is broken (best_mv = 0) at a good share of the blocks
    _, _, base_err, _, _ = oracle.first_pass_inter_frame(sb, lb, None, lsb, B, bs, blocks, rows, cols, oq, np.full(n, 2**31 - 1, np.int64), thr, skip_zero, tj, t0,
                                                         t1, bd=bd)
    intra = (base_err.astype(np.int64) + rng.integers(-40, 120, n)).astype(np.int32)
    intra[rng.random(n) < 0.2] = 0
    want = oracle.first_pass_inter_frame(sb, lb, gb if golden else None, lsb, B, bs, blocks, rows, cols, oq, intra, thr, skip_zero, tj, t0, t1, bd=bd)
    fp = capi.FirstPassParams(rows, cols, thr, skip_zero)
    d_b, d_i = ctx.to_device(blocks), ctx.to_device(intra)
    outs = [ctx.malloc(n * 4) for _ in range(5)]
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    ctx.first_pass_inter_frame(ps, 1, pl, 0, pg if golden else None, 1, pls, 1, bs, bs, q, fp, d_b, d_i, outs[0], outs[2], outs[1], outs[3], outs[4], d_j,
                               d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    got = (ctx.from_device(outs[0], (n, 2), np.int16), ctx.from_device(outs[1], (n, 2), np.int16), ctx.from_device(outs[2], (n,), np.int32),
           ctx.from_device(outs[3], (n,), np.int32), ctx.from_device(outs[4], (n,), np.int32))
    for name, g, w in zip(("best_mv", "full_mv", "motion_error", "gf_motion_error", "raw_motion_error"), got, want):
        assert np.array_equal(g, w), (name, np.flatnonzero((g != w).reshape(n, -1).any(1))[:8])
    best = want[0].reshape(rows, cols, 2)
    assert best.any() and (best == 0).all(2).any()                       # the chain both carries MVs and is reset
    moved_prev = (best[:, :-1] != 0).any(2)
    assert moved_prev.sum() > n // 8                                        # a good share of blocks searched from a non-zero best_ref_mv
    assert (want[4][: 4 * cols] == 0).all() and not want[1][: 4 * cols].any()   # raw error 0 -> no search, MV 0
    if golden:
        assert (want[3] != want[2]).any()
    for d in [d_b, d_i, d_j, d_c0, d_c1] + outs:
        ctx.free(d)
    for r in rings:
        ctx.planes_free(r)


def test_optional_outputs_and_bad_arguments(hip, ctx):
    capi = hip.capi
    W, H, B, bs = 64, 32, 32, 16
    ps, pl = ctx.planes_alloc(W, H, B, 8, 1), ctx.planes_alloc(W, H, B, 8, 1)
    other = ctx.planes_alloc(W + 16, H, B, 8, 1)
    img = (np.arange(W * H).reshape(H, W) * 7 % 251).astype(np.uint8)
    ctx.planes_upload(ps, 0, img); ctx.planes_upload(pl, 0, np.roll(img, 1, 1))
    rows, cols = H // bs, W // bs
    n = rows * cols
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % cols) * bs, (np.arange(n) // cols) * bs
    blocks["col_min"], blocks["col_max"], blocks["row_min"], blocks["row_max"] = -8, 8, -8, 8
    q = capi.SearchParams.make("NSTEP_FPF", 4, capi.MV_COST_NONE)
    d_b, d_i = ctx.to_device(blocks), ctx.to_device(np.full(n, 1 << 30, np.int32))
    d_mv, d_e = ctx.malloc(n * 4), ctx.malloc(n * 4)
    ctx.first_pass_inter_frame(ps, 0, pl, 0, None, 0, pl, 0, bs, bs, q, capi.FirstPassParams(rows, cols, 0, 0), d_b, d_i, d_mv, d_e)
    assert ctx.from_device(d_e, (n,), np.int32).min() >= 0
    ctx.first_pass_inter_frame(ps, 0, pl, 0, None, 0, pl, 0, bs, bs, q, capi.FirstPassParams(0, cols, 0, 0), d_b, d_i, d_mv, d_e)   # empty raster
    with pytest.raises(capi.AomHipError):
        ctx.first_pass_inter_frame(ps, 0, other, 0, None, 0, pl, 0, bs, bs, q, capi.FirstPassParams(rows, cols, 0, 0), d_b, d_i, d_mv, d_e)
    with pytest.raises(capi.AomHipError):
        ctx.first_pass_inter_frame(ps, 0, pl, 1, None, 0, pl, 0, bs, bs, q, capi.FirstPassParams(rows, cols, 0, 0), d_b, d_i, d_mv, d_e)
    for d in (d_b, d_i, d_mv, d_e):
        ctx.free(d)
    for r in (ps, pl, other):
        ctx.planes_free(r)
