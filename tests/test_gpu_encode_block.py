"""aomhip_encode_inter_blocks_batch (csrc/encode_block.hip): prediction -> residual -> forward transform + quantise -> inverse + add for square
inter blocks in ONE kernel, against (a) the three device calls it replaces, in sequence, and (b) the oracle's restatement of the same chain
(av1/encoder/encodemb.c:343-470 behind av1_enc_build_inter_predictor).  Bit-exact: reconstruction, qcoeff, dqcoeff, eob."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TX_OF = {8: 1, 16: 2, 32: 3}
# (block size, TX_TYPE): DCT_DCT everywhere; the ADST / FLIPADST / identity kinds where the size has them (av1_get_fwd_txfm_cfg)
CASES = [(8, 0), (16, 0), (32, 0), (8, 1), (8, 6), (8, 9), (8, 15), (16, 3), (16, 4), (16, 5), (16, 9), (16, 10), (16, 13), (32, 9)]


def _setup(hip, oracle, ctx, bd, bw, seed, W=384, H=224, border=64, noise=24, qindex=90):
    rng = np.random.default_rng(seed)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 2, bd, shift=(2, -3), frac8=(3, 5))
    hi = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    ref = np.clip(ref.astype(np.int32) + rng.integers(-noise, noise + 1, ref.shape) * (1 << (bd - 8)), 0, hi).astype(dt)
    src = src.astype(dt)
    src[: H // 4] = rng.integers(0, hi + 1, (H // 4, W)).astype(dt)            # a band nothing predicts: large residuals, long eobs
    gc, gr = W // bw, H // bw
    n = gc * gr
    blocks = np.zeros(n, hip.capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bw
    ext = border - 8
    lim = lambda lo, hi_: (np.maximum(lo, -1023), np.minimum(hi_, 1023))
    blocks["col_min"], blocks["col_max"] = lim(-(blocks["bx"] + ext), W - blocks["bx"] - bw + ext)
    blocks["row_min"], blocks["row_max"] = lim(-(blocks["by"] + ext), H - blocks["by"] - bw + ext)
    mv = np.stack([rng.integers(blocks["row_min"] * 8, blocks["row_max"] * 8 + 1), rng.integers(blocks["col_min"] * 8, blocks["col_max"] * 8 + 1)], 1).astype(np.int16)
    mv[::7] &= ~7                                                              # some full-pel MVs (the copy / one-direction cases of the facade)
    mv[::11, 0] &= ~7
    # the last block row: zero MVs onto a reference that equals the source there -> residual 0 -> eob 0 (the inverse's skip branch)
    last = blocks["by"] == (gr - 1) * bw
    mv[last] = 0
    ref[(gr - 1) * bw:] = src[(gr - 1) * bw:]
    q = oracle.build_quantizer_y(bd, qindex)
    return src, ref, blocks, mv, q, gc, n, W, H, border


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("bw,tx_type", CASES)
def test_fused_block_kernel_equals_the_three_calls_and_the_oracle(hip, oracle, ctx, bd, bw, tx_type):
    capi = hip.capi
    src, ref, blocks, mv, q, gc, n, W, H, border = _setup(hip, oracle, ctx, bd, bw, 100 * bd + bw + tx_type)
    qp = capi.QuantParams.from_tables(q)
    nc, tx = bw * bw, TX_OF[bw]
    ps, pr, pp, pf = (ctx.planes_alloc(W, H, border, bd, 1) for _ in range(4))
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    d_b, d_mv = ctx.to_device(blocks), ctx.to_device(mv)
    d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    d_q2, d_dq2, d_e2 = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    # the chain
    ctx.build_inter_pred_batch(pr, 0, pp, 0, bw, bw, d_b, d_mv, n, 0, 0)
    pred_g = ctx.planes_download(pp, 0)[border:border + H, border:border + W].copy()
    ctx.subtract_xform_quant_batch(ps, pp, 0, tx, None, n, gc, tx_type, qp, None, d_q, d_dq, d_e)
    ctx.inv_txfm_add_batch(d_dq, tx, None, n, gc, tx_type, d_e, pp, 0)
    rec_chain = ctx.planes_download(pp, 0)[border:border + H, border:border + W].copy()
    q_chain, dq_chain, e_chain = ctx.from_device(d_q, (n, nc), np.int32), ctx.from_device(d_dq, (n, nc), np.int32), ctx.from_device(d_e, (n,), np.uint16)
    # the fused call
    ctx.memset(d_q2, 0xEE, n * nc * 4); ctx.memset(d_dq2, 0xEE, n * nc * 4)
    ctx.encode_inter_blocks_batch(ps, 0, pr, 0, pf, 0, bw, d_b, d_mv, n, qp, d_q2, d_dq2, d_e2, 0, 0, tx_type)
    rec_f = ctx.planes_download(pf, 0)[border:border + H, border:border + W].copy()
    q_f, dq_f, e_f = ctx.from_device(d_q2, (n, nc), np.int32), ctx.from_device(d_dq2, (n, nc), np.int32), ctx.from_device(d_e2, (n,), np.uint16)
    assert np.array_equal(e_f, e_chain) and np.array_equal(q_f, q_chain) and np.array_equal(dq_f, dq_chain)
    assert np.array_equal(rec_f, rec_chain)
    assert (e_chain == 0).any() and (e_chain > nc // 4).any()                  # both branches of the inverse ran
    assert (rec_chain != pred_g).any()
    # the oracle's chain
    rb = oracle.extend_plane(ref, border, pr.stride)
    pred_o = oracle.build_inter_pred(rb, border, W, H, bw, bw, blocks, mv, 0, 0, bd)
    assert np.array_equal(pred_g, pred_o)
    residual = (src.astype(np.int32) - pred_o.astype(np.int32)).astype(np.int16)
    _, q_o, dq_o, e_o = oracle.xform_quant_batch(residual, tx, None, n, gc, tx_type, q, bd > 8, n * nc, False, 8)
    rec_o = oracle.inv_txfm_add_batch(dq_o, tx, None, n, gc, tx_type, e_o, pred_o, bd)
    assert np.array_equal(e_f, e_o) and np.array_equal(q_f.ravel(), q_o) and np.array_equal(dq_f.ravel(), dq_o)
    assert np.array_equal(rec_f, rec_o.astype(rec_f.dtype))
    # coefficient outputs are optional
    ctx.memset(d_e2, 0, 2 * n)
    ctx.encode_inter_blocks_batch(ps, 0, pr, 0, pf, 0, bw, d_b, d_mv, n, qp, None, None, d_e2, 0, 0, tx_type)
    assert np.array_equal(ctx.from_device(d_e2, (n,), np.uint16), e_chain)
    assert np.array_equal(ctx.planes_download(pf, 0)[border:border + H, border:border + W], rec_chain)
    for d in (d_b, d_mv, d_q, d_dq, d_e, d_q2, d_dq2, d_e2):
        ctx.free(d)
    for p in (ps, pr, pp, pf):
        ctx.planes_free(p)


@pytest.mark.parametrize("fx,fy", [(1, 0), (2, 2), (3, 1)])
def test_fused_block_kernel_takes_the_other_interpolation_filters(hip, oracle, ctx, fx, fy):
    capi = hip.capi
    bd, bw = 10, 16
    src, ref, blocks, mv, q, gc, n, W, H, border = _setup(hip, oracle, ctx, bd, bw, 7 + fx)
    qp = capi.QuantParams.from_tables(q)
    ps, pr, pp, pf = (ctx.planes_alloc(W, H, border, bd, 1) for _ in range(4))
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    d_b, d_mv = ctx.to_device(blocks), ctx.to_device(mv)
    d_q, d_dq, d_e, d_e2 = ctx.malloc(n * 1024), ctx.malloc(n * 1024), ctx.malloc(2 * n), ctx.malloc(2 * n)
    ctx.build_inter_pred_batch(pr, 0, pp, 0, bw, bw, d_b, d_mv, n, fx, fy)
    ctx.subtract_xform_quant_batch(ps, pp, 0, 2, None, n, gc, 0, qp, None, d_q, d_dq, d_e)
    ctx.inv_txfm_add_batch(d_dq, 2, None, n, gc, 0, d_e, pp, 0)
    ctx.encode_inter_blocks_batch(ps, 0, pr, 0, pf, 0, bw, d_b, d_mv, n, qp, None, None, d_e2, fx, fy, 0)
    vis = (slice(border, border + H), slice(border, border + W))
    assert np.array_equal(ctx.planes_download(pf, 0)[vis], ctx.planes_download(pp, 0)[vis])
    assert np.array_equal(ctx.from_device(d_e2, (n,), np.uint16), ctx.from_device(d_e, (n,), np.uint16))
    for d in (d_b, d_mv, d_q, d_dq, d_e, d_e2):
        ctx.free(d)
    for p in (ps, pr, pp, pf):
        ctx.planes_free(p)


def test_fused_block_kernel_rejects_what_it_does_not_cover(hip, ctx):
    import ctypes as C
    capi = hip.capi
    ps, pr = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 10, 1)
    qp = capi.QuantParams()
    d = ctx.malloc(64)
    call = lambda s, r, o, bw, tt: capi.lib.aomhip_encode_inter_blocks_batch(ctx.h, C.byref(s), 0, C.byref(r), 0, C.byref(o), 0, bw, d, d, 1, 0, 0, tt,
                                                                             C.byref(qp), None, None, d)
    ps2 = ctx.planes_alloc(64, 64, 32, 8, 1)
    assert call(ps, pr, ps2, 16, 0) == capi.ERR_INVALID          # mixed bit depths
    assert call(ps, ps2, ps2, 16, 0) == capi.ERR_INVALID         # reconstruction over the reference frame
    assert call(ps, ps2, ps, 4, 0) == capi.ERR_INVALID           # 4x4 / 64x64 are not covered
    assert call(ps, ps2, ps, 64, 0) == capi.ERR_INVALID
    assert call(ps, ps2, ps, 32, 1) == capi.ERR_INVALID          # TX_32X32 has no ADST
    ctx.free(d)
    for p in (ps, pr, ps2):
        ctx.planes_free(p)
