"""End-to-end parity of the chained encode inner loop (BASELINE configs[4] at a reduced frame size): every stage
runs on the device and feeds the next through HBM; the oracle runs the same chain on the host; the final
CDEF output plane and every intermediate must be bit-identical.
  search (mcomp.c) -> full-pel prediction -> subtract + fwd txfm + quantize_b (encodemb.c:323) ->
  inverse txfm + add (idct.c:304) -> deblock (av1_loopfilter.c) -> CDEF (cdef_block.c)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd,W,H", [(8, 256, 192), (10, 384, 256), (10, 704, 448)])
def test_chain_matches_oracle(hip, oracle, ctx, bd, W, H):
    rng = np.random.default_rng(bd + W)
    border, bw = 160, 16
    src, ref = hip.synth.shifted_smooth_pair(W, H, 3, bd, shift=(3, -2))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-6, 7, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr, pp, po = (ctx.planes_alloc(W, H, border, bd, 1) for _ in range(4))
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    gc, gr = W // bw, H // bw
    n = gc * gr
    blocks = np.zeros(n, hip.capi.search_block_dtype)
    blocks["bx"] = (np.arange(n) % gc) * bw; blocks["by"] = (np.arange(n) // gc) * bw
    for i in range(n):
        lim = oracle.mv_limits_for_block(int(blocks["bx"][i]), int(blocks["by"][i]), bw, bw, W, H, border)
        blocks["row_min"][i], blocks["row_max"][i], blocks["col_min"][i], blocks["col_max"][i] = lim
    # --- device chain
    d_b, d_mv, d_c = ctx.to_device(blocks), ctx.malloc(n * 4), ctx.malloc(n * 4)
    ctx.fullpel_diamond_batch(ps, pr, 0, bw, bw, 0, 4, 3, d_b, n, d_mv, d_c)
    ctx.build_pred_fullpel(pr, 0, pp, 0, bw, bw, d_b, d_mv, n)
    nc = 256
    q = oracle.build_quantizer_y(bd, 100)
    d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    ctx.subtract_xform_quant_batch(ps, pp, 0, 2, None, n, gc, 0, hip.capi.QuantParams.from_tables(q), None, d_q, d_dq, d_e)
    ctx.inv_txfm_add_batch(d_dq, 2, None, n, gc, 0, d_e, pp, 0)
    params = np.zeros((H // 4, W // 4, 4), np.uint8)
    params[:, 2::2, 0] = 8; params[:, 2::2, 1] = 32; params[2::2, :, 2] = 8; params[2::2, :, 3] = 32
    d_params = ctx.to_device(params)
    recon_g = ctx.planes_download(pp, 0)[border:border + H, border:border + W].copy()
    ctx.deblock_plane(pp, 0, d_params, W // 4, 0, 3)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    pri, sec = np.full((fbh, fbw), 4, np.uint8), np.full((fbh, fbw), 2, np.uint8)
    skip = np.zeros((H // 8, W // 8), np.uint8)
    d_pri, d_sec, d_skip = ctx.to_device(pri), ctx.to_device(sec), ctx.to_device(skip)
    ctx.cdef_luma_plane(pp, 0, po, 0, d_pri, d_sec, fbw, d_skip, 6)
    mv_g = ctx.from_device(d_mv, (n, 2), np.int16)
    eob_g = ctx.from_device(d_e, (n,), np.uint16)
    lf_g = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
    out_g = ctx.planes_download(po, 0)[border:border + H, border:border + W]
    # --- oracle chain
    mv_o, _ = oracle.fullpel_diamond_batch(sb, rb, border, bw, bw, blocks, 0, 4, 3, bd)
    assert np.array_equal(mv_g, mv_o)
    pred_o = np.zeros_like(src)
    for i in range(n):
        x, y = int(blocks["bx"][i]), int(blocks["by"][i])
        pred_o[y:y + bw, x:x + bw] = rb[border + y + mv_o[i, 0]:border + y + mv_o[i, 0] + bw,
                                        border + x + mv_o[i, 1]:border + x + mv_o[i, 1] + bw]
    residual = (src.astype(np.int32) - pred_o.astype(np.int32)).astype(np.int16)
    _, _, dq_o, eob_o = oracle.xform_quant_batch(residual, 2, None, n, gc, 0, q, bd > 8, n * nc, False, 4)
    assert np.array_equal(eob_g, eob_o) and eob_o.max() > 0
    recon_o = oracle.inv_txfm_add_batch(dq_o, 2, None, n, gc, 0, eob_o, pred_o, bd)
    assert np.array_equal(recon_g, recon_o)
    lf_o = oracle.deblock_plane(recon_o, params, 0, bd, order=0)
    assert np.array_equal(lf_g, lf_o)
    out_o, _, _ = oracle.cdef_plane_luma(lf_o, pri, sec, skip, 6, bd)
    assert np.array_equal(out_g, out_o)
    # the chain reconstructs the source closely (fine quantiser, search converged on the synthetic shift)
    mse = np.mean((out_g.astype(np.float64) - src) ** 2)
    assert 10 * np.log10(((1 << bd) - 1) ** 2 / max(mse, 1e-9)) > 30
    for d in (d_b, d_mv, d_c, d_q, d_dq, d_e, d_params, d_pri, d_sec, d_skip):
        ctx.free(d)
    for p in (ps, pr, pp, po):
        ctx.planes_free(p)
