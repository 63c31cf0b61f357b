"""BASELINE.json's configurations at their FULL sizes, every block checked against the oracle (the single-GPU legs; the
multi-GPU legs only change who owns which tile column -- tests/test_gpu_exchange.py, tests/test_exchange_plan.py):
  configs[2]  av1_fwd_txfm2d 4x4 / 8x8 / 16x16 / 32x32 + aom_quantize_b over a whole 1920x1088 8-bit residual plane;
  configs[3]  full-pel diamond search + bilinear sub-pel refinement of all 32 400 16x16 blocks of a 3840x2160 10-bit pair;
  configs[4]  the encode inner loop on a whole 3840x2160 10-bit 4:2:0 frame: search -> 8-tap prediction -> subtract + forward
              transform + quantise -> inverse transform + reconstruction -> deblocking -> CDEF, luma AND the chroma pair
              (8x8 chroma blocks predicted with the luma MVs, length-6 chroma deblocking, CDEF with the luma directions)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tx_size", [0, 1, 2, 3])
def test_config2_every_square_size_over_a_1080p_plane(hip, oracle, ctx, tx_size):
    rng = np.random.default_rng(20 + tx_size)
    W, H = 1920, 1088
    residual = ((rng.integers(0, 1 << 16, (H, W)) & 511) - 256).astype(np.int16)
    residual[:, : W // 4] //= 16                       # a quiet quarter: small coefficients, short eobs
    w = 4 << tx_size
    gc, n, nc = W // w, (W // w) * (H // w), min(w * w, 1024)
    q = oracle.build_quantizer_y(8, 100)
    d_res = ctx.to_device(residual)
    d_c, d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    ctx.xform_quant_batch(d_res, W, tx_size, None, n, gc, 0, hip.capi.QuantParams.from_tables(q), False, d_c, d_q, d_dq, d_e)
    gcf, gq = ctx.from_device(d_c, (n * nc,), np.int32), ctx.from_device(d_q, (n, nc), np.int32)
    gdq, ge = ctx.from_device(d_dq, (n, nc), np.int32), ctx.from_device(d_e, (n,), np.uint16)
    wc, wq, wdq, we = oracle.xform_quant_batch(residual, tx_size, None, n, gc, 0, q, False, n * nc, True, threads=8)
    assert np.array_equal(gcf, wc) and np.array_equal(gq.ravel(), wq) and np.array_equal(gdq.ravel(), wdq) and np.array_equal(ge, we)
    scan, _ = oracle.get_scan(tx_size, 0)
    nz = gq[:, scan] != 0
    assert np.array_equal(ge, np.where(nz.any(1), nc - np.argmax(nz[:, ::-1], axis=1), 0))   # eob == last non-zero in scan order + 1
    assert 0 < ge.min() if tx_size == 3 else True
    assert ge.max() > nc // 2
    for d in (d_res, d_c, d_q, d_dq, d_e):
        ctx.free(d)


def _grid_blocks(hip, W, H, bs, border):
    gc, gr = W // bs, H // bs
    n = gc * gr
    b = np.zeros(n, hip.capi.search_block_dtype)
    b["bx"], b["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    ext = border - 8
    b["col_min"] = np.maximum(-(b["bx"] + ext), -1023); b["col_max"] = np.minimum(W - b["bx"] - bs + ext, 1023)
    b["row_min"] = np.maximum(-(b["by"] + ext), -1023); b["row_max"] = np.minimum(H - b["by"] - bs + ext, 1023)
    return b, gc


def _subpel_list(blocks, mv):
    sp = blocks.copy()
    sp["start_row"], sp["start_col"] = mv[:, 0] * 8, mv[:, 1] * 8
    for k in ("row_min", "row_max", "col_min", "col_max"):
        sp[k] = np.clip(blocks[k].astype(np.int32) * 8, -16383, 16383)
    return sp


def test_config3_search_of_every_block_of_a_4k_10bit_pair(hip, oracle, ctx):
    W, H, bd, border, bs = 3840, 2160, 10, 160, 16
    rng = np.random.default_rng(33)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 1, bd, shift=(4, -3), frac8=(3, 5))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-5, 6, ref.shape), 0, 1023).astype(np.uint16)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    blocks, _ = _grid_blocks(hip, W, H, bs, border)
    n = len(blocks)
    assert n == 32400
    d_b, d_mv, d_cost = ctx.to_device(blocks), ctx.malloc(n * 4), ctx.malloc(n * 4)
    ctx.fullpel_diamond_batch(ps, pr, 0, bs, bs, 0, 4, hip.capi.MV_COST_L1_HDRES, d_b, n, d_mv, d_cost)
    mv, cost = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_cost, (n,), np.int32)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    wmv, wcost = oracle.fullpel_diamond_batch(sb, rb, border, bs, bs, blocks, 0, 4, 3, bd, threads=8)
    assert np.array_equal(mv, wmv) and np.array_equal(cost, wcost)
    vals, counts = np.unique(mv, axis=0, return_counts=True)                # the search converges on the synthetic (non-zero) shift
    assert counts.max() > 0.8 * n and vals[np.argmax(counts)].any()
    sp = _subpel_list(blocks, mv)
    d_sp = ctx.to_device(sp)
    d_smv, d_err, d_dist, d_sse = (ctx.malloc(n * 4) for _ in range(4))
    ctx.subpel_bilinear_batch(ps, pr, 0, bs, bs, hip.capi.MV_COST_L1_HDRES, 2, 1, 0, d_sp, n, d_smv, d_err, d_dist, d_sse)
    got = (ctx.from_device(d_smv, (n, 2), np.int16), ctx.from_device(d_err, (n,), np.uint32), ctx.from_device(d_dist, (n,), np.int32),
           ctx.from_device(d_sse, (n,), np.uint32))
    want = oracle.subpel_bilinear_batch(sb, rb, border, bs, bs, sp, 3, 2, 1, 0, bd, threads=8)
    for g, w_, name in zip(got, want, ("mv", "err", "distortion", "sse")):
        assert np.array_equal(g, w_), name
    assert (got[0] & 7).any()                                                # eighth-pel positions are reached
    for d in (d_b, d_mv, d_cost, d_sp, d_smv, d_err, d_dist, d_sse):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_config4_inner_loop_of_a_whole_4k_10bit_420_frame(hip, oracle, ctx):
    W, H, bd, border, bs = 3840, 2160, 10, 160, 16
    CW, CH, cb = W // 2, H // 2, 8
    rng = np.random.default_rng(44)
    capi = hip.capi
    src, ref = hip.synth.shifted_smooth_pair(W, H, 2, bd, shift=(2, 3), frac8=(4, 2))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-5, 6, ref.shape), 0, 1023).astype(np.uint16)
    # chroma pair: the luma content at half resolution, different offsets / noise per plane
    csrc = [np.clip(src[::2, ::2].astype(np.int32) // 2 + off, 0, 1023).astype(np.uint16) for off in (200, 330)]
    cref = [np.clip(ref[::2, ::2].astype(np.int32) // 2 + off + rng.integers(-3, 4, (CH, CW)), 0, 1023).astype(np.uint16) for off in (200, 330)]
    q = oracle.build_quantizer_y(bd, 100)
    qp = capi.QuantParams.from_tables(q)
    blocks, gc = _grid_blocks(hip, W, H, bs, border)
    n = len(blocks)
    cblocks = blocks.copy()
    cblocks["bx"] //= 2; cblocks["by"] //= 2

    # ------------------------------------------------------------------ luma, device
    ps, pr, pp, po = (ctx.planes_alloc(W, H, border, bd, 1) for _ in range(4))
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    d_b, d_cb = ctx.to_device(blocks), ctx.to_device(cblocks)
    d_mv, d_cost = ctx.malloc(n * 4), ctx.malloc(n * 4)
    ctx.fullpel_diamond_batch(ps, pr, 0, bs, bs, 0, 4, capi.MV_COST_L1_HDRES, d_b, n, d_mv, d_cost)
    mv = ctx.from_device(d_mv, (n, 2), np.int16)
    sp = _subpel_list(blocks, mv)
    d_sp = ctx.to_device(sp)
    d_smv, d_err, d_dist, d_sse = (ctx.malloc(n * 4) for _ in range(4))
    ctx.subpel_bilinear_batch(ps, pr, 0, bs, bs, capi.MV_COST_L1_HDRES, 2, 1, 0, d_sp, n, d_smv, d_err, d_dist, d_sse)
    smv = ctx.from_device(d_smv, (n, 2), np.int16)
    ctx.build_inter_pred_batch(pr, 0, pp, 0, bs, bs, d_b, d_smv, n, 0, 0)
    pred_g = ctx.planes_download(pp, 0)[border:border + H, border:border + W].copy()
    d_q, d_dq, d_e = ctx.malloc(n * 256 * 4), ctx.malloc(n * 256 * 4), ctx.malloc(2 * n)
    ctx.subtract_xform_quant_batch(ps, pp, 0, 2, None, n, gc, 0, qp, None, d_q, d_dq, d_e)
    eob_g = ctx.from_device(d_e, (n,), np.uint16)
    ctx.inv_txfm_add_batch(d_dq, 2, None, n, gc, 0, d_e, pp, 0)
    recon_g = ctx.planes_download(pp, 0)[border:border + H, border:border + W].copy()
    params = np.zeros((H // 4, W // 4, 4), np.uint8)
    params[:, 2::2, 0] = 8; params[:, 2::2, 1] = 32; params[2::2, :, 2] = 8; params[2::2, :, 3] = 32
    d_params = ctx.to_device(params)
    ctx.deblock_plane(pp, 0, d_params, W // 4, 0, 3)
    lf_g = ctx.planes_download(pp, 0)[border:border + H, border:border + W].copy()
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    pri, sec = np.full((fbh, fbw), 4, np.uint8), np.full((fbh, fbw), 2, np.uint8)
    skip = np.zeros((H // 8, W // 8), np.uint8)
    d_pri, d_sec, d_skip = ctx.to_device(pri), ctx.to_device(sec), ctx.to_device(skip)
    d_dir, d_var = ctx.malloc((H // 8) * (W // 8)), ctx.malloc((H // 8) * (W // 8) * 4)
    ctx.cdef_luma_plane(pp, 0, po, 0, d_pri, d_sec, fbw, d_skip, 6, d_dir, d_var)
    out_g = ctx.planes_download(po, 0)[border:border + H, border:border + W].copy()
    dir_g = ctx.from_device(d_dir, (H // 8, W // 8), np.uint8)

    # ------------------------------------------------------------------ luma, oracle
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    wmv, _ = oracle.fullpel_diamond_batch(sb, rb, border, bs, bs, blocks, 0, 4, 3, bd, threads=8)
    assert np.array_equal(mv, wmv)
    wsmv = oracle.subpel_bilinear_batch(sb, rb, border, bs, bs, sp, 3, 2, 1, 0, bd, threads=8)[0]
    assert np.array_equal(smv, wsmv) and (smv & 7).any()
    pred_o = oracle.build_inter_pred(rb, border, W, H, bs, bs, blocks, wsmv, 0, 0, bd)
    assert np.array_equal(pred_g, pred_o)
    residual = (src.astype(np.int32) - pred_o.astype(np.int32)).astype(np.int16)
    _, _, dq_o, eob_o = oracle.xform_quant_batch(residual, 2, None, n, gc, 0, q, True, n * 256, False, 8)
    assert np.array_equal(eob_g, eob_o) and eob_o.max() > 0
    recon_o = oracle.inv_txfm_add_batch(dq_o, 2, None, n, gc, 0, eob_o, pred_o, bd)
    assert np.array_equal(recon_g, recon_o)
    lf_o = oracle.deblock_plane(recon_o, params, 0, bd, order=0)
    assert np.array_equal(lf_g, lf_o)
    out_o, dir_o, _ = oracle.cdef_plane_luma(lf_o, pri, sec, skip, 6, bd)
    assert np.array_equal(out_g, out_o) and np.array_equal(dir_g, dir_o)
    mse = np.mean((out_g.astype(np.float64) - src) ** 2)
    assert 10 * np.log10(1023.0 ** 2 / max(mse, 1e-9)) > 30

    # ------------------------------------------------------------------ the chroma pair (4:2:0): 8x8 blocks, luma MVs
    cparams = np.zeros((CH // 4, CW // 4, 4), np.uint8)
    cparams[:, 2::2, 0] = 6; cparams[:, 2::2, 1] = 32; cparams[2::2, :, 2] = 6; cparams[2::2, :, 3] = 32
    d_cparams = ctx.to_device(cparams)
    cs, cr, cp, co = (ctx.planes_alloc(CW, CH, border // 2, bd, 1) for _ in range(4))
    d_cq, d_cdq, d_ce = ctx.malloc(n * 64 * 4), ctx.malloc(n * 64 * 4), ctx.malloc(2 * n)
    cgc = CW // cb
    for plane in range(2):
        ctx.planes_upload(cs, 0, csrc[plane]); ctx.planes_upload(cr, 0, cref[plane])
        ctx.build_inter_pred_batch(cr, 0, cp, 0, cb, cb, d_cb, d_smv, n, 0, 0, 1, 1)
        cpred_g = ctx.planes_download(cp, 0)[border // 2:border // 2 + CH, border // 2:border // 2 + CW].copy()
        ctx.subtract_xform_quant_batch(cs, cp, 0, 1, None, n, cgc, 0, qp, None, d_cq, d_cdq, d_ce)
        ceob_g = ctx.from_device(d_ce, (n,), np.uint16)
        ctx.inv_txfm_add_batch(d_cdq, 1, None, n, cgc, 0, d_ce, cp, 0)
        ctx.deblock_plane(cp, 0, d_cparams, CW // 4, 0, 3)
        clf_g = ctx.planes_download(cp, 0)[border // 2:border // 2 + CH, border // 2:border // 2 + CW].copy()
        ctx.cdef_chroma_plane(cp, 0, co, 0, 1, 1, d_dir, d_pri, d_sec, fbw, d_skip, 6)
        cout_g = ctx.planes_download(co, 0)[border // 2:border // 2 + CH, border // 2:border // 2 + CW].copy()

        crb = oracle.extend_plane(cref[plane], border // 2, cr.stride)
        cpred_o = oracle.build_inter_pred(crb, border // 2, CW, CH, cb, cb, cblocks, wsmv, 0, 0, bd, 1, 1)
        assert np.array_equal(cpred_g, cpred_o), plane
        cres = (csrc[plane].astype(np.int32) - cpred_o.astype(np.int32)).astype(np.int16)
        _, _, cdq_o, ceob_o = oracle.xform_quant_batch(cres, 1, None, n, cgc, 0, q, True, n * 64, False, 8)
        assert np.array_equal(ceob_g, ceob_o), plane
        crecon_o = oracle.inv_txfm_add_batch(cdq_o, 1, None, n, cgc, 0, ceob_o, cpred_o, bd)
        clf_o = oracle.deblock_plane(crecon_o, cparams, 0, bd, order=0)
        assert np.array_equal(clf_g, clf_o), plane
        assert not np.array_equal(clf_o, crecon_o)
        assert np.array_equal(cout_g, oracle.cdef_plane_chroma(clf_o, 1, 1, dir_o, pri, sec, skip, 6, bd)), plane
    for d in (d_b, d_cb, d_mv, d_cost, d_sp, d_smv, d_err, d_dist, d_sse, d_q, d_dq, d_e, d_params, d_pri, d_sec, d_skip, d_dir, d_var,
              d_cparams, d_cq, d_cdq, d_ce):
        ctx.free(d)
    for p in (ps, pr, pp, po, cs, cr, cp, co):
        ctx.planes_free(p)
