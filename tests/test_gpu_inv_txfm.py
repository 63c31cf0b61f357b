"""Parity of the HIP inverse-transform + reconstruction kernel with the oracle (bit-exact): all 19 sizes x
every servable TX_TYPE x bit depth 8/10/12, random and extreme coefficients (test/av1_inv_txfm2d_test.cc
:340-404 uses full-range and > int16 inputs), eob skipping, and the whole encode-side chain
subtract -> forward -> quantise -> inverse -> reconstruct against the oracle's chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nonoverlapping_blocks(hip, rng, W, H, w, h, types, nc):
    xs, ys = np.meshgrid(np.arange(0, W - w + 1, w), np.arange(0, H - h + 1, h))
    xs, ys = xs.ravel(), ys.ravel()
    keep = rng.random(xs.size) < 0.8
    xs, ys = xs[keep], ys[keep]
    n = xs.size
    b = np.zeros(n, hip.capi.txb_dtype)
    b["x"], b["y"] = xs, ys
    b["tx_type"] = rng.choice(types, n)
    b["out_offset"] = rng.permutation(n) * nc
    return b


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("tx_size", range(19))
def test_inverse_all_sizes_types(hip, oracle, ctx, tx_size, bd):
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    types = [t for t in range(16) if oracle.lib.orc_txfm_valid(tx_size, t)]
    nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
    rng = np.random.default_rng(tx_size * 3 + bd)
    W, H, border = 192, 128, 32
    pred = hip.synth.lcg_frame(W, H, 9, 0, bd)
    blocks = _nonoverlapping_blocks(hip, rng, W, H, w, h, types, nc)
    n = len(blocks)
    total = n * nc
    for case in range(3):
        if case == 0:    # plausible dequantised coefficients: sparse, decaying
            dq = (rng.integers(-(1 << (bd + 3)), 1 << (bd + 3), total) * (rng.random(total) < 0.3)).astype(np.int32)
        elif case == 1:  # dense, up to the bd+8 input clamp and beyond it
            dq = rng.integers(-(1 << (bd + 9)), 1 << (bd + 9), total).astype(np.int32)
        else:            # DC only / extremes
            dq = np.zeros(total, np.int32)
            dq[::nc] = rng.choice([-(1 << (bd + 7)), (1 << (bd + 7)) - 1, 1, -1, 4095], n)
        eob = rng.integers(0, 3, n).astype(np.uint16)  # some blocks flagged eob == 0
        p = ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(p, 0, pred)
        d_dq, d_b, d_e = ctx.to_device(dq), ctx.to_device(blocks), ctx.to_device(eob)
        ctx.inv_txfm_add_batch(d_dq, tx_size, d_b, n, 0, 0, d_e, p, 0)
        got = ctx.planes_download(p, 0)[border:border + H, border:border + W]
        want = oracle.inv_txfm_add_batch(dq, tx_size, blocks, n, 0, 0, eob, pred, bd)
        assert np.array_equal(got, want), (tx_size, bd, case)
        ctx.planes_free(p)
        for d in (d_dq, d_b, d_e):
            ctx.free(d)


@pytest.mark.parametrize("bd", [8, 10])
def test_encode_chain_reconstruction(hip, oracle, ctx, bd):
    """encodemb.c encode_block order: subtract -> av1_xform_quant -> av1_inverse_transform_block, all on the
    device, vs the oracle chain; then the reconstruction error must be small at a fine quantiser."""
    rng = np.random.default_rng(100 + bd)
    W, H, border = 256, 128, 32
    src, _ = hip.synth.shifted_smooth_pair(W, H, 3, bd)
    pred = np.clip(src.astype(np.int32) + rng.integers(-12, 13, (H, W)), 0, (1 << bd) - 1).astype(src.dtype)
    ps, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pp, 0, pred)
    residual = (src.astype(np.int32) - pred.astype(np.int32)).astype(np.int16)
    recon_want = pred.copy()
    for tx_size, region in [(2, (0, 128)), (1, (128, 192)), (3, (192, 256))]:  # 16x16 | 8x8 | 32x32 column bands
        w = oracle.TX_W[tx_size]
        nc = w * w
        xs, ys = np.meshgrid(np.arange(region[0], region[1], w), np.arange(0, H, w))
        n = xs.size
        blocks = np.zeros(n, hip.capi.txb_dtype)
        blocks["x"], blocks["y"] = xs.ravel(), ys.ravel()
        blocks["tx_type"] = rng.choice([0, 1, 2, 3, 9] if w < 32 else [0, 9], n)
        blocks["out_offset"] = np.arange(n) * nc
        q = oracle.build_quantizer_y(bd, 12)
        d_b = ctx.to_device(blocks)
        d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
        ctx.subtract_xform_quant_batch(ps, pp, 0, tx_size, d_b, n, 0, 0, hip.capi.QuantParams.from_tables(q), None, d_q,
                                       d_dq, d_e)
        ctx.inv_txfm_add_batch(d_dq, tx_size, d_b, n, 0, 0, d_e, pp, 0)
        _, wq, wdq, we = oracle.xform_quant_batch(residual, tx_size, blocks, n, 0, 0, q, bd > 8, n * nc, False, 4)
        assert np.array_equal(ctx.from_device(d_dq, (n * nc,), np.int32), wdq)
        recon_want = oracle.inv_txfm_add_batch(wdq, tx_size, blocks, n, 0, 0, we, recon_want, bd)
        for d in (d_b, d_q, d_dq, d_e):
            ctx.free(d)
    got = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
    assert np.array_equal(got, recon_want)
    assert np.abs(got.astype(np.int64) - src.astype(np.int64)).max() <= (3 if bd == 8 else 12)
    ctx.planes_free(ps); ctx.planes_free(pp)
