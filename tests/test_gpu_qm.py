"""Quantisation matrices on the device (aomhip_quantize_b_qm_batch and the two composites) against (a) the interpreted reference's
aom_[highbd_]quantize_b_helper_c with qm_ptr / iqm_ptr (tests/golden/ref_eval_qm.npz: 240 cases, directly) and (b) the oracle on whole planes."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_matrix_quantiser_reproduces_the_interpreted_reference(hip, ctx):
    z = np.load(os.path.join(HERE, "golden", "ref_eval_qm.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    capi = hip.capi
    for k, c in enumerate(cases):
        n = c["n"]
        qp = capi.QuantParams.from_tables({m: np.array(v, np.int16) for m, v in c["tables"].items()})
        d_c = ctx.to_device(np.ascontiguousarray(z["c%d" % k], np.int32))
        d_qm, d_iqm = ctx.to_device(z["qm_" + c["matrix"]]), ctx.to_device(z["iqm_" + c["matrix"]])
        d_q, d_dq, d_e = ctx.malloc(4 * n), ctx.malloc(4 * n), ctx.malloc(2)
        ctx.quantize_b_qm_batch(d_c, c["tx_size"], None, 1, 0, qp, c["hbd"], d_qm, d_iqm, d_q, d_dq, d_e)
        assert np.array_equal(ctx.from_device(d_q, (n,), np.int32), z["q%d" % k]), c
        assert np.array_equal(ctx.from_device(d_dq, (n,), np.int32), z["d%d" % k]), c
        assert int(ctx.from_device(d_e, (1,), np.uint16)[0]) == c["eob"], c
        for d in (d_c, d_qm, d_iqm, d_q, d_dq, d_e):
            ctx.free(d)


@pytest.mark.parametrize("tx_size,bd", [(0, 8), (1, 10), (2, 8), (2, 10), (3, 10), (7, 8), (4, 10)])
def test_transform_and_matrix_quantiser_over_a_plane_equal_the_oracle(hip, oracle, ctx, tx_size, bd):
    z = np.load(os.path.join(HERE, "golden", "ref_eval_qm.npz"))
    rng = np.random.default_rng(tx_size * 10 + bd)
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    nc = min(w, 32) * min(h, 32)
    # a real matrix pair where the fixture has this size (level 8 luma), a synthetic one otherwise
    key = "%d_8_0" % tx_size
    qm = z["qm_" + key] if "qm_" + key in z else rng.integers(16, 200, nc).astype(np.uint8)
    iqm = z["iqm_" + key] if "iqm_" + key in z else rng.integers(8, 64, nc).astype(np.uint8)
    W, H = 8 * w, 6 * h
    span = 255 << (bd - 8)
    residual = rng.integers(-span, span + 1, (H, W)).astype(np.int16)
    residual[:, : W // 2] //= 24
    gc, n = W // w, (W // w) * (H // h)
    q = oracle.build_quantizer_y(bd, 110)
    qp = hip.capi.QuantParams.from_tables(q)
    d_res, d_qm, d_iqm = ctx.to_device(residual), ctx.to_device(qm), ctx.to_device(iqm)
    d_c, d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    ctx.xform_quant_qm_batch(d_res, W, tx_size, None, n, gc, 0, qp, bd > 8, d_qm, d_iqm, d_c, d_q, d_dq, d_e)
    gq, gdq, ge = ctx.from_device(d_q, (n * nc,), np.int32), ctx.from_device(d_dq, (n * nc,), np.int32), ctx.from_device(d_e, (n,), np.uint16)
    _, wq, wdq, we = oracle.xform_quant_batch(residual, tx_size, None, n, gc, 0, q, bd > 8, n * nc, False, 4, qm=qm, iqm=iqm)
    assert np.array_equal(gq, wq) and np.array_equal(gdq, wdq) and np.array_equal(ge, we)
    _, fq, _, _ = oracle.xform_quant_batch(residual, tx_size, None, n, gc, 0, q, bd > 8, n * nc, False, 4)
    assert not np.array_equal(fq, wq)                                    # the matrices change the levels
    # flat matrices (NULL pointers) == the plain call
    ctx.xform_quant_qm_batch(d_res, W, tx_size, None, n, gc, 0, qp, bd > 8, None, None, d_c, d_q, d_dq, d_e)
    assert np.array_equal(ctx.from_device(d_q, (n * nc,), np.int32), fq)
    for d in (d_res, d_qm, d_iqm, d_c, d_q, d_dq, d_e):
        ctx.free(d)


def test_subtract_form_and_argument_checks(hip, oracle, ctx):
    capi = hip.capi
    rng = np.random.default_rng(3)
    W, H, bd, border = 128, 64, 10, 32
    src = rng.integers(0, 1024, (H, W)).astype(np.uint16)
    pred = np.clip(src.astype(np.int32) + rng.integers(-40, 41, (H, W)), 0, 1023).astype(np.uint16)
    ps, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pp, 0, pred)
    qm, iqm = rng.integers(20, 120, 256).astype(np.uint8), rng.integers(10, 60, 256).astype(np.uint8)
    q = oracle.build_quantizer_y(bd, 80)
    qp = capi.QuantParams.from_tables(q)
    n, gc = (W // 16) * (H // 16), W // 16
    d_qm, d_iqm = ctx.to_device(qm), ctx.to_device(iqm)
    d_c, d_q, d_dq, d_e = ctx.malloc(n * 1024), ctx.malloc(n * 1024), ctx.malloc(n * 1024), ctx.malloc(2 * n)
    ctx.subtract_xform_quant_qm_batch(ps, pp, 0, 2, None, n, gc, 0, qp, d_qm, d_iqm, d_c, d_q, d_dq, d_e)
    residual = (src.astype(np.int32) - pred.astype(np.int32)).astype(np.int16)
    _, wq, wdq, we = oracle.xform_quant_batch(residual, 2, None, n, gc, 0, q, True, n * 256, False, 2, qm=qm, iqm=iqm)
    assert np.array_equal(ctx.from_device(d_q, (n * 256,), np.int32), wq) and np.array_equal(ctx.from_device(d_dq, (n * 256,), np.int32), wdq)
    assert np.array_equal(ctx.from_device(d_e, (n,), np.uint16), we)
    import ctypes as C
    assert capi.lib.aomhip_subtract_xform_quant_qm_batch(ctx.h, C.byref(ps), C.byref(pp), 0, 2, None, n, gc, 0, C.byref(qp), d_qm, d_iqm, None, d_q, d_dq,
                                                         d_e) == capi.ERR_INVALID      # d_coeff is required
    assert capi.lib.aomhip_quantize_b_qm_batch(ctx.h, d_c, 19, None, n, 0, C.byref(qp), 1, d_qm, d_iqm, d_q, d_dq, d_e) == capi.ERR_INVALID
    for d in (d_qm, d_iqm, d_c, d_q, d_dq, d_e):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pp)
