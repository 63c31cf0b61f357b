"""aomhip_quantize_b_adaptive_qm_batch (csrc/xform_quant.hip) against (a) the interpreted reference's aom_[highbd_]quantize_b_adaptive_helper_c with
qm_ptr / iqm_ptr (tests/golden/ref_eval_qm_adaptive.npz: 180 cases, directly) and (b) the oracle on lists of blocks."""
import numpy as np
import pytest

from test_golden_qm_adaptive import load, orc_adaptive_qm

pytestmark = pytest.mark.gpu


def test_device_adaptive_matrix_quantiser_reproduces_the_interpreted_reference(hip, ctx):
    z, cases = load()
    capi = hip.capi
    for c in cases:
        k, n = c["k"], c["n"]
        qp = capi.QuantParams.from_tables({m: np.array(v, np.int16) for m, v in c["tables"].items()})
        d_c = ctx.to_device(np.ascontiguousarray(z["c%d" % k], np.int32))
        d_qm, d_iqm = ctx.to_device(z["qm_" + c["matrix"]]), ctx.to_device(z["iqm_" + c["matrix"]])
        d_q, d_dq, d_e = ctx.malloc(4 * n), ctx.malloc(4 * n), ctx.malloc(2)
        ctx.quantize_b_adaptive_qm_batch(d_c, c["tx_size"], None, 1, 0, qp, c["hbd"], d_qm, d_iqm, d_q, d_dq, d_e)
        assert np.array_equal(ctx.from_device(d_q, (n,), np.int32), z["q%d" % k]), c
        assert np.array_equal(ctx.from_device(d_dq, (n,), np.int32), z["d%d" % k]), c
        assert int(ctx.from_device(d_e, (1,), np.uint16)[0]) == c["eob"], c
        for d in (d_c, d_qm, d_iqm, d_q, d_dq, d_e):
            ctx.free(d)


@pytest.mark.parametrize("tx_size,bd", [(0, 8), (1, 10), (2, 8), (2, 12), (3, 10), (7, 8), (4, 10), (9, 8)])
def test_lists_of_blocks_equal_the_oracle(hip, oracle, ctx, tx_size, bd):
    capi = hip.capi
    rng = np.random.default_rng(tx_size * 13 + bd)
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    nc = min(w, 32) * min(h, 32)
    ls = (w * h > 256) + (w * h > 1024)
    qm, iqm = rng.integers(16, 200, nc).astype(np.uint8), rng.integers(8, 64, nc).astype(np.uint8)
    hbd = bd > 8
    q = oracle.build_quantizer_y(bd, 110)
    tabs = {m: [int(q[m][0]), int(q[m][1])] for m in q}
    qp = capi.QuantParams.from_tables(q)
    nb = 97
    span = (1 << (bd + 7)) - 1
    coeff = rng.integers(-span, span + 1, (nb, nc)).astype(np.int32)
    coeff[rng.random((nb, nc)) < 0.8] //= 256
    coeff[3] = 0
    coeff[4] = 0; coeff[4, 1] = int(q["zbin"][1]) * 32 // int(qm[1]) + 2          # one small coefficient: the single-coefficient rule
    d_c, d_qm, d_iqm = ctx.to_device(coeff), ctx.to_device(qm), ctx.to_device(iqm)
    d_q, d_dq, d_e = ctx.malloc(4 * nb * nc), ctx.malloc(4 * nb * nc), ctx.malloc(2 * nb)
    for tx_type in ((0, 10, 11) if nc <= 256 else (0,)):
        ctx.quantize_b_adaptive_qm_batch(d_c, tx_size, None, nb, tx_type, qp, hbd, d_qm, d_iqm, d_q, d_dq, d_e)
        qg, dg, e = ctx.from_device(d_q, (nb, nc), np.int32), ctx.from_device(d_dq, (nb, nc), np.int32), ctx.from_device(d_e, (nb,), np.uint16)
        scan, _ = oracle.get_scan(tx_size, tx_type)
        for i in range(nb):
            wq, wd, we = orc_adaptive_qm(oracle, coeff[i], tabs, scan, ls, hbd, qm, iqm)
            assert np.array_equal(qg[i], wq) and np.array_equal(dg[i], wd) and int(e[i]) == we, (tx_type, i)
        assert e[3] == 0
    # flat matrices: the plain adaptive quantiser
    flat = ctx.to_device(np.full(nc, 32, np.uint8))
    d_q2, d_dq2, d_e2 = ctx.malloc(4 * nb * nc), ctx.malloc(4 * nb * nc), ctx.malloc(2 * nb)
    ctx.quantize_b_adaptive_qm_batch(d_c, tx_size, None, nb, 0, qp, hbd, flat, None, d_q, d_dq, d_e)
    ctx.quantize_b_adaptive_batch(d_c, tx_size, None, nb, 0, qp, hbd, d_q2, d_dq2, d_e2)
    assert np.array_equal(ctx.from_device(d_q, (nb, nc), np.int32), ctx.from_device(d_q2, (nb, nc), np.int32))
    assert np.array_equal(ctx.from_device(d_e, (nb,), np.uint16), ctx.from_device(d_e2, (nb,), np.uint16))
    for d in (d_c, d_qm, d_iqm, d_q, d_dq, d_e, flat, d_q2, d_dq2, d_e2):
        ctx.free(d)
