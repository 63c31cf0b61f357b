"""aomhip_quantize_lp_batch (csrc/xform_quant.hip) against (a) the interpreted reference's av1_quantize_lp_c / av1_block_error_lp_c
(tests/golden/ref_eval_quant_lp.npz, directly) and (b) the oracle on lists of blocks, list and grid mode."""
import json

import numpy as np
import pytest

from test_golden_quant_lp import GOLD, oracle_quantize_lp

pytestmark = pytest.mark.gpu


def _qp(capi, tabs):
    t = dict(tabs, zbin=[0, 0], quant_shift=[0, 0])
    return capi.QuantParams.from_tables({m: np.array(v, np.int16) for m, v in t.items()})


def test_device_lp_quantiser_reproduces_the_interpreted_reference(hip, ctx):
    z = np.load(GOLD)
    cases = json.loads(bytes(z["cases"]))
    for c in cases:
        k, n = c["k"], c["n"]
        d_c = ctx.to_device(np.ascontiguousarray(z["c%d" % k], np.int16))
        d_q, d_dq, d_e, d_err = ctx.malloc(2 * n), ctx.malloc(2 * n), ctx.malloc(2), ctx.malloc(8)
        ctx.quantize_lp_batch(d_c, c["tx_size"], None, 1, c["tx_type"], _qp(hip.capi, c["tables"]), d_q, d_dq, d_e, d_err)
        assert np.array_equal(ctx.from_device(d_q, (n,), np.int16), z["q%d" % k]), c
        assert np.array_equal(ctx.from_device(d_dq, (n,), np.int16), z["d%d" % k]), c
        assert int(ctx.from_device(d_e, (1,), np.uint16)[0]) == c["eob"], c
        assert int(ctx.from_device(d_err, (1,), np.int64)[0]) == c["block_error"], c
        for d in (d_c, d_q, d_dq, d_e, d_err):
            ctx.free(d)


@pytest.mark.parametrize("tx_size", [0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16])
def test_lists_of_blocks_equal_the_oracle(hip, oracle, ctx, tx_size):
    capi = hip.capi
    rng = np.random.default_rng(500 + tx_size)
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    nc = w * h
    dq = np.array([rng.integers(4, 300), rng.integers(4, 500)], np.int64)
    tabs = {"round": ((64 * dq) >> 7).tolist(), "quant": np.minimum((1 << 16) // dq, 32767).tolist(), "dequant": dq.tolist()}
    nb = 101
    coeff = rng.integers(-32768, 32768, (nb, nc)).astype(np.int16)
    coeff[rng.random((nb, nc)) < 0.7] //= 256
    coeff[3] = 0
    # list mode: the blocks' coefficients in a shuffled order, every block with its own transform type
    order = rng.permutation(nb)
    types = rng.choice([0, 10, 11] if nc <= 256 else [0], nb)
    blocks = np.zeros(nb, capi.txb_dtype)
    blocks["tx_type"], blocks["out_offset"] = types, order * nc
    packed = np.zeros((nb, nc), np.int16)
    packed[order] = coeff
    d_c, d_b = ctx.to_device(packed), ctx.to_device(blocks)
    d_q, d_dq, d_e, d_err = ctx.malloc(2 * nb * nc), ctx.malloc(2 * nb * nc), ctx.malloc(2 * nb), ctx.malloc(8 * nb)
    ctx.quantize_lp_batch(d_c, tx_size, d_b, nb, 0, _qp(capi, tabs), d_q, d_dq, d_e, d_err)
    q, dqc = ctx.from_device(d_q, (nb, nc), np.int16)[order], ctx.from_device(d_dq, (nb, nc), np.int16)[order]
    e, err = ctx.from_device(d_e, (nb,), np.uint16), ctx.from_device(d_err, (nb,), np.int64)
    for i in range(nb):
        scan, _ = oracle.get_scan(tx_size, int(types[i]))
        wq, wd, we, werr = oracle_quantize_lp(coeff[i], tabs, scan)
        assert np.array_equal(q[i], wq) and np.array_equal(dqc[i], wd) and int(e[i]) == we and int(err[i]) == werr, (i, types[i])
    assert e[3] == 0 and err[3] == 0 and e.max() > nc // 4
    # grid mode without the distortion output
    ctx.memset(d_q, 0x11, 2 * nb * nc)
    ctx.quantize_lp_batch(d_c, tx_size, None, nb, 0, _qp(capi, tabs), d_q, d_dq, d_e)
    scan, _ = oracle.get_scan(tx_size, 0)
    g = ctx.from_device(d_q, (nb, nc), np.int16)
    for i in (0, 17, nb - 1):
        assert np.array_equal(g[i], oracle_quantize_lp(packed[i], tabs, scan)[0])
    for d in (d_c, d_b, d_q, d_dq, d_e, d_err):
        ctx.free(d)


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    qp = _qp(capi, {"round": [1, 1], "quant": [100, 100], "dequant": [8, 8]})
    d = ctx.malloc(64)
    with pytest.raises(capi.AomHipError):
        ctx.quantize_lp_batch(d, 19, None, 1, 0, qp, d, d, d)
    with pytest.raises(capi.AomHipError):
        ctx.quantize_lp_batch(d, 0, None, 1, 16, qp, d, d, d)
    ctx.free(d)
