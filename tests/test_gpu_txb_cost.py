"""aomhip_cost_coeffs_txb_batch / _laplacian_batch (csrc/xform_quant.hip) against (a) warehouse_efficients_txb interpreted on random cost tables
(tests/golden/ref_eval_txb_cost.npz, directly) and (b) the oracle on lists of blocks with mixed transform types."""
import numpy as np
import pytest

from test_golden_nzmap import TXH, TXW
from test_golden_txb_cost import load, oracle_cost, oracle_cost_laplacian, oracle_entropy_ctx

pytestmark = pytest.mark.gpu


def test_device_matches_the_interpreted_function(hip, ctx):
    z, cases = load()
    for c in cases:
        coeff = np.ascontiguousarray(z["c%d" % c["k"]], np.int32)
        d_q, d_t = ctx.to_device(coeff), ctx.to_device(np.ascontiguousarray(z["t%d" % c["k"]], np.int32))
        d_e, d_x, d_o = ctx.to_device(np.array([c["eob"]], np.uint16)), ctx.to_device(np.array([c["txb_skip_ctx"], c["dc_sign_ctx"]], np.uint8)), ctx.malloc(4)
        ctx.cost_coeffs_txb_batch(d_q, c["tx_size"], None, 1, c["tx_type"], d_e, d_x, d_t, d_o)
        assert int(ctx.from_device(d_o, (1,), np.int32)[0]) == c["cost"], c
        ctx.cost_coeffs_txb_batch(d_q, c["tx_size"], None, 1, c["tx_type"], d_e, d_x, d_t, d_o, laplacian=True)
        assert int(ctx.from_device(d_o, (1,), np.int32)[0]) == c["cost_laplacian"], c
        ctx.txb_entropy_context_batch(d_q, c["tx_size"], None, 1, c["tx_type"], d_e, d_o)
        assert int(ctx.from_device(d_o, (1,), np.uint8)[0]) == c["entropy_ctx"], c
        for d in (d_q, d_t, d_e, d_x, d_o):
            ctx.free(d)


@pytest.mark.parametrize("tx_size", list(range(19)))
def test_lists_of_blocks_equal_the_oracle(hip, oracle, ctx, tx_size):
    capi = hip.capi
    rng = np.random.default_rng(1700 + tx_size)
    W, H = TXW[tx_size], TXH[tx_size]
    w, h = min(W, 32), min(H, 32)
    n = w * h
    nb = 53
    one_d = W <= 16 and H <= 16
    types = rng.choice([0, 3, 9, 10, 11, 12, 13, 14, 15] if one_d else [0, 1, 5], nb)
    if W > 32 or H > 32:
        types[:] = 0
    costs = rng.integers(1, 6000, 966).astype(np.int32)
    coeff = np.zeros((nb, n), np.int32)
    eobs = np.zeros(nb, np.int64)
    ctxs = np.stack([rng.integers(0, 13, nb), rng.integers(0, 3, nb)], 1).astype(np.uint8)
    for i in range(nb):
        scan, _ = oracle.get_scan(tx_size, int(types[i]))
        eobs[i] = [n, 1, 0, 2, 3, 4, 5][i] if i < 7 else int(rng.integers(1, n + 1))
        e = int(eobs[i])
        vals = rng.choice([0, 0, 1, 2, 3, 5, 14, 15, 16, 130, 40000], n) * rng.choice([-1, 1], n)
        coeff[i, scan[:e]] = vals[:e]
        if e:
            coeff[i, scan[e - 1]] = int(rng.choice([-1, 2, -7, 15]))
    order = rng.permutation(nb)
    blocks = np.zeros(nb, capi.txb_dtype)
    blocks["tx_type"], blocks["out_offset"] = types, order * n
    packed = np.zeros((nb, n), np.int32)
    packed[order] = coeff
    d_q, d_b, d_t = ctx.to_device(packed), ctx.to_device(blocks), ctx.to_device(costs)
    d_e, d_x, d_o = ctx.to_device(eobs.astype(np.uint16)), ctx.to_device(ctxs), ctx.malloc(4 * nb)
    ctx.cost_coeffs_txb_batch(d_q, tx_size, d_b, nb, 0, d_e, d_x, d_t, d_o)
    got = ctx.from_device(d_o, (nb,), np.int32)
    for i in range(nb):
        want = oracle_cost(coeff[i], int(eobs[i]), tx_size, int(types[i]), int(ctxs[i, 0]), int(ctxs[i, 1]), costs)
        assert int(got[i]) == want, (i, types[i], eobs[i])
    ctx.cost_coeffs_txb_batch(d_q, tx_size, d_b, nb, 0, d_e, d_x, d_t, d_o, laplacian=True)
    got = ctx.from_device(d_o, (nb,), np.int32)
    for i in range(nb):
        assert int(got[i]) == oracle_cost_laplacian(coeff[i], int(eobs[i]), tx_size, int(types[i]), int(ctxs[i, 0]), costs), (i, types[i], eobs[i])
    ctx.txb_entropy_context_batch(d_q, tx_size, d_b, nb, 0, d_e, d_o)
    got8 = ctx.from_device(d_o, (nb,), np.uint8)
    for i in range(nb):
        assert int(got8[i]) == oracle_entropy_ctx(coeff[i], int(eobs[i]), tx_size, int(types[i])), (i, types[i], eobs[i])
    for d in (d_q, d_b, d_t, d_e, d_x, d_o):
        ctx.free(d)


def test_small_entropy_contexts_match_the_interpreted_function(hip, ctx):
    import json
    z, _ = load()
    for c in json.loads(bytes(z["small_ctx"])):
        d_q, d_e, d_o = ctx.to_device(np.array(c["coeff"], np.int32)), ctx.to_device(np.array([c["eob"]], np.uint16)), ctx.malloc(4)
        ctx.txb_entropy_context_batch(d_q, 0, None, 1, 0, d_e, d_o)
        assert int(ctx.from_device(d_o, (1,), np.uint8)[0]) == c["entropy_ctx"], c
        for d in (d_q, d_e, d_o):
            ctx.free(d)


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    d = ctx.malloc(8192)
    for args in ((d, 19, None, 1, 0, d, d, d, d), (d, 2, None, 1, 16, d, d, d, d), (d, 2, None, 1, 0, d, None, d, d), (d, 2, None, 1, 0, d, d, None, d)):
        with pytest.raises(capi.AomHipError):
            ctx.cost_coeffs_txb_batch(*args)
    ctx.free(d)
