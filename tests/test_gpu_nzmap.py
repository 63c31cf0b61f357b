"""aomhip_get_nz_map_contexts_batch (csrc/xform_quant.hip) behind aomhip_txb_init_levels_batch against (a) av1_get_nz_map_contexts_c interpreted
(tests/golden/ref_eval_nzmap.npz, directly) and (b) the oracle on lists of blocks with mixed transform types."""
import numpy as np
import pytest

from test_golden_nzmap import TXH, TXW, load, oracle_contexts

pytestmark = pytest.mark.gpu


def _run(ctx, coeff, tx_size, types, eobs, blocks=None):
    nb = coeff.shape[0]
    w, h = min(TXW[tx_size], 32), min(TXH[tx_size], 32)
    lp, cp = (h + 4) * (w + 4) + 16 + 5, w * h + 3          # pitches with slack
    d_c, d_l, d_x, d_e = ctx.to_device(coeff), ctx.malloc(lp * nb), ctx.malloc(cp * nb), ctx.to_device(np.asarray(eobs, np.uint16))
    ctx.memset(d_x, 0xF9, cp * nb)                          # -7
    ctx.txb_init_levels_batch(d_c, w, h, None, nb, d_l, lp)
    d_b = ctx.to_device(blocks) if blocks is not None else None
    ctx.get_nz_map_contexts_batch(d_l, lp, tx_size, d_b, nb, int(types[0]) if blocks is None else 0, d_e, d_x, cp)
    out = ctx.from_device(d_x, (nb, cp), np.int8)
    for d in (d_c, d_l, d_x, d_e, d_b):
        if d:
            ctx.free(d)
    assert (out[:, w * h:] == -7).all()
    return out[:, :w * h]


def test_device_matches_the_interpreted_function(hip, ctx):
    z, cases = load()
    for c in cases:
        coeff = np.ascontiguousarray(z["c%d" % c["k"]], np.int32)[None, :]
        got = _run(ctx, coeff, c["tx_size"], [c["tx_type"]], [c["eob"]])
        assert np.array_equal(got[0], z["x%d" % c["k"]]), c


@pytest.mark.parametrize("tx_size", list(range(19)))
def test_lists_of_blocks_equal_the_oracle(hip, oracle, ctx, tx_size):
    capi = hip.capi
    rng = np.random.default_rng(900 + tx_size)
    W, H = TXW[tx_size], TXH[tx_size]
    w, h = min(W, 32), min(H, 32)
    n = w * h
    nb = 61
    one_d = W <= 16 and H <= 16
    types = rng.choice([0, 3, 9, 10, 11, 12, 13, 14, 15] if one_d else [0, 1, 5], nb)
    if W > 32 or H > 32:
        types[:] = 0
    coeff = np.zeros((nb, n), np.int32)
    eobs = np.zeros(nb, np.int64)
    for i in range(nb):
        scan, _ = oracle.get_scan(tx_size, int(types[i]))
        eobs[i] = [n, 1, 0, 2][i] if i < 4 else int(rng.integers(1, n + 1))
        e = int(eobs[i])
        vals = rng.choice([0, 0, 1, 2, 3, 5, 40, 3000], n) * rng.choice([-1, 1], n)
        coeff[i, scan[:e]] = vals[:e]
        if e:
            coeff[i, scan[e - 1]] = -2
    blocks = np.zeros(nb, capi.txb_dtype)
    blocks["tx_type"] = types
    got = _run(ctx, coeff, tx_size, types, eobs, blocks)
    for i in range(nb):
        want = oracle_contexts(coeff[i], tx_size, int(types[i]), int(eobs[i]))
        assert np.array_equal(got[i], want), (i, types[i], eobs[i])
    assert (got[2] == -7).all()          # eob 0: nothing written


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    d = ctx.malloc(4096)
    for args in ((d, 100, 2, None, 1, 0, d, d, 256), (d, 416, 2, None, 1, 0, d, d, 255), (d, 416, 19, None, 1, 0, d, d, 256), (d, 416, 2, None, 1, 16, d, d, 256)):
        with pytest.raises(capi.AomHipError):
            ctx.get_nz_map_contexts_batch(*args)
    ctx.free(d)
