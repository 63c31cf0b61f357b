"""The coefficient coder's rate: the restatement against warehouse_efficients_txb (av1/encoder/txb_rdopt.c:450-544) interpreted on random cost tables
(tests/golden/gen_ref_eval_txb_cost.py), bit-exact."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc
from test_golden_nzmap import TXH, TXW

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_txb_cost.npz")


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def oracle_cost(coeff, eob, tx_size, tx_type, skip_ctx, dc_ctx, costs):
    scan, _ = orc.get_scan(tx_size, tx_type)
    sc = np.ascontiguousarray(scan, np.int16)
    tx_class = 0 if tx_type < 10 else (2 if tx_type % 2 == 0 else 1)
    co, cs = np.ascontiguousarray(coeff, np.int32), np.ascontiguousarray(costs, np.int32)
    return int(orc.lib.orc_cost_coeffs_txb(C.c_void_p(co.ctypes.data), eob, TXW[tx_size], TXH[tx_size], tx_class, C.c_void_p(sc.ctypes.data), skip_ctx, dc_ctx,
                                           C.c_void_p(cs.ctypes.data)))


def oracle_cost_laplacian(coeff, eob, tx_size, tx_type, skip_ctx, costs):
    scan, _ = orc.get_scan(tx_size, tx_type)
    sc = np.ascontiguousarray(scan, np.int16)
    tx_class = 0 if tx_type < 10 else (2 if tx_type % 2 == 0 else 1)
    co, cs = np.ascontiguousarray(coeff, np.int32), np.ascontiguousarray(costs, np.int32)
    return int(orc.lib.orc_cost_coeffs_txb_laplacian(C.c_void_p(co.ctypes.data), eob, tx_class, C.c_void_p(sc.ctypes.data), skip_ctx, C.c_void_p(cs.ctypes.data)))


def oracle_entropy_ctx(coeff, eob, tx_size, tx_type):
    scan, _ = orc.get_scan(tx_size, tx_type)
    sc, co = np.ascontiguousarray(scan, np.int16), np.ascontiguousarray(coeff, np.int32)
    return int(orc.lib.orc_get_txb_entropy_context(C.c_void_p(co.ctypes.data), C.c_void_p(sc.ctypes.data), eob))


def test_entropy_context_matches_the_reference():
    z, cases = load()
    seen = set()
    for c in cases:
        got = oracle_entropy_ctx(z["c%d" % c["k"]], c["eob"], c["tx_size"], c["tx_type"])
        assert got == c["entropy_ctx"], c
        seen.add(got)
    for c in json.loads(bytes(z["small_ctx"])):          # below the saturation: small sums, the three DC signs
        got = oracle_entropy_ctx(np.array(c["coeff"], np.int32), c["eob"], 0, 0)
        assert got == c["entropy_ctx"], c
        seen.add(got)
    assert len(seen) >= 9 and oracle_entropy_ctx(np.zeros(16, np.int32), 0, 0, 0) == 0


def test_laplacian_rate_matches_the_reference():
    z, cases = load()
    for c in cases:
        got = oracle_cost_laplacian(z["c%d" % c["k"]], c["eob"], c["tx_size"], c["tx_type"], c["txb_skip_ctx"], z["t%d" % c["k"]])
        assert got == c["cost_laplacian"], c
    assert len({c["cost_laplacian"] for c in cases}) > 90


def test_rate_matches_the_reference():
    z, cases = load()
    assert len(cases) >= 99
    golomb = 0
    for c in cases:
        coeff, costs = z["c%d" % c["k"]], z["t%d" % c["k"]]
        assert oracle_cost(coeff, c["eob"], c["tx_size"], c["tx_type"], c["txb_skip_ctx"], c["dc_sign_ctx"], costs) == c["cost"], c
        golomb += int(np.abs(coeff).max() >= 15)
    assert golomb >= 40
    # eob == 0 (av1_cost_coeffs_txb, :611-613): the skip cost alone
    costs = z["t0"]
    assert oracle_cost(np.zeros(16, np.int32), 0, 0, 0, 5, 1, costs) == int(costs[5 * 2 + 1])
