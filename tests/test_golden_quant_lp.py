"""The low-precision quantiser's restatement against the interpreted reference (tests/golden/gen_ref_eval_quant_lp.py): av1_quantize_lp_c
(av1/encoder/av1_quantize.c:212-240) and av1_block_error_lp_c (av1/encoder/rdopt.c:650-660), bit-exact."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_quant_lp.npz")
i16p = C.POINTER(C.c_int16)


def _p(a):
    return a.ctypes.data_as(i16p)


def oracle_quantize_lp(c, tables, scan):
    n = c.size
    q, d, eob = np.full(n, 0x55, np.int16), np.full(n, 0x55, np.int16), C.c_uint16(77)
    t = {m: np.asarray(v, np.int16) for m, v in tables.items()}
    sc = np.ascontiguousarray(scan, np.int16)
    orc.lib.orc_quantize_lp(_p(c), C.c_ssize_t(n), _p(t["round"]), _p(t["quant"]), _p(q), _p(d), _p(t["dequant"]), C.byref(eob), _p(sc))
    orc.lib.orc_block_error_lp.restype = C.c_int64
    err = orc.lib.orc_block_error_lp(_p(c), _p(d), C.c_ssize_t(n))
    return q, d, eob.value, err


def test_quantize_lp_matches_reference():
    z = np.load(GOLD)
    cases = json.loads(bytes(z["cases"]))
    assert len(cases) >= 50
    clamp = 0
    for cs in cases:
        k = cs["k"]
        c = np.ascontiguousarray(z["c%d" % k])
        scan, _ = orc.get_scan(cs["tx_size"], cs["tx_type"])
        q, d, eob, err = oracle_quantize_lp(c, cs["tables"], scan)
        assert np.array_equal(q, z["q%d" % k]), cs
        assert np.array_equal(d, z["d%d" % k]), cs
        assert eob == cs["eob"] and err == cs["block_error"], cs
        dq = np.full(c.size, cs["tables"]["dequant"][1], np.int64)
        dq[0] = cs["tables"]["dequant"][0]
        clamp += int(np.any(q.astype(np.int64) * dq != d))
    assert clamp >= 10         # the truncated int16 dqcoeff store is exercised
