"""oracle/aomref_yrd.c against av1_estimate_txfm_yrd interpreted with its callees (tests/golden/gen_ref_eval_yrd.py -> ref_eval_yrd.npz): rate,
distortion, sse, skip flag and the returned rd cost of inter blocks 4x4 .. 128x128 (one to four transform blocks with the entropy-context update
between them), 8 / 10-bit."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_yrd.npz")


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def oracle_yrd(z, c, lossless=0):
    k = c["k"]
    res = np.ascontiguousarray(z["res%d" % k], np.int16)
    q = orc.build_quantizer_y(c["bd"], c["qindex"])
    tabs = np.ascontiguousarray(np.stack([np.asarray(q[n], np.int16)[:2] for n in ("zbin", "round", "quant", "quant_shift", "dequant")]))
    above, left = (np.ascontiguousarray(z["%s%d" % (n, k)]) for n in ("above", "left"))
    costs = np.ascontiguousarray(z["costs%d" % k], np.int32)
    out = np.zeros(4, np.int64)
    f = orc.lib.orc_estimate_txfm_yrd
    f.restype = C.c_int64
    rd = f(C.c_void_p(res.ctypes.data), res.shape[1], c["bw"], c["bh"], c["bd"], int(c["bd"] > 8), C.c_void_p(tabs.ctypes.data), C.c_void_p(above.ctypes.data),
           C.c_void_p(left.ctypes.data), C.c_void_p(costs.ctypes.data), c["tx_type_rate"], c["tx_size_rate"], c["no_skip_txfm_rate"],
           c["skip_txfm_rate"], c["rdmult"], lossless, C.c_void_p(out.ctypes.data))
    return int(rd), [int(v) for v in out]


def test_estimate_txfm_yrd_matches_the_reference():
    z, cases = load()
    assert len(cases) >= 50
    multi = skipped = forced = 0
    for c in cases:
        rd, (rate, skip, dist, sse) = oracle_yrd(z, c)
        assert (rd, rate, skip, dist, sse) == (int(c["rd"]), c["rate"], c["skip_txfm"], int(c["dist"]), int(c["sse"])), c
        multi += len(c["eobs"]) > 1
        skipped += all(e == 0 for e in c["eobs"])
        forced += c["skip_txfm"] == 1 and any(e > 0 for e in c["eobs"])     # the forced-skip check at the end of the function took over
    assert multi >= 7 and skipped >= 5 and forced >= 1


def test_second_mv_choice_is_a_strict_rd_comparison():
    """motion_search_facade.c:404-423: `if (tmp_rd < rd)` on RDCOST(rdmult, mv_rate + rate, dist) -- the arithmetic is pinned by the rd values of the
    fixture above (RDCOST with rate 0 extra); here only the comparison's direction and the tie."""
    f = orc.lib.orc_second_mv_rd_choice
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64]
    assert f(100, 10, 500, 1000, 10, 500, 999) == 1 and f(100, 10, 500, 1000, 10, 500, 1000) == 0 and f(100, 10, 500, 1000, 12, 499, 1000) == 0
    assert f(512, 0, 1, 0, 0, 0, 0) == 1    # ROUND_POWER_OF_TWO(512, 9) = 1 > 0
