"""The coefficient contexts' restatement against av1_get_nz_map_contexts_c interpreted (tests/golden/gen_ref_eval_nzmap.py), bit-exact, on the level
maps of orc_txb_init_levels."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_nzmap.npz")
TXW = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TXH = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def oracle_contexts(coeff, tx_size, tx_type, eob, fill=-7):
    W, H = TXW[tx_size], TXH[tx_size]
    w, h = min(W, 32), min(H, 32)
    lv = np.full((w + 4) * (h + 4) + 16, 0x55, np.uint8)
    orc.lib.orc_txb_init_levels.restype = None
    orc.lib.orc_txb_init_levels(C.c_void_p(np.ascontiguousarray(coeff, np.int32).ctypes.data), w, h, C.c_void_p(lv.ctypes.data))
    scan, _ = orc.get_scan(tx_size, tx_type)
    sc = np.ascontiguousarray(scan, np.int16)
    out = np.full(w * h, fill, np.int8)
    tx_class = 0 if tx_type < 10 else (2 if tx_type % 2 == 0 else 1)        # tx_type_to_class (txb_common.h:30-48): V_* -> VERT, H_* -> HORIZ
    orc.lib.orc_get_nz_map_contexts.restype = None
    orc.lib.orc_get_nz_map_contexts(C.c_void_p(lv.ctypes.data), C.c_void_p(sc.ctypes.data), eob, W, H, tx_class, C.c_void_p(out.ctypes.data))
    return out


def test_contexts_match_the_reference():
    z, cases = load()
    assert len(cases) >= 100
    seen = set()
    for c in cases:
        got = oracle_contexts(z["c%d" % c["k"]], c["tx_size"], c["tx_type"], c["eob"])
        assert np.array_equal(got, z["x%d" % c["k"]]), c
        seen |= set(int(v) for v in got if v >= 0)
    assert len(seen) >= 38          # of the 42 contexts (26 two-dimensional + 16 one-dimensional)
