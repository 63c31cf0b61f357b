"""The variance-based partitioning's leaf statistics: the restatement against fill_variance_8x8avg / compute_minmax_8x8 / fill_variance_4x4avg
(av1/encoder/var_based_part.c:255-430) interpreted (tests/golden/gen_ref_eval_vbp.py), bit-exact."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_vbp.npz")


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def _vp(a):
    return C.c_void_p(a.ctypes.data)


def oracle_8x8(src, dst, x16, y16, hbd, pw, ph, x0=0, y0=0):
    """src / dst: planes whose (y0, x0) is the superblock's first pixel -> (sum[4], sse[4], minmax)"""
    S = src.shape[1]
    off = (y0 * S + x0) * src.itemsize
    s, q = np.zeros(4, np.int32), np.zeros(4, np.uint32)
    orc.lib.orc_vbp_fill_8x8avg(C.c_void_p(src.ctypes.data + off), S, C.c_void_p(dst.ctypes.data + off), S, x16, y16, hbd, pw, ph, _vp(s), _vp(q))
    mm = orc.lib.orc_vbp_minmax_8x8(C.c_void_p(src.ctypes.data + off), S, C.c_void_p(dst.ctypes.data + off), S, x16, y16, hbd, pw, ph)
    return s.tolist(), q.tolist(), int(mm)


def oracle_4x4(src, x8, y8, hbd, pw, ph, bo, x0=0, y0=0):
    S = src.shape[1]
    off = (y0 * S + x0) * src.itemsize
    s, q = np.zeros(4, np.int32), np.zeros(4, np.uint32)
    orc.lib.orc_vbp_fill_4x4avg(C.c_void_p(src.ctypes.data + off), S, x8, y8, hbd, pw, ph, bo, _vp(s), _vp(q))
    return s.tolist(), q.tolist()


def planes_of(z, bd):
    dt = np.uint8 if bd == 8 else np.uint16
    return np.ascontiguousarray(z["src%d" % bd].astype(dt)), np.ascontiguousarray(z["dst%d" % bd].astype(dt))


def test_leaf_statistics_match_the_reference():
    z, cases = load()
    assert len(cases) >= 88
    outside = big = 0
    for c in cases:
        src, dst = planes_of(z, c["bd"])
        if c["kind"] == "8x8":
            s, q, mm = oracle_8x8(src, dst, c["x16"], c["y16"], int(c["bd"] > 8), c["pw"], c["ph"])
            assert (s, q, mm) == (c["sum"], c["sse"], c["minmax"]), c
            big += int(abs(mm) > 255)
        else:
            s, q = oracle_4x4(src, c["x8"], c["y8"], int(c["bd"] > 8), c["pw"], c["ph"], c["border_offset"])
            assert (s, q) == (c["sum"], c["sse"]), c
        outside += int(0 in c["sse"])
    assert outside >= 20 and big >= 1
