"""The projection-based motion estimation's restatement against the interpreted reference (tests/golden/gen_ref_eval_intpro.py):
av1_int_pro_motion_estimation (av1/encoder/mcomp.c:1897-2105) as written, vector and SAD, bit-exact."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_intpro.npz")


def load():
    z = np.load(GOLD)
    meta = json.loads(bytes(z["meta"]))
    return z, meta


def oracle_int_pro(src_b, ref_b, border, bx, by, w, h, bd, limits_rrcc, ref_mv):
    """src_b / ref_b: bordered planes; limits_rrcc = (row_min, row_max, col_min, col_max) as x->mv_limits; -> (best_sad, [row, col])"""
    S = src_b.shape[1]
    es = src_b.itemsize
    off = ((border + by) * S + border + bx) * es
    lim = np.array([limits_rrcc[2], limits_rrcc[3], limits_rrcc[0], limits_rrcc[1]], np.int32)
    rm, out = np.asarray(ref_mv, np.int16), np.zeros(2, np.int16)
    f = orc.lib.orc_int_pro_motion_estimation
    f.restype = C.c_uint
    sad = f(C.c_void_p(src_b.ctypes.data + off), S, C.c_void_p(ref_b.ctypes.data + off), S, w, h, bd, C.c_void_p(lim.ctypes.data), C.c_void_p(rm.ctypes.data),
            C.c_void_p(out.ctypes.data))
    return int(sad), out.tolist()


def planes(z, bd):
    dt = np.uint8 if bd == 8 else np.uint16
    return np.ascontiguousarray(z["src%d" % bd].astype(dt)), np.ascontiguousarray(z["ref%d" % bd].astype(dt))


def test_int_pro_motion_estimation_matches_the_reference():
    z, meta = load()
    cases = meta["cases"]
    assert len(cases) >= 20
    moved = clamped = 0
    for c in cases:
        s, r = planes(z, c["bd"])
        sad, mv = oracle_int_pro(s, r, meta["border"], c["bx"], c["by"], c["w"], c["h"], c["bd"], c["limits"], c["ref_mv"])
        assert (sad, mv) == (c["best_sad"], c["mv"]), c
        moved += int(mv != [0, 0])
        clamped += int(c["limits"] == [-2, 3, -1, 2] or abs(c["ref_mv"][0]) > 8000)
    assert moved >= 12 and clamped >= 6
