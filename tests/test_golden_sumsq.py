"""The residual statistics' restatement against aom_sum_squares_2d_i16_c / aom_sum_sse_2d_i16_c interpreted (tests/golden/gen_ref_eval_sumsq.py)."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_sumsq.npz")


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def oracle_sum_sse(plane, x, y, w, h, sum_in=0):
    S = plane.shape[1]
    sm = C.c_int(sum_in)
    f = orc.lib.orc_sum_sse_2d_i16
    f.restype = C.c_uint64
    ss = f(C.c_void_p(plane.ctypes.data + 2 * (y * S + x)), S, w, h, C.byref(sm))
    return int(ss), sm.value


def test_sum_of_squares_matches_the_reference():
    z, cases = load()
    assert len(cases) >= 54
    for c in cases:
        plane = np.ascontiguousarray(z[c["plane"]])
        ss, sm = oracle_sum_sse(plane, c["x"], c["y"], c["w"], c["h"], c["sum_in"])
        assert (ss, sm) == (int(c["ss"]), c["sum_out"]), c
    assert max(int(c["ss"]) for c in cases) > 1 << 36
