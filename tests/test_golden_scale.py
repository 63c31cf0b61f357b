"""The oracle's scaled-reference predictor (orc_convolve_2d_scale, oracle/aomref_convolve.c) against the reference's own av1_convolve_2d_scale_c /
av1_highbd_convolve_2d_scale_c (av1/common/convolve.c) interpreted where they lie: tests/golden/ref_eval_scale.npz (gen_ref_eval_scale.py), 42 cases --
single and compound (CONV_BUF, average, distance weights), steps 512 .. 2048 with unequal x / y, the four filter families -- bit for bit."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_scale.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def orc_scale(oracle, planes, c):
    """-> (CONV_BUF after the first reference or None, prediction)"""
    lib = oracle.lib
    lib.orc_convolve_2d_scale.restype = None
    lib.orc_convolve_2d_scale.argtypes = [C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 16 + [C.c_void_p, C.c_int]
    bd, w, h = c["bd"], c["w"], c["h"]
    dt = np.uint8 if bd == 8 else np.uint16
    dst = np.zeros((h, w), dt)
    conv = np.zeros((h, w), np.uint16)
    first = None
    wts = c["weights"]
    for r in range(2 if c["compound"] else 1):
        pl = np.ascontiguousarray(planes[r], dt)
        x0, y0 = c["pos"][r]
        lib.orc_convolve_2d_scale(pl.ctypes.data + (y0 * pl.shape[1] + x0) * pl.itemsize, pl.shape[1], dst.ctypes.data, w, w, h, c["fx"], c["fy"], c["subs"][r][0],
                                  c["xs"], c["subs"][r][1], c["ys"], int(bd > 8), bd, c["compound"], r, int(wts is not None), wts[0] if wts else 0,
                                  wts[1] if wts else 0, conv.ctypes.data, w)
        if c["compound"] and r == 0:
            first = conv.copy()
    return first, dst


def test_oracle_scaled_predictor_reproduces_the_interpreted_reference(oracle):
    z, cases = load()
    assert len(cases) == 42
    kinds = set()
    for c in cases:
        planes = [z["p%d_%d" % (c["bd"], r)] for r in range(2)]
        conv, pred = orc_scale(oracle, planes, c)
        if c["compound"]:
            assert np.array_equal(conv.ravel(), z["c%d" % c["k"]]), c
        assert np.array_equal(pred.ravel().astype(np.uint16), z["d%d" % c["k"]]), c
        kinds.add((c["compound"], c["weights"] is not None))
    assert kinds == {(0, False), (1, False), (1, True)}
