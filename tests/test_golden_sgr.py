"""The oracle's self-guided restoration filter (orc_selfguided_restoration, oracle/aomref_sgr.c) against the reference's own
av1_selfguided_restoration_c (av1/common/restoration.c:871-915) interpreted where it lies: tests/golden/ref_eval_sgr.npz
(tests/golden/gen_ref_eval_sgr.py), 23 units (8 / 10 / 12 bits, the three radius combinations, odd sizes, flat and extreme areas), bit for bit;
and the property that lets the device tile a restoration unit freely: a unit's output does not depend on how it is cut into processing units."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_sgr.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def orc_sgr(oracle, img, bd, x0, y0, w, h, idx):
    """flt0, flt1 of the w x h unit at (x0, y0) of img (which holds >= 3 pixels around it)."""
    lib = oracle.lib
    lib.orc_selfguided_restoration.restype = None
    lib.orc_selfguided_restoration.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    dt = np.uint8 if bd == 8 else np.uint16
    a = np.ascontiguousarray(img, dt)
    f0, f1 = np.full((h, w), -7, np.int32), np.full((h, w), -7, np.int32)
    lib.orc_selfguided_restoration(a.ctypes.data + (y0 * a.shape[1] + x0) * a.itemsize, int(bd > 8), w, h, a.shape[1], f0.ctypes.data, f1.ctypes.data, w, idx, bd)
    return f0, f1


def test_oracle_self_guided_filter_reproduces_the_interpreted_reference(oracle):
    z, cases = load()
    assert len(cases) == 23
    combos = set()
    for c in cases:
        k = c["k"]
        f0, f1 = orc_sgr(oracle, z["img%d" % k], c["bd"], 3, 3, c["w"], c["h"], c["idx"])
        assert np.array_equal(f0.ravel(), z["f0_%d" % k]), c
        assert np.array_equal(f1.ravel(), z["f1_%d" % k]), c
        combos.add((bool((z["f0_%d" % k] != -7).any()), bool((z["f1_%d" % k] != -7).any())))
    assert combos == {(True, True), (True, False), (False, True)}     # a switched-off filter leaves its output untouched


def test_output_does_not_depend_on_the_processing_unit_tiling(oracle):
    rng = np.random.default_rng(8)
    for bd, idx in ((8, 3), (10, 12), (12, 15)):
        mx = (1 << bd) - 1
        img = rng.integers(0, mx + 1, (150, 200))
        whole = orc_sgr(oracle, img, bd, 5, 4, 160, 128, idx)
        for (tw, th) in ((64, 64), (32, 16), (160, 2)):     # (tile heights even: the r[0] filter's row parity is relative to the unit's top)
            for y in range(0, 128, th):
                for x in range(0, 160, tw):
                    w, h = min(tw, 160 - x), min(th, 128 - y)
                    part = orc_sgr(oracle, img, bd, 5 + x, 4 + y, w, h, idx)
                    for p, q in zip(part, whole):
                        assert np.array_equal(p, q[y:y + h, x:x + w]), (bd, idx, tw, th, x, y)
