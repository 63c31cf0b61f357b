"""The oracle's two loop-restoration filters as the search applies them -- orc_apply_selfguided_restoration and orc_wiener_convolve_add_src
(oracle/aomref_sgr.c) -- against the reference's own av1_apply_selfguided_restoration_c and the two passes of av1_[highbd_]wiener_convolve_add_src_c,
interpreted where they lie (tests/golden/ref_eval_lr_apply.npz, gen_ref_eval_lr_apply.py): 15 + 12 cases, 8 / 10 / 12 bits, bit for bit."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_lr_apply.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def orc_lr(oracle, img, c, x0=3, y0=3):
    lib = oracle.lib
    bd, w, h = c["bd"], c["w"], c["h"]
    dt = np.uint8 if bd == 8 else np.uint16
    a = np.ascontiguousarray(img, dt)
    out = np.zeros((h, w), dt)
    at = a.ctypes.data + (y0 * a.shape[1] + x0) * a.itemsize
    if c["kind"] == "sgr":
        lib.orc_apply_selfguided_restoration.restype = None
        lib.orc_apply_selfguided_restoration.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        xqd = np.array(c["xqd"], np.int32)
        lib.orc_apply_selfguided_restoration(at, int(bd > 8), w, h, a.shape[1], c["idx"], xqd.ctypes.data, out.ctypes.data, w, bd)
    else:
        lib.orc_wiener_convolve_add_src.restype = None
        lib.orc_wiener_convolve_add_src.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        fx, fy = np.array(c["fx"], np.int16), np.array(c["fy"], np.int16)
        lib.orc_wiener_convolve_add_src(at, int(bd > 8), a.shape[1], out.ctypes.data, w, fx.ctypes.data, fy.ctypes.data, w, h, bd)
    return out


def test_oracle_restoration_filters_reproduce_the_interpreted_reference(oracle):
    z, cases = load()
    assert len(cases) == 27
    kinds = {"sgr": 0, "wiener": 0}
    for c in cases:
        out = orc_lr(oracle, z["img%d" % c["k"]], c)
        assert np.array_equal(out.ravel().astype(np.uint16), z["out%d" % c["k"]]), c
        kinds[c["kind"]] += 1
    assert kinds == {"sgr": 15, "wiener": 12}
