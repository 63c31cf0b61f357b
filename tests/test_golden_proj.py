"""The oracle's self-guided projection statistics (orc_calc_proj_params / orc_pixel_proj_error, oracle/aomref_lrstats.c) against the reference's own
av1_calc_proj_params[_high_bd]_c and av1_[lowbd|highbd]_pixel_proj_error_c (av1/encoder/pickrst.c:226-370,470-657) interpreted where they lie:
tests/golden/ref_eval_proj.npz (tests/golden/gen_ref_eval_proj.py): 72 units x (H, C, 5 errors), the three radius branches, 8 / 10 / 12 bits."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_proj.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def bind(oracle):
    lib = oracle.lib
    lib.orc_calc_proj_params.restype = None
    lib.orc_calc_proj_params.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]
    lib.orc_pixel_proj_error.restype = C.c_int64
    lib.orc_pixel_proj_error.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int]
    return lib


def planes_of(z, c):
    dt = np.uint8 if c["bd"] == 8 else np.uint16
    k = c["k"]
    return (np.ascontiguousarray(z["s%d" % k], dt), np.ascontiguousarray(z["d%d" % k], dt), np.ascontiguousarray(z["f0_%d" % k], np.int32),
            np.ascontiguousarray(z["f1_%d" % k], np.int32))


def test_oracle_projection_statistics_reproduce_the_interpreted_reference(oracle):
    z, cases = load()
    lib = bind(oracle)
    assert len(cases) == 72
    branches = set()
    for c in cases:
        src, dat, f0, f1 = planes_of(z, c)
        H, Cc = np.zeros(4, np.int64), np.zeros(2, np.int64)
        r0, r1 = c["r"]
        lib.orc_calc_proj_params(src.ctypes.data, c["w"], c["h"], c["S"], dat.ctypes.data, c["S"], f0.ctypes.data, c["FS"], f1.ctypes.data, c["FS"], int(c["bd"] > 8),
                                 r0, r1, H.ctypes.data, Cc.ctypes.data)
        assert H.tolist() == c["H"] and Cc.tolist() == c["C"], c
        for (xq0, xq1), want in zip(c["xq"], c["err"]):
            got = lib.orc_pixel_proj_error(src.ctypes.data, c["w"], c["h"], c["S"], dat.ctypes.data, c["S"], f0.ctypes.data, c["FS"], f1.ctypes.data, c["FS"],
                                           int(c["bd"] > 8), r0, r1, xq0, xq1)
            assert got == want, (c["k"], xq0, xq1)
        branches.add((r0 > 0, r1 > 0))
    assert branches == {(True, True), (True, False), (False, True)}
