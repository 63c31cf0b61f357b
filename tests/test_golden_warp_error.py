"""The global-motion model error's restatement against the interpreted reference (tests/golden/gen_ref_eval_warp_error.py): av1_get_shear_params,
av1_warp_error (av1/encoder/global_motion.c:128-224) and av1_segmented_frame_error (av1/common/warped_motion.c:400-460,687-760), bit-exact."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_warp_error.npz")
INT64_MAX = (1 << 63) - 1


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def planes_of(z, c):
    dt = np.uint8 if c["bd"] == 8 else np.uint16
    return np.ascontiguousarray(z["ref%d" % c["bd"]].astype(dt)), np.ascontiguousarray(z["cur%d" % c["bd"]].astype(dt))


def oracle_shear(mat):
    m, out = np.asarray(mat, np.int32), np.zeros(4, np.int16)
    ok = orc.lib.orc_get_shear_params(m.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return int(ok), out


def oracle_warp_error(c, ref, cur, shear, best_error=INT64_MAX):
    m, seg = np.asarray(c["mat"], np.int32), np.asarray(c["seg"], np.uint8)
    sh = np.asarray(shear, np.int16)
    f = orc.lib.orc_warp_error
    f.restype = C.c_int64
    return int(f(m.ctypes.data_as(C.c_void_p), sh.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p), int(c["bd"] > 8), c["W"], c["H"], c["W"],
                 cur.ctypes.data_as(C.c_void_p), c["p_col"], c["p_row"], c["pw"], c["ph"], c["W"], c["ss"], c["ss"], c["bd"], C.c_int64(best_error),
                 seg.ctypes.data_as(C.c_void_p), c["seg_stride"]))


def oracle_frame_error(c, ref, cur):
    seg = np.asarray(c["seg"], np.uint8)
    f = orc.lib.orc_segmented_frame_error
    f.restype = C.c_int64
    return int(f(ref.ctypes.data_as(C.c_void_p), int(c["bd"] > 8), c["W"], cur.ctypes.data_as(C.c_void_p), c["W"], c["H"], c["W"], c["bd"],
                 seg.ctypes.data_as(C.c_void_p), c["seg_stride"]))


def test_shear_parameters_and_model_error_match_the_reference():
    z, cases = load()
    assert len(cases) >= 17
    seen = {"invalid": 0, "bounded": 0, "frame": 0}
    for c in cases:
        ref, cur = planes_of(z, c)
        ok, shear = oracle_shear(c["mat"])
        if c["mat"][2] > 0:                       # (is_affine_valid fails before anything is written)
            assert shear.tolist() == c["shear"], c["k"]
        assert ok == c["valid"], c["k"]
        if not ok:
            assert int(c["error"]) == INT64_MAX
            seen["invalid"] += 1
            continue
        assert oracle_warp_error(c, ref, cur, shear) == int(c["error"]), c["k"]
        if "best_error" in c:
            assert oracle_warp_error(c, ref, cur, shear, int(c["best_error"])) == INT64_MAX == int(c["error_bounded"])
            seen["bounded"] += 1
        if "frame_error" in c:
            assert oracle_frame_error(c, ref, cur) == int(c["frame_error"]), c["k"]
            seen["frame"] += 1
    assert seen["invalid"] >= 2 and seen["bounded"] >= 4 and seen["frame"] >= 6
