"""oracle/aomref_tf.c against tests/golden/ref_eval_tf_apply.npz -- the reference's own 12-tap facade, tf_apply_temporal_filter_self,
av1_apply_temporal_filter_c and tf_normalize_filtered_frame, interpreted where they lie (tests/golden/gen_ref_eval_tf_apply.py).
Integer results (predictor pixels, accum, count, normalised pixels) must be bit-exact; the weights go through libm's exp() on both
sides (the same libm here), so they are compared exactly too."""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fx():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_tf_apply.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def _p(a):
    return C.c_void_p(a.ctypes.data)


def test_convolve12_equals_the_interpreted_facade(oracle, fx):
    z, cases = fx
    lib = oracle.lib
    lib.orc_convolve_sr12.restype = None
    n = 0
    for c in cases:
        if c["kind"] != "convolve12":
            continue
        bd, w, h = c["bd"], c["w"], c["h"]
        plane = z["p%d" % bd].astype(np.uint8 if bd == 8 else np.uint16)
        S = plane.shape[1]
        esz = plane.itemsize
        dst = np.zeros((h, w), plane.dtype)
        lib.orc_convolve_sr12(C.c_void_p(plane.ctypes.data + (c["y0"] * S + c["x0"]) * esz), S, _p(dst), w, w, h, c["sx"], c["sy"], int(bd > 8), bd)
        assert np.array_equal(dst.ravel(), z["c%d" % c["k"]].astype(plane.dtype)), c
        n += 1
    assert n == 30


def test_apply_self_filter_normalize_equal_the_interpreted_reference(oracle, fx):
    z, cases = fx
    lib = oracle.lib
    for f in (lib.orc_tf_apply_block, lib.orc_tf_apply_self_block, lib.orc_tf_normalize_block):
        f.restype = None
        f.argtypes = None
    n = 0
    for c in cases:
        if c["kind"] != "apply":
            continue
        k, bd, P, ssx, ssy = c["k"], c["bd"], c["planes"], c["ss_x"], c["ss_y"]
        dt = np.uint8 if bd == 8 else np.uint16
        e16 = int(bd > 8)
        frames = [[np.ascontiguousarray(z["f%d_%d_%d" % (k, p, f)].astype(dt)) for p in range(P)] for f in range(3)]
        strides = (C.c_int * 3)(*([frames[1][p].shape[1] for p in range(P)] + [0] * (3 - P)))
        outs = [np.zeros_like(frames[1][p]) for p in range(P)]
        noise = (C.c_double * 3)(*c["noise"])
        pels = 1024 + (2 * (32 >> ssx) * (32 >> ssy) if P == 3 else 0)
        for b in c["blocks"]:
            r, cc = b["mb_row"], b["mb_col"]
            accum, count = np.zeros(3 * 1024, np.uint32), np.zeros(3 * 1024, np.uint16)
            fp = (C.c_void_p * 3)(*([frames[1][p].ctypes.data for p in range(P)] + [None] * (3 - P)))
            lib.orc_tf_apply_self_block(fp, strides, C.c_int(P), C.c_int(ssx), C.c_int(ssy), C.c_int(r), C.c_int(cc), _p(accum), _p(count), C.c_int(e16))
            for f, ref in zip((0, 2), b["refs"]):
                pred = np.ascontiguousarray(z["pred%d_%d_%d_%d" % (k, r, cc, f)].astype(dt))
                mvs = np.ascontiguousarray(ref["mvs"], np.int16)
                mses = np.ascontiguousarray(ref["mses"], np.int32)
                lib.orc_tf_apply_block(fp, strides, C.c_int(c["w"]), C.c_int(c["h"]), C.c_int(P), C.c_int(ssx), C.c_int(ssy), C.c_int(r), C.c_int(cc),
                                       noise, _p(mvs), _p(mses), C.c_int(c["q"]), C.c_int(c["strength"]), _p(pred), _p(accum), _p(count),
                                       C.c_int(e16), C.c_int(bd))
                assert np.array_equal(accum[:pels], z["accum%d_%d_%d_%d" % (k, r, cc, f)]), (k, r, cc, f)
                assert np.array_equal(count[:pels], z["count%d_%d_%d_%d" % (k, r, cc, f)]), (k, r, cc, f)
                n += 1
            op = (C.c_void_p * 3)(*([outs[p].ctypes.data for p in range(P)] + [None] * (3 - P)))
            lib.orc_tf_normalize_block(op, strides, C.c_int(P), C.c_int(ssx), C.c_int(ssy), C.c_int(r), C.c_int(cc), _p(accum), _p(count), C.c_int(e16))
        for p in range(P):
            assert np.array_equal(outs[p], z["out%d_%d" % (k, p)].astype(dt)), (k, p)
        # the weights really vary (a fixture where every weight is 0 or 1000 would pin nothing)
        w = z["count%d_%d_%d_%d" % (k, 1, 1, 2)][:1024].astype(np.int64) - 1000
        assert len(np.unique(w)) > 100, np.unique(w)
    assert n == 20
