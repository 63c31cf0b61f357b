"""The oracle's warped-motion predictor (orc_warp_affine, oracle/aomref_warp.c) against the reference's own av1_warp_affine_c /
av1_highbd_warp_affine_c (av1/common/warped_motion.c:264-393,538-675) interpreted where they lie: tests/golden/ref_eval_warp.npz
(tests/golden/gen_ref_eval_warp.py), 42 cases -- 8 / 10 / 12 bits, luma and 4:2:0 geometry, footprints that leave the frame -- bit for bit."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_warp.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def orc_warp(oracle, plane, bd, c):
    lib = oracle.lib
    lib.orc_warp_affine.restype = None
    lib.orc_warp_affine.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] + [C.c_int] * 13
    h, w = plane.shape
    ref = np.ascontiguousarray(plane, np.uint8 if bd == 8 else np.uint16)
    mat = np.array(c["mat"], np.int32)
    out = np.zeros((c["ph"], c["pw"]), ref.dtype)
    a, b, g, d = c["shear"]
    lib.orc_warp_affine(mat.ctypes.data, ref.ctypes.data, int(bd > 8), w, h, w, out.ctypes.data, c["p_col"], c["p_row"], c["pw"], c["ph"], c["pw"], c["ss"], c["ss"],
                        bd, c["round_0"], a, b, g, d)
    return out


def test_oracle_warp_reproduces_the_interpreted_reference(oracle):
    z, cases = load()
    assert len(cases) == 42
    clamped = identity = 0
    for c in cases:
        plane = z["ref%d" % c["bd"]]
        got = orc_warp(oracle, plane, c["bd"], c)
        want = z["d%d" % c["k"]].reshape(c["ph"], c["pw"])
        assert np.array_equal(got.astype(np.uint16), want), c
        clamped += int(abs(c["mat"][0]) > (40 << 16) or abs(c["mat"][1]) > (40 << 16))
        if c["mat"] == [0, 0, 1 << 16, 0, 0, 1 << 16] and c["ss"] == 0:
            # the identity model: the phase-0 kernel of Warped_Filters is { 0, 0, 0, 127, 1, 0, 0, 0 }, not a pure copy -- within 2 of the block
            blk = plane[c["p_row"]:c["p_row"] + c["ph"], c["p_col"]:c["p_col"] + c["pw"]].astype(np.int32)
            assert np.abs(want.astype(np.int32) - blk).max() <= max(4, (1 << c["bd"]) // 64)
            identity += 1
    assert clamped >= 6 and identity >= 2


def orc_warp_compound(oracle, planes, bd, c):
    """Both calls of a compound: reference 0 into the block's CONV_BUF, reference 1 blended in -> (conv buffer, prediction)."""
    lib = oracle.lib
    lib.orc_warp_affine_compound.restype = None
    lib.orc_warp_affine_compound.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] + [C.c_int] * 17 + [C.c_void_p, C.c_int]
    dt = np.uint8 if bd == 8 else np.uint16
    out = np.zeros((c["ph"], c["pw"]), dt)
    conv = np.zeros((c["ph"], c["pw"]), np.uint16)
    wts = c["weights"]
    for r in range(2):
        ref = np.ascontiguousarray(planes[r], dt)
        h, w = ref.shape
        mat = np.array(c["mat"][r], np.int32)
        a, b, g, d = c["shear"][r]
        lib.orc_warp_affine_compound(mat.ctypes.data, ref.ctypes.data, int(bd > 8), w, h, w, out.ctypes.data, c["p_col"], c["p_row"], c["pw"], c["ph"], c["pw"],
                                     c["ss"], c["ss"], bd, c["round_0"], a, b, g, d, r, int(wts is not None), wts[0] if wts else 0, wts[1] if wts else 0,
                                     conv.ctypes.data, c["pw"])
        if r == 0:
            first = conv.copy()
    return first, out


def test_oracle_compound_warp_reproduces_the_interpreted_reference(oracle):
    z = np.load(os.path.join(HERE, "golden", "ref_eval_warp_compound.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    assert len(cases) == 24
    weighted = 0
    for c in cases:
        planes = [z["ref%d_%d" % (c["bd"], r)] for r in range(2)]
        conv, pred = orc_warp_compound(oracle, planes, c["bd"], c)
        assert np.array_equal(conv.ravel(), z["c%d" % c["k"]]), c
        assert np.array_equal(pred.ravel().astype(np.uint16), z["d%d" % c["k"]]), c
        weighted += c["weights"] is not None
    assert weighted == 12
