"""av1_selectSamples and av1_find_projection (the local warp model's least-squares fit) interpreted (tests/golden/ref_eval_warpfit.npz, generator
tests/golden/gen_ref_eval_warpfit.py) against the oracle's restatement (oracle/aomref_warpfit.c) and -- no GPU needed -- the product's host functions
aomhip_select_samples / aomhip_find_projection (aom-av1-psy_amd/host/warp_model.c)."""
import ctypes as C
import json
import os

import numpy as np

import pyoracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_warpfit.npz"))
    return json.loads(bytes(z["cases"]).decode())


def oracle_fit(c):
    pts, pin = np.array(c["pts"], np.int32), np.array(c["pts_inref"], np.int32)
    sel = c["n"]
    if c["n"] > 1:
        sel = orc.lib.orc_select_samples(c["mv"][0], c["mv"][1], pts.ctypes.data_as(C.c_void_p), pin.ctypes.data_as(C.c_void_p), c["n"], c["w"], c["h"])
    mat, sh = np.array([0, 0, 1 << 16, 0, 0, 1 << 16], np.int32), np.zeros(4, np.int16)
    bad = orc.lib.orc_find_projection(sel, pts.ctypes.data_as(C.c_void_p), pin.ctypes.data_as(C.c_void_p), c["w"], c["h"], c["mv"][0], c["mv"][1],
                                      mat.ctypes.data_as(C.c_void_p), sh.ctypes.data_as(C.c_void_p), c["mi_row"], c["mi_col"])
    return sel, pts, pin, bad, mat, sh


def check(c, sel, pts, pin, bad, mat, sh):
    assert sel == c["selected"], c["k"]
    assert pts[:2 * sel].tolist() == c["sel_pts"][:2 * sel] and pin[:2 * sel].tolist() == c["sel_pts_inref"][:2 * sel], c["k"]
    assert bad == c["invalid"], (c["k"], bad, c["invalid"])
    # (a singular system returns before anything is written: the model is as it was)
    assert mat.tolist() == c["mat"], (c["k"], mat.tolist(), c["mat"])
    if not c["invalid"]:
        assert sh.tolist() == c["shear"], c["k"]


def test_oracle_fit_equals_the_interpreted_reference():
    cases = load()
    for c in cases:
        check(c, *oracle_fit(c))
    assert sum(c["invalid"] for c in cases) >= 20 and sum(c["selected"] < c["n"] for c in cases) >= 50 and len(cases) >= 250
    assert sum(1 for c in cases if not c["invalid"] and abs(c["mat"][0]) < (1 << 23) - 1 and abs(c["mat"][1]) < (1 << 23) - 1) >= 40   # unclamped translations too


def test_host_fit_equals_the_interpreted_reference():
    import importlib
    capi = importlib.import_module("aom-av1-psy_amd.capi")
    for c in load():
        pts, pin = np.array(c["pts"], np.int32), np.array(c["pts_inref"], np.int32)
        sel = c["n"]
        if c["n"] > 1:
            sel = capi.select_samples(c["mv"], pts, pin, c["n"], c["w"], c["h"])
        rec = np.zeros(1, capi.warp_model_dtype)
        rec["mat"][0] = [0, 0, 1 << 16, 0, 0, 1 << 16]
        ok = capi.find_projection(sel, pts, pin, c["w"], c["h"], c["mv"], rec, c["mi_row"], c["mi_col"])
        check(c, sel, pts, pin, int(not ok), rec["mat"][0], np.array([rec[f][0] for f in ("alpha", "beta", "gamma", "delta")], np.int16))
