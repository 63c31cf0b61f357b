"""The small members of the named files (oracle/aomref_misc.c) against the reference's own functions interpreted
(tests/golden/gen_ref_eval_leftovers.py -> ref_eval_leftovers.npz): aom_get_mb_ss, aom_mse_wxh_16bit / _16xh_ / _highbd,
aom_[highbd_]comp_mask_pred, av1_return_max / _min_sub_pixel_mv."""
import json
import os

import numpy as np

import pyoracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_leftovers.npz")


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def test_get_mb_ss_and_the_16_bit_mse_match_the_reference():
    z, cases = load()
    n = 0
    for c in cases:
        if c["kind"] == "mb_ss":
            assert orc.get_mb_ss(z[c["a"]]) == c["out"], c
            n += 1
        elif c["kind"] == "mse_wxh":
            dst, src = np.ascontiguousarray(z["mse_dst%d" % c["bd"]]), np.ascontiguousarray(z["mse_src%d" % c["bd"]])
            got = orc.mse_wxh_16bit(dst[c["y"]:, c["x"]:], src[c["y"]:, c["x"]:], c["w"], c["h"])
            assert got == int(c["out"]), c
            n += 1
        elif c["kind"] == "mse_16xh":
            dst = np.ascontiguousarray(z["mse_dst8"])
            assert orc.mse_16xh_16bit(dst[c["y"]:, c["x"]:], z[c["src"]], c["w"], c["h"]) == int(c["out"]), c
            n += 1
    assert n >= 4 + 36 + 5
    assert any(c["kind"] == "mb_ss" and c["out"] == 0 for c in cases)   # 256 * 2^30 wraps to 0 in the reference's unsigned sum


def test_comp_mask_pred_matches_the_reference():
    z, cases = load()
    n = 0
    for c in cases:
        if c["kind"] != "comp_mask":
            continue
        k = c["k"]
        got = orc.comp_mask_pred(z["cmp_pred%d" % k], np.ascontiguousarray(z["cmp_ref%d" % k]), np.ascontiguousarray(z["cmp_mask%d" % k]), c["invert"])
        assert np.array_equal(got, z["cmp_out%d" % k]), c
        n += 1
    assert n == 36


def test_extreme_sub_pixel_mvs_match_the_reference():
    _, cases = load()
    n = odd = 0
    for c in cases:
        if c["kind"] != "extreme_mv":
            continue
        for want_max, key in ((1, "max"), (0, "min")):
            e, mv = orc.return_extreme_sub_pixel_mv(c["limits"], c["allow_hp"], want_max)
            assert [e, mv[0], mv[1]] == c[key], c
        odd += (not c["allow_hp"]) and any(v & 1 for v in c["limits"])
        n += 1
    assert n == 80 and odd >= 20
