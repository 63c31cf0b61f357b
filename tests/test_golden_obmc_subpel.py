"""The oracle's OBMC sub-pel search (oracle/aomref_mcomp.c orc_obmc_subpel_tree_batch) against the values obtained by interpreting the reference's
av1_find_best_obmc_sub_pixel_tree_up itself (tests/golden/ref_eval_obmc_subpel.npz, generator tests/golden/gen_ref_eval_obmc_subpel.py): both error
forms -- USE_2_TAPS_ORIG (centre at ref->buf, osvf + estimate_obmc_mvcost) and USE_8_TAPS (up-sampled prediction + ovf + mv_err_cost_)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name="ref_eval_obmc_subpel.npz"):
    z = np.load(os.path.join(HERE, "golden", name))
    return z, json.loads(bytes(z["meta"]).decode())


def subpel_block(c):
    """the record of the sub-pel entry points: start MV and limits in 1/8 pel (SubpelMvLimits as the reference computed them)"""
    dt = np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")])
    b = np.zeros(1, dt)
    blk, lim = c["block"], c["subpel_limits"]
    for n, v in zip(dt.names, (blk[0], blk[1], blk[2] * 8, blk[3] * 8, blk[4], blk[5], lim[0], lim[1], lim[2], lim[3])):
        b[n] = v
    return b


def test_obmc_subpel_tree_matches_reference_evaluation(oracle):
    z, meta = load()
    n = up = moved = 0
    for c in meta["cases"]:
        k = c["k"]
        mv, err, dist, sse = oracle.obmc_subpel_tree_batch(z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], subpel_block(c), z["ws%d" % k][None], z["om%d" % k][None],
                                                           cost_type=c["cost_type"], error_per_bit=c["error_per_bit"], mvjcost=z["mvjcost"], mvcost0=z["mvcost0"],
                                                           mvcost1=z["mvcost1"], iters_per_step=c["iters"], allow_hp=c["allow_hp"], forced_stop=c["forced_stop"],
                                                           subpel_search_type=c["subpel_search_type"], bd=c["bd"], threads=1)
        assert (list(map(int, mv[0])), int(err[0]), int(dist[0]), int(sse[0])) == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        n += 1
        up += c["subpel_search_type"] == 3
        moved += c["mv"] != [c["block"][2] * 8, c["block"][3] * 8]
    assert n >= 50 and up >= 24 and moved >= n // 2


def test_obmc_subpel_tree_with_2_and_4_taps_matches_reference_evaluation(oracle):
    """USE_4_TAPS / USE_2_TAPS in upsampled_obmc_pref_error: tests/golden/ref_eval_obmc_subpel_taps.npz (generator gen_ref_eval_subpel_taps.py)."""
    z, meta = load("ref_eval_obmc_subpel_taps.npz")
    n = {1: 0, 2: 0}
    for c in meta["cases"]:
        k = c["k"]
        mv, err, dist, sse = oracle.obmc_subpel_tree_batch(z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], subpel_block(c), z["ws%d" % k][None], z["om%d" % k][None],
                                                           cost_type=c["cost_type"], error_per_bit=c["error_per_bit"], mvjcost=z["mvjcost"], mvcost0=z["mvcost0"],
                                                           mvcost1=z["mvcost1"], iters_per_step=c["iters"], allow_hp=c["allow_hp"], forced_stop=c["forced_stop"],
                                                           subpel_search_type=c["subpel_search_type"], bd=c["bd"], threads=1)
        assert (list(map(int, mv[0])), int(err[0]), int(dist[0]), int(sse[0])) == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        n[c["subpel_search_type"]] += 1
    assert n[1] >= 12 and n[2] >= 12
