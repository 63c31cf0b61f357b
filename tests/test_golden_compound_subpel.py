"""The oracle's sub-pel trees on a compound prediction (oracle/aomref_mcomp.c orc_compound_subpel_tree_batch) against the values obtained by
interpreting the reference's av1_find_best_sub_pixel_tree{_pruned_more,_pruned,} themselves with ms_buffers.second_pred [/ mask / inv_mask]
(tests/golden/ref_eval_compound_subpel.npz, generator tests/golden/gen_ref_eval_compound_subpel.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_compound_subpel.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def subpel_block(c):
    dt = np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")])
    b = np.zeros(1, dt)
    blk, lim = c["block"], c["subpel_limits"]
    for n, v in zip(dt.names, (blk[0], blk[1], blk[2] * 8, blk[3] * 8, blk[4], blk[5], lim[0], lim[1], lim[2], lim[3])):
        b[n] = v
    return b


def test_compound_subpel_trees_match_reference_evaluation(oracle):
    z, meta = load()
    n = up = masked = 0
    for c in meta["cases"]:
        k = c["k"]
        dt = np.uint8 if c["bd"] == 8 else np.uint16
        mask = z["mask%d" % k][None] if c["masked"] else None
        mv, err, dist, sse = oracle.compound_subpel_tree_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], subpel_block(c),
                                                               z["sp%d" % k].astype(dt)[None], mask, c["inv"], tree=c["tree"],
                                                               subpel_search_type=c["subpel_search_type"], cost_type=c["cost_type"],
                                                               error_per_bit=c["error_per_bit"], mvjcost=z["mvjcost"], mvcost0=z["mvcost0"], mvcost1=z["mvcost1"],
                                                               iters_per_step=c["iters"], allow_hp=c["allow_hp"], forced_stop=c["forced_stop"], bd=c["bd"], threads=1)
        assert (list(map(int, mv[0])), int(err[0]), int(dist[0]), int(sse[0])) == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        n += 1
        up += c["subpel_search_type"] == 3
        masked += c["masked"]
    assert n >= 60 and up >= 12 and masked >= 30


def test_tree_with_2_and_4_tap_upsampled_error_matches_reference_evaluation(oracle):
    """av1_find_best_sub_pixel_tree with USE_4_TAPS (speed 1 - 2) and USE_2_TAPS: tests/golden/ref_eval_subpel_taps.npz
    (generator tests/golden/gen_ref_eval_subpel_taps.py) -- single-reference, averaged and masked compounds."""
    z = np.load(os.path.join(HERE, "golden", "ref_eval_subpel_taps.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    n = {}
    for c in meta["cases"]:
        k = c["k"]
        dt = np.uint8 if c["bd"] == 8 else np.uint16
        kw = dict(subpel_search_type=c["subpel_search_type"], cost_type=c["cost_type"], error_per_bit=c["error_per_bit"], mvjcost=z["mvjcost"],
                  mvcost0=z["mvcost0"], mvcost1=z["mvcost1"], allow_hp=c["allow_hp"], forced_stop=c["forced_stop"], bd=c["bd"], threads=1)
        if c["compound"]:
            mask = z["mask%d" % k][None] if c["masked"] else None
            mv, err, dist, sse = oracle.compound_subpel_tree_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], subpel_block(c),
                                                                   z["sp%d" % k].astype(dt)[None], mask, c["inv"], tree=c["tree"], iters_per_step=c["iters"], **kw)
        else:
            mv, err, dist, sse = oracle.subpel_tree_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], subpel_block(c),
                                                          tree=c["tree"], iters=c["iters"], **kw)
        assert (list(map(int, mv[0])), int(err[0]), int(dist[0]), int(sse[0])) == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        key = (c["subpel_search_type"], c["compound"], c["masked"])
        n[key] = n.get(key, 0) + 1
    assert len(n) == 6 and min(n.values()) >= 4 and sum(n.values()) >= 32
