"""The oracle's compound-reference refinement and OBMC full-pel searches (oracle/aomref_mcomp.c) against the values obtained by interpreting
the reference's av1_refining_search_8p_c / av1_get_mvpred_compound_var / av1_obmc_full_pixel_search themselves
(tests/golden/ref_eval_compound_search.npz, generator tests/golden/gen_ref_eval_compound_search.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_compound_search.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def blocks_of(blk):
    dt = np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")])
    b = np.zeros(1, dt)
    for n, v in zip(dt.names, blk):
        b[n] = v
    return b


def tables(z):
    return dict(mvjcost=z["mvjcost"], mvcost0=z["mvcost0"], mvcost1=z["mvcost1"])


def test_refining_search_8p_matches_reference_evaluation(oracle):
    z, meta = load()
    n = masked = 0
    for c in meta["cases"]:
        if c["kind"] != "refine8p":
            continue
        k = c["k"]
        dt = np.uint8 if c["bd"] == 8 else np.uint16
        mask = z["mask%d" % k][None] if c["masked"] else None
        mv, sad, var = oracle.refining_search_8p_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], blocks_of(c["block"]),
                                                       z["sp%d" % k].astype(dt)[None], mask, c["inv"], cost_type=c["cost_type"],
                                                       sad_per_bit=c["sad_per_bit"], error_per_bit=c["error_per_bit"], bd=c["bd"], threads=1, **tables(z))
        assert (list(map(int, mv[0])), int(sad[0]), int(var[0])) == (c["mv"], c["sad"], c["var"]), c
        n += 1
        masked += c["masked"]
    assert n >= 40 and masked >= 16


def test_obmc_full_pixel_search_matches_reference_evaluation(oracle):
    z, meta = load()
    n = fast = 0
    for c in meta["cases"]:
        if c["kind"] != "obmc":
            continue
        k = c["k"]
        mv, cost = oracle.obmc_full_pixel_search_batch(z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], blocks_of(c["block"]), z["ws%d" % k][None],
                                                       z["om%d" % k][None], c["method"], c["step_param"], c["fast"], cost_type=c["cost_type"],
                                                       sad_per_bit=c["sad_per_bit"], error_per_bit=c["error_per_bit"], bd=c["bd"], threads=1, **tables(z))
        assert (list(map(int, mv[0])), int(cost[0])) == (c["mv"], c["cost"]), c
        n += 1
        fast += c["fast"]
    assert n >= 32 and fast >= 12
