"""The oracle's av1_full_pixel_search on a compound prediction (oracle/aomref_mcomp.c: orc_compound_full_pixel_search_batch) against the values
obtained by interpreting the reference's av1_full_pixel_search itself with ms_buffers.second_pred / mask set
(tests/golden/ref_eval_compound_fullpel.npz, generator tests/golden/gen_ref_eval_compound_fullpel.py): the compound diamond runs, the plain
mesh passes that follow them, the second-best MV."""
import json
import os

import numpy as np

from test_golden_compound_search import blocks_of

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_compound_fullpel.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def params_of(oracle, c):
    return oracle.search_params(c["method"], c["step_param"], c["cost_type"], c["sad_per_bit"], c["error_per_bit"], False, c.get("run_mesh", 0),
                                c.get("prune_mesh", 0), c.get("mesh_diff_thr", 0), c.get("force_mesh_thresh", 2147483647), 0, c.get("mesh"),
                                no_cost_list=1)


def test_compound_full_pixel_search_matches_reference_evaluation(oracle):
    z, meta = load()
    n = masked = meshed = differs = 0
    for c in meta["cases"]:
        k = c["k"]
        dt = np.uint8 if c["bd"] == 8 else np.uint16
        mask = z["mask%d" % k][None] if c["masked"] else None
        q = params_of(oracle, c)
        src, ref = z["src%d" % c["bd"]], z["ref%d" % c["bd"]]
        mv, cost, sec = oracle.compound_full_pixel_search_batch(src, ref, meta["border"], c["w"], c["h"], blocks_of(c["block"]), q,
                                                                z["sp%d" % k].astype(dt)[None], mask, c["inv"], z["mvjcost"], z["mvcost0"], z["mvcost1"],
                                                                bd=c["bd"], threads=1)
        got = (list(map(int, mv[0])), int(cost[0]), list(map(int, sec[0])))
        assert got == (c["mv"], c["cost"], c["second_best"]), (c, got)
        # the single-reference search of the same block is a different search (the fixture would not notice a dropped second_pred otherwise)
        mv1, cost1, _, _ = oracle.full_pixel_search_batch(src, ref, meta["border"], c["w"], c["h"], blocks_of(c["block"]), q, z["mvjcost"], z["mvcost0"],
                                                          z["mvcost1"], bd=c["bd"], threads=1)
        differs += int(list(map(int, mv1[0])) != c["mv"] or int(cost1[0]) != c["cost"])
        n += 1
        masked += c["masked"]
        meshed += int("mesh" in c)
    assert n >= 30 and masked >= 18 and meshed >= 8 and differs >= 18
