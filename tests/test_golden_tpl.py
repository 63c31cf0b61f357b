"""The selection glue of the inter leg of TPL's mode_estimation in the oracle (tpl_prune / tpl_best_of / tpl_best_ref, oracle/pyoracle.py) against the
reference's own statements interpreted (av1/encoder/tpl_model.c:706-765; tests/golden/gen_ref_eval_tpl.py -> ref_eval_tpl.npz)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_tpl.npz"))
    return json.loads(bytes(z["cases"]).decode())


def test_prune_ranking_and_cuts(oracle):
    cs = cases()["prune"]
    assert len(cs) == 160
    ties = cuts = 0
    for c in cs:
        assert oracle.tpl_prune(c["sads"], c["prune"]) == c["order"], c
        ties += len(set(c["sads"])) < len(c["sads"])
        cuts += len(c["order"]) < min(4 - c["prune"], len(c["sads"]))
    assert ties > 20 and cuts > 10          # equal SADs and the 20 % rule both occur


def test_best_candidate_and_best_reference(oracle):
    cs = cases()
    for c in cs["best_of"]:
        k = oracle.tpl_best_of(c["errs"])
        assert ([c["rows"][k], c["cols"][k]] if k is not None else [0, 0]) == c["best"], c
    assert any(oracle.tpl_best_of(c["errs"]) is None for c in cs["best_of"])
    for c in cs["best_ref"]:
        rf, cost, pe = oracle.tpl_best_ref(c["costs"], c["have"])
        assert rf == c["best_rf"] and cost == c["best_cost"], c
        assert [p if p is not None else -7 for p in pe] == c["pred_error"], c      # untouched entries keep the fixture's fill value
        if rf >= 0:
            assert c["best_mv_row"] == 100 + rf
    assert any(c["best_rf"] == -1 for c in cs["best_ref"]) and any(min(c["costs"]) == 0 for c in cs["best_ref"])
