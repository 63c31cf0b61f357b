"""av1_refine_warped_mv AS IT IS WRITTEN (interpreted with compute_motion_cost, av1_selectSamples, av1_find_projection, the vtable's variance and
mv_err_cost_: tests/golden/ref_eval_refine_warped.npz, generator tests/golden/gen_ref_eval_refine_warped.py) against the oracle's statement-by-statement
composition (oracle.refine_warped_mv) -- the final MV, model, num_proj_ref and cost, and the model of every predictor the reference asked for, in order."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
COST = {"ENTROPY": 0, "L1_LOWRES": 1, "L1_MIDRES": 2, "L1_HDRES": 3, "NONE": 4}


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_refine_warped.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def block_of(c, dtype):
    b = np.zeros(1, dtype)
    b["bx"], b["by"], b["mv_row"], b["mv_col"], b["ref_row"], b["ref_col"] = c["bx"], c["by"], c["mv"][0], c["mv"][1], c["ref_mv"][0], c["ref_mv"][1]
    b["col_min"], b["col_max"], b["row_min"], b["row_max"] = c["limits"]
    b["total_samples"], b["num_proj_ref"] = c["total_samples"], c["num_proj_ref"]
    n = c["total_samples"]
    b["pts"][0, :2 * n], b["pts_inref"][0, :2 * n] = c["pts"], c["pts_inref"]
    b["model"]["mat"][0] = c["start_model"]["mat"]
    for f, v in zip(("alpha", "beta", "gamma", "delta"), c["start_model"]["shear"]):
        b["model"][f][0] = v
    return b


def test_oracle_composition_equals_the_interpreted_function(oracle):
    import importlib
    capi = importlib.import_module("aom-av1-psy_amd.capi")
    z, meta = load()
    B, W, H = meta["border"], meta["width"], meta["height"]
    moved = cut_short = 0
    for c in meta["cases"]:
        bd = c["bd"]
        src, ref = z["src%d" % bd][B:B + H, B:B + W], z["ref%d" % bd][B:B + H, B:B + W]
        b = block_of(c, capi.warp_refine_block_dtype)[0]
        models = []

        def pred_fn(mat, shear):
            models.append((np.asarray(mat).tolist(), np.asarray(shear).tolist()))
            return oracle.warp_block_pred(ref, bd, mat, shear, c["bx"], c["by"], c["w"], c["h"])
        got = oracle.refine_warped_mv(src, ref, bd, c["w"], c["h"], b, c["allow_hp"], COST[c["cost_type"]], c["error_per_bit"], z["mvjcost"], z["mvcost0"], z["mvcost1"],
                                      pred_fn=pred_fn)
        assert (got["mv"], got["bestmse"], got["num_proj_ref"]) == (c["best_mv"], c["bestmse"], c["best_num_proj_ref"]), (c["k"], got, c["best_mv"], c["bestmse"])
        assert (got["mat"], got["shear"]) == (c["best_model"]["mat"], c["best_model"]["shear"]), c["k"]
        assert models == [(k_["mat"], k_["shear"]) for k_ in c["calls"]], c["k"]
        # the predictor is built while mbmi->mv[0] still holds the CENTRE of the round (only wm_params changes per candidate)
        assert all(k_["mv_in_mbmi"] in ([c["mv"]] + [list(m) for m in got["measured"]]) for k_ in c["calls"])
        moved += got["mv"] != c["mv"]
        cut_short += len(c["calls"]) < 9
    assert moved >= 15 and cut_short >= 4 and len(meta["cases"]) >= 20
