"""TPL's mode_estimation AS IT IS WRITTEN, run for whole rows of blocks in raster order with tpl_model_store between them (interpreted:
tests/golden/ref_eval_tpl_mode.npz, generator tests/golden/gen_ref_eval_tpl_mode.py -- intra costs given, get_rate_distortion scripted), against the
oracle's composition tpl_mode_estimation_rows: candidate gathering from the neighbours' stored stats (is_alike_mv), the prune_starting_mv block,
motion_estimation per surviving candidate in the pruned order, predictor + DCT SATD per reference, best reference, the mode decision, and what the
function hands get_rate_distortion."""
import json
import os

import numpy as np

from test_golden_joint import TAPS, TREES

HERE = os.path.dirname(os.path.abspath(__file__))
INT_MAX = 2147483647
NEWMV = 16


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_tpl_mode.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def frame_inputs(oracle, z, meta, fi):
    f = meta["frames"][fi]
    cfg = f["config"]
    step = min(cfg["reduce_first_step_size"], 9)        # AOMMIN(step_param, MAX_MVSEARCH_STEPS - 2) (tpl_model.c:264-265)
    q = oracle.search_params(cfg["search_method"], step, 0, f["sadperbit"], f["errorperbit"], 0, 0, 0, 4, INT_MAX, 0, meta["mesh"],
                             no_cost_list=int(not cfg["use_fullpel_costlist"]))
    sub = dict(tree=TREES[cfg["subpel_search_method"]], cost_type=4, error_per_bit=f["errorperbit"], iters=2, allow_hp=1, forced_stop=cfg["subpel_force_stop"],
               subpel_search_type=TAPS["USE_2_TAPS"])
    positions = [(b["mi_col"] * 4, b["mi_row"] * 4) for b in f["blocks"]]
    return f, cfg, q, sub, positions


def searched_centres(b, n_refs):
    """per reference (in search order) the centre MVs motion_estimation was called with, from the block's log"""
    out, cur = [], []
    for e in b["log"]:
        if e[0] == "me":
            cur.append((e[1], e[2]))
        elif e[0] == "pred":
            out.append(cur)
            cur = []
    assert len(out) == n_refs
    return out


def test_rows_of_blocks_match_the_interpreted_function(oracle):
    z, meta = load()
    refs = meta["refs"]
    seen = dict(newmv=0, intra=0, second_ref=0, pruned=0, alike=0, multi=0)
    for fi in range(len(meta["frames"])):
        f, cfg, q, sub, positions = frame_inputs(oracle, z, meta, fi)
        bd = cfg["bd"]
        got = oracle.tpl_mode_estimation_rows(z["src_%d" % fi], [z["ref0_%d" % fi], z["ref1_%d" % fi]], meta["border"], meta["width"], meta["height"], meta["bs"],
                                              positions, [b["limits"] for b in f["blocks"]], [b["intra_costs"] for b in f["blocks"]], q, sub,
                                              use_cost_list=cfg["use_fullpel_costlist"], prune_starting_mv=cfg["prune_starting_mv"],
                                              skip_alike_starting_mv=cfg["skip_alike_starting_mv"], mvjcost=z["mvjcost"], mvcost0=z["mvcost0"], mvcost1=z["mvcost1"],
                                              bd=bd, threads=1)
        for b, g in zip(f["blocks"], got):
            s, where = b["stats"], (fi, b["mi_row"], b["mi_col"])
            assert b["intra_calls"] == (3 if cfg["prune_intra_modes"] else 13)
            for k, r in enumerate(refs):
                assert g["mv"][k].tolist() == s["mv"][r] and int(g["pred_error"][k]) == s["pred_error"][r], (where, r, g["mv"][k], s["mv"][r], g["pred_error"][k])
            for r in range(7):
                if r not in refs:
                    assert s["mv"][r] == [-32768, -32768] and s["pred_error"][r] == 0          # a reference that does not exist (:634-637)
            rfi = [refs[g["ref_frame_index"][0]] if g["ref_frame_index"][0] >= 0 else -1, -1]     # (the composition numbers the references it was given)
            assert (g["intra_cost"], g["inter_cost"], rfi) == (s["intra_cost"], s["inter_cost"], s["ref_frame_index"]), (where, g, s)
            # the candidates: all of them searched in gathering order without pruning, at most 4 - prune_starting_mv of them (a subset) with it
            centres = searched_centres(b, len(refs))
            for k in range(len(refs)):
                if cfg["prune_starting_mv"] == 0:
                    assert centres[k] == g["candidates"][k], (where, k, centres[k], g["candidates"][k])
                else:
                    assert set(centres[k]) <= set(g["candidates"][k]) and 1 <= len(centres[k]) <= 4 - cfg["prune_starting_mv"]
                    seen["pruned"] += len(centres[k]) < len(g["candidates"][k])
                seen["multi"] += len(centres[k]) > 1
            # what the function hands get_rate_distortion: first the source reference of the best single reference with the decided mode (:896-910),
            # then the reconstructed one for the final encode (:922-938; the best inter reference even when an intra mode won)
            rd = [e for e in b["log"] if e[0] == "rd"]
            assert [e[1] for e in rd] == [g["best_mode"]] * 2 and [e[2] for e in rd] == ["src%d" % refs[g["best_rf"]], "rec%d" % refs[g["best_rf"]]], (where, rd, g)
            assert s["srcrf_rate"] == rd[0][4] if g["best_mode"] == NEWMV else s["srcrf_rate"] == rd[1][4]       # (:911, :944-948)
            assert s["recrf_rate"] == max(s["srcrf_rate"], rd[1][4]) and s["recrf_dist"] == max(s["srcrf_dist"], rd[1][5] << 4)
            assert b["stored"]["mv"] == s["mv"] and b["stored"]["inter_cost"] == max(1, s["inter_cost"])
            if g["best_mode"] == NEWMV:
                assert b["mi_ref_frame"][0] == refs[g["best_rf"]] + 1 and b["mi_mv"] == s["mv"][refs[g["best_rf"]]]
                seen["newmv"] += 1
                seen["second_ref"] += g["best_rf"] == 1
            else:
                seen["intra"] += 1
            n_nb = (b["mi_row"] > 0) + (b["mi_col"] > 0) + (b["mi_row"] > 0 and b["mi_col"] + 4 < meta["width"] // 4)
            seen["alike"] += any(len(c) < 1 + n_nb for c in g["candidates"])
    assert seen["newmv"] >= 20 and seen["intra"] >= 10 and seen["second_ref"] >= 4 and seen["pruned"] >= 4 and seen["alike"] >= 10 and seen["multi"] >= 20, seen


def test_candidate_gathering_rules(oracle):
    g = oracle.tpl_gather_candidates
    assert g(None, None, None, 0) == [(0, 0)]
    assert g((0, 0), (3, -2), (3, -2), 0) == [(0, 0), (3, -2)]                          # threshold 1: only equal MVs are alike
    assert g((63, 63), (64, 0), (-64, 200), 1) == [(0, 0), (64, 0), (-64, 200)]        # 8 << 3: both components must be closer than 64
    assert g((100, 127), (128, 0), (227, 128), 2) == [(0, 0), (128, 0), (227, 128)]    # 16 << 3
    d = oracle.tpl_mode_decision
    assert d([500, 400, 400], 0, 400, None)["best_mode"] == 1                          # first smallest intra cost; inter must be SMALLER
    assert d([500, 400, 400], 0, 399, None) == dict(best_mode=16, intra_cost=400, inter_cost=399, ref_frame_index=[0, -1])
    assert d([0, 5], -1, 2147483647, None) == dict(best_mode=0, intra_cost=1, inter_cost=1, ref_frame_index=[-1, -1])
