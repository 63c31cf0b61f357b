"""av1_single_motion_search AS IT IS WRITTEN (interpreted: tests/golden/ref_eval_single_caller.npz, generator
tests/golden/gen_ref_eval_single_caller.py) against the oracle's compositions: single_motion_search_batch for SIMPLE_TRANSLATION (two start
candidates and the weight rule, search_range narrowing, one cost list, try_second on one fractional list, force_integer_mv) and
obmc_full_pixel_search_batch + obmc_subpel_tree_batch for OBMC_CAUSAL."""
import json
import os

import numpy as np

from test_golden_joint import BLOCK_DT, TAPS, TREES

HERE = os.path.dirname(os.path.abspath(__file__))
INT_MAX = 2147483647


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_single_caller.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def rawpel(v):
    return (v + 3 + (v >= 0)) >> 3   # get_fullmv_from_mv (mv.h:65-70)


def step_param_of(oracle, c):
    """:229-243: the search_range narrowing of step_param on the method's own site table"""
    step = c["step"]
    sr = c.get("search_range", INT_MAX)
    if sr < INT_MAX:
        ns, _, rad, _ = oracle.search_sites(c["method"])
        if sr < 1:
            step = ns
        else:
            while rad[ns - step - 1] > (sr << 1) and ns - step - 1 > 0:
                step += 1
    return step


def second_candidate(c):
    """:271-290: cand[1] is searched unless the weight rule stops after cand[0]"""
    if c["cand2"] is None:
        return None
    w0, w1 = c["weights"]
    return None if 4 * w0 > 3 * (w0 + w1) else c["cand2"]


def run_case(oracle, z, meta, c):
    bd, w, h = c["bd"], c["w"], c["h"]
    src, ref = z["src%d" % bd], z["ref%d" % bd]
    B = meta["border"]
    tabs = dict(mvjcost=z["mvjcost"], mvcost0=z["mvcost0"], mvcost1=z["mvcost1"])
    b = np.zeros(1, BLOCK_DT)
    b["bx"], b["by"] = c["bx"], c["by"]
    b["ref_row"], b["ref_col"] = c["ref_mv"]
    b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
    step = step_param_of(oracle, c)
    sst = TAPS[c["taps"]] if c["accurate"] else 0
    if c["mode"] == "SIMPLE":
        b["start_row"], b["start_col"] = rawpel(c["ref_mv"][0]), rawpel(c["ref_mv"][1])
        q = oracle.search_params(c["method"], step, 0, c["sadperbit"], c["errorperbit"], 0, 0, 0, 4, c.get("mesh_thr", INT_MAX), 0, meta["mesh"],
                                 no_cost_list=int(not c["costlist"]))
        sub = dict(tree=TREES[c["tree"]], cost_type=0, error_per_bit=c["errorperbit"], iters=2, allow_hp=1, forced_stop=0, subpel_search_type=sst)
        c2 = second_candidate(c)
        out = oracle.single_motion_search_batch(src, ref, B, w, h, b, q, sub, None if c2 is None else np.array([c2], np.int16), use_cost_list=c["costlist"],
                                                try_second_mv=int(bool(c["accurate"])), force_integer_mv=c.get("force_int", 0), bd=bd, threads=1, **tabs)
        return dict(best_mv=out["best_mv"][0].tolist(), rate_mv=int(out["rate_mv"][0]), pred_sse=int(out["pred_sse"][0]))
    # OBMC_CAUSAL (:291-294, :432-436): start = get_fullmv_from_mv(mbmi->mv[0]), obmc full-pel search, av1_find_best_obmc_sub_pixel_tree_up
    b["start_row"], b["start_col"] = rawpel(c["mi_mv"][0]), rawpel(c["mi_mv"][1])
    fl = b.copy()
    fl["row_min"], fl["row_max"], fl["col_min"], fl["col_max"] = oracle.set_mv_search_range(tuple(c["limits"]), *c["ref_mv"])
    ws, om = z["ws%d" % c["k"]][None], z["om%d" % c["k"]][None]
    fmv, _ = oracle.obmc_full_pixel_search_batch(ref, B, w, h, fl, ws, om, c["method"], step, c.get("fast_obmc", 0), cost_type=0, sad_per_bit=c["sadperbit"],
                                                 error_per_bit=c["errorperbit"], bd=bd, threads=1, **tabs)
    sl = b.copy()
    sl["row_min"], sl["row_max"], sl["col_min"], sl["col_max"] = oracle.set_subpel_mv_search_range(tuple(c["limits"]), *c["ref_mv"])
    sl["start_row"], sl["start_col"] = int(fmv[0, 0]) * 8, int(fmv[0, 1]) * 8
    mv, _, _, sse = oracle.obmc_subpel_tree_batch(ref, B, w, h, sl, ws, om, cost_type=0, error_per_bit=c["errorperbit"], iters_per_step=2, allow_hp=1, forced_stop=0,
                                                  subpel_search_type=sst, bd=bd, threads=1, **tabs)
    rate = oracle.mv_bit_cost(int(mv[0, 0]), int(mv[0, 1]), c["ref_mv"][0], c["ref_mv"][1], z["mvjcost"], z["mvcost0"], z["mvcost1"])
    return dict(best_mv=mv[0].tolist(), rate_mv=int(rate), pred_sse=int(sse[0]))


def test_single_motion_search_matches_the_interpreted_caller(oracle):
    z, meta = load()
    n = {}
    for c in meta["cases"]:
        got = run_case(oracle, z, meta, c)
        assert got == {k: c[k] for k in got}, (c, got)
        n[c["mode"]] = n.get(c["mode"], 0) + 1
    assert n.get("SIMPLE", 0) >= 10 and n.get("OBMC", 0) >= 4
    # both outcomes of the weight rule and both outcomes of the narrowing occur
    simple = [c for c in meta["cases"] if c["mode"] == "SIMPLE"]
    assert {second_candidate(c) is None for c in simple if c["cand2"] is not None} == {True, False}
    assert any(step_param_of(oracle, c) != c["step"] for c in simple)
