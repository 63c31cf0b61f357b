"""The SEQUENCING of the compound searches: the oracle's compositions (oracle/pyoracle.py joint_motion_search_batch /
compound_single_motion_search_batch) against av1_joint_motion_search (both branches) and av1_compound_single_motion_search interpreted AS THEY ARE
WRITTEN (tests/golden/ref_eval_joint.npz, generator tests/golden/gen_ref_eval_joint.py): iteration loop and early-outs, the limits of the
ms-params builders, try_second, the update rule, rate_mv."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TREES = {"SUBPEL_TREE": 2, "SUBPEL_TREE_PRUNED": 1, "SUBPEL_TREE_PRUNED_MORE": 0}
TAPS = {"USE_2_TAPS_ORIG": 0, "USE_2_TAPS": 1, "USE_4_TAPS": 2, "USE_8_TAPS": 3}
BLOCK_DT = np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")])


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_joint.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def block_of(c):
    b = np.zeros(1, BLOCK_DT)
    b["bx"], b["by"] = c["bx"], c["by"]
    b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]    # the RAW x->mv_limits
    return b


def params_of(oracle, c, mesh):
    """what av1_make_default_fullpel_ms_params / _subpel_ms_params derive from the speed features of the case (mcomp.c:95-195)"""
    full = oracle.search_params("NSTEP", 5, 0, c["sadperbit"], c["errorperbit"], 0, 0, 0, 4, c.get("mesh_thr", 2147483647), 0, mesh, no_cost_list=1)
    sub = dict(tree=TREES[c["tree"]], subpel_search_type=TAPS[c["taps"]], error_per_bit=c["errorperbit"], iters_per_step=2, allow_hp=1)
    return full, sub


def run_case(oracle, z, meta, c):
    bd = c["bd"]
    full, sub = params_of(oracle, c, meta["mesh"])
    src, refs = z["src%d" % bd], (z["ref0_%d" % bd], z["ref1_%d" % bd])
    tabs = dict(mvjcost=z["mvjcost"], mvcost0=z["mvcost0"], mvcost1=z["mvcost1"])
    mask = z["mask%d" % c["k"]][None] if c["masked"] else None
    B, W, H = meta["border"], meta["width"], meta["height"]
    if c["fn"] == "joint":
        mv, rate, err, _ = oracle.joint_motion_search_batch(src, refs[0], refs[1], B, W, H, c["w"], c["h"], block_of(c), [c["ref_mv"]], [c["cur_in"]], mask,
                                                            cost_type=0, sad_per_bit=c["sadperbit"], sub=sub, force_integer_mv=c.get("force_int", 0), bd=bd,
                                                            threads=1, full=full if c["ext"] else None, allow_second_mv=c["second"], **tabs)
        return dict(cur_out=mv[0].tolist(), rate_mv=int(rate[0]), err=int(err[0]))
    ri = c["ref_idx"]
    dt = np.uint8 if bd == 8 else np.uint16
    mv, rate, sme = oracle.compound_single_motion_search_batch(src, refs[ri], B, W, H, c["w"], c["h"], block_of(c), [c["ref_mv"][ri]], [c["cur_in"][ri]], full, sub,
                                                               second_pred=z["sp%d" % c["k"]].astype(dt)[None], mask=mask, ref_idx=ri,
                                                               force_integer_mv=c.get("force_int", 0), bd=bd, threads=1, **tabs)
    return dict(this_out=mv[0].tolist(), rate_mv=int(rate[0]), err=int(sme[0]))


def test_compound_search_sequencing_matches_the_interpreted_callers(oracle):
    z, meta = load()
    kinds = {}
    for c in meta["cases"]:
        got = run_case(oracle, z, meta, c)
        want = {k: c[k] for k in got}
        assert got == want, (c, got)
        key = (c["fn"], c.get("ext", -1))
        kinds[key] = kinds.get(key, 0) + 1
    assert kinds.get(("joint", 0), 0) >= 4 and kinds.get(("joint", 1), 0) >= 8 and kinds.get(("single", -1), 0) >= 6
    # the early-outs and the loop are exercised: the refining cases use between one and four predictor builds
    npred = {len(c["predictors"]) for c in meta["cases"] if c["fn"] == "joint"}
    assert len(npred) >= 2, npred
