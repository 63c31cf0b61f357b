#!/usr/bin/env python3
"""Golden vectors for the pieces av1_single_motion_search (av1/encoder/motion_search_facade.c:120-495) adds on top of the plain full-pel /
sub-pel searches, obtained by interpreting the reference's own functions (tests/golden/ref_c_eval.py, harness of gen_ref_eval_mcomp.py):

  * the three sub-pel trees called TWICE on one `last_mv_search_list` (av1_set_fractional_mv + check_repeated_mv_and_update,
    mcomp.c:2816-2828): first from the full-pel winner, then from a second start -- av1_full_pixel_search's second_best_mv, the winner
    itself (repeats at once), or a neighbour of it (repeats at a later iteration or not at all) -- recording what the second call
    returns and leaves behind when it stops with INT_MAX;
  * av1_mv_bit_cost (mcomp.c:261-266) with MV_COST_WEIGHT for the resulting MVs.

`int_mv` is a union in the reference (av1/common/mv.h:36-41); the evaluator sees it as an opaque type with the one member these functions
touch, `as_mv` (same storage); av1_set_fractional_mv's store through `as_int` is written as the two halves.  Output: tests/golden/ref_eval_single.npz (planes and cost tables are those of ref_eval_mcomp.npz, regenerated
from the same seeds, and stored again so the file stands alone)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_ref_eval_mcomp as G  # noqa: E402

R = G.R
INVALID = -32768
TREES = ("av1_find_best_sub_pixel_tree_pruned_more", "av1_find_best_sub_pixel_tree_pruned", "av1_find_best_sub_pixel_tree")


def main():
    ev = G.make_evaluator()
    imv = ev.structs["<opaque>int_mv"]
    imv.fields = [("as_mv", ev.structs["mv"])]
    arrays, cases = {}, []
    mvc = G.synth_mv_costs(7)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    harness = {}
    for bd in (8, 10):
        s, r = G.synth_planes(bd, 100 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s, r
        harness[bd] = G.Harness(ev, bd, s, r, mvc)
    rng = np.random.default_rng(20261101)
    t0 = time.time()
    for bd in (8, 10):
        hs = harness[bd]
        for fn in TREES:
            for cost_type in ("ENTROPY", "L1_HDRES"):
                for trial in range(8 if bd == 8 else 3):
                    w, h = (16, 16) if trial % 2 == 0 else (8, 8)
                    bx, by = int(rng.integers(0, (G.W - w) // 4 + 1)) * 4, int(rng.integers(0, (G.H - h) // 4 + 1)) * 4
                    refmv = (int(rng.integers(-30, 31)), int(rng.integers(-30, 31)))
                    start = ((refmv[0] + 3 + (refmv[0] >= 0)) >> 3, (refmv[1] + 3 + (refmv[1] >= 0)) >> 3)
                    blk = (bx, by, start[0], start[1], refmv[0], refmv[1]) + G.limits(bx, by, w, h, 24)
                    # av1_full_pixel_search with second_best_mv
                    ms = hs.fullpel_params(blk, w, h, "NSTEP", cost_type, sad_per_bit=25, error_per_bit=70)
                    st = hs.mv_struct("FULLPEL_MV", blk[2], blk[3])
                    best, second = ev.new("FULLPEL_MV"), ev.new("FULLPEL_MV")
                    cl = ev.array([0] * 5, "int")
                    step_param = int(rng.integers(1, 5))
                    cost = ev.call("av1_full_pixel_search", st.buf[0], ms, step_param, cl, best, second)
                    fmv = [ev.get(best, "row"), ev.get(best, "col")]
                    try:
                        smv = [ev.get(second, "row"), ev.get(second, "col")]
                    except R.CError:
                        smv = [INVALID, INVALID]
                    allow_hp, forced_stop, iters = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.integers(1, 3))
                    sp = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
                    ev.set(sp, "allow_hp", allow_hp); ev.set(sp, "forced_stop", forced_stop); ev.set(sp, "iters_per_step", iters)
                    use_cl = trial >= 2 and not fn.endswith("tree")
                    if use_cl:
                        ev.set(sp, "cost_list", cl)
                    fl = ev.new("FullMvLimits")
                    for k, v in zip(("row_min", "row_max", "col_min", "col_max"), blk[6:]):
                        ev.set(fl, k, v)
                    rm = hs.mv_struct("MV", blk[4], blk[5])
                    ev.interp.call("av1_set_subpel_mv_search_range", [(ev.field(sp, "mv_limits"), R.PTR), (fl, R.PTR), (rm, R.PTR)])
                    hs.cost_params(sp, "mv_cost_params.", cost_type, blk[4], blk[5], 25, 70)
                    ev.set(sp, "var_params.vfp", hs.vtable(w, h))
                    ev.set(sp, "var_params.subpel_search_type", hs.const("USE_2_TAPS_ORIG"))
                    ev.set(sp, "var_params.ms_buffers.ref", hs.buf2d(hs.refp, by, bx)); ev.set(sp, "var_params.ms_buffers.src", hs.buf2d(hs.srcp, by, bx))
                    ev.set(sp, "var_params.w", w); ev.set(sp, "var_params.h", h)
                    lim = [ev.get(sp, "mv_limits." + k) for k in ("row_min", "row_max", "col_min", "col_max")]
                    # the second start: second_best_mv when there is one, else the winner itself / a neighbour of it
                    kind2 = ("second_best", "same", "neighbour")[min(trial, 2)] if smv[0] != INVALID else ("same", "neighbour")[min(trial, 1)]
                    if kind2 == "second_best":
                        s2 = smv
                    elif kind2 == "same":
                        s2 = fmv
                    else:
                        d = [(0, 1), (1, 0), (-1, 0), (0, -1), (1, 1), (-1, -1)][trial % 6]
                        s2 = [fmv[0] + d[0], fmv[1] + d[1]]
                    s2 = [min(max(s2[0], lim[0] >> 3), lim[1] >> 3), min(max(s2[1], lim[2] >> 3), lim[3] >> 3)]
                    lst = ev.interp.alloc(("arr", imv, 3), True)
                    for k in range(3):      # av1_set_fractional_mv (mcomp.h:338-342): as_int = INVALID_MV = 0x80008000, i.e. row = col = -32768
                        ev.set(lst, "[%d].as_mv.row" % k, INVALID); ev.set(lst, "[%d].as_mv.col" % k, INVALID)
                    calls = []
                    # calls 3 and 4 repeat the first search with the list's first one / two entries reset: the same trajectory now passes
                    # iteration 0 (and 1) and is stopped at the next one it has -- the mid-way exits of every tree
                    for ci, start_full in enumerate((fmv, s2, fmv, fmv)):
                        for k in range(ci - 1 if ci >= 2 else 0):
                            ev.set(lst, "[%d].as_mv.row" % k, INVALID); ev.set(lst, "[%d].as_mv.col" % k, INVALID)
                        before = [[ev.get(lst, "[%d].as_mv.row" % k), ev.get(lst, "[%d].as_mv.col" % k)] for k in range(3)]
                        stm = hs.mv_struct("MV", start_full[0] * 8, start_full[1] * 8)
                        bestm = ev.new("MV")
                        ev.set(bestm, "row", 12345); ev.set(bestm, "col", 12345)
                        dist, sse = ev.array([0], "int"), ev.array([0], "unsigned int")
                        err = ev.call(fn, G.make_xd(ev, bd), None, sp, stm.buf[0], bestm, dist, sse, lst.deref()[0])   # the array decays to int_mv *
                        after = [[ev.get(lst, "[%d].as_mv.row" % k), ev.get(lst, "[%d].as_mv.col" % k)] for k in range(3)]
                        calls.append(dict(start=[start_full[0] * 8, start_full[1] * 8], err=err, mv=[ev.get(bestm, "row"), ev.get(bestm, "col")],
                                          distortion=dist.buf[0], sse=sse.buf[0], list_before=before, list_after=after))
                    rate = []
                    for c in calls:                                                 # av1_mv_bit_cost(.., MV_COST_WEIGHT)
                        m = hs.mv_struct("MV", c["mv"][0], c["mv"][1])
                        mvcost = ev.interp.alloc(("arr", ("ptr", R.I32), 2), True)
                        ev.set(mvcost, "[0]", hs.comp[0].add(hs.mv_max)); ev.set(mvcost, "[1]", hs.comp[1].add(hs.mv_max))
                        rate.append(ev.call("av1_mv_bit_cost", m, rm, hs.joint, mvcost.deref()[0], 108))
                    cases.append(dict(fn=fn, bd=bd, w=w, h=h, block=list(blk), step_param=step_param, cost_type=G.COST_TYPES[cost_type], sad_per_bit=25,
                                      error_per_bit=70, full_mv=fmv, full_cost=cost, second_best=smv, cost_list=list(cl.buf), use_cost_list=int(use_cl),
                                      allow_hp=allow_hp, forced_stop=forced_stop, iters=iters, subpel_limits=lim, second_start=kind2, calls=calls,
                                      mv_bit_cost=rate))
                    print(fn[-12:], bd, cost_type, trial, kind2, [c["err"] for c in calls], "%.0f s" % (time.time() - t0), flush=True)
    arrays["cases"] = np.frombuffer(json.dumps(dict(W=G.W, H=G.H, border=G.BORDER, cases=cases)).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_single.npz"), **arrays)
    n_max = [sum(c["calls"][k]["err"] == 2147483647 for c in cases) for k in range(4)]
    print("%d cases, calls stopped by the list (per call index): %s, %.0f s" % (len(cases), n_max, time.time() - t0))


if __name__ == "__main__":
    main()
