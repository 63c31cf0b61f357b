#!/usr/bin/env python3
"""Golden vectors for the sub-pel search of a COMPOUND prediction, obtained by interpreting av1/encoder/mcomp.c itself (build container only;
tests/golden/ref_c_eval.py, harness of gen_ref_eval_mcomp.py):

  av1_find_best_sub_pixel_tree_pruned_more / _pruned / av1_find_best_sub_pixel_tree (mcomp.c:2844-3133) with var_params.ms_buffers.second_pred
  [/ mask, mask_stride, inv_mask] set as av1_set_ms_compound_refs does (mcomp.h:152-166) -- the find_fractional_mv_step call of
  av1_joint_motion_search / av1_compound_single_motion_search (motion_search_facade.c:496-870): every error is vfp->svaf or vfp->msvf
  (estimated_pref_error, :2311-2337) or, for the tree with USE_8_TAPS, aom_[highbd_]comp_avg_upsampled_pred / comp_mask_upsampled_pred + vf
  (upsampled_pref_error, :2339-2428).

The vtable members are the reference's own functions (svaf = aom_[highbd_10_]sub_pixel_avg_variance{W}x{H}_c, msvf =
aom_[highbd_10_]masked_sub_pixel_variance{W}x{H}_c).  Output: tests/golden/ref_eval_compound_subpel.npz.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as M  # noqa: E402

TREES = ["av1_find_best_sub_pixel_tree_pruned_more", "av1_find_best_sub_pixel_tree_pruned", "av1_find_best_sub_pixel_tree"]


def main(plan=None, out="ref_eval_compound_subpel.npz", seed=20261301):
    """plan: (bd, w, h, tree, subpel_search_type, masked[, compound = 1]) per case; the default is what produced ref_eval_compound_subpel.npz
    (gen_ref_eval_subpel_taps.py passes its own: the USE_2_TAPS / USE_4_TAPS forms, single-reference and compound)."""
    ev = M.make_evaluator(with_compound=True)
    arrays, cases = {}, []
    rng = np.random.default_rng(seed)
    mvc = M.synth_mv_costs(17)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    harness = {}
    for bd in (8, 10):
        s, r = M.synth_planes(bd, 700 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s, r
        harness[bd] = M.Harness(ev, bd, s, r, mvc)
    W, H, B = M.W, M.H, M.BORDER
    t0 = time.time()
    k = 0
    given = plan
    plan = []
    for bd in (8, 10) if given is None else ():
        for (w, h) in ((8, 8), (16, 16), (16, 8), (32, 16)):
            for tree in range(3):
                for masked in (0, 1):
                    plan.append((bd, w, h, tree, 0, masked))
            for masked in (0, 1):
                plan.append((bd, w, h, 2, 3, masked))        # the tree with the up-sampled error
    for entry in (plan if given is None else given):
        bd, w, h, tree, sst, masked = entry[:6]
        compound = entry[6] if len(entry) > 6 else 1
        hs = harness[bd]
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        cost_type = ("ENTROPY", "L1_HDRES", "NONE")[k % 3]
        inv = int(masked and (k % 2))
        bx, by = int(rng.integers(1, (W - w) // 4)) * 4, int(rng.integers(1, (H - h) // 4)) * 4
        refmv = (int(rng.integers(-30, 31)), int(rng.integers(-30, 31)))
        lim = M.limits(bx, by, w, h, 20)
        full = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
        blk = (bx, by, full[0], full[1], refmv[0], refmv[1]) + lim
        allow_hp, forced_stop, iters = int(rng.integers(0, 2)), int(rng.integers(0, 2)), 1 + (k % 2)
        sp = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
        ev.set(sp, "allow_hp", allow_hp); ev.set(sp, "forced_stop", forced_stop); ev.set(sp, "iters_per_step", iters)
        fl = ev.new("FullMvLimits")
        for kk, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
            ev.set(fl, kk, v)
        rm = hs.mv_struct("MV", refmv[0], refmv[1])
        ev.interp.call("av1_set_subpel_mv_search_range", [(ev.field(sp, "mv_limits"), R.PTR), (fl, R.PTR), (rm, R.PTR)])
        epb = int(rng.integers(30, 110))
        hs.cost_params(sp, "mv_cost_params.", cost_type, refmv[0], refmv[1], 25, epb)
        vfp = hs.vtable(w, h)
        names = (dict(svaf="aom_sub_pixel_avg_variance%dx%d_c", msvf="aom_masked_sub_pixel_variance%dx%d_c") if bd == 8 else
                 dict(svaf="aom_highbd_10_sub_pixel_avg_variance%dx%d_c", msvf="aom_highbd_10_masked_sub_pixel_variance%dx%d_c"))
        for kk, pat in names.items():
            fn = pat % (w, h)
            assert fn in ev.funcs, fn
            ev.set(vfp, kk, R.FuncRef(fn))
        ev.set(sp, "var_params.vfp", vfp)
        ev.set(sp, "var_params.subpel_search_type", hs.const(("USE_2_TAPS_ORIG", "USE_2_TAPS", "USE_4_TAPS", "USE_8_TAPS")[sst]))
        ev.set(sp, "var_params.ms_buffers.ref", hs.buf2d(hs.refp, by, bx)); ev.set(sp, "var_params.ms_buffers.src", hs.buf2d(hs.srcp, by, bx))
        ev.set(sp, "var_params.w", w); ev.set(sp, "var_params.h", h)
        # the other reference's predictor: the reference block near the full-pel MV plus noise (an input here)
        refpl = arrays["ref%d" % bd]
        oy, ox = by + full[0] + int(rng.integers(-2, 3)), bx + full[1] + int(rng.integers(-2, 3))
        spred = np.clip(refpl[B + oy:B + oy + h, B + ox:B + ox + w].astype(np.int32) + rng.integers(-(6 << (bd - 8)), (6 << (bd - 8)) + 1, (h, w)), 0, mx).astype(np.uint16)
        if compound:
            ev.set(sp, "var_params.ms_buffers.second_pred", ev.array(spred.ravel(), ct))
        mask = None
        if masked and compound:
            ramp = np.clip((np.arange(w)[None, :] * 2 + np.arange(h)[:, None] - (w + h) // 2) * 4 + 32 + rng.integers(-3, 4, (h, w)), 0, 64)
            mask = ramp.astype(np.uint8)
            ev.set(sp, "var_params.ms_buffers.mask", ev.array(mask.ravel(), "uint8_t"))
            ev.set(sp, "var_params.ms_buffers.mask_stride", w)
            ev.set(sp, "var_params.ms_buffers.inv_mask", inv)
        start = hs.mv_struct("MV", full[0] * 8, full[1] * 8)
        best = ev.new("MV")
        dist, sse = ev.array([0], "int"), ev.array([0], "unsigned int")
        t1 = time.time()
        err = ev.call(TREES[tree], M.make_xd(ev, bd), None, sp, start.buf[0], best, dist, sse, None)
        sl = [ev.get(sp, "mv_limits." + kk) for kk in ("row_min", "row_max", "col_min", "col_max")]
        arrays["sp%d" % k] = spred
        if mask is not None:
            arrays["mask%d" % k] = mask
        cases.append(dict(k=k, bd=bd, w=w, h=h, block=list(blk), tree=tree, subpel_search_type=sst, masked=int(mask is not None), inv=inv, compound=compound, cost_type=M.COST_TYPES[cost_type],
                          error_per_bit=epb, allow_hp=allow_hp, forced_stop=forced_stop, iters=iters, subpel_limits=sl,
                          mv=[ev.get(best, "row"), ev.get(best, "col")], err=err, distortion=dist.buf[0], sse=sse.buf[0]))
        print(k, bd, w, h, TREES[tree][25:], sst, "masked" if masked else "avg", inv, cost_type, cases[-1]["mv"], err, "%.0f s" % (time.time() - t1), flush=True)
        k += 1
    meta = dict(border=B, width=W, height=H, generated_by="tests/golden/gen_ref_eval_compound_subpel.py" if given is None else "tests/golden/gen_ref_eval_subpel_taps.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, out), **arrays)
    print("wrote %s: %d cases, %.0f s" % (out, len(cases), time.time() - t0))


if __name__ == "__main__":
    main()
