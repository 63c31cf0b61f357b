#!/usr/bin/env python3
"""Golden vectors for the OBMC sub-pel search, obtained by interpreting av1/encoder/mcomp.c itself (build container only;
tests/golden/ref_c_eval.py, harness of gen_ref_eval_mcomp.py):

  av1_find_best_obmc_sub_pixel_tree_up (mcomp.c:3588-3633) with var_params.ms_buffers.wsrc / obmc_mask, both error forms:
      subpel_search_type USE_2_TAPS_ORIG -- setup_obmc_center_error (vfp->ovf at ms_buffers->ref->buf) and obmc_check_better_fast
          (vfp->osvf + estimate_obmc_mvcost);
      USE_8_TAPS -- upsampled_obmc_pref_error (aom_[highbd_]upsampled_pred, then vfp->ovf) + mv_err_cost_.

The vtable members are the reference's own functions: ovf = aom_obmc_variance{W}x{H}_c, osvf = aom_obmc_sub_pixel_variance{W}x{H}_c;
10-bit: aom_highbd_10_obmc_[sub_pixel_]variance{W}x{H}_c.

Output: tests/golden/ref_eval_obmc_subpel.npz.
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


def pred_buffer_adaptation(ev):
    """upsampled_obmc_pref_error declares `uint8_t pred[2 * MAX_SB_SQUARE]` and, for high bit depth, uses it as MAX_SB_SQUARE uint16_t
    through CONVERT_TO_BYTEPTR / CONVERT_TO_SHORTPTR (mcomp.c:3334-3342).  The evaluator's buffers are typed (a uint8_t array cannot hold
    10-bit values), so for the high-bit-depth cases the function is re-read with that ONE declaration as `uint16_t pred[MAX_SB_SQUARE]`
    -- the same bytes in C, every other token the reference's.  Returns {8: original text, 10: adapted text} of the function."""
    src = open(M.REF + "av1/encoder/mcomp.c").read()
    a = src.index("static int upsampled_obmc_pref_error(")
    b = src.index("static unsigned int setup_obmc_center_error(")
    fn = src[a:b]
    decl = "DECLARE_ALIGNED(16, uint8_t, pred[2 * MAX_SB_SQUARE]);"
    assert fn.count(decl) == 1
    return {8: fn, 10: fn.replace(decl, "DECLARE_ALIGNED(16, uint16_t, pred[MAX_SB_SQUARE]);")}


def main(plan=None, out="ref_eval_obmc_subpel.npz", seed=20261201):
    """plan: (bd, w, h, SUBPEL_SEARCH_TYPE name, trial) per case; the default is what produced ref_eval_obmc_subpel.npz."""
    ev = M.make_evaluator(with_compound=True)
    pred_text = pred_buffer_adaptation(ev)
    loaded_for = None
    arrays, cases = {}, []
    rng = np.random.default_rng(seed)
    mvc = M.synth_mv_costs(13)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    harness = {}
    for bd in (8, 10):
        s, r = M.synth_planes(bd, 500 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s, r
        harness[bd] = M.Harness(ev, bd, s, r, mvc)
    W, H, B = M.W, M.H, M.BORDER
    t0 = time.time()
    k = 0
    given, plan = plan, []
    for bd in (8, 10) if given is None else ():
        for (w, h) in ((8, 8), (16, 16), (16, 8), (8, 16), (32, 16)):
            for stype in ("USE_2_TAPS_ORIG", "USE_8_TAPS"):
                for trial in range(3):
                    plan.append((bd, w, h, stype, trial))
    for (bd, w, h, stype, trial) in (plan if given is None else given):
        hs = harness[bd]
        if loaded_for != bd:
            ev.load_text(pred_text[bd], "mcomp.c:upsampled_obmc_pref_error")
            loaded_for = bd
        mx = (1 << bd) - 1
        # estimate_obmc_mvcost asserts on the L1 types: ENTROPY / NONE for the bilinear form; mv_err_cost_ of the 8-tap form takes all
        cost_type = ("ENTROPY", "NONE", "ENTROPY")[trial] if stype == "USE_2_TAPS_ORIG" else ("ENTROPY", "L1_HDRES", "NONE")[trial]
        bx, by = int(rng.integers(1, (W - w) // 4)) * 4, int(rng.integers(1, (H - h) // 4)) * 4
        refmv = (int(rng.integers(-30, 31)), int(rng.integers(-30, 31)))
        lim = M.limits(bx, by, w, h, 20)
        full = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))) if trial < 2 else (0, 0)         # the full-pel MV the sub-pel search starts from
        blk = (bx, by, full[0], full[1], refmv[0], refmv[1]) + lim
        allow_hp, forced_stop, iters = int(rng.integers(0, 2)), int(rng.integers(0, 2)), 2 - (trial & 1)
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
        names = (dict(ovf="aom_obmc_variance%dx%d_c", osvf="aom_obmc_sub_pixel_variance%dx%d_c") if bd == 8 else
                 dict(ovf="aom_highbd_10_obmc_variance%dx%d_c", osvf="aom_highbd_10_obmc_sub_pixel_variance%dx%d_c"))
        for kk, pat in names.items():
            fn = pat % (w, h)
            assert fn in ev.funcs, fn
            ev.set(vfp, kk, R.FuncRef(fn))
        ev.set(sp, "var_params.vfp", vfp)
        ev.set(sp, "var_params.subpel_search_type", hs.const(stype))
        ev.set(sp, "var_params.ms_buffers.ref", hs.buf2d(hs.refp, by, bx))
        ev.set(sp, "var_params.w", w); ev.set(sp, "var_params.h", h)
        # calc_target_weighted_pred's outputs as in gen_ref_eval_compound_search.py
        srcpl = arrays["src%d" % bd]
        sblk = srcpl[B + by:B + by + h, B + bx:B + bx + w].astype(np.int64)
        om = np.full((h, w), 4096, np.int64)
        om[:h // 2, :] = (np.linspace(36, 64, h // 2).astype(np.int64)[:, None]) * 64
        om[:, :w // 2] = np.minimum(om[:, :w // 2], (np.linspace(34, 64, w // 2).astype(np.int64)[None, :]) * 64)
        nb = np.clip(sblk + rng.integers(-(10 << (bd - 8)), (10 << (bd - 8)) + 1, (h, w)), 0, mx)
        ws = sblk * 4096 - nb * (4096 - om)
        ev.set(sp, "var_params.ms_buffers.wsrc", ev.array(ws.ravel(), "int32_t"))
        ev.set(sp, "var_params.ms_buffers.obmc_mask", ev.array(om.ravel(), "int32_t"))
        start = hs.mv_struct("MV", full[0] * 8, full[1] * 8)
        best = ev.new("MV")
        dist, sse = ev.array([0], "int"), ev.array([0], "unsigned int")
        t1 = time.time()
        err = ev.call("av1_find_best_obmc_sub_pixel_tree_up", M.make_xd(ev, bd), None, sp, start.buf[0], best, dist, sse, None)
        sl = [ev.get(sp, "mv_limits." + kk) for kk in ("row_min", "row_max", "col_min", "col_max")]
        arrays["ws%d" % k], arrays["om%d" % k] = ws.astype(np.int32), om.astype(np.int32)
        cases.append(dict(k=k, bd=bd, w=w, h=h, block=list(blk), subpel_search_type=("USE_2_TAPS_ORIG", "USE_2_TAPS", "USE_4_TAPS", "USE_8_TAPS").index(stype), cost_type=M.COST_TYPES[cost_type],
                          error_per_bit=epb, allow_hp=allow_hp, forced_stop=forced_stop, iters=iters, subpel_limits=sl,
                          mv=[ev.get(best, "row"), ev.get(best, "col")], err=err, distortion=dist.buf[0], sse=sse.buf[0]))
        print(k, bd, w, h, stype, cost_type, cases[-1]["mv"], err, "%.0f s" % (time.time() - t1), flush=True)
        k += 1
    meta = dict(border=B, width=W, height=H, generated_by="tests/golden/gen_ref_eval_obmc_subpel.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, out), **arrays)
    print("wrote %s: %d cases, %.0f s" % (out, len(cases), time.time() - t0))


if __name__ == "__main__":
    main()
