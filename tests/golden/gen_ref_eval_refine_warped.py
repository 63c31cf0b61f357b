#!/usr/bin/env python3
"""Golden vectors for av1_refine_warped_mv AS IT IS WRITTEN (av1/encoder/mcomp.c:3197-3293: compute_motion_cost and the two rounds over the four
neighbours), obtained by interpreting the function (build container only; tests/golden/ref_c_eval.py) together with av1_selectSamples
(av1/common/mvref_common.c:1083-1104), av1_find_projection / find_affine_int / av1_get_shear_params (av1/common/warped_motion.c), the variance of the
block size's vtable and mv_err_cost_.

Supplied as inputs / adaptations:
  * av1_enc_build_inter_predictor writes what the oracle's warped predictor (oracle/aomref_warp.c, pinned by ref_eval_warp.npz: av1_warp_affine_c
    interpreted) gives for mbmi->wm_params on the block's rectangle -- as gen_ref_eval_joint.py does for the convolve predictor; the fixture records
    the model of every call;
  * the starting model (mbmi->wm_params, num_proj_ref) is what the reference's own av1_selectSamples + av1_find_projection give at the starting MV, as
    motion_mode_rd prepares it; blocks whose starting model is unusable are not cases;
  * MACROBLOCKD / MB_MODE_INFO are views with the members the function reads (mi, mi_row, mi_col, plane[0].dst; mv[0], wm_params, num_proj_ref);
    int_mv is a struct holding as_mv.

Output: tests/golden/ref_eval_refine_warped.npz.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402
import gen_ref_eval_composites as C  # noqa: E402
from gen_ref_eval_yrd import cut  # noqa: E402
import pyoracle as orc  # noqa: E402  (the warped predictor only)

REF = G.REF
W, H, BORDER = G.W, G.H, G.BORDER
I32, U8, PTR = R.I32, R.U8, R.PTR


def main():
    ev = G.make_evaluator()
    # (mcomp.c was read without aom/aom_image.h:208 and av1/common/warped_motion.h:28-30: the two constants its text names, before the header's macros exist)
    ev.load_text("enum { AOM_PLANE_Y = 0, SAMPLES_ARRAY_SIZE = 16 };\n", "aom_image.h:AOM_PLANE_Y, warped_motion.h:SAMPLES_ARRAY_SIZE")
    # (TransformationType comes from flow_estimation.h, which the motion-search evaluator does not load: mv.h's WarpedMotionParams is read again after it)
    for f in ("aom_dsp/flow_estimation/flow_estimation.h", "av1/common/convolve.h"):
        ev.load(REF + f)
    ev.structs.pop("<opaque>WarpedMotionParams", None)
    ev.load_text(cut(open(REF + "av1/common/mv.h").read(), "typedef struct {\n  int32_t wmmat[6];"), "mv.h:WarpedMotionParams")
    for n in ("av1_warp_affine", "av1_highbd_warp_affine", "av1_calc_frame_error"):
        ev.define(n, n + "_c")
    n0 = len(ev.skipped)
    for f in ("av1/common/warped_motion.h", "av1/common/warped_motion.c"):
        ev.load(REF + f)
    ev.define("LEAST_SQUARES_SAMPLES_MAX", "(8)")
    ev.load_text(cut(open(REF + "av1/common/mvref_common.c").read(), "uint8_t av1_selectSamples("), "mvref_common.c:av1_selectSamples")
    assert "av1_refine_warped_mv" in ev.funcs and "av1_find_projection" in ev.funcs and "compute_motion_cost" in ev.funcs
    imv = C.view(ev, "int_mv", [("as_mv", ev.structs["mv"])])
    wm_t = ev.typedefs["WarpedMotionParams"]
    mbmi = ev.structs["<opaque>MB_MODE_INFO"]
    mbmi.fields = [("use_intrabc", U8), ("mv", ("arr", imv, 2)), ("wm_params", wm_t), ("num_proj_ref", U8)]
    pdp_t = C.view(ev, "macroblockd_plane", [("dst", ev.structs["buf_2d"])], opaque=False)
    xd_t = ev.structs["<opaque>MACROBLOCKD"]
    xd_t.fields = xd_t.fields + [("plane", ("arr", pdp_t, 3))]
    state = {}
    pyc = ev.interp.pycalls

    def build_pred(it, a):   # av1_enc_build_inter_predictor(cm, xd, mi_row, mi_col, ctx, bsize, plane_from, plane_to)
        wm = ev.field(state["mi"], "wm_params")
        mat = [int(ev.get(wm, "wmmat[%d]" % i)) for i in range(6)]
        sh = [int(ev.get(wm, f)) for f in ("alpha", "beta", "gamma", "delta")]
        assert (int(a[2][0]), int(a[3][0])) == (state["by"] // 4, state["bx"] // 4) and int(a[6][0]) == 0 and int(a[7][0]) == 0
        pred = orc.warp_block_pred(state["ref_vis"], state["bd"], mat, sh, state["bx"], state["by"], state["w"], state["h"])
        dst = state["dst"]
        for i, v in enumerate(pred.ravel()):
            dst.add(i).store(int(v), I32)
        state["calls"].append(dict(mat=mat, shear=sh, mv_in_mbmi=[int(ev.get(state["mi"], "mv[0].as_mv.row")), int(ev.get(state["mi"], "mv[0].as_mv.col"))]))
        return (None, R.VOID)
    pyc["av1_enc_build_inter_predictor"] = build_pred
    ev.funcs.pop("av1_enc_build_inter_predictor", None)
    arrays, cases = {}, []
    mvc = G.synth_mv_costs(41)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    rng = np.random.default_rng(20261009)
    t0 = time.time()
    k = 0
    for bd in (8, 10):
        s_, r_ = G.synth_planes(bd, 500 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s_, r_
        hs = G.Harness(ev, bd, s_, r_, mvc)
        ref_vis = np.ascontiguousarray(r_[BORDER:BORDER + H, BORDER:BORDER + W])
        ct = "uint8_t" if bd == 8 else "uint16_t"
        tried = 0
        while len([c for c in cases if c["bd"] == bd]) < (14 if bd == 8 else 10) and tried < 80:
            tried += 1
            w, h = [(16, 16), (8, 8), (16, 8), (32, 16), (8, 16), (32, 32)][tried % 6]
            bx, by = int(rng.integers(1, (W - w) // 8)) * 8, int(rng.integers(1, (H - h) // 8)) * 8
            mv = [int(rng.integers(-40, 41)), int(rng.integers(-40, 41))]
            refmv = [mv[0] + int(rng.integers(-12, 13)), mv[1] + int(rng.integers(-12, 13))]
            n = int(rng.integers(1, 9)) if tried % 5 else 1
            pts = np.zeros((n, 2), np.int64)
            pts[:, 0] = rng.integers(-8 * 48, 8 * (w + 24), n)
            pts[:, 1] = rng.integers(-8 * 48, 8 * (h + 24), n)
            a = np.array([[1.0 + rng.normal(0, 0.03), rng.normal(0, 0.03)], [rng.normal(0, 0.03), 1.0 + rng.normal(0, 0.03)]])
            ctr = np.array([w * 4.0, h * 4.0])
            pin = np.rint((pts - ctr) @ a.T + ctr + np.array([mv[1], mv[0]]) + rng.normal(0, 2.5, (n, 2))).astype(np.int64)
            cost_type = ("ENTROPY", "L1_HDRES", "NONE", "L1_LOWRES")[tried % 4]
            allow_hp = int(tried % 3 != 0)
            sp = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
            ev.set(sp, "allow_hp", allow_hp)
            lim = (mv[1] - int(rng.integers(0, 4)), mv[1] + int(rng.integers(0, 4)), mv[0] - int(rng.integers(0, 4)), mv[0] + int(rng.integers(0, 4))) if tried % 7 == 0 \
                else (mv[1] - 64, mv[1] + 64, mv[0] - 64, mv[0] + 64)
            for kk, v in zip(("col_min", "col_max", "row_min", "row_max"), lim):
                ev.set(sp, "mv_limits." + kk, v)
            hs.cost_params(sp, "mv_cost_params.", cost_type, refmv[0], refmv[1], 20, int(rng.integers(30, 100)))
            epb = int(ev.get(sp, "mv_cost_params.error_per_bit"))
            ev.set(sp, "var_params.vfp", hs.vtable(w, h))
            ev.set(sp, "var_params.ms_buffers.src", hs.buf2d(hs.srcp, by, bx))
            xd = G.make_xd(ev, bd)
            mi = ev.get(xd, "mi").deref()[0]
            ev.set(xd, "mi_row", by // 4); ev.set(xd, "mi_col", bx // 4)
            dst = ev.array([0] * (w * h), ct)
            ev.set(xd, "plane[0].dst.buf", dst); ev.set(xd, "plane[0].dst.stride", w)
            ev.set(mi, "mv[0].as_mv.row", mv[0]); ev.set(mi, "mv[0].as_mv.col", mv[1])
            bs = ev.globs["BLOCK_%dX%d" % (w, h)].buf[0]
            # the starting model, as the caller prepares it: selectSamples (when more than one sample) + find_projection at the starting MV
            P, Q = ev.array(pts.ravel(), "int"), ev.array(pin.ravel(), "int")
            m = hs.mv_struct("MV", mv[0], mv[1])
            np0 = int(ev.call("av1_selectSamples", m, P, Q, n, bs)) if n > 1 else 1
            wm = ev.field(mi, "wm_params")
            ev.set(wm, "wmtype", 3)
            if int(ev.call("av1_find_projection", np0, P, Q, bs, mv[0], mv[1], wm, by // 4, bx // 4)):
                continue
            ev.set(mi, "num_proj_ref", np0)
            start_model = dict(mat=[int(ev.get(wm, "wmmat[%d]" % i)) for i in range(6)], shear=[int(ev.get(wm, f)) for f in ("alpha", "beta", "gamma", "delta")])
            state.update(mi=mi, bx=bx, by=by, w=w, h=h, bd=bd, ref_vis=ref_vis, dst=dst, calls=[])
            P0, Q0 = ev.array(pts.ravel(), "int"), ev.array(pin.ravel(), "int")      # (pts0 / pts_inref0: the samples as found, not yet selected)
            t1 = time.time()
            mse = int(ev.call("av1_refine_warped_mv", xd, None, sp, bs, P0, Q0, n))
            rec = dict(k=k, bd=bd, w=w, h=h, bx=bx, by=by, mv=mv, ref_mv=refmv, limits=list(lim), cost_type=cost_type, error_per_bit=epb, allow_hp=allow_hp,
                       total_samples=n, pts=pts.ravel().tolist(), pts_inref=pin.ravel().tolist(), num_proj_ref=np0, start_model=start_model,
                       bestmse=mse & 0xFFFFFFFF, best_mv=[int(ev.get(mi, "mv[0].as_mv.row")), int(ev.get(mi, "mv[0].as_mv.col"))],
                       best_model=dict(mat=[int(ev.get(wm, "wmmat[%d]" % i)) for i in range(6)], shear=[int(ev.get(wm, f)) for f in ("alpha", "beta", "gamma", "delta")]),
                       best_num_proj_ref=int(ev.get(mi, "num_proj_ref")), calls=state["calls"])
            cases.append(rec)
            print(k, bd, w, h, n, cost_type, allow_hp, mv, "->", rec["best_mv"], rec["bestmse"], len(rec["calls"]), "%.0f s (%.0f)" % (time.time() - t1, time.time() - t0), flush=True)
            k += 1
    meta = dict(border=BORDER, width=W, height=H, generated_by="tests/golden/gen_ref_eval_refine_warped.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_refine_warped.npz"), **arrays)
    print("wrote ref_eval_refine_warped.npz: %d cases, moved %d" % (len(cases), sum(c["best_mv"] != c["mv"] for c in cases)))


if __name__ == "__main__":
    main()
