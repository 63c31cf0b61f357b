#!/usr/bin/env python3
"""Golden vectors for the SEQUENCING of the compound motion searches, obtained by interpreting the reference's functions themselves (build
container only; tests/golden/ref_c_eval.py, views of gen_ref_eval_composites.py):

  av1_joint_motion_search                 av1/encoder/motion_search_facade.c:496-702   both branches: av1_refining_search_8p_c
                                          (disable_extensive_joint_motion_search) and av1_full_pixel_search(.., 5, ..) with try_second
  av1_compound_single_motion_search       :703-801 (second_pred handed in)

The functions run as they are written -- iteration loop, early-outs, limits through av1_make_default_fullpel_ms_params /
av1_make_default_subpel_ms_params, av1_set_ms_compound_refs, the update rule, av1_mv_bit_cost -- with every search they call interpreted too
(mcomp.c with the reference's own sdaf / msdf / svaf / msvf members).  Adaptations, all of the frame plumbing or of the evaluator's memory model:
  * the predictor of the OTHER reference (av1_init_inter_params + get_conv_params + av1_enc_build_one_inter_predictor, :585-595) is one call of
    a supplied function that writes what the oracle's prediction (pinned by ref_eval_convolve.npz) gives for that MV: it is an INPUT of the
    search.  av1_get_ref_mv returns two given MVs; av1_get_scaled_ref_frame returns NULL (unscaled references); av1_num_planes is 1;
  * int_mv is a struct holding as_mv (no unions in the evaluator): `a.as_int == b.as_int` is rewritten as the comparison of both components,
    `!= INVALID_MV` as "not both INVALID_MV_ROW_COL" (mv.h:26-34: the same bits);
  * second_pred16, a byte array C reuses as uint16_t[] for high bit depth, is declared with the plane's pixel type (the evaluator's buffers are typed).

Output: tests/golden/ref_eval_joint.npz.
"""
import json
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402
import gen_ref_eval_composites as C  # noqa: E402
import gen_ref_eval_compound_search as CS  # noqa: E402
import pyoracle as orc  # noqa: E402  (the predictor of the other reference only)

REF = G.REF
W, H, BORDER = G.W, G.H, G.BORDER
I32, U8, PTR = R.I32, R.U8, R.PTR


def adapt(text, name, pixt):
    t = text
    t = t.replace("int av1_joint_motion_search(", "int %s(" % name).replace("int av1_compound_single_motion_search(", "int %s(" % name)
    t = re.sub(r"const int_interpfilters interp_filters =\s*av1_broadcast_interp_filter\(EIGHTTAP_REGULAR\);", "", t)
    t = t.replace("InterPredParams inter_pred_params;", "")
    t = re.sub(r"av1_init_inter_params\(&inter_pred_params,.*?interp_filters\);", "", t, flags=re.S)
    t = t.replace("inter_pred_params.conv_params = get_conv_params(0, 0, xd->bd);", "")
    t, n = re.subn(r"av1_enc_build_one_inter_predictor\(second_pred, pw, &cur_mv\[!id\]\.as_mv,\s*&inter_pred_params\);",
                   "joint_build_second_pred(second_pred, pw, &cur_mv[!id].as_mv, !id);", t)
    t = re.sub(r"DECLARE_ALIGNED\(16, uint8_t, second_pred16\[MAX_SB_SQUARE \* sizeof\(uint16_t\)\]\);", "%s second_pred16[MAX_SB_SQUARE];" % pixt, t)
    t = t.replace("uint8_t *second_pred = get_buf_by_bd(xd, second_pred16);", "%s *second_pred = second_pred16;" % pixt)
    eq = lambda a, b: "(%s.as_mv.row == %s.as_mv.row && %s.as_mv.col == %s.as_mv.col)" % (a, b, a, b)
    t = re.sub(r"(\w+(?:\[!?\w+\])?)\.as_int == (\w+(?:\[!?\w+\])?)\.as_int", lambda m: eq(m.group(1), m.group(2)), t)
    t = t.replace("second_best_mv.as_int != INVALID_MV", "!(second_best_mv.as_mv.row == INVALID_MV_ROW_COL && second_best_mv.as_mv.col == INVALID_MV_ROW_COL)")
    t = re.sub(r"(\w+)\.as_int != (\w+)\.as_int", lambda m: "!" + eq(m.group(1), m.group(2)), t)
    # (an initializer list of struct VALUES: written as two assignments -- the evaluator flattens initializer lists into scalars)
    t = t.replace("const int_mv init_mv[2] = { cur_mv[0], cur_mv[1] };", "int_mv init_mv[2]; init_mv[0] = cur_mv[0]; init_mv[1] = cur_mv[1];")
    assert ".as_int" not in t, re.findall(r".{30}\.as_int.{20}", t)
    return t


def main():
    ev = CS.make_evaluator()
    C.view(ev, "int_mv", [("as_mv", ev.structs["mv"])])
    ev.define("as_fullmv", "as_mv")
    R.ALIASED_STRUCTS.add(frozenset(("mv", "fullpel_mv")))
    ev.load(REF + "av1/common/common_data.c")
    C.grab(ev, "av1/encoder/encoder.h", "cond_cost_list_const")
    C.grab(ev, "av1/encoder/encoder.h", "cond_cost_list")
    ev.load_text(re.search(r"enum \{\s*SS_CFG_SRC = 0,.*?\} UENUM1BYTE\(SS_CFG_OFFSET\);", open(REF + "av1/encoder/encoder.h").read(), re.S).group(0)
                 .replace("UENUM1BYTE(SS_CFG_OFFSET)", "SS_CFG_OFFSET_e"), "encoder.h:SS_CFG_OFFSET")
    C.grab(ev, "av1/encoder/motion_search_facade.h", "av1_get_search_site_config")
    C.grab(ev, "av1/common/mv.h", "convert_fullmv_to_mv")
    enc = C.Encoder(ev)
    # the two members these functions read beyond the composites' views
    ev.structs["MV_SPEED_FEATURES"].fields.append(("disable_extensive_joint_motion_search", I32))
    comp_t = C.view(ev, "INTERINTER_COMPOUND_DATA_view", [("type", I32)], opaque=False)
    enc.mbmi_t.fields.append(("interinter_comp", comp_t))
    state = {}
    pyc = ev.interp.pycalls
    pyc["av1_get_scaled_ref_frame"] = lambda it, a: (None, PTR)
    pyc["av1_num_planes"] = lambda it, a: (1, I32)
    pyc["has_second_ref"] = lambda it, a: (1, I32)
    pyc["is_interintra_mode"] = lambda it, a: (0, I32)
    pyc["av1_setup_pre_planes"] = lambda it, a: (None, R.VOID)
    pyc["use_fine_search_interval"] = lambda it, a: (0, I32)

    def build_second_pred(it, a):
        dst, pw, mvp, other = a[0][0], int(a[1][0]), a[2][0], int(a[3][0])
        mv = mvp.deref()[0]
        row, col = int(mv.f["row"].deref()[0]), int(mv.f["col"].deref()[0])
        w, h, bx, by = state["w"], state["h"], state["bx"], state["by"]
        assert pw == w
        blk = np.zeros(1, orc.search_block_dtype if hasattr(orc, "search_block_dtype") else [("bx", "<i2"), ("by", "<i2")])
        blk["bx"], blk["by"] = bx, by
        plane = orc.build_inter_pred(state["refs"][other], BORDER, W, H, w, h, blk, [(row, col)], 0, 0, bd=state["bd"])
        pred = plane[by:by + h, bx:bx + w]
        state["preds"].append(dict(other=other, mv=[row, col]))
        for i, v in enumerate(pred.ravel()):
            dst.add(i).store(int(v), I32)
        return (None, R.VOID)
    pyc["joint_build_second_pred"] = build_second_pred
    ev.load_text("int_mv g_joint_ref_mv[2];\nstatic int_mv av1_get_ref_mv(const MACROBLOCK *x, int ref_idx) { return g_joint_ref_mv[ref_idx]; }\n", "joint:av1_get_ref_mv")
    for name in ("EIGHTTAP_REGULAR",):
        if name not in ev.globs:
            ev.define(name, "(0)")
    text = open(REF + "av1/encoder/motion_search_facade.c").read()
    joint = re.search(r"int av1_joint_motion_search\([^;{]*\)\s*\{.*?\n}\n", text, re.S).group(0)
    single = re.search(r"int av1_compound_single_motion_search\([^;{]*\)\s*\{.*?\n}\n", text, re.S).group(0)
    for bd, pixt in ((8, "uint8_t"), (10, "uint16_t")):
        ev.load_text(adapt(joint, "joint_bd%d" % bd, pixt), "motion_search_facade.c:av1_joint_motion_search")
        ev.load_text(adapt(single, "single_bd%d" % bd, pixt), "motion_search_facade.c:av1_compound_single_motion_search")
    for fn in list(pyc):
        ev.funcs.pop(fn, None)
    if ev.skipped:
        bad = [s for s in ev.skipped if s[0].startswith("motion_search_facade.c") or s[0].startswith("joint:")]
        assert not bad, bad
    arrays, cases = {}, []
    mvc = G.synth_mv_costs(17)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    rng = np.random.default_rng(20261301)
    t0 = time.time()
    mesh = [(12, 4), (6, 2), (4, 1), (3, 1)]
    k = 0
    for bd in (8, 10):
        # the source lies between two references that moved in opposite directions
        base, _ = G.synth_planes(bd, 700 + bd)
        src = base
        mx = (1 << bd) - 1
        vis = base[BORDER:BORDER + H, BORDER:BORDER + W].astype(np.int32)
        def moved(dy, dx, seed):   # the visible area displaced + noise, then the border replicated (the planes' model: aom_extend_frame_borders)
            r = np.random.default_rng(seed)
            v = np.roll(np.roll(vis, dy, 0), dx, 1) + r.integers(-(2 << (bd - 8)), (2 << (bd - 8)) + 1, vis.shape)
            return np.pad(np.clip(v, 0, mx).astype(base.dtype), BORDER, mode="edge")
        ref0, ref1 = moved(2, -3, 1), moved(-3, 2, 2)
        arrays["src%d" % bd], arrays["ref0_%d" % bd], arrays["ref1_%d" % bd] = src, ref0, ref1
        hs0, hs1 = G.Harness(ev, bd, src, ref0, mvc), G.Harness(ev, bd, src, ref1, mvc)
        state["refs"], state["bd"] = (ref0, ref1), bd
        specs = [dict(fn="joint", ext=0, second=0, masked=0, w=16, h=16, tree="SUBPEL_TREE", taps="USE_2_TAPS"),
                 dict(fn="joint", ext=1, second=1, masked=0, w=16, h=16, tree="SUBPEL_TREE", taps="USE_8_TAPS"),
                 dict(fn="joint", ext=1, second=1, masked=1, w=8, h=8, tree="SUBPEL_TREE_PRUNED", taps="USE_2_TAPS", mesh_thr=2000),
                 dict(fn="joint", ext=0, second=1, masked=1, w=16, h=8, tree="SUBPEL_TREE_PRUNED_MORE", taps="USE_2_TAPS"),
                 dict(fn="joint", ext=1, second=0, masked=0, w=8, h=16, tree="SUBPEL_TREE", taps="USE_4_TAPS", force_int=1),
                 dict(fn="joint", ext=1, second=1, masked=0, w=8, h=8, tree="SUBPEL_TREE", taps="USE_4_TAPS", far=1),
                 dict(fn="joint", ext=0, second=0, masked=0, w=8, h=8, tree="SUBPEL_TREE_PRUNED", taps="USE_2_TAPS", converged=1),
                 dict(fn="joint", ext=1, second=1, masked=1, w=16, h=8, tree="SUBPEL_TREE", taps="USE_2_TAPS_ORIG", converged=1),
                 dict(fn="single", masked=1, ref_idx=0, w=16, h=16, tree="SUBPEL_TREE", taps="USE_4_TAPS"),
                 dict(fn="single", masked=1, ref_idx=1, w=8, h=8, tree="SUBPEL_TREE_PRUNED", taps="USE_8_TAPS", mesh_thr=1500),
                 dict(fn="single", masked=0, ref_idx=1, w=16, h=8, tree="SUBPEL_TREE", taps="USE_2_TAPS", force_int=1)]
        for spec in specs:
            w, h = spec["w"], spec["h"]
            sf = dict(search_method="NSTEP", subpel_search_method=spec["tree"], use_accurate_subpel_search=spec["taps"], sadperbit=int(rng.integers(10, 40)),
                      errorperbit=int(rng.integers(30, 100)), force_integer_mv=spec.get("force_int", 0), mesh=mesh,
                      exhaustive_searches_thresh=spec.get("mesh_thr", C.INT_MAX))
            cpi, x = enc.make(hs0, bd, W, H, sf, 30, mvc, sizes=((16, 16), (8, 8), (16, 8), (8, 16)))
            for (ww, hh) in ((16, 16), (8, 8), (16, 8), (8, 16)):   # the compound members of every vtable entry the call may touch
                vt = ev.field(ev.get(cpi, "ppi"), "fn_ptr[%d]" % hs0.const(G.BSIZE[(ww, hh)]))
                CS.extend_vtable(ev, vt, bd, ww, hh)
            ev.set(cpi, "sf.mv_sf.disable_extensive_joint_motion_search", 0 if spec.get("ext") else 1)
            bx, by = int(rng.integers(1, (W - w) // 8)) * 8, int(rng.integers(1, (H - h) // 8)) * 8
            state.update(w=w, h=h, bx=bx, by=by, preds=[])
            lim = G.limits(bx, by, w, h, 30)
            for kk, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
                ev.set(x, "mv_limits." + kk, v)
            off = (BORDER + by) * hs0.S + BORDER + bx
            ev.set(x, "plane[0].src.buf", hs0.srcp.add(off)); ev.set(x, "plane[0].src.stride", hs0.S)
            for r_, hsx in enumerate((hs0, hs1)):
                p = "e_mbd.plane[0].pre[%d]." % r_
                ev.set(x, p + "buf", hsx.refp.add(off)); ev.set(x, p + "buf0", hsx.refp.add(off)); ev.set(x, p + "stride", hsx.S)
                ev.set(x, p + "width", W); ev.set(x, p + "height", H)
            ev.set(x, "e_mbd.mi_row", by // 4); ev.set(x, "e_mbd.mi_col", bx // 4)
            ev.set(enc.mi, "ref_frame[0]", 1); ev.set(enc.mi, "ref_frame[1]", 4)
            ev.set(enc.mi, "interinter_comp.type", hs0.const("COMPOUND_DIFFWTD" if spec["masked"] else "COMPOUND_AVERAGE"))
            true = (np.array([2 * 8, -3 * 8]), np.array([-3 * 8, 2 * 8]))   # what moved() gives each reference
            spread = 60 if spec.get("far") else 26
            cur = [(true[r_] + rng.integers(-spread, spread + 1, 2)).tolist() for r_ in range(2)]
            refmv = [rng.integers(-40, 41, 2).tolist() for _ in range(2)]
            g = ev.globs["g_joint_ref_mv"]
            for r_ in range(2):
                ev.set(g, "[%d].as_mv.row" % r_, refmv[r_][0]); ev.set(g, "[%d].as_mv.col" % r_, refmv[r_][1])
            mask = None
            maskp = None
            if spec["masked"]:
                mask = np.clip((np.arange(w)[None, :] * 64 // w + rng.integers(-6, 7, (h, w))), 0, 64).astype(np.uint8)
                maskp = ev.array(mask.ravel(), "uint8_t")
            rate = ev.array([0], "int")
            t1 = time.time()
            rec = dict(k=k, bd=bd, bx=bx, by=by, limits=list(lim), ref_mv=refmv, cur_in=cur, sadperbit=sf["sadperbit"], errorperbit=sf["errorperbit"], **spec)
            if spec["fn"] == "joint":
                cm = ev.interp.alloc(("arr", ev.typedefs["int_mv"], 2), True)
                for r_ in range(2):
                    ev.set(cm, "[%d].as_mv.row" % r_, cur[r_][0]); ev.set(cm, "[%d].as_mv.col" % r_, cur[r_][1])
                for rep in range(2 if spec.get("converged") else 1):   # converged: the result of a first call is the input of the recorded one
                    if rep:
                        cur = [[ev.get(cm, "[%d].as_mv.row" % r_), ev.get(cm, "[%d].as_mv.col" % r_)] for r_ in range(2)]
                        rec["cur_in"] = cur
                        state["preds"] = []
                    err = ev.call("joint_bd%d" % bd, cpi, x, hs0.const(G.BSIZE[(w, h)]), ev.field(cm, "[0]"), maskp, w if mask is not None else 0, rate,
                                  spec["second"])
                rec.update(cur_out=[[ev.get(cm, "[%d].as_mv.row" % r_), ev.get(cm, "[%d].as_mv.col" % r_)] for r_ in range(2)], err=err, rate_mv=rate.buf[0],
                           predictors=state["preds"])
            else:
                ri = spec["ref_idx"]
                this = G.Harness.mv_struct(hs0, "MV", cur[ri][0], cur[ri][1])
                oy, ox = by + (cur[1 - ri][0] >> 3), bx + (cur[1 - ri][1] >> 3)
                sp = np.clip((ref1 if ri == 0 else ref0)[BORDER + oy:BORDER + oy + h, BORDER + ox:BORDER + ox + w].astype(np.int32) +
                             rng.integers(-(4 << (bd - 8)), (4 << (bd - 8)) + 1, (h, w)), 0, mx).astype(np.uint16)
                arrays["sp%d" % k] = sp
                SP = ev.array(sp.ravel(), "uint8_t" if bd == 8 else "uint16_t")
                err = ev.call("single_bd%d" % bd, cpi, x, hs0.const(G.BSIZE[(w, h)]), this, SP, maskp, w if mask is not None else 0, rate, ri)
                rec.update(this_out=[ev.get(this, "row"), ev.get(this, "col")], err=err, rate_mv=rate.buf[0])
            if mask is not None:
                arrays["mask%d" % k] = mask
            cases.append(rec)
            print(k, spec["fn"], bd, w, h, rec.get("cur_out") or rec.get("this_out"), err, "%.0f s (%.0f)" % (time.time() - t1, time.time() - t0), flush=True)
            k += 1
    meta = dict(border=BORDER, width=W, height=H, mesh=mesh, generated_by="tests/golden/gen_ref_eval_joint.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_joint.npz"), **arrays)
    print("wrote ref_eval_joint.npz: %d cases" % len(cases))


if __name__ == "__main__":
    main()
