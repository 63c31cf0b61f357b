#!/usr/bin/env python3
"""Golden vectors for av1_single_motion_search AS IT IS WRITTEN (av1/encoder/motion_search_facade.c:120-495), obtained by interpreting the function
(build container only; tests/golden/ref_c_eval.py, views of gen_ref_eval_composites.py): step_param and the search_range narrowing, the two start
candidates and the weight rule, ONE cost_list for both full-pel searches, best / second-best bookkeeping, the sub-pel search on
fractional_ms_list, try_second with disable_second_mv == 1, convert_fullmv_to_mv, *rate_mv, x->pred_sse[ref] -- SIMPLE_TRANSLATION and
OBMC_CAUSAL.  ref_eval_single.npz (round 3) holds the callees on one shared list; here the caller runs.

Supplied as inputs / adaptations (frame plumbing and the evaluator's memory model):
  * av1_get_ref_mv returns a given MV; av1_get_scaled_ref_frame returns NULL; av1_num_planes is 1; get_mv_candidate_from_tpl returns a given
    second candidate with its weights (the TPL statistics are the encoder's state); mode_info / args are NULL with skip_newmv_in_drl == 0 and
    skip_fullpel_search_using_startmv == 0 (the early exits on the mode loop's state, :300-341, :447-483, are the caller's);
  * the RD branch of the second-MV decision (disable_second_mv == 0: av1_enc_build_inter_predictor + av1_estimate_txfm_yrd, :376-391, :404-423)
    is cut from the text: every case runs disable_second_mv == 1 (the var comparison, :424-430);
  * int_mv is a struct holding as_mv: as_int comparisons / assignments are written component-wise (mv.h:26-34: the same bits); `cand` has 3
    entries (cand_cnt <= 2) and is zeroed by a loop instead of av1_zero on MAX_TPL_BLK_IN_SB^2 + 1.

Output: tests/golden/ref_eval_single_caller.npz.
"""
import json
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402
import gen_ref_eval_composites as C  # noqa: E402
import gen_ref_eval_compound_search as CS  # noqa: E402
import gen_ref_eval_obmc_subpel as OS  # noqa: E402

REF = G.REF
W, H, BORDER = G.W, G.H, G.BORDER
I32, U8, PTR = R.I32, R.U8, R.PTR
OPER = r"((?:\w+(?:\[[^\]]+\])*(?:\.|->))*\w+(?:\[[^\]]+\])*)(\.|->)as_int"


def adapt(text, keep_rd=False):
    t = text.replace("void av1_single_motion_search(", "void single_ms(")
    t = t.replace("struct buf_2d backup_yv12[MAX_MB_PLANE] = { { 0, 0, 0, 0, 0 } };", "struct buf_2d backup_yv12[MAX_MB_PLANE];")
    t, n = re.subn(r"cand_mv_t cand\[MAX_TPL_BLK_IN_SB \* MAX_TPL_BLK_IN_SB \+ 1\];\s*av1_zero\(cand\);",
                   "cand_mv_t cand[3]; for (int z_ = 0; z_ < 3; z_++) { cand[z_].fmv.as_mv.row = 0; cand[z_].fmv.as_mv.col = 0; cand[z_].weight = 0; }", t)
    assert n == 1
    inv = lambda a, acc: "(%s%sas_mv.row == INVALID_MV_ROW_COL && %s%sas_mv.col == INVALID_MV_ROW_COL)" % (a, acc, a, acc)
    setinv = lambda a, acc: "%s%sas_mv.row = INVALID_MV_ROW_COL; %s%sas_mv.col = INVALID_MV_ROW_COL" % (a, acc, a, acc)
    t = t.replace("best_mv->as_int = second_best_mv.as_int = INVALID_MV;", setinv("best_mv", "->") + "; " + setinv("second_best_mv", ".") + ";")
    if keep_rd:
        # the RD branch stays (gen_ref_eval_single_rd.py): only orig_dst -- the predictor's destination, an argument of a supplied function -- goes
        a = t.index("struct macroblockd_plane *p = xd->plane;")
        b = t.index("int64_t rd = INT64_MAX;", a)
        t = t[:a] + t[b:]
        assert t.count("&orig_dst") == 2
        t = t.replace("&orig_dst", "NULL")
    else:
        # the RD branch of the second-MV decision
        a = t.index("struct macroblockd_plane *p = xd->plane;")
        b = t.index("MV this_best_mv;", a)
        t = t[:a] + t[b:]
        a = t.index("if (!cpi->sf.mv_sf.disable_second_mv) {\n                // If cpi->sf.mv_sf.disable_second_mv is 0")
        b = t.index("} else {\n                // If cpi->sf.mv_sf.disable_second_mv = 1", a)
        t = t[:a] + "{" + t[b + len("} else {"):]
    eq = lambda a1, c1, a2, c2: "(%s%sas_mv.row == %s%sas_mv.row && %s%sas_mv.col == %s%sas_mv.col)" % (a1, c1, a2, c2, a1, c1, a2, c2)
    t = re.sub(OPER + r" == INVALID_MV", lambda m: inv(m.group(1), m.group(2)), t)
    t = re.sub(OPER + r" != INVALID_MV", lambda m: "!" + inv(m.group(1), m.group(2)), t)
    t = re.sub(OPER + r" == " + OPER, lambda m: eq(m.group(1), m.group(2), m.group(3), m.group(4)), t)
    t = re.sub(OPER + r" != " + OPER, lambda m: "!" + eq(m.group(1), m.group(2), m.group(3), m.group(4)), t)
    t = re.sub(OPER + r" = INVALID_MV;", lambda m: "{ " + setinv(m.group(1), m.group(2)) + "; }", t)
    t = re.sub(OPER + r" = " + OPER + ";", lambda m: "%s%sas_mv = %s%sas_mv;" % (m.group(1), m.group(2), m.group(3), m.group(4)), t)
    assert "as_int" not in t, re.findall(r".{40}as_int.{20}", t)
    assert "BUFFER_SET" not in t and (keep_rd or "RD_STATS" not in t)
    return t


def setup(keep_rd=False, before_function=None):
    """-> (ev, enc, state, pred_text): the evaluator with av1_single_motion_search loaded; before_function(ev, enc): declarations the kept text needs"""
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
    for f in ("auto_mv_step_size", "full_pixel_search_level", "skip_fullpel_search_using_startmv", "disable_second_mv"):
        ev.structs["MV_SPEED_FEATURES"].fields.append((f, I32))
    inter_sf_t = C.view(ev, "INTER_MODE_SPEED_FEATURES", [("skip_newmv_in_drl", I32)], opaque=False)
    ev.structs["SPEED_FEATURES"].fields.append(("inter_sf", inter_sf_t))
    enc.mbmi_t.fields.append(("ref_mv_idx", I32))
    ev.structs["AV1_COMMON"].fields.append(("show_frame", I32)) if "AV1_COMMON" in ev.structs else ev.structs["<opaque>AV1_COMMON"].fields.append(("show_frame", I32))
    enc.x_t.fields.append(("max_mv_context", ("arr", I32, 8)))
    fmv_t = ev.typedefs["FULLPEL_MV"]
    C.view(ev, "HandleInterModeArgs", [("start_mv_cnt", I32), ("start_mv_stack", ("arr", fmv_t, 8))])
    C.view(ev, "inter_mode_info", [("full_search_mv", ev.typedefs["int_mv"]), ("full_mv_rate", I32), ("full_mv_bestsme", I32), ("drl_cost", I32), ("skip", I32)])
    state = {}
    pyc = ev.interp.pycalls
    pyc["av1_get_scaled_ref_frame"] = lambda it, a: (None, PTR)
    pyc["av1_num_planes"] = lambda it, a: (1, I32)
    pyc["av1_setup_pre_planes"] = lambda it, a: (None, R.VOID)
    pyc["use_fine_search_interval"] = lambda it, a: (0, I32)

    def tpl_cand(it, a):   # get_mv_candidate_from_tpl(cpi, x, bsize, ref, cand, &cnt, &total_weight)
        cand, cnt, tot = a[4][0], a[5][0], a[6][0]
        c2 = state["cand2"]
        if c2 is not None:
            e = cand.add(1).deref()[0]
            mv = e.f["fmv"].deref()[0].f["as_mv"].deref()[0]
            mv.f["row"].store(c2[0], I32); mv.f["col"].store(c2[1], I32)
            e.f["weight"].store(state["w1"], I32)
            cand.deref()[0].f["weight"].store(state["w0"], I32)
            cnt.store(2, I32)
            tot.store(state["w0"] + state["w1"], I32)
        return (None, R.VOID)
    pyc["get_mv_candidate_from_tpl"] = tpl_cand
    ev.load_text("typedef struct { int_mv fmv; int weight; } cand_mv_t;\nint_mv g_single_ref_mv;\n"
                 "static int_mv av1_get_ref_mv(const MACROBLOCK *x, int ref_idx) { return g_single_ref_mv; }\n"
                 "static void av1_set_fractional_mv(int_mv *l) { for (int z = 0; z < 3; z++) { l[z].as_mv.row = INVALID_MV_ROW_COL; l[z].as_mv.col = INVALID_MV_ROW_COL; } }\n",
                 "single:helpers")
    for name, val in (("SIMPLE_TRANSLATION", 0), ("OBMC_CAUSAL", 1)):
        if name not in ev.globs:
            ev.define(name, "(%d)" % val)
    text = open(REF + "av1/encoder/motion_search_facade.c").read()
    fn = re.search(r"void av1_single_motion_search\([^;{]*\)\s*\{.*?\n}\n", text, re.S).group(0)
    if before_function:
        before_function(ev, enc)
    ev.load_text(adapt(fn, keep_rd), "motion_search_facade.c:av1_single_motion_search")
    for f in list(pyc):
        ev.funcs.pop(f, None)
    bad = [s for s in ev.skipped if s[0].startswith("motion_search_facade.c") or s[0].startswith("single:")]
    assert not bad, bad
    pred_text = OS.pred_buffer_adaptation(ev)   # upsampled_obmc_pref_error's pred[] typed per bit depth (see gen_ref_eval_obmc_subpel.py)
    return ev, enc, state, pred_text


MESH = [(12, 4), (6, 2), (4, 1), (3, 1)]


def main():
    ev, enc, state, pred_text = setup()
    arrays, cases = {}, []
    mvc = G.synth_mv_costs(23)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    rng = np.random.default_rng(20261501)
    t0 = time.time()
    mesh = [(12, 4), (6, 2), (4, 1), (3, 1)]
    k = 0
    for bd in (8, 10):
        s_, r_ = G.synth_planes(bd, 900 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s_, r_
        hs = G.Harness(ev, bd, s_, r_, mvc)
        ev.load_text(pred_text[bd], "mcomp.c:upsampled_obmc_pref_error")
        specs = [dict(mode="SIMPLE", method="NSTEP", step=3, tree="SUBPEL_TREE", taps="USE_8_TAPS", accurate=1, cand2=1, w=(3, 2), costlist=0, w_h=(16, 16)),
                 dict(mode="SIMPLE", method="DIAMOND", step=4, tree="SUBPEL_TREE_PRUNED", taps="USE_2_TAPS_ORIG", accurate=0, cand2=1, w=(9, 1), costlist=1, w_h=(8, 8)),
                 dict(mode="SIMPLE", method="NSTEP", step=2, tree="SUBPEL_TREE", taps="USE_4_TAPS", accurate=1, cand2=0, w=(1, 0), costlist=0, w_h=(16, 8), search_range=6),
                 dict(mode="SIMPLE", method="BIGDIA", step=3, tree="SUBPEL_TREE_PRUNED_MORE", taps="USE_2_TAPS", accurate=1, cand2=1, w=(2, 5), costlist=1, w_h=(8, 16),
                      mesh_thr=3000),
                 dict(mode="SIMPLE", method="NSTEP", step=5, tree="SUBPEL_TREE", taps="USE_2_TAPS", accurate=1, cand2=0, w=(1, 0), costlist=0, w_h=(16, 16), force_int=1,
                      search_range=0),
                 dict(mode="OBMC", method="NSTEP", step=4, tree="SUBPEL_TREE", taps="USE_8_TAPS", accurate=1, cand2=0, w=(1, 0), costlist=0, w_h=(16, 16)),
                 dict(mode="OBMC", method="DIAMOND", step=5, tree="SUBPEL_TREE", taps="USE_2_TAPS_ORIG", accurate=0, cand2=0, w=(1, 0), costlist=0, w_h=(8, 8), fast_obmc=1)]
        for spec in specs:
            w, h = spec["w_h"]
            sf = dict(search_method=spec["method"], subpel_search_method=spec["tree"], use_accurate_subpel_search=spec["taps"], sadperbit=int(rng.integers(10, 40)),
                      errorperbit=int(rng.integers(30, 100)), force_integer_mv=spec.get("force_int", 0), mesh=mesh, use_fullpel_costlist=spec["costlist"],
                      exhaustive_searches_thresh=spec.get("mesh_thr", C.INT_MAX), obmc_full_pixel_search_level=spec.get("fast_obmc", 0))
            cpi, x = enc.make(hs, bd, W, H, sf, 30, mvc, sizes=((16, 16), (8, 8), (16, 8), (8, 16)))
            for (ww, hh) in ((16, 16), (8, 8), (16, 8), (8, 16)):
                vt = ev.field(ev.get(cpi, "ppi"), "fn_ptr[%d]" % hs.const(G.BSIZE[(ww, hh)]))
                CS.extend_vtable(ev, vt, bd, ww, hh)
                osvf = ("aom_obmc_sub_pixel_variance%dx%d_c" if bd == 8 else "aom_highbd_10_obmc_sub_pixel_variance%dx%d_c") % (ww, hh)
                assert osvf in ev.funcs, osvf
                ev.set(vt, "osvf", R.FuncRef(osvf))
            ev.set(cpi, "mv_search_params.mv_step_param", spec["step"])
            ev.set(cpi, "sf.mv_sf.disable_second_mv", 1)
            ev.set(cpi, "sf.mv_sf.use_accurate_subpel_search", hs.const(spec["taps"]) if spec["accurate"] else 0)
            bx, by = int(rng.integers(1, (W - w) // 8)) * 8, int(rng.integers(1, (H - h) // 8)) * 8
            lim = G.limits(bx, by, w, h, 30)
            for kk, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
                ev.set(x, "mv_limits." + kk, v)
            off = (BORDER + by) * hs.S + BORDER + bx
            ev.set(x, "plane[0].src.buf", hs.srcp.add(off)); ev.set(x, "plane[0].src.stride", hs.S)
            p = "e_mbd.plane[0].pre[0]."
            ev.set(x, p + "buf", hs.refp.add(off)); ev.set(x, p + "buf0", hs.refp.add(off)); ev.set(x, p + "stride", hs.S); ev.set(x, p + "width", W); ev.set(x, p + "height", H)
            ev.set(x, "e_mbd.mi_row", by // 4); ev.set(x, "e_mbd.mi_col", bx // 4)
            ev.set(enc.mi, "ref_frame[0]", 1); ev.set(enc.mi, "ref_frame[1]", -1)
            obmc = spec["mode"] == "OBMC"
            ev.set(enc.mi, "motion_mode", 1 if obmc else 0)
            refmv = rng.integers(-48, 49, 2).tolist()
            g = ev.globs["g_single_ref_mv"]
            ev.set(g, "as_mv.row", refmv[0]); ev.set(g, "as_mv.col", refmv[1])
            mimv = rng.integers(-40, 41, 2).tolist()
            ev.set(enc.mi, "mv[0].as_mv.row", mimv[0]); ev.set(enc.mi, "mv[0].as_mv.col", mimv[1])
            cand2 = rng.integers(-7, 8, 2).tolist() if spec["cand2"] else None
            state.update(cand2=cand2, w0=spec["w"][0], w1=spec["w"][1])
            rec = dict(k=k, bd=bd, w=w, h=h, bx=bx, by=by, limits=list(lim), ref_mv=refmv, mi_mv=mimv, cand2=cand2, sadperbit=sf["sadperbit"], errorperbit=sf["errorperbit"],
                       **{kk: v for kk, v in spec.items() if kk not in ("w_h", "cand2", "w")}, weights=list(spec["w"]))
            if obmc:
                srcpl = arrays["src%d" % bd]
                mx = (1 << bd) - 1
                sblk = srcpl[BORDER + by:BORDER + by + h, BORDER + bx:BORDER + bx + w].astype(np.int64)
                om = np.full((h, w), 4096, np.int64)
                om[:h // 2, :] = (np.linspace(36, 64, h // 2).astype(np.int64)[:, None]) * 64
                om[:, :w // 2] = np.minimum(om[:, :w // 2], (np.linspace(34, 64, w // 2).astype(np.int64)[None, :]) * 64)
                nb = np.clip(sblk + rng.integers(-(10 << (bd - 8)), (10 << (bd - 8)) + 1, (h, w)), 0, mx)
                ws = sblk * 4096 - nb * (4096 - om)
                ev.set(x, "obmc_buffer.wsrc", ev.array(ws.ravel().astype(np.int64), "int32_t")); ev.set(x, "obmc_buffer.mask", ev.array(om.ravel(), "int32_t"))
                arrays["ws%d" % k], arrays["om%d" % k] = ws.astype(np.int32), om.astype(np.int32)
            ev.set(x, "pred_sse[1]", 0)
            rate = ev.array([0], "int")
            best = ev.new("int_mv")
            t1 = time.time()
            ev.call("single_ms", cpi, x, hs.const(G.BSIZE[(w, h)]), 0, rate, spec.get("search_range", C.INT_MAX), None, best, None)
            rec.update(best_mv=[ev.get(best, "as_mv.row"), ev.get(best, "as_mv.col")], rate_mv=rate.buf[0], pred_sse=int(ev.get(x, "pred_sse[1]")))
            cases.append(rec)
            print(k, spec["mode"], bd, w, h, spec["method"], rec["best_mv"], rec["rate_mv"], rec["pred_sse"], "%.0f s (%.0f)" % (time.time() - t1, time.time() - t0), flush=True)
            k += 1
    meta = dict(border=BORDER, width=W, height=H, mesh=mesh, generated_by="tests/golden/gen_ref_eval_single_caller.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_single_caller.npz"), **arrays)
    print("wrote ref_eval_single_caller.npz: %d cases" % len(cases))


if __name__ == "__main__":
    main()
