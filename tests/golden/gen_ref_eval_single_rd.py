#!/usr/bin/env python3
"""Golden vectors for the RD form of av1_single_motion_search's second-MV decision (av1/encoder/motion_search_facade.c:367-430 with
sf.mv_sf.disable_second_mv == 0), obtained by interpreting the WHOLE function with that branch kept (build container only;
tests/golden/ref_c_eval.py; the rest of the function's setup is gen_ref_eval_single_caller.py's): try_second, the first candidate's rd
(mbmi->mv[0] = best_mv, predictor, subtract, av1_estimate_txfm_yrd with max_txsize_rect_lookup[bsize], RDCOST(x->rdmult, mv rate + rate, dist)),
the sub-pel search from second_best_mv when av1_is_subpelmv_in_range, its rd, `tmp_rd < rd`, x->pred_sse[ref].

Supplied as inputs / adaptations, beyond gen_ref_eval_single_caller.py's:
  * av1_enc_build_inter_predictor and av1_subtract_plane do nothing, and av1_estimate_txfm_yrd returns RD_STATS that are a FIXED FUNCTION OF
    mbmi->mv[0] at the time the predictor was asked for (scripted_stats below; the fixture records every call's MV, tx_size and ref_best_rd):
    what the three functions compute is pinned elsewhere (the convolve fixtures, ref_eval_golden.npz's subtract, ref_eval_yrd.npz) -- here the
    caller's sequencing around them is what runs.  `orig_dst` (the predictor's destination argument) is cut, NULL passed instead.
  * RD_STATS is a view with rate and dist; av1_init_rd_stats zeroes them.

Output: tests/golden/ref_eval_single_rd.npz.
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
import gen_ref_eval_single_caller as SC  # noqa: E402
import gen_ref_eval_yrd as Y  # noqa: E402  (cut())

W, H, BORDER = G.W, G.H, G.BORDER
I32, I64 = R.I32, R.I64


def scripted_stats(row, col):
    """the stand-in for av1_estimate_txfm_yrd's result at a motion vector (1/8 pel): (rate, dist)"""
    return 300 + (row * 73 + col * 151) % 977, 1500 + (row * 331 + col * 17) % 2903


def main():
    def before(ev, enc):
        C.view(ev, "RD_STATS", [("rate", I32), ("dist", I64)], opaque=False)
        enc.x_t.fields.append(("rdmult", I32))
        for name, val in (("AV1_PROB_COST_SHIFT", 9), ("RDDIV_BITS", 7), ("INT64_MAX", "0x7fffffffffffffffLL")):    # cost.h:25, rd.h:28
            ev.define(name, "(%s)" % val)
        # TX_SIZE is a UENUM1BYTE enum the evaluator skipped: enumerators in declaration order (av1/common/enums.h), then the table's own text
        for i, n in enumerate(("TX_4X4", "TX_8X8", "TX_16X16", "TX_32X32", "TX_64X64", "TX_4X8", "TX_8X4", "TX_8X16", "TX_16X8", "TX_16X32", "TX_32X16", "TX_32X64",
                               "TX_64X32", "TX_4X16", "TX_16X4", "TX_8X32", "TX_32X8", "TX_16X64", "TX_64X16")):
            ev.define(n, "(%d)" % i)
        ev.define("TX_SIZE", "int")
        ev.load_text(Y.cut(open(G.REF + "av1/common/common_data.h").read(), "static const TX_SIZE max_txsize_rect_lookup[BLOCK_SIZES_ALL] ="),
                     "common_data.h:max_txsize_rect_lookup")
        ev.load_text(re.search(r"#define RDCOST\(RM, R, D\).*?\n\n", open(G.REF + "av1/encoder/rd.h").read(), re.S).group(0) +
                     "static void av1_init_rd_stats(RD_STATS *s) { s->rate = 0; s->dist = 0; }\n", "rd.h:RDCOST")

    ev, enc, state, _ = SC.setup(keep_rd=True, before_function=before)
    pyc = ev.interp.pycalls
    log = {}

    def predictor(it, a):
        log["mv"] = (int(ev.get(enc.mi, "mv[0].as_mv.row")), int(ev.get(enc.mi, "mv[0].as_mv.col")))
        log["events"].append(["pred", *log["mv"]])
        return (None, R.VOID)

    def subtract(it, a):
        log["events"].append(["subtract", int(a[1][0]), int(a[2][0])])
        return (None, R.VOID)

    def yrd(it, a):   # av1_estimate_txfm_yrd(cpi, x, rd_stats, ref_best_rd, bs, tx_size)
        rate, dist = scripted_stats(*log["mv"])
        st = a[2][0].deref()[0]
        st.f["rate"].store(rate, I32); st.f["dist"].store(dist, I64)
        log["events"].append(["yrd", log["mv"][0], log["mv"][1], int(a[3][0]) == (1 << 63) - 1, int(a[4][0]), int(a[5][0])])
        return (0, I64)
    pyc["av1_enc_build_inter_predictor"], pyc["av1_subtract_plane"], pyc["av1_estimate_txfm_yrd"] = predictor, subtract, yrd
    for f in ("av1_enc_build_inter_predictor", "av1_subtract_plane", "av1_estimate_txfm_yrd"):
        ev.funcs.pop(f, None)
    arrays, cases = {}, []
    mvc = G.synth_mv_costs(23)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    rng = np.random.default_rng(20261006)
    t0 = time.time()
    k = 0
    sizes = ((16, 16), (8, 8), (16, 8), (8, 16))
    for bd in (8, 10):
        s_, r_ = G.synth_planes(bd, 900 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s_, r_
        hs = G.Harness(ev, bd, s_, r_, mvc)
        specs = [dict(method="NSTEP", step=3, tree="SUBPEL_TREE", taps="USE_8_TAPS", cand2=1, w=(3, 2), costlist=0),
                 dict(method="DIAMOND", step=4, tree="SUBPEL_TREE_PRUNED", taps="USE_2_TAPS", cand2=0, w=(1, 0), costlist=1),
                 dict(method="NSTEP", step=2, tree="SUBPEL_TREE_PRUNED_MORE", taps="USE_4_TAPS", cand2=1, w=(2, 5), costlist=1),
                 dict(method="NSTEP", step=4, tree="SUBPEL_TREE", taps="USE_2_TAPS", cand2=0, w=(1, 0), costlist=0),
                 dict(method="BIGDIA", step=3, tree="SUBPEL_TREE", taps="USE_8_TAPS", cand2=0, w=(1, 0), costlist=0),      # (second_best_mv stays invalid)
                 dict(method="NSTEP", step=3, tree="SUBPEL_TREE", taps="USE_8_TAPS", cand2=0, w=(1, 0), costlist=0, disable_second_mv=2)]   # no second search
        for spec in specs:
            for (w, h) in (sizes if spec["method"] == "NSTEP" and "disable_second_mv" not in spec else sizes[:1]):
                sf = dict(search_method=spec["method"], subpel_search_method=spec["tree"], use_accurate_subpel_search=spec["taps"], sadperbit=int(rng.integers(10, 40)),
                          errorperbit=int(rng.integers(30, 100)), force_integer_mv=0, mesh=SC.MESH, use_fullpel_costlist=spec["costlist"],
                          exhaustive_searches_thresh=C.INT_MAX, obmc_full_pixel_search_level=0)
                cpi, x = enc.make(hs, bd, W, H, sf, 30, mvc, sizes=sizes)
                for (ww, hh) in sizes:
                    CS.extend_vtable(ev, ev.field(ev.get(cpi, "ppi"), "fn_ptr[%d]" % hs.const(G.BSIZE[(ww, hh)])), bd, ww, hh)
                ev.set(cpi, "mv_search_params.mv_step_param", spec["step"])
                ev.set(cpi, "sf.mv_sf.disable_second_mv", spec.get("disable_second_mv", 0))
                ev.set(cpi, "sf.mv_sf.use_accurate_subpel_search", hs.const(spec["taps"]))
                rdmult = int(rng.integers(100, 3000))
                ev.set(x, "rdmult", rdmult)
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
                ev.set(enc.mi, "motion_mode", 0)
                refmv = rng.integers(-48, 49, 2).tolist()
                g = ev.globs["g_single_ref_mv"]
                ev.set(g, "as_mv.row", refmv[0]); ev.set(g, "as_mv.col", refmv[1])
                mimv = rng.integers(-40, 41, 2).tolist()
                ev.set(enc.mi, "mv[0].as_mv.row", mimv[0]); ev.set(enc.mi, "mv[0].as_mv.col", mimv[1])
                cand2 = rng.integers(-7, 8, 2).tolist() if spec["cand2"] else None
                state.update(cand2=cand2, w0=spec["w"][0], w1=spec["w"][1])
                log.clear(); log["events"] = []
                ev.set(x, "pred_sse[1]", 0)
                rate = ev.array([0], "int")
                best = ev.new("int_mv")
                t1 = time.time()
                ev.call("single_ms", cpi, x, hs.const(G.BSIZE[(w, h)]), 0, rate, C.INT_MAX, None, best, None)
                rec = dict(k=k, bd=bd, w=w, h=h, bx=bx, by=by, limits=list(lim), ref_mv=refmv, cand2=cand2, sadperbit=sf["sadperbit"], errorperbit=sf["errorperbit"],
                           rdmult=rdmult, weights=list(spec["w"]), bsize=int(hs.const(G.BSIZE[(w, h)])), **{kk: v for kk, v in spec.items() if kk not in ("cand2", "w")},
                           best_mv=[ev.get(best, "as_mv.row"), ev.get(best, "as_mv.col")], rate_mv=rate.buf[0], pred_sse=int(ev.get(x, "pred_sse[1]")),
                           mbmi_mv_after=[int(ev.get(enc.mi, "mv[0].as_mv.row")), int(ev.get(enc.mi, "mv[0].as_mv.col"))], events=log["events"])
                cases.append(rec)
                print(k, bd, w, h, spec["method"], rec["best_mv"], rec["rate_mv"], rec["pred_sse"], len(rec["events"]), "%.0f s (%.0f)" % (time.time() - t1, time.time() - t0),
                      flush=True)
                k += 1
    meta = dict(border=BORDER, width=W, height=H, mesh=SC.MESH, generated_by="tests/golden/gen_ref_eval_single_rd.py",
                scripted_stats="rate = 300 + (row * 73 + col * 151) % 977, dist = 1500 + (row * 331 + col * 17) % 2903", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_single_rd.npz"), **arrays)
    print("wrote ref_eval_single_rd.npz: %d cases" % len(cases))


if __name__ == "__main__":
    main()
