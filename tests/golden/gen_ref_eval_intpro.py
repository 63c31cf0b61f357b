#!/usr/bin/env python3
"""Golden vectors for av1_int_pro_motion_estimation AS IT IS WRITTEN (av1/encoder/mcomp.c:1897-2105), obtained by interpreting the function with
aom_int_pro_row_c / aom_int_pro_col_c / aom_vector_var_c (aom_dsp/avg.c:536-581) and the SAD members of cpi->ppi->fn_ptr[bsize] under it (build
container only; tests/golden/ref_c_eval.py, views of gen_ref_eval_composites.py).

Supplied as inputs / adaptations (frame plumbing and the evaluator's memory model):
  * av1_get_scaled_ref_frame returns NULL (an unscaled reference); the rtcd names aom_int_pro_row / _col / aom_vector_var are the C versions;
  * int_mv is a struct holding as_mv (as_fullmv aliases it): `best_int_mv->as_int != 0` is written on the two members (mv.h:26-34: the same bits).

Output: tests/golden/ref_eval_intpro.npz.
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

REF = G.REF
W, H, BORDER = G.W, G.H, G.BORDER
I32, PTR = R.I32, R.PTR


G.BSIZE.update({(16, 32): "BLOCK_16X32", (64, 64): "BLOCK_64X64"})     # (the harness's name table holds the sizes of the earlier generators)


def adapt(text):
    t = text.replace("unsigned int av1_int_pro_motion_estimation(", "unsigned int int_pro_me(")
    t = t.replace("struct buf_2d backup_yv12[MAX_MB_PLANE] = { { 0, 0, 0, 0, 0 } };", "struct buf_2d backup_yv12[MAX_MB_PLANE];")
    t, n = re.subn(r"best_int_mv->as_int != 0", "(best_int_mv->as_mv.row != 0 || best_int_mv->as_mv.col != 0)", t)
    assert n == 1 and "as_int" not in t
    return t


def main():
    ev = G.make_evaluator()
    C.view(ev, "int_mv", [("as_mv", ev.structs["mv"])])
    ev.define("as_fullmv", "as_mv")
    R.ALIASED_STRUCTS.add(frozenset(("mv", "fullpel_mv")))
    ev.load(REF + "av1/common/common_data.c")
    C.grab(ev, "av1/common/mv.h", "convert_fullmv_to_mv")
    for n in ("aom_int_pro_row", "aom_int_pro_col", "aom_vector_var"):
        ev.define(n, n + "_c")
    ev.load(REF + "aom_dsp/avg.c")
    enc = C.Encoder(ev)
    pyc = ev.interp.pycalls
    pyc["av1_get_scaled_ref_frame"] = lambda it, a: (None, PTR)
    pyc["av1_setup_pre_planes"] = lambda it, a: (None, R.VOID)
    text = open(REF + "av1/encoder/mcomp.c").read()
    vm = re.search(r"static int vector_match\([^;{]*\)\s*\{.*?\n}\n", text, re.S).group(0)
    fn = re.search(r"unsigned int av1_int_pro_motion_estimation\([^;{]*\)\s*\{.*?\n}\n", text, re.S).group(0)
    ev.load_text(vm, "mcomp.c:vector_match")
    ev.load_text(adapt(fn), "mcomp.c:av1_int_pro_motion_estimation")
    for f in list(pyc):
        ev.funcs.pop(f, None)
    bad = [s for s in ev.skipped if s[0].startswith("mcomp.c:")]
    assert not bad, bad
    arrays, cases = {}, []
    mvc = G.synth_mv_costs(29)
    rng = np.random.default_rng(20261111)
    t0 = time.time()
    k = 0
    sizes = ((16, 16), (32, 32), (32, 16), (16, 32), (64, 64))
    for bd in (8, 10):
        s_, r_ = G.synth_planes(bd, 1300 + bd)
        if bd == 8:
            # a second reference whose content is the source displaced by a known vector plus noise, so that the projections find something
            src_in = s_[BORDER:BORDER + H, BORDER:BORDER + W].astype(np.int64)
            pad = np.pad(src_in, 16, mode="reflect")
            moved = pad[16 - 5:16 - 5 + H, 16 + 9:16 + 9 + W] + rng.integers(-6, 7, (H, W))
            r_ = np.pad(np.clip(moved, 0, 255).astype(r_.dtype), BORDER, mode="edge")     # (borders replicated, like every plane of these fixtures)
            assert r_.shape == s_.shape
        arrays["src%d" % bd], arrays["ref%d" % bd] = s_, r_
        hs = G.Harness(ev, bd, s_, r_, mvc)
        cpi, x = enc.make(hs, bd, W, H, dict(search_method="NSTEP"), 30, mvc, sizes=sizes)
        for (w, h) in (sizes if bd == 8 else sizes[:2]):
            for trial in range(4 if (bd == 8 and w * h <= 1024) else (2 if bd == 8 else 1)):
                bx = int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = int(rng.integers(0, (H - h) // 4 + 1)) * 4
                if trial == 1:
                    bx, by = 0, 0                       # the window leaves the frame: the border is read (the reference reads it too)
                lim = list(G.limits(bx, by, w, h, 30))
                if trial == 2:
                    lim = [-2, 3, -1, 2]                # a tight window: the final clamp_mv moves the vector
                for kk, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
                    ev.set(x, "mv_limits." + kk, v)
                off = (BORDER + by) * hs.S + BORDER + bx
                ev.set(x, "plane[0].src.buf", hs.srcp.add(off)); ev.set(x, "plane[0].src.stride", hs.S)
                p = "e_mbd.plane[0].pre[0]."
                ev.set(x, p + "buf", hs.refp.add(off)); ev.set(x, p + "buf0", hs.refp.add(off)); ev.set(x, p + "stride", hs.S)
                ev.set(x, p + "width", W); ev.set(x, p + "height", H)
                ev.set(x, "e_mbd.mi_row", by // 4); ev.set(x, "e_mbd.mi_col", bx // 4)
                ev.set(enc.mi, "ref_frame[0]", 1); ev.set(enc.mi, "ref_frame[1]", -1)
                ev.set(enc.mi, "mv[0].as_mv.row", 77); ev.set(enc.mi, "mv[0].as_mv.col", -77)
                refmv = rng.integers(-64, 65, 2).tolist() if trial != 3 else [-(1023 * 8 + 40), 1023 * 8 + 24]
                rm = ev.new("MV")
                ev.set(rm, "row", refmv[0]); ev.set(rm, "col", refmv[1])
                t1 = time.time()
                sad = ev.call("int_pro_me", cpi, x, hs.const(G.BSIZE[(w, h)]), by // 4, bx // 4, rm)
                rec = dict(k=k, bd=bd, w=w, h=h, bx=bx, by=by, limits=lim, ref_mv=refmv, best_sad=int(sad),
                           mv=[int(ev.get(enc.mi, "mv[0].as_mv.row")), int(ev.get(enc.mi, "mv[0].as_mv.col"))])
                cases.append(rec)
                print(rec, "%.0f s (%.0f)" % (time.time() - t1, time.time() - t0), flush=True)
                k += 1
    meta = dict(border=BORDER, width=W, height=H, generated_by="tests/golden/gen_ref_eval_intpro.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_intpro.npz"), **arrays)
    print("wrote ref_eval_intpro.npz: %d cases" % len(cases))


if __name__ == "__main__":
    main()
