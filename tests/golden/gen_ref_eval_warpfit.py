#!/usr/bin/env python3
"""Golden vectors of the local warp model's fit from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_warpfit.npz   av1_selectSamples (av1/common/mvref_common.c:1083-1104: the samples whose motion differs from the block's MV by no more than
                         clamp(max(bw, bh), 16, 112) in the L1 norm, compacted in place, at least one kept) and av1_find_projection
                         (av1/common/warped_motion.c:894-1015: find_affine_int -- the 2 x 2 least-squares systems with their LS_* fixed-point products,
                         resolve_divisor_64 on the determinant, the clamped model and its translation -- then av1_get_shear_params' verdict): random
                         neighbourhoods of 1 .. 8 samples around blocks of 8x8 .. 128x128 at positions across a 4K frame, motion from a random affine
                         model plus noise, outliers, duplicated and collinear samples (singular systems), extreme MVs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402,F401
from gen_ref_eval_golden import evaluator, save, REF  # noqa: E402
from gen_ref_eval_yrd import cut  # noqa: E402

SIZES = [(8, 8), (16, 16), (32, 32), (64, 64), (128, 128), (16, 8), (8, 16), (32, 16), (16, 64), (64, 32)]


def make_evaluator():
    ev = evaluator(["aom_dsp/flow_estimation/flow_estimation.h", "av1/common/filter.h", "av1/common/convolve.h", "av1/common/mv.h", "av1/common/common_data.h"])
    for n in ("av1_warp_affine", "av1_highbd_warp_affine", "av1_calc_frame_error"):
        ev.define(n, n + "_c")
    for f in ("av1/common/warped_motion.h", "av1/common/warped_motion.c"):
        ev.load(REF + f)
    ev.define("LEAST_SQUARES_SAMPLES_MAX", "(8)")
    ev.load_text(cut(open(REF + "av1/common/mvref_common.c").read(), "uint8_t av1_selectSamples("), "mvref_common.c:av1_selectSamples")
    return ev


def bsize_of(ev, w, h):
    return ev.globs["BLOCK_%dX%d" % (w, h)].buf[0]


def main():
    ev = make_evaluator()
    assert "av1_find_projection" in ev.funcs and "av1_selectSamples" in ev.funcs and "find_affine_int" in ev.funcs
    rng = np.random.default_rng(20261008)
    cases = []
    for k in range(260):
        w, h = SIZES[k % len(SIZES)]
        n = int(rng.integers(1, 9))
        mi_row, mi_col = int(rng.integers(0, 540 - h // 4)), int(rng.integers(0, 960 - w // 4))
        mv = [int(rng.integers(-200, 201)), int(rng.integers(-200, 201))]      # (row, col), 1/8 pel
        if k % 13 == 0:
            mv = [int(rng.choice([-2000, 2000])), int(rng.integers(-200, 201))]
        # neighbour-block centres relative to the block's top-left pixel, 1/8 pel (av1_findSamples): above / left of the block, a few inside-ish
        pts = np.zeros((n, 2), np.int64)
        pts[:, 0] = rng.integers(-8 * 64, 8 * (w + 32), n)
        pts[:, 1] = rng.integers(-8 * 64, 8 * (h + 32), n)
        # their positions in the reference: an affine motion around the block's MV plus noise
        a = np.array([[1.0 + rng.normal(0, 0.04), rng.normal(0, 0.04)], [rng.normal(0, 0.04), 1.0 + rng.normal(0, 0.04)]])
        ctr = np.array([w * 4.0, h * 4.0])
        pin = (pts - ctr) @ a.T + ctr + np.array([mv[1], mv[0]]) + rng.normal(0, 3 if k % 3 else 12, (n, 2))
        pin = np.rint(pin).astype(np.int64)
        if k % 7 == 3 and n > 1:
            pin[0] += rng.integers(150, 400, 2) * rng.choice([-1, 1], 2)        # an outlier selectSamples drops (or LS_MV_MAX does)
        if k % 11 == 5 and n > 1:
            pts[1:] = pts[0]; pin[1:] = pin[0]                                    # one point repeated: a singular system
        if k % 17 == 6 and n > 2:
            pts[:, 1] = pts[:, 0] * 2 + 5                                         # collinear samples
        P, Q = ev.array(pts.ravel(), "int"), ev.array(pin.ravel(), "int")
        m = ev.new("MV")
        ev.set(m, "row", mv[0]); ev.set(m, "col", mv[1])
        bs = bsize_of(ev, w, h)
        sel = int(ev.call("av1_selectSamples", m, P, Q, n, bs)) if n > 1 else n   # (the callers skip the call for a single sample)
        sp, sq = [int(v) for v in P.buf], [int(v) for v in Q.buf]
        wm = ev.new("WarpedMotionParams")
        ev.set(wm, "wmtype", 3)
        for i, v in enumerate((0, 0, 1 << 16, 0, 0, 1 << 16)):
            ev.set(wm, "wmmat[%d]" % i, v)
        bad = int(ev.call("av1_find_projection", sel, P, Q, bs, mv[0], mv[1], wm, mi_row, mi_col))
        rec = dict(k=k, w=w, h=h, n=n, mi_row=mi_row, mi_col=mi_col, mv=mv, pts=pts.ravel().tolist(), pts_inref=pin.ravel().tolist(), selected=sel,
                   sel_pts=sp, sel_pts_inref=sq, invalid=bad, mat=[int(ev.get(wm, "wmmat[%d]" % i)) for i in range(6)],
                   shear=[int(ev.get(wm, f)) for f in ("alpha", "beta", "gamma", "delta")])
        cases.append(rec)
        if k % 20 == 0:
            print(k, w, h, n, sel, bad, rec["mat"], flush=True)
    print("invalid:", sum(c["invalid"] for c in cases), "dropped samples:", sum(c["selected"] < c["n"] for c in cases))
    save("ref_eval_warpfit.npz", {}, cases)


if __name__ == "__main__":
    main()
