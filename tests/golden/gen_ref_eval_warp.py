#!/usr/bin/env python3
"""Golden vectors of the warped-motion predictor from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_warp.npz   av1_warp_affine_c / av1_highbd_warp_affine_c (av1/common/warped_motion.c:264-393,538-675) with conv_params->is_compound == 0
                      (get_conv_params_no_round), the shear parameters from the reference's own av1_get_shear_params (:212-259) on
                      random valid affine models: 8 / 10 / 12 bits, luma and 4:2:0 chroma geometry, blocks 8x8 .. 32x16, blocks whose
                      footprint leaves the frame on every side (the clamps of both passes).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def shear_of(mat):
    """Shear parameters of an affine model as av1_get_shear_params derives them (spec 7.11.3.6) -- here only a source of VALID inputs for the
    function under test, which takes alpha .. delta as arguments: near-identity models, rounded to WARP_PARAM_REDUCE_BITS."""
    def red(v):
        v = int(np.clip(v, -32768, 32767))
        r = (abs(v) + 32) >> 6
        return (r if v >= 0 else -r) * 64
    alpha, beta = mat[2] - (1 << 16), mat[3]
    gamma = int(round(mat[4] * 65536.0 / mat[2]))
    delta = mat[5] - int(round(mat[3] * mat[4] / float(mat[2]))) - (1 << 16)
    a, b, g, d = red(alpha), red(beta), red(gamma), red(delta)
    assert 4 * abs(a) + 7 * abs(b) < (1 << 16) and 4 * abs(g) + 4 * abs(d) < (1 << 16)   # is_affine_shear_allowed (:197-210)
    return a, b, g, d


def main():
    ev = evaluator(["av1/common/filter.h", "av1/common/convolve.h", "av1/common/mv.h", "av1/common/warped_motion.h", "av1/common/warped_motion.c"])
    rng = np.random.default_rng(20261102)
    arrays, cases = {}, []
    W, H = 96, 64
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        plane = rng.integers(0, mx + 1, (H, W))
        plane[:6] = np.where(rng.integers(0, 2, (6, W)) > 0, mx, 0)   # extreme rows: the offsets and clips must hold
        arrays["ref%d" % bd] = plane.astype(np.uint16)
        P = ev.array(plane.ravel(), ct)
        cpv = ev.call("get_conv_params_no_round", 0, 0, 0, 0, 0, bd)     # is_compound = 0: round_0 = 3 (5 at 12 bits)
        cp = R.Ptr([cpv], 0, cpv.st)
        for trial in range(14):
            # an affine model near the identity with a translation of a few pixels (WARPEDMODEL_PREC_BITS = 16)
            mat = [int(rng.integers(-6 << 16, 6 << 16)), int(rng.integers(-6 << 16, 6 << 16)),
                   (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12)), int(rng.integers(-(1 << 12), 1 << 12)),
                   int(rng.integers(-(1 << 12), 1 << 12)), (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12))]
            if trial % 5 == 4:
                mat[int(rng.integers(0, 2))] += int(rng.choice([-1, 1])) * (70 << 16)   # far outside the frame: the clamps of both passes
            if trial == 0:
                mat = [0, 0, 1 << 16, 0, 0, 1 << 16]                                    # the identity: the prediction is the reference block
            alpha, beta, gamma, delta = shear_of(mat)
            ssx = ssy = 1 if trial % 3 == 2 else 0                                      # 4:2:0 chroma geometry (the plane passed IS the chroma plane)
            pw, ph = [(8, 8), (16, 8), (8, 16), (16, 16), (32, 16)][trial % 5]
            p_col, p_row = int(rng.integers(0, (W - pw) // 8 + 1)) * 8, int(rng.integers(0, (H - ph) // 8 + 1)) * 8
            if trial % 7 == 3:
                p_col, p_row = 0, 0
            if trial % 7 == 5:
                p_col, p_row = W - pw, H - ph
            M = ev.array(mat, "int32_t")
            dst = ev.array([0] * (pw * ph), ct)
            args = [M, P, W, H, W, dst, p_col, p_row, pw, ph, pw, ssx, ssy]
            if bd > 8:
                args.append(bd)
            args += [cp, alpha, beta, gamma, delta]
            ev.call("av1_warp_affine_c" if bd == 8 else "av1_highbd_warp_affine_c", *args)
            arrays["d%d" % k] = np.asarray(dst.buf, np.uint16)
            cases.append({"k": k, "bd": bd, "mat": mat, "shear": [alpha, beta, gamma, delta], "p_col": p_col, "p_row": p_row, "pw": pw, "ph": ph, "ss": ssx,
                          "round_0": 5 if bd == 12 else 3})
            k += 1
    save("ref_eval_warp.npz", arrays, cases)
    # ---- compound: the first reference into the CONV_BUF (do_average 0), the second blended in (do_average 1): plain average and distance weights
    arrays2, cases2 = {}, []
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        planes = [rng.integers(0, mx + 1, (H, W)) for _ in range(2)]
        planes[0][:6] = np.where(rng.integers(0, 2, (6, W)) > 0, mx, 0)
        for r in range(2):
            arrays2["ref%d_%d" % (bd, r)] = planes[r].astype(np.uint16)
        P = [ev.array(pl.ravel(), ct) for pl in planes]
        for trial in range(8):
            pw, ph = [(8, 8), (16, 8), (8, 16), (16, 16)][trial % 4]
            p_col, p_row = int(rng.integers(0, (W - pw) // 8 + 1)) * 8, int(rng.integers(0, (H - ph) // 8 + 1)) * 8
            if trial == 5:
                p_col, p_row = 0, 0
            ssx = ssy = 1 if trial % 3 == 2 else 0
            wts = None if trial % 2 == 0 else [(9, 7), (4, 12), (13, 3), (2, 14)][trial // 2]
            mats, shears = [], []
            for r in range(2):
                mat = [int(rng.integers(-6 << 16, 6 << 16)), int(rng.integers(-6 << 16, 6 << 16)), (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12)),
                       int(rng.integers(-(1 << 12), 1 << 12)), int(rng.integers(-(1 << 12), 1 << 12)), (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12))]
                if trial == 6 and r == 1:
                    mat[0] += 70 << 16
                mats.append(mat); shears.append(list(shear_of(mat)))
            buf16 = ev.array([0] * (pw * ph), "uint16_t")
            dst = ev.array([0] * (pw * ph), ct)
            for r in range(2):
                cpv = ev.call("get_conv_params_no_round", r, 0, buf16, pw, 1, bd)
                cp = R.Ptr([cpv], 0, cpv.st)
                if wts:
                    ev.set(cp, "use_dist_wtd_comp_avg", 1); ev.set(cp, "fwd_offset", wts[0]); ev.set(cp, "bck_offset", wts[1])
                args = [ev.array(mats[r], "int32_t"), P[r], W, H, W, dst, p_col, p_row, pw, ph, pw, ssx, ssy]
                if bd > 8:
                    args.append(bd)
                args += [cp] + shears[r]
                ev.call("av1_warp_affine_c" if bd == 8 else "av1_highbd_warp_affine_c", *args)
                if r == 0:
                    arrays2["c%d" % k] = np.asarray(buf16.buf, np.uint16).copy()
            arrays2["d%d" % k] = np.asarray(dst.buf, np.uint16)
            cases2.append({"k": k, "bd": bd, "mat": mats, "shear": shears, "p_col": p_col, "p_row": p_row, "pw": pw, "ph": ph, "ss": ssx, "weights": wts,
                           "round_0": 5 if bd == 12 else 3})
            k += 1
    save("ref_eval_warp_compound.npz", arrays2, cases2)


if __name__ == "__main__":
    main()
