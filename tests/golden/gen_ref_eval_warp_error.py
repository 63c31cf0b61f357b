#!/usr/bin/env python3
"""Golden vectors of the global-motion search's model error from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_warp_error.npz   av1_warp_error (av1/encoder/global_motion.c:128-224) -- av1_get_shear_params on the model (its four values and its verdict
                            recorded), then 32 x 32 tiles of av1_[highbd_]warp_affine_c against the frame through error_measure_lut, only where the
                            segment map holds inliers, INT64_MAX once the sum passes best_error -- and av1_segmented_frame_error
                            (av1/common/warped_motion.c:400-460,687-760): 8 / 10 / 12 bits, frames whose last tile column / row is partial (also not a
                            multiple of 8), sparse segment maps, 4:2:0 geometry, a region that starts inside the frame, invalid models.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402,F401
from gen_ref_eval_golden import evaluator, save, REF  # noqa: E402


def main():
    ev = evaluator(["aom_dsp/flow_estimation/flow_estimation.h", "av1/common/filter.h", "av1/common/convolve.h", "av1/common/mv.h"])
    for n in ("av1_warp_affine", "av1_highbd_warp_affine", "av1_calc_frame_error"):     # the rtcd names -> the C versions
        ev.define(n, n + "_c")
    for f in ("av1/common/warped_motion.h", "av1/common/warped_motion.c", "av1/encoder/global_motion.h", "av1/encoder/global_motion.c"):
        ev.load(REF + f)
    rng = np.random.default_rng(20261110)
    arrays, cases = {}, []
    INT64_MAX = (1 << 63) - 1
    k = 0
    for bd, (W, H) in ((8, (80, 72)), (10, (76, 70)), (12, (72, 40))):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        # a smooth-ish picture plus noise, and the "current" frame = a shifted copy plus noise, so that the models' errors differ
        base = (rng.integers(0, mx + 1, (H // 8 + 2, W // 8 + 2)).repeat(8, 0).repeat(8, 1)[:H + 8, :W + 8]).astype(np.int64)
        ref = np.clip(base[4:H + 4, 4:W + 4] + rng.integers(-(mx >> 4), (mx >> 4) + 1, (H, W)), 0, mx)
        cur = np.clip(base[2:H + 2, 5:W + 5] + rng.integers(-(mx >> 4), (mx >> 4) + 1, (H, W)), 0, mx)
        cur[:3, :7] = mx - ref[:3, :7]                                  # large differences: the far ends of the table
        arrays["ref%d" % bd], arrays["cur%d" % bd] = ref.astype(np.uint16), cur.astype(np.uint16)
        Rf, Cf = ev.array(ref.ravel(), ct), ev.array(cur.ravel(), ct)
        sw, sh = (W + 31) // 32, (H + 31) // 32
        trials = 7 if bd == 8 else 5
        for trial in range(trials):
            mat = [int(rng.integers(-3 << 16, 3 << 16)), int(rng.integers(-3 << 16, 3 << 16)),
                   (1 << 16) + int(rng.integers(-(1 << 11), 1 << 11)), int(rng.integers(-(1 << 11), 1 << 11)),
                   int(rng.integers(-(1 << 11), 1 << 11)), (1 << 16) + int(rng.integers(-(1 << 11), 1 << 11))]
            wmtype = 3
            if trial == 0:
                mat, wmtype = [0, 0, 1 << 16, 0, 0, 1 << 16], 0            # IDENTITY
            if trial == 1:
                mat, wmtype = [mat[0], mat[1], 1 << 16, 0, 0, 1 << 16], 1  # TRANSLATION
            if trial == 2:
                mat[4], mat[5], wmtype = -mat[3], mat[2], 2                # ROTZOOM
            if trial == 5:
                mat[3] = 1 << 14                                           # a shear av1_get_shear_params refuses
            if trial == 6:
                mat[2] = 0                                                 # is_affine_valid fails
            seg = np.ones((sh, sw), np.uint8)
            if trial % 2 == 1:
                seg = (rng.random((sh, sw)) < 0.6).astype(np.uint8)
                seg[0, 0] = 1
            ss = 1 if trial == 3 else 0
            p_col, p_row, pw, ph = (0, 0, W, H) if trial != 4 else (32, 32, W - 32, H - 32)
            wm = ev.new("WarpedMotionParams")
            for i, v in enumerate(mat):
                ev.set(wm, "wmmat[%d]" % i, v)
            ev.set(wm, "wmtype", wmtype)
            S = ev.array(seg.ravel(), "uint8_t")
            err = ev.call("av1_warp_error", wm, int(bd > 8), bd, Rf, W, H, W, Cf, p_col, p_row, pw, ph, W, ss, ss, INT64_MAX, S, sw)
            shear = [int(ev.get(wm, f)) for f in ("alpha", "beta", "gamma", "delta")]
            case = {"k": k, "bd": bd, "W": W, "H": H, "mat": mat, "wmtype": wmtype, "shear": shear, "valid": int(err != INT64_MAX), "ss": ss,
                    "p_col": p_col, "p_row": p_row, "pw": pw, "ph": ph, "seg": seg.ravel().tolist(), "seg_stride": sw, "error": str(int(err))}
            if err != INT64_MAX and trial in (2, 3):
                # the early exit: the same call with a bound below the total
                case["best_error"] = str(int(err) // 2)
                case["error_bounded"] = str(int(ev.call("av1_warp_error", wm, int(bd > 8), bd, Rf, W, H, W, Cf, p_col, p_row, pw, ph, W, ss, ss,
                                                        int(err) // 2, S, sw)))
                assert case["error_bounded"] == str(INT64_MAX)
            if trial < 2:
                case["frame_error"] = str(int(ev.call("av1_segmented_frame_error", int(bd > 8), bd, Rf, W, Cf, W, H, W, S, sw)))
            cases.append(case)
            print(case["k"], bd, trial, case["valid"], case["error"], case.get("frame_error"), flush=True)
            k += 1
    save("ref_eval_warp_error.npz", arrays, cases)


if __name__ == "__main__":
    main()
