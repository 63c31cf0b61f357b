#!/usr/bin/env python3
"""Golden vectors of the two loop-restoration filters as the search applies them, from the interpreted reference (build container only):

  ref_eval_lr_apply.npz   (a) av1_apply_selfguided_restoration_c (av1/common/restoration.c:917-956, with av1_decode_xq and the filter) on units with
                          their 3-pixel surround, 8 / 10 / 12 bits, the three radius combinations, xqd at the corners of its range;
                          (b) the Wiener filter av1_[highbd_]wiener_convolve_add_src_c (av1/common/convolve.c:1093-1257): its two passes
                          convolve_add_src_horiz_hip / _vert_hip (and the highbd pair) interpreted where they lie and composed as the function
                          composes them for steps of 16 -- the function itself locates its kernels through the ADDRESS of the filter array
                          (get_filter_base masks the pointer to a 256-byte boundary), which has no meaning in the evaluator's pointer model;
                          with x_step_q4 = y_step_q4 = 16 the phase is the filter's own row, i.e. offset 0 of a one-row table.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def main():
    ev = evaluator(["av1/common/filter.h", "av1/common/convolve.h", "av1/common/restoration.h", "av1/common/restoration.c", "aom_dsp/aom_convolve.c",
                    "av1/common/convolve.c"])
    rng = np.random.default_rng(20261108)
    arrays, cases = {}, []
    k = 0
    tmp_n = 2 * 406 * 398      # SGRPROJ_TMPBUF_SIZE / sizeof(int32_t) for 256-pixel units (restoration.h:80-92)
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        # ---- (a) self-guided apply
        for (w, h), idx, xqd in (((32, 24), 3, (-20, 90)), ((17, 9), 12, (0, 95)), ((24, 16), 15, (-96, 0)), ((40, 20), 7, (31, -32)), ((16, 16), 0, (-96, 95))):
            S, Hh = w + 8, h + 6
            img = np.clip(rng.integers(0, mx + 1, (Hh, S)) // 2 + mx // 4, 0, mx).astype(np.int64)
            img[Hh // 2: Hh // 2 + 2, S // 2: S // 2 + 3] = rng.choice([0, mx], (2, 3))
            P = ev.array(img.ravel(), ct)
            dst = ev.array([0] * (w * h), ct)
            tmp = ev.array([0] * tmp_n, "int32_t")
            ev.call("av1_apply_selfguided_restoration_c", P.add(3 * S + 3), w, h, S, idx, ev.array(list(xqd), "int"), dst, w, tmp, bd, int(bd > 8))
            arrays["img%d" % k] = img.astype(np.uint16)
            arrays["out%d" % k] = np.asarray(dst.buf, np.int64).astype(np.uint16)
            cases.append({"k": k, "kind": "sgr", "bd": bd, "w": w, "h": h, "S": S, "idx": idx, "xqd": list(xqd)})
            k += 1
            print(k, "sgr", bd, w, h, flush=True)
        # ---- (b) Wiener
        round_0 = 5 if bd == 12 else 3
        round_1 = 14 - round_0
        for (w, h) in ((32, 16), (8, 8), (64, 24), (20, 12)):
            S, Hh = w + 10, h + 8
            img = rng.integers(0, mx + 1, (Hh, S)).astype(np.int64)
            img[:5] = np.where(rng.integers(0, 2, (5, S)) > 0, mx, 0)
            filts = []
            for _ in range(2):   # symmetric 7 taps in their coded ranges, centre = -2 (t0 + t1 + t2) (the stored form), tap 7 = 0
                t0, t1, t2 = int(rng.integers(-5, 11)), int(rng.integers(-23, 9)), int(rng.integers(-17, 47))
                filts.append([t0, t1, t2, -2 * (t0 + t1 + t2), t2, t1, t0, 0])
            P = ev.array(img.ravel(), ct)
            temp = ev.array([0] * (128 * (h + 8 + 1)), "uint16_t")
            dst = ev.array([0] * (w * h), ct)
            FX, FY = ev.array(filts[0], "int16_t"), ev.array(filts[1], "int16_t")
            fx = R.Ptr(FX.buf, 0, FX.t, (8,)); fy = R.Ptr(FY.buf, 0, FY.t, (8,))
            src = P.add(3 * S + 3)
            if bd == 8:
                ih = h + 7
                ev.call("convolve_add_src_horiz_hip", src.add(-3 * S), S, temp, 128, fx, 0, 16, w, ih, round_0)
                ev.call("convolve_add_src_vert_hip", temp.add(128 * 3), 128, dst, w, fy, 0, 16, w, h, round_1)
            else:
                ih = h + 8
                ev.call("highbd_convolve_add_src_horiz_hip", src.add(-3 * S), S, temp, 128, fx, 0, 16, w, ih, round_0, bd)
                ev.call("highbd_convolve_add_src_vert_hip", temp.add(128 * 3), 128, dst, w, fy, 0, 16, w, h, round_1, bd)
            arrays["img%d" % k] = img.astype(np.uint16)
            arrays["out%d" % k] = np.asarray(dst.buf, np.int64).astype(np.uint16)
            cases.append({"k": k, "kind": "wiener", "bd": bd, "w": w, "h": h, "S": S, "fx": filts[0], "fy": filts[1]})
            k += 1
            print(k, "wiener", bd, w, h, flush=True)
    save("ref_eval_lr_apply.npz", arrays, cases)


if __name__ == "__main__":
    main()
