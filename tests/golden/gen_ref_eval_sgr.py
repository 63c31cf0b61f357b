#!/usr/bin/env python3
"""Golden vectors of the self-guided restoration filter from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_sgr.npz   av1_selfguided_restoration_c (av1/common/restoration.c:871-915 with boxsum, calculate_intermediate_result and the two
                     *_internal functions) on units of several sizes (odd ones too) with their 3-pixel surround: 8 / 10 / 12 bits, parameter
                     sets with both radii, r[1] only and r[0] only, flat areas (z = 0: A = 1) and extreme pixels next to each other.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def main():
    ev = evaluator(["av1/common/restoration.h", "av1/common/restoration.c"])
    rng = np.random.default_rng(20261104)
    arrays, cases = {}, []
    k = 0
    plan = {8: [((64, 64), (0, 12)), ((24, 16), (5, 14, 9)), ((17, 9), (3, 10, 15)), ((50, 33), (7,))],
            10: [((64, 64), (2,)), ((24, 16), (0, 13, 15)), ((17, 9), (9, 11, 14))],
            12: [((32, 32), (4, 10)), ((24, 16), (1, 12, 14)), ((17, 9), (8, 15))]}
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        for (w, h), idxs in plan[bd]:
            S, Hh = w + 6 + 2, h + 6
            for idx in idxs:
                yy, xx = np.mgrid[0:Hh, 0:S]
                img = np.clip((np.sin(xx / 5.0) + np.cos(yy / 4.0) + 2) * 0.25 * mx + rng.integers(-mx // 10, mx // 10 + 1, (Hh, S)), 0, mx).astype(np.int64)
                img[: Hh // 3, : S // 3] = img[0, 0]                                   # a flat corner: p = 0, z = 0
                img[Hh // 2: Hh // 2 + 3, S // 2: S // 2 + 4] = rng.choice([0, mx], (3, 4))   # extremes side by side
                P = ev.array(img.ravel(), ct)
                f0, f1 = ev.array([-7] * (w * h), "int32_t"), ev.array([-7] * (w * h), "int32_t")
                ev.call("av1_selfguided_restoration_c", P.add(3 * S + 3), w, h, S, f0, f1, w, idx, bd, int(bd > 8))
                arrays["img%d" % k] = img.astype(np.uint16)
                arrays["f0_%d" % k], arrays["f1_%d" % k] = np.asarray(f0.buf, np.int64).astype(np.int32), np.asarray(f1.buf, np.int64).astype(np.int32)
                cases.append({"k": k, "bd": bd, "w": w, "h": h, "S": S, "idx": idx})
                k += 1
                print(k, bd, w, h, idx, flush=True)
    save("ref_eval_sgr.npz", arrays, cases)


if __name__ == "__main__":
    main()
