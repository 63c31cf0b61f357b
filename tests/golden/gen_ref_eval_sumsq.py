#!/usr/bin/env python3
"""Golden vectors of the residual statistics from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_sumsq.npz   aom_sum_squares_2d_i16_c and aom_sum_sse_2d_i16_c (aom_dsp/sum_squares.c:16-30,75-90) on blocks of an int16 residual plane:
                       4x4 .. 64x64 incl. rectangles, 8-bit-range, 12-bit-range and full-int16 residuals (the products' int range), a non-zero
                       incoming *sum.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def main():
    ev = evaluator(["aom_dsp/sum_squares.c"])
    rng = np.random.default_rng(20261115)
    arrays, cases = {}, []
    S = 96
    for name, amp in (("r8", 255), ("r12", 4095), ("r16", 32767)):
        plane = rng.integers(-amp, amp + 1, (S, S))
        if name == "r16":
            plane[:8, :8] = -32768
        arrays[name] = plane.astype(np.int16)
        P = ev.array(plane.ravel(), "int16_t")
        for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 32), (4, 16), (64, 16)):
            for trial in range(2):
                x, y = (0, 0) if trial == 0 else (int(rng.integers(0, S - w + 1)), int(rng.integers(0, S - h + 1)))
                sub = R_add(P, y * S + x)
                ss = ev.call("aom_sum_squares_2d_i16_c", sub, S, w, h)
                start = int(rng.integers(-1000, 1000)) * trial
                sm = ev.array([start], "int")
                ss2 = ev.call("aom_sum_sse_2d_i16_c", sub, S, w, h, sm)
                assert ss == ss2
                cases.append({"plane": name, "w": w, "h": h, "x": x, "y": y, "ss": str(int(ss)), "sum_in": start, "sum_out": int(sm.buf[0])})
    save("ref_eval_sumsq.npz", arrays, cases)


def R_add(p, k):
    return p.add(k)


if __name__ == "__main__":
    main()
