#!/usr/bin/env python3
"""Golden vectors of the wedge-mask helpers from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_wedge.npz   av1_wedge_sse_from_residuals_c, av1_wedge_sign_from_residuals_c, av1_wedge_compute_delta_squares_c
                       (av1/encoder/wedge_utils.c:52-125) -- what pick_wedge (av1/encoder/compound_type.c) evaluates per wedge
                       index before the masked compound search: N = 64 .. 1024 (multiples of 64, as the callers guarantee),
                       residuals of 8- and 12-bit range incl. values that reach the int16 clamp, masks 0 .. 64.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def main():
    ev = evaluator([])
    ev.define("WEDGE_WEIGHT_BITS", "6")                       # av1/common/reconinter.h:34 (the header's structs are outside the evaluator's subset)
    ev.define("MAX_MASK_VALUE", "(1 << WEDGE_WEIGHT_BITS)")   # av1/encoder/wedge_utils.c:22
    ev.load("/root/reference/av1/encoder/wedge_utils.c")
    rng = np.random.default_rng(20261101)
    arrays, cases = {}, []
    k = 0
    for N in (64, 128, 256, 512, 1024):
        for bits in (8, 12, 15):
            for trial in range(2):
                lim = (1 << bits) - 1
                r1 = rng.integers(-lim, lim + 1, N)
                d = rng.integers(-lim, lim + 1, N)
                m = rng.integers(0, 65, N)
                if trial == 1:   # a ramp like a real wedge, and exact 0 / 64 runs
                    m = np.clip(np.arange(N) % 32 * 3 - 10, 0, 64)
                a, b = rng.integers(-lim, lim + 1, N), rng.integers(-lim, lim + 1, N)
                R1, D, M = ev.array(r1, "int16_t"), ev.array(d, "int16_t"), ev.array(m, "uint8_t")
                sse = int(ev.call("av1_wedge_sse_from_residuals_c", R1, D, M, N))
                A, B = ev.array(a, "int16_t"), ev.array(b, "int16_t")
                DS = ev.array([0] * N, "int16_t")
                ev.call("av1_wedge_compute_delta_squares_c", DS, A, B, N)
                ds = np.asarray(DS.buf, np.int64)
                acc = int((ds * m).sum())
                signs = []
                for limit in (acc - 1, acc, acc + 1, 0, -(1 << 40)):
                    signs.append(int(ev.call("av1_wedge_sign_from_residuals_c", ev.array(ds, "int16_t"), M, N, limit)))
                arrays["r1_%d" % k], arrays["d_%d" % k], arrays["m_%d" % k] = r1.astype(np.int16), d.astype(np.int16), m.astype(np.uint8)
                arrays["a_%d" % k], arrays["b_%d" % k], arrays["ds_%d" % k] = a.astype(np.int16), b.astype(np.int16), ds.astype(np.int16)
                cases.append({"k": k, "N": N, "bits": bits, "sse": sse, "acc": acc, "limits": [acc - 1, acc, acc + 1, 0, -(1 << 40)], "signs": signs})
                k += 1
    save("ref_eval_wedge.npz", arrays, cases)


if __name__ == "__main__":
    main()
