#!/usr/bin/env python3
"""Golden vectors of the scaled-reference predictor from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_scale.npz   av1_convolve_2d_scale_c / av1_highbd_convolve_2d_scale_c (av1/common/convolve.c) with get_conv_params_no_round: single
                       reference and the compound pair (CONV_BUF then average / distance weights), 8 / 10 / 12 bits, steps from 1:2 up-scaling
                       (x_step_qn 512) to 2:1 down-scaling (2048) incl. unequal x / y steps, all four filter families, the 4-tap sets of narrow blocks.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def main():
    ev = evaluator(["av1/common/filter.h", "av1/common/convolve.h", "aom_dsp/aom_convolve.c", "av1/common/convolve.c"])
    rng = np.random.default_rng(20261105)
    arrays, cases = {}, []
    S, ROWS = 160, 150
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        planes = []
        for r in range(2):
            base = rng.integers(0, mx + 1, (ROWS, S))
            base[:20] = np.where(rng.integers(0, 2, (20, S)) > 0, mx, 0)
            planes.append(base)
            arrays["p%d_%d" % (bd, r)] = base.astype(np.uint16)
        P = [ev.array(pl.ravel(), ct) for pl in planes]
        trials = [((8, 8), 1024, 1024), ((16, 16), 2048, 2048), ((16, 8), 512, 512), ((4, 8), 1536, 1024), ((8, 4), 1024, 1365), ((32, 16), 1820, 1138),
                  ((16, 32), 683, 2048), ((4, 4), 2048, 512)]
        for ti, ((w, h), xs, ys) in enumerate(trials if bd != 12 else trials[:5]):
            for compound in (0, 1):
                fxi, fyi = [(0, 0), (1, 2), (2, 1), (3, 3)][(ti + compound) % 4]
                fp = [ev.call("av1_get_interp_filter_params_with_block_size", fxi, w), ev.call("av1_get_interp_filter_params_with_block_size", fyi, h)]
                wts = [(9, 7), (4, 12)][ti % 2] if (compound and ti % 3 == 1) else None
                pos, subs = [], []
                for r in range(2):
                    x0 = int(rng.integers(4, S - (w * xs >> 10) - 14)); y0 = int(rng.integers(4, ROWS - (h * ys >> 10) - 14))
                    if ti % 4 == 0 and r == 0:
                        y0 = int(rng.integers(4, 12))                      # the extreme band
                    pos.append((x0, y0)); subs.append((int(rng.integers(0, 1024)), int(rng.integers(0, 1024))))
                buf16 = ev.array([0] * (w * h), "uint16_t")
                dst = ev.array([0] * (w * h), ct)
                for r in range(2 if compound else 1):
                    cpv = ev.call("get_conv_params_no_round", r, 0, buf16, w, compound, bd)
                    cp = R.Ptr([cpv], 0, cpv.st)
                    if wts:
                        ev.set(cp, "use_dist_wtd_comp_avg", 1); ev.set(cp, "fwd_offset", wts[0]); ev.set(cp, "bck_offset", wts[1])
                    args = [P[r].add(pos[r][1] * S + pos[r][0]), S, dst, w, w, h, fp[0], fp[1], subs[r][0], xs, subs[r][1], ys, cp]
                    if bd > 8:
                        args.append(bd)
                    ev.call("av1_convolve_2d_scale_c" if bd == 8 else "av1_highbd_convolve_2d_scale_c", *args)
                    if compound and r == 0:
                        arrays["c%d" % k] = np.asarray(buf16.buf, np.uint16).copy()
                arrays["d%d" % k] = np.asarray(dst.buf, np.uint16)
                cases.append({"k": k, "bd": bd, "w": w, "h": h, "xs": xs, "ys": ys, "fx": fxi, "fy": fyi, "compound": compound, "weights": wts, "pos": pos, "subs": subs})
                k += 1
                print(k, bd, w, h, xs, ys, compound, flush=True)
    save("ref_eval_scale.npz", arrays, cases)


if __name__ == "__main__":
    main()
