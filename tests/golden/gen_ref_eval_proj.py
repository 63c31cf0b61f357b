#!/usr/bin/env python3
"""Golden vectors of the self-guided filter's projection statistics from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_proj.npz   av1_calc_proj_params_c / _high_bd_c (av1/encoder/pickrst.c:470-657: H[2][2], C[2] of get_proj_subspace) and
                      av1_lowbd_pixel_proj_error_c / av1_highbd_pixel_proj_error_c (:226-370: get_pixel_proj_error of finer_search) with the
                      reference's own av1_sgr_params[] entries: both radii set, only r[0], only r[1] (the three branches), 8 / 10 / 12 bits,
                      strided buffers, several xq per unit incl. the corners of its range.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def main():
    ev = evaluator(["av1/common/restoration.h", "av1/common/restoration.c", "av1/encoder/pickrst.h", "av1/encoder/pickrst.c"])
    g = ev.globs.get("av1_sgr_params")
    radii = []
    for s in g.buf:
        r = s.f["r"]
        radii.append([int(r.buf[r.off]), int(r.buf[r.off + 1])])
    rng = np.random.default_rng(20261103)
    arrays, cases = {"sgr_r": np.array(radii, np.int32)}, []
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        for (w, h, S, FS) in ((24, 16, 32, 24), (64, 64, 64, 72), (13, 9, 20, 16), (56, 40, 64, 56)):
            for ep in (0, 3, 10, 12, 14, 15):     # 0-9: both radii; 10-13: r[1] only... as av1_sgr_params declares them
                src = rng.integers(0, mx + 1, (h, S))
                dat = np.clip(src + rng.integers(-mx // 16, mx // 16 + 1, (h, S)), 0, mx)
                # flt = (dat << SGRPROJ_RST_BITS) + a filter correction, inside the reference's asserted 15-bit range only at 8 / 10 bits;
                # at 12 bits it reaches 2^16 like the real filter output
                f0 = (dat << 4) + rng.integers(-mx, mx + 1, (h, S))
                f1 = (dat << 4) + rng.integers(-mx, mx + 1, (h, S))
                f0p, f1p = np.zeros((h, FS), np.int64), np.zeros((h, FS), np.int64)
                f0p[:, :S][:, :min(S, FS)] = f0[:, :min(S, FS)]
                f1p[:, :S][:, :min(S, FS)] = f1[:, :min(S, FS)]
                SRC, DAT = ev.array(src.ravel(), ct), ev.array(dat.ravel(), ct)
                F0, F1 = ev.array(f0p.ravel(), "int32_t"), ev.array(f1p.ravel(), "int32_t")
                prm = R.Ptr(g.buf, ep, g.t)
                Hb, Cb = ev.array([0, 0, 0, 0], "int64_t"), ev.array([0, 0], "int64_t")
                ev.call("av1_calc_proj_params_c" if bd == 8 else "av1_calc_proj_params_high_bd_c", SRC, w, h, S, DAT, S, F0, FS, F1, FS,
                        R.Ptr(Hb.buf, 0, Hb.t, (2,)), Cb, prm)
                errs, xqs = [], []
                for xq in ([0, 0], [-96, 224], [31, -32], [int(rng.integers(-96, 32)), int(rng.integers(-32, 96))], [127, 127]):
                    XQ = ev.array(xq, "int")
                    e = ev.call("av1_lowbd_pixel_proj_error_c" if bd == 8 else "av1_highbd_pixel_proj_error_c", SRC, w, h, S, DAT, S, F0, FS, F1, FS, XQ, prm)
                    errs.append(int(e)); xqs.append(xq)
                arrays["s%d" % k], arrays["d%d" % k] = src.astype(np.uint16), dat.astype(np.uint16)
                arrays["f0_%d" % k], arrays["f1_%d" % k] = f0p.astype(np.int32), f1p.astype(np.int32)
                cases.append({"k": k, "bd": bd, "w": w, "h": h, "S": S, "FS": FS, "ep": ep, "r": radii[ep], "H": [int(x) for x in Hb.buf], "C": [int(x) for x in Cb.buf],
                              "xq": xqs, "err": errs})
                k += 1
    save("ref_eval_proj.npz", arrays, cases)


if __name__ == "__main__":
    main()
