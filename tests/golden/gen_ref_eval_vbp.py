#!/usr/bin/env python3
"""Golden vectors of the variance-based partitioning's leaf statistics from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_vbp.npz   fill_variance_8x8avg, compute_minmax_8x8 and fill_variance_4x4avg (av1/encoder/var_based_part.c:255-430) with aom_avg_8x8[_quad] /
                     aom_avg_4x4 / aom_minmax_8x8 and the high-bit-depth forms (aom_dsp/avg.c:18-100) under them: 8 and 10 bits, 16 x 16 blocks wholly
                     inside the superblock's visible part, cut by its right / bottom edge and outside, the 4 x 4 border offset.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402,F401
from gen_ref_eval_golden import evaluator, save, REF  # noqa: E402


def main():
    ev = evaluator([])
    for n in ("aom_avg_8x8", "aom_avg_4x4", "aom_avg_8x8_quad", "aom_minmax_8x8", "aom_highbd_avg_8x8", "aom_highbd_avg_4x4", "aom_highbd_minmax_8x8"):
        ev.define(n, n + "_c")
    ev.define("YV12_FLAG_HIGHBITDEPTH", "8")          # aom_scale/yv12config.h:127
    ev.load(REF + "aom_dsp/avg.c")
    enc_h = open(REF + "av1/encoder/encoder.h").read()
    ev.load_text(re.search(r"typedef struct \{\s*// TODO\(kyslov\): consider changing to 64bit.*?\} VP16x16;", enc_h, re.S).group(0), "encoder.h:VP16x16")
    ev.load(REF + "av1/encoder/var_based_part.h")
    text = open(REF + "av1/encoder/var_based_part.c").read()
    for name in ("fill_variance", "all_blks_inside", "fill_variance_8x8avg_highbd", "fill_variance_8x8avg_lowbd", "fill_variance_8x8avg", "compute_minmax_8x8",
                 "fill_variance_4x4avg"):
        m = re.search(r"static (?:AOM_INLINE )?(?:void|int) %s\([^;{]*\)\s*\{.*?\n}\n" % name, text, re.S)
        assert m, name
        ev.load_text(m.group(0), "var_based_part.c:" + name)
    bad = [s for s in ev.skipped if s[0].startswith("var_based_part.c:")]
    assert not bad, bad
    rng = np.random.default_rng(20261112)
    arrays, cases = {}, []
    S = 80                                              # a 64 x 64 superblock in an 80-wide buffer (the reads past the visible part are the border's)
    k = 0
    for bd in (8, 10):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        src = rng.integers(0, mx + 1, (S, S))
        dst = np.clip(src + rng.integers(-(mx >> 3), (mx >> 3) + 1, (S, S)), 0, mx)
        dst[:16, :16] = mx - src[:16, :16]              # far apart: the averages' difference at its extremes, max - min beyond 255 at 10 bits
        src[16:24, 16:24] = dst[16:24, 16:24]           # min = max = 0
        arrays["src%d" % bd], arrays["dst%d" % bd] = src.astype(np.uint16), dst.astype(np.uint16)
        Sp, Dp = ev.array(src.ravel(), ct), ev.array(dst.ravel(), ct)
        hb = 8 if bd > 8 else 0
        for (pw, ph) in ((64, 64), (40, 64), (64, 20), (24, 8)):
            for (x16, y16) in ((0, 0), (16, 16), (32, 0), (48, 48), (16, 0), (0, 16)):
                vst = ev.new("VP16x16")
                ev.call("fill_variance_8x8avg", Sp, S, Dp, S, x16, y16, vst, hb, pw, ph)
                sums = [int(ev.get(vst, "split[%d].part_variances.none.sum_error" % i)) for i in range(4)]
                sses = [int(ev.get(vst, "split[%d].part_variances.none.sum_square_error" % i)) for i in range(4)]
                mm = int(ev.call("compute_minmax_8x8", Sp, S, Dp, S, x16, y16, hb, pw, ph))
                cases.append({"k": k, "kind": "8x8", "bd": bd, "x16": x16, "y16": y16, "pw": pw, "ph": ph, "sum": sums, "sse": sses, "minmax": mm})
                k += 1
        for (pw, ph, bo) in ((64, 64, 0), (64, 64, 4), (20, 36, 0), (20, 36, 4)):
            for (x8, y8) in ((0, 0), (8, 24), (16, 32), (56, 56), (16, 8)):
                vst = ev.new("VP8x8")
                ev.call("fill_variance_4x4avg", Sp, S, x8, y8, vst, hb, pw, ph, bo)
                sums = [int(ev.get(vst, "split[%d].part_variances.none.sum_error" % i)) for i in range(4)]
                sses = [int(ev.get(vst, "split[%d].part_variances.none.sum_square_error" % i)) for i in range(4)]
                cases.append({"k": k, "kind": "4x4", "bd": bd, "x8": x8, "y8": y8, "pw": pw, "ph": ph, "border_offset": bo, "sum": sums, "sse": sses})
                k += 1
    save("ref_eval_vbp.npz", arrays, cases)


if __name__ == "__main__":
    main()
