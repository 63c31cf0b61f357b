#!/usr/bin/env python3
"""Golden vectors of the small members of the files the north star names, from the interpreted reference (build container only; see
ref_c_eval.py) -> tests/golden/ref_eval_leftovers.npz:

  aom_get_mb_ss_c                                aom_dsp/variance.c:46-54
  aom_mse_wxh_16bit_c / aom_mse_16xh_16bit_c / aom_mse_wxh_16bit_highbd_c      aom_dsp/variance.c:1258-1297
  aom_comp_mask_pred_c / aom_highbd_comp_mask_pred_c                           aom_dsp/variance.c:773-791,841-862
  av1_return_max_sub_pixel_mv / av1_return_min_sub_pixel_mv                    av1/encoder/mcomp.c:3139-3190 (lower_mv_precision: mvref_common.h:88-97)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402
from gen_ref_eval_golden import save  # noqa: E402


def main():
    ev = G.make_evaluator(with_compound=True)
    # lower_mv_precision is a static inline of av1/common/mvref_common.h, a header whose other members need the whole
    # of AV1_COMMON: its own text is cut out of it and loaded by itself
    text = open(G.REF + "av1/common/mvref_common.h").read()
    a = text.index("static INLINE void lower_mv_precision(MV *mv, int allow_hp, int is_integer)")   # (integer_mv_precision: av1/common/mv.h, loaded)
    b = text.index("static INLINE int8_t get_uni_comp_ref_idx")
    ev.load_text(text[a:b], "av1/common/mvref_common.h:lower_mv_precision")
    rng = np.random.default_rng(20261201)
    arrays, cases = {}, []

    # ---- aom_get_mb_ss
    for k, amp in enumerate((255, 4095, 32767, 32767)):
        a16 = rng.integers(-amp, amp + 1, 256).astype(np.int16)
        if k == 3:
            a16[:] = -32768   # 256 * 2^30 = 2^38: the unsigned sum wraps
        arrays["mbss%d" % k] = a16
        v = ev.call("aom_get_mb_ss_c", ev.array(a16, "int16_t"))
        cases.append({"kind": "mb_ss", "a": "mbss%d" % k, "out": int(v) & 0xffffffff})

    # ---- aom_mse_wxh_16bit (8-bit dst), _highbd (16-bit dst), aom_mse_16xh_16bit
    S = 40
    for bd in (8, 10, 12):
        dmax = 255 if bd == 8 else (1 << bd) - 1
        dst = rng.integers(0, dmax + 1, (S, S)).astype(np.uint8 if bd == 8 else np.uint16)
        src = np.clip(dst.astype(np.int64) + rng.integers(-60 * (1 << (bd - 8)), 60 * (1 << (bd - 8)) + 1, (S, S)), 0, (1 << max(bd, 12)) - 1).astype(np.uint16)
        arrays["mse_dst%d" % bd], arrays["mse_src%d" % bd] = dst, src
        D = ev.array(dst.ravel(), "uint8_t" if bd == 8 else "uint16_t")
        Q = ev.array(src.ravel(), "uint16_t")
        for (w, h) in ((4, 4), (8, 8), (8, 4), (4, 8), (16, 16), (16, 8)):
            for trial in range(2):
                x, y = (0, 0) if trial == 0 else (int(rng.integers(0, S - w + 1)), int(rng.integers(0, S - h + 1)))
                fn = "aom_mse_wxh_16bit_c" if bd == 8 else "aom_mse_wxh_16bit_highbd_c"
                v = ev.call(fn, D.add(y * S + x), S, Q.add(y * S + x), S, w, h)
                cases.append({"kind": "mse_wxh", "bd": bd, "w": w, "h": h, "x": x, "y": y, "out": str(int(v))})
        if bd == 8:
            for (w, h) in ((4, 4), (8, 8), (4, 8), (8, 4), (16, 16)):
                packed = rng.integers(0, 256, 16 // w * w * h).astype(np.uint16) + rng.integers(0, 3, 16 // w * w * h).astype(np.uint16) * 256
                name = "mse16_src_%dx%d" % (w, h)
                arrays[name] = packed
                x, y = int(rng.integers(0, S - 16 + 1)), int(rng.integers(0, S - h + 1))
                v = ev.call("aom_mse_16xh_16bit_c", D.add(y * S + x), S, ev.array(packed, "uint16_t"), w, h)
                cases.append({"kind": "mse_16xh", "w": w, "h": h, "x": x, "y": y, "src": name, "out": str(int(v))})

    # ---- aom_comp_mask_pred / aom_highbd_comp_mask_pred
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        dt = np.uint8 if bd == 8 else np.uint16
        for (w, h) in ((8, 8), (16, 8), (8, 16), (16, 16), (32, 16), (4, 4)):
            for inv in (0, 1):
                rs, ms = w + int(rng.integers(0, 9)), w + int(rng.integers(0, 5))
                pred = rng.integers(0, mx + 1, (h, w)).astype(dt)
                ref = rng.integers(0, mx + 1, (h, rs)).astype(dt)
                mask = rng.integers(0, 65, (h, ms)).astype(np.uint8)
                if k % 5 == 0:
                    mask[:, : w // 2] = 64; mask[:, w // 2:] = 0
                out = ev.array(np.zeros(w * h, dt), ct)
                fn = "aom_comp_mask_pred_c" if bd == 8 else "aom_highbd_comp_mask_pred_c"
                ev.call(fn, out, ev.array(pred.ravel(), ct), w, h, ev.array(ref.ravel(), ct), rs, ev.array(mask.ravel(), "uint8_t"), ms, inv)
                arrays["cmp_pred%d" % k], arrays["cmp_ref%d" % k], arrays["cmp_mask%d" % k] = pred, ref, mask
                arrays["cmp_out%d" % k] = np.array(out.buf[: w * h], dt).reshape(h, w)
                cases.append({"kind": "comp_mask", "bd": bd, "w": w, "h": h, "invert": inv, "k": k})
                k += 1

    # ---- av1_return_max_sub_pixel_mv / av1_return_min_sub_pixel_mv
    for i in range(40):
        lim = [-int(rng.integers(1, 3000)), int(rng.integers(1, 3000)), -int(rng.integers(1, 3000)), int(rng.integers(1, 3000))]   # col_min, col_max, row_min, row_max
        if i < 6:
            lim = [[-1, 1, -1, 1], [-7, 9, -5, 3], [0, 0, 0, 0], [1, 5, 3, 9], [-9, -3, -7, -1], [-8, 8, -8, 8]][i]
        for allow_hp in (0, 1):
            ms = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
            ev.set(ms, "allow_hp", allow_hp)
            for kname, v in zip(("col_min", "col_max", "row_min", "row_max"), lim):
                ev.set(ms, "mv_limits." + kname, int(v))
            outs = []
            for fn in ("av1_return_max_sub_pixel_mv", "av1_return_min_sub_pixel_mv"):
                best = ev.new("MV")
                ev.set(best, "row", 12345); ev.set(best, "col", -12345)
                start = ev.new("MV")
                r = ev.call(fn, None, None, ms, start.buf[0], best, None, None, None)
                outs.append([int(r), ev.get(best, "row"), ev.get(best, "col")])
            cases.append({"kind": "extreme_mv", "limits": lim, "allow_hp": allow_hp, "max": outs[0], "min": outs[1]})
    save("ref_eval_leftovers.npz", arrays, cases)


if __name__ == "__main__":
    main()
