#!/usr/bin/env python3
"""Golden vectors of the ADAPTIVE quantiser WITH quantisation matrices from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_qm_adaptive.npz   aom_quantize_b_adaptive_helper_c / aom_highbd_quantize_b_adaptive_helper_c (aom_dsp/quantize.c:16-105,173-258) with
                             qm_ptr / iqm_ptr of av1/common/quant_common.c (levels 0, 8, 14; luma and chroma): TX_4X4 / 8X8 / 16X16 / 32X32 / 8X16,
                             inputs that exercise the pre-scan (a tail inside the widened dead zone) and the single-coefficient rule.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ref_eval_golden import evaluator, save  # noqa: E402
from gen_ref_eval_qm import matrices, offset_of  # noqa: E402


def main():
    import pyoracle as orc   # quantiser tables and scan orders as INPUTS (pinned separately)
    mats = matrices()
    ev = evaluator(["aom_dsp/quantize.h", "aom_dsp/quantize.c"])
    rng = np.random.default_rng(20261107)
    arrays, cases = {}, []
    k = 0
    for tx_size, n, ls in ((0, 16, 0), (1, 64, 0), (2, 256, 0), (3, 1024, 1), (7, 128, 0)):
        off = offset_of(tx_size)
        scan, iscan = orc.get_scan(tx_size, 0)
        for level in (0, 8, 14):
            for plane in (0, 1):
                qm = mats["wt_matrix_ref"][level, plane, off:off + n].copy()
                iqm = mats["iwt_matrix_ref"][level, plane, off:off + n].copy()
                mkey = "%d_%d_%d" % (tx_size, level, plane)
                arrays["qm_" + mkey], arrays["iqm_" + mkey] = qm, iqm
                for hbd in (0, 1):
                    bd = 10 if hbd else 8
                    q = orc.build_quantizer_y(bd, 60 if (level + plane) % 2 else 150)
                    zb = int(q["zbin"][1])
                    for kind in ("tail", "single", "random"):
                        span = (1 << (bd + 7)) - 1
                        if kind == "random":
                            c = rng.integers(-span, span + 1, n)
                            c[rng.random(n) < 0.6] //= 64
                        elif kind == "tail":   # significant head, then a tail just inside / outside the widened dead zone
                            c = np.zeros(n, np.int64)
                            order = np.asarray(scan)
                            head = n // 3
                            c[order[:head]] = rng.integers(-span // 8, span // 8 + 1, head)
                            edge = (zb * 32 + ((int(q["dequant"][1]) * 325 + 64) >> 7)) // np.maximum(qm[order[head:]].astype(np.int64), 1)
                            c[order[head:]] = (edge + rng.integers(-2, 3, n - head)) * np.sign(rng.integers(-1, 2, n - head))
                        else:   # exactly one coefficient that quantises to +-1: the SKIP_EOB_FACTOR_ADJUST rule
                            c = np.zeros(n, np.int64)
                            pos = int(rng.integers(0, n))
                            c[pos] = (zb * 32 // max(int(qm[pos]), 1) + int(rng.integers(0, 12))) * int(rng.choice([-1, 1]))
                        fn = "aom_highbd_quantize_b_adaptive_helper_c" if hbd else "aom_quantize_b_adaptive_helper_c"
                        qc, dq, eob = ev.array([0x55] * n, "int32_t"), ev.array([0x55] * n, "int32_t"), ev.array([77], "uint16_t")
                        t = {m: ev.array(q[m], "int16_t") for m in q}
                        ev.call(fn, ev.array(c, "int32_t"), n, t["zbin"], t["round"], t["quant"], t["quant_shift"], qc, dq, t["dequant"], eob,
                                ev.array(scan, "int16_t"), ev.array(iscan, "int16_t"), ev.array(qm, "uint8_t"), ev.array(iqm, "uint8_t"), ls)
                        arrays["c%d" % k] = np.asarray(c, np.int32)
                        arrays["q%d" % k] = np.asarray(qc.buf, np.int32)
                        arrays["d%d" % k] = np.asarray(dq.buf, np.int32)
                        cases.append({"k": k, "fn": fn, "matrix": mkey, "tx_size": tx_size, "n": n, "log_scale": ls, "hbd": hbd, "bd": bd, "qm_level": level, "plane": plane,
                                      "kind": kind, "eob": int(eob.buf[0]), "tables": {m: [int(q[m][0]), int(q[m][1])] for m in q}})
                        k += 1
    save("ref_eval_qm_adaptive.npz", arrays, cases)


if __name__ == "__main__":
    main()
