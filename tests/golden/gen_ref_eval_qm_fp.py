#!/usr/bin/env python3
"""Golden vectors of the `fp` quantiser WITH quantisation matrices from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_qm_fp.npz   quantize_fp_helper_c / highbd_quantize_fp_helper_c (av1/encoder/av1_quantize.c:71-199) with qm_ptr / iqm_ptr of
                       av1/common/quant_common.c (levels 0, 8, 14; luma and chroma), TX_4X4 / 8X8 / 16X16 / 32X32 / 8X16, log_scale 0 and 1, and
                       one-sided calls (only qm, only iqm) -- the AV1_XFORM_QUANT_FP flavour when enable_qm is on.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ref_eval_golden import evaluator, save  # noqa: E402
from gen_ref_eval_qm import matrices, offset_of  # noqa: E402


def main():
    import pyoracle as orc   # scan orders as INPUTS (pinned separately)
    mats = matrices()
    ev = evaluator(["aom_dsp/quantize.h", "av1/encoder/av1_quantize.h", "av1/encoder/av1_quantize.c"])
    rng = np.random.default_rng(20261106)
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
                    # round_fp / quant_fp as av1_build_quantizer makes them: quant_fp = (1 << 16) / dequant, round_fp = (64 * dequant) >> 7
                    dq = np.array([rng.integers(8, 200), rng.integers(8, 300)], np.int64) << (bd - 8)
                    tabs = {"round": (64 * dq) >> 7, "quant": (1 << 16) // dq, "dequant": dq, "zbin": dq * 0, "quant_shift": dq * 0}
                    for kind, which in (("random", "both"), ("near_thresh", "both"), ("random", "qm_only"), ("random", "iqm_only")):
                        if which != "both" and (level, plane) != (8, 0):
                            continue
                        span = (1 << (bd + 7)) - 1
                        if kind == "random":
                            c = rng.integers(-span, span + 1, n)
                            c[rng.random(n) < 0.6] //= 64
                            c[:3] = (span, -span, 32767 if not hbd else span)     # the int16 clamp of the low-bit-depth form
                        else:   # around the weighted dead zone: |c| * wt ~ dequant << (4 - log_scale)
                            c = (rng.integers(-3, 4, n) + np.sign(rng.integers(-1, 2, n)) * ((int(dq[1]) << (4 - ls)) // np.maximum(qm.astype(np.int64), 1))).astype(np.int64)
                        fn = "highbd_quantize_fp_helper_c" if hbd else "quantize_fp_helper_c"
                        qc, dqc, eob = ev.array([0x55] * n, "int32_t"), ev.array([0x55] * n, "int32_t"), ev.array([77], "uint16_t")
                        t = {m: ev.array(v, "int16_t") for m, v in tabs.items()}
                        QM = ev.array(qm, "uint8_t") if which != "iqm_only" else 0
                        IQM = ev.array(iqm, "uint8_t") if which != "qm_only" else 0
                        ev.call(fn, ev.array(c, "int32_t"), n, t["zbin"], t["round"], t["quant"], t["quant_shift"], qc, dqc, t["dequant"], eob,
                                ev.array(scan, "int16_t"), ev.array(iscan, "int16_t"), QM, IQM, ls)
                        arrays["c%d" % k] = np.asarray(c, np.int32)
                        arrays["q%d" % k] = np.asarray(qc.buf, np.int32)
                        arrays["d%d" % k] = np.asarray(dqc.buf, np.int32)
                        cases.append({"k": k, "fn": fn, "matrix": mkey, "which": which, "tx_size": tx_size, "n": n, "log_scale": ls, "hbd": hbd, "bd": bd,
                                      "qm_level": level, "plane": plane, "kind": kind, "eob": int(eob.buf[0]),
                                      "tables": {m: [int(v[0]), int(v[1])] for m, v in tabs.items()}})
                        k += 1
    save("ref_eval_qm_fp.npz", arrays, cases)


if __name__ == "__main__":
    main()
