#!/usr/bin/env python3
"""Golden vectors of the low-precision quantiser from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_quant_lp.npz   av1_quantize_lp_c (av1/encoder/av1_quantize.c:212-240) and av1_block_error_lp_c (av1/encoder/rdopt.c:650-660) on int16
                          coefficients: TX_4X4 / 8X8 / 16X16 / 32X32 / 8X16, default and 1-D scans, inputs at the int16 clamp and with dqcoeff
                          products that leave int16 (the reference stores them truncated).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def main():
    import pyoracle as orc   # scan orders as INPUTS (pinned separately)
    ev = evaluator(["aom_dsp/quantize.h", "av1/encoder/av1_quantize.h", "av1/encoder/av1_quantize.c"])
    ev2 = evaluator(["aom_dsp/quantize.h", "av1/encoder/rdopt.c"])
    rng = np.random.default_rng(20261109)
    arrays, cases = {}, []
    k = 0
    for tx_size, n in ((0, 16), (1, 64), (2, 256), (3, 1024), (7, 128)):
        for tx_type in ((0, 10, 11) if n <= 256 else (0,)):
            scan, iscan = orc.get_scan(tx_size, tx_type)
            for trial in range(5):
                dq = np.array([rng.integers(4, 400), rng.integers(4, 600)], np.int64)
                tabs = {"round": (64 * dq) >> 7, "quant": np.minimum((1 << 16) // dq, 32767), "dequant": dq}
                c = rng.integers(-2000, 2001, n)
                c[rng.random(n) < 0.5] //= 32
                if trial >= 3:
                    c[:4] = (32767, -32768, 32700, -32700)        # |c| + round at the clamp
                if trial == 4:                                    # tables as given: qcoeff * dequant leaves int16 and is stored truncated
                    tabs["quant"] = np.array([32767, 30000], np.int64)
                qc, dqc, eob = ev.array([0x55] * n, "int16_t"), ev.array([0x55] * n, "int16_t"), ev.array([77], "uint16_t")
                t = {m: ev.array(v, "int16_t") for m, v in tabs.items()}
                C_ = ev.array(c, "int16_t")
                ev.call("av1_quantize_lp_c", C_, n, t["round"], t["quant"], qc, dqc, t["dequant"], eob, ev.array(scan, "int16_t"), ev.array(iscan, "int16_t"))
                arrays["c%d" % k] = np.asarray(c, np.int16)
                arrays["q%d" % k] = np.asarray(qc.buf, np.int64).astype(np.int16)
                arrays["d%d" % k] = np.asarray(dqc.buf, np.int64).astype(np.int16)
                err = ev2.call("av1_block_error_lp_c", ev2.array(c, "int16_t"), ev2.array(arrays["d%d" % k], "int16_t"), n)
                cases.append({"k": k, "block_error": int(err), "tx_size": tx_size, "tx_type": tx_type, "n": n, "eob": int(eob.buf[0]),
                              "tables": {m: [int(v[0]), int(v[1])] for m, v in tabs.items()}})
                k += 1
    save("ref_eval_quant_lp.npz", arrays, cases)


if __name__ == "__main__":
    main()
