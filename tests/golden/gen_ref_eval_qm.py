#!/usr/bin/env python3
"""Golden vectors of the quantisers WITH quantisation matrices, obtained by interpreting the reference's own C functions
(build container only; tests/golden/ref_c_eval.py):

  ref_eval_qm.npz   aom_quantize_b_helper_c / aom_highbd_quantize_b_helper_c (aom_dsp/quantize.c:108-169,261-316) called with the
                    qm_ptr / iqm_ptr of av1/common/quant_common.c -- levels 0, 8 and 14 (15 = flat, NULL pointers), luma and chroma sets,
                    TX_4X4 / 8X8 / 16X16 / 32X32 / 8X16, log_scale 0 and 1.

The matrices are DATA of the bitstream format (AV1 spec section 7.12.2 tables): they are read out of the reference's initialisers
(wt_matrix_ref / iwt_matrix_ref, quant_common.c:323-) and sliced per transform size the way av1_qm_init (:283-312) lays them out
(TX sizes in enum order, sizes tx_size_2d[t], 64-point sizes reuse their adjusted size) -- the slices used are stored in the fixture so the
tests need nothing but the .npz.

Usage: python tests/golden/gen_ref_eval_qm.py"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_ref_eval_golden import evaluator, save, REF  # noqa: E402

TX_2D = [16, 64, 256, 1024, 4096, 32, 32, 128, 128, 512, 512, 2048, 2048, 64, 64, 256, 256, 1024, 1024]
ADJ = {4: 3, 11: 3, 12: 3, 18: 10, 17: 9}   # av1_get_adjusted_tx_size (blockd.h:1365-1374)


def matrices():
    text = open(REF + "av1/common/quant_common.c").read()
    out = {}
    for name in ("iwt_matrix_ref", "wt_matrix_ref"):
        m = re.search(r"static const qm_val_t %s\[NUM_QM_LEVELS - 1\]\[2\]\[QM_TOTAL_SIZE\] = \{(.*?)\n\};" % name, text, re.S)
        body = re.sub(r"/\*.*?\*/|//[^\n]*", "", m.group(1), flags=re.S)
        vals = np.array([int(v) for v in re.findall(r"\d+", body)], np.uint8)
        out[name] = vals.reshape(15, 2, 3344)
    return out


def offset_of(tx_size):   # av1_qm_init: `current` when t == tx_size
    cur = 0
    for t in range(19):
        if ADJ.get(t, t) != t:
            continue
        if t == tx_size:
            return cur
        cur += TX_2D[t]
    raise ValueError(tx_size)


def main():
    import pyoracle as orc   # quantiser tables and scan orders as INPUTS (pinned separately)
    mats = matrices()
    ev = evaluator(["aom_dsp/quantize.h", "aom_dsp/quantize.c"])
    rng = np.random.default_rng(20261003)
    arrays, cases = {}, []
    k = 0
    for tx_size, n, ls in ((0, 16, 0), (1, 64, 0), (2, 256, 0), (3, 1024, 1), (7, 128, 0)):
        off = offset_of(tx_size)
        scan, iscan = orc.get_scan(tx_size, 0)
        for level in (0, 8, 14):
            for plane in (0, 1):
                qm = mats["wt_matrix_ref"][level, plane, off:off + n].copy()
                iqm = mats["iwt_matrix_ref"][level, plane, off:off + n].copy()
                for hbd in (0, 1):
                    for qindex in (60, 150):
                        bd = 10 if hbd else 8
                        q = orc.build_quantizer_y(bd, qindex)
                        for kind in ("random", "near_zbin"):
                            span = (1 << (bd + 7)) - 1
                            if kind == "random":
                                c = rng.integers(-span, span + 1, n)
                                c[rng.random(n) < 0.6] //= 64
                            else:   # around the matrix-scaled dead zone: |c| * wt ~ zbin * 32
                                zb = int(q["zbin"][1])
                                c = (rng.integers(-3, 4, n) + np.sign(rng.integers(-1, 2, n)) * (zb * 32 // np.maximum(qm.astype(np.int64), 1))).astype(np.int64)
                            fn = "aom_highbd_quantize_b_helper_c" if hbd else "aom_quantize_b_helper_c"
                            qc, dq, eob = ev.array([0x55] * n, "int32_t"), ev.array([0x55] * n, "int32_t"), ev.array([77], "uint16_t")
                            t = {m: ev.array(q[m], "int16_t") for m in q}
                            ev.call(fn, ev.array(c, "int32_t"), n, t["zbin"], t["round"], t["quant"], t["quant_shift"], qc, dq, t["dequant"], eob,
                                    ev.array(scan, "int16_t"), ev.array(iscan, "int16_t"), ev.array(qm, "uint8_t"), ev.array(iqm, "uint8_t"), ls)
                            arrays["c%d" % k] = np.asarray(c, np.int32)
                            arrays["q%d" % k] = np.asarray(qc.buf, np.int32)
                            arrays["d%d" % k] = np.asarray(dq.buf, np.int32)
                            mkey = "%d_%d_%d" % (tx_size, level, plane)
                            arrays["qm_" + mkey], arrays["iqm_" + mkey] = qm, iqm
                            cases.append({"fn": fn, "matrix": mkey, "tx_size": tx_size, "n": n, "log_scale": ls, "hbd": hbd, "bd": bd, "qindex": qindex, "qm_level": level,
                                          "plane": plane, "kind": kind, "eob": int(eob.buf[0]),
                                          "tables": {m: [int(q[m][0]), int(q[m][1])] for m in q}})
                            k += 1
    save("ref_eval_qm.npz", arrays, cases)


if __name__ == "__main__":
    main()
