#!/usr/bin/env python3
"""Golden vectors of the coefficient contexts from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_nzmap.npz   av1_get_nz_map_contexts_c (av1/encoder/encodetxb.c:222-267) on level maps from av1_txb_init_levels_c (:238-254), with get_nz_mag /
                       get_nz_map_ctx_from_stats and the av1_nz_map_ctx_offset tables (av1/common/txb_common.h:150-224, txb_common.c): square, 1:2, 1:4
                       and 64-point transform sizes, the three transform classes with their scan orders, ends of block from 1 to the full block.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402,F401
from gen_ref_eval_golden import evaluator, save, REF  # noqa: E402

TXW = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TXH = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


def main():
    import pyoracle as orc   # scan orders as INPUTS (pinned separately)
    ev = evaluator([])
    # TX_SIZE / TX_CLASS are UENUM1BYTE enums (av1/common/enums.h:174-197, av1/common/entropy.h:57-62), a form the evaluator skips: the types as int, the enumerators
    # av1_get_adjusted_tx_size and get_nz_map_ctx_from_stats name with the values of their declaration order
    for n in ("TX_SIZE", "TX_CLASS"):
        ev.define(n, "int")
    for i, n in enumerate(("TX_4X4", "TX_8X8", "TX_16X16", "TX_32X32", "TX_64X64", "TX_4X8", "TX_8X4", "TX_8X16", "TX_16X8", "TX_16X32", "TX_32X16", "TX_32X64",
                           "TX_64X32", "TX_4X16", "TX_16X4", "TX_8X32", "TX_32X8", "TX_16X64", "TX_64X16", "TX_SIZES_ALL")):
        ev.define(n, "(%d)" % i)
    for i, n in enumerate(("TX_CLASS_2D", "TX_CLASS_HORIZ", "TX_CLASS_VERT")):
        ev.define(n, "(%d)" % i)
    for f in ("av1/common/common_data.h", "av1/common/common_data.c"):
        ev.load(REF + f)
    blockd = open(REF + "av1/common/blockd.h").read()
    ev.load_text(re.search(r"static INLINE TX_SIZE av1_get_adjusted_tx_size\(TX_SIZE tx_size\) \{.*?\n}\n", blockd, re.S).group(0), "blockd.h:av1_get_adjusted_tx_size")
    ev.load_text("typedef int8_t ENTROPY_CONTEXT; typedef struct { int txb_skip_ctx; int dc_sign_ctx; } TXB_CTX;\n", "blockd.h:types")   # (parameter types of functions not called here)
    ent = open(REF + "av1/common/entropy.h").read()
    ev.load_text("\n".join(re.findall(r"#define (?:SIG_COEF_CONTEXTS\w*|TXB_SKIP_CONTEXTS|DC_SIGN_CONTEXTS|LEVEL_CONTEXTS|BR_CDF_SIZE|COEFF_BASE_RANGE|NUM_BASE_LEVELS) [^\n]*", ent)) + "\n",
                 "entropy.h:context counts")
    ev.load(REF + "av1/common/txb_common.h")
    ev.load(REF + "av1/common/txb_common.c")
    text = open(REF + "av1/encoder/encodetxb.c").read()
    for pat in (r"static INLINE int get_nz_map_ctx\([^;{]*\)\s*\{.*?\n}\n", r"void av1_txb_init_levels_c\([^;{]*\)\s*\{.*?\n}\n",
                r"void av1_get_nz_map_contexts_c\([^;{]*\)\s*\{.*?\n}\n"):
        ev.load_text(re.search(pat, text, re.S).group(0), "encodetxb.c:" + pat[:40])
    bad = [s for s in ev.skipped if s[0].startswith("encodetxb.c")]
    assert not bad, bad
    rng = np.random.default_rng(20261113)
    arrays, cases = {}, []
    k = 0
    for tx_size in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18):
        W, H = TXW[tx_size], TXH[tx_size]
        w, h = min(W, 32), min(H, 32)
        n = w * h
        for tx_type in ((0, 10, 11) if n <= 256 and W <= 16 and H <= 16 else (0,)):
            tx_class = 0 if tx_type < 10 else (2 if tx_type == 10 else 1)     # tx_type_to_class: V_DCT -> TX_CLASS_VERT (2), H_DCT -> TX_CLASS_HORIZ (1)
            scan, _ = orc.get_scan(tx_size, tx_type)
            for trial in range(3 if n <= 256 else 2):
                eob = [n, max(1, n // 5), 1][trial]
                coeff = np.zeros(n, np.int64)
                mags = rng.choice([0, 0, 1, 1, 2, 3, 4, 9, 200], n)
                coeff[scan[:eob]] = (mags * rng.choice([-1, 1], n))[:eob]
                coeff[scan[eob - 1]] = 1 + trial
                lv = ev.array([0x55] * ((w + 4) * (h + 4) + 16), "uint8_t")
                ev.call("av1_txb_init_levels_c", ev.array(coeff, "int32_t"), w, h, lv)
                ctxs = ev.array([-7] * n, "int8_t")
                ev.call("av1_get_nz_map_contexts_c", lv, ev.array(scan, "int16_t"), eob, tx_size, tx_class, ctxs)
                arrays["c%d" % k] = coeff.astype(np.int32)
                arrays["x%d" % k] = np.asarray(ctxs.buf, np.int64).astype(np.int8)
                cases.append({"k": k, "tx_size": tx_size, "tx_type": tx_type, "tx_class": tx_class, "w": w, "h": h, "eob": eob})
                k += 1
        print(tx_size, k, flush=True)
    save("ref_eval_nzmap.npz", arrays, cases)


if __name__ == "__main__":
    main()
