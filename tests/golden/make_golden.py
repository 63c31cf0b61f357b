#!/usr/bin/env python3
"""Generates the committed golden fixtures from the reference sources (build container only).

  txfm1d_golden.npz   inputs + outputs of every straight-line 1-D butterfly network of the
                      reference (av1_fdct4..64, av1_fadst8/16, av1_idct4..64, av1_iadst8/16),
                      obtained by evaluating the reference's own statements
                      (ref_txfm1d_eval.py), several input magnitudes, cos_bit 10..13,
                      inverse clamp widths 0/16/18/20.
  table_checksums.json  sha256 of every constant table parsed out of the reference:
                      cospi, sinpi, the six Dc/Ac_Qlookup tables, all scan / iscan arrays and the
                      (tx_size, tx_type) -> scan-name map of av1_scan_orders.

Fixtures are data (inputs and expected outputs); no reference source text is stored.
"""
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_txfm1d_eval import REF, evaluate, load_reference_networks, parse_int_table  # noqa: E402


def sha(vals):
    return hashlib.sha256(np.asarray(vals, dtype=np.int64).tobytes()).hexdigest()


def main():
    fns, cospi, sinpi = load_reference_networks()
    out = {}
    rng = np.random.default_rng(20261001)
    for name, fn in sorted(fns.items()):
        size = fn[0]
        inv = name.startswith("av1_i")
        xs = []
        for mag in (8, 11, 14, 17, 22, 31):
            xs.append(rng.integers(-(1 << (mag - 1)), 1 << (mag - 1), size=(6, size), dtype=np.int64))
        edge = np.zeros((4, size), np.int64)
        edge[0, :] = 1023; edge[1, :] = -1024; edge[2, 0] = 32767; edge[3, ::2] = 511
        x = np.concatenate(xs + [edge])
        out[name + "/in"] = x.astype(np.int32)
        for cb in ((12,) if inv else (10, 11, 12, 13)):
            for clamp in ((0, 16, 18, 20) if inv else (0,)):
                y = evaluate(fn, x, cb, cospi[cb - 10], clamp)
                out["%s/cb%d/clamp%d" % (name, cb, clamp)] = y.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "txfm1d_golden.npz"), **out)

    sums = {"cospi": sha(cospi), "sinpi": sha(sinpi)}
    for n in ("dc_qlookup_QTX", "dc_qlookup_10_QTX", "dc_qlookup_12_QTX", "ac_qlookup_QTX", "ac_qlookup_10_QTX",
              "ac_qlookup_12_QTX"):
        sums[n] = sha(parse_int_table(REF + "/av1/common/quant_common.c", "const int16_t " + n))
    scan_src = open(REF + "/av1/common/scan.c").read()
    for full in re.findall(r"const int16_t, ((?:av1_)?(?:default|mcol|mrow)_i?scan_\d+x\d+)\[", scan_src):
        sums[full] = sha(parse_int_table(REF + "/av1/common/scan.c", full))
    body = scan_src[scan_src.index("const SCAN_ORDER av1_scan_orders"):]
    ents = re.findall(r"\{\s*(\w+_scan_\w+),\s*(av1_\w+_iscan_\w+)\s*\}", body)
    assert len(ents) == 19 * 16
    sums["av1_scan_orders"] = [list(e) for e in ents]
    json.dump(sums, open(os.path.join(HERE, "table_checksums.json"), "w"), indent=0, sort_keys=True)
    print("wrote txfm1d_golden.npz (%d arrays), table_checksums.json (%d entries)" % (len(out), len(sums)))


if __name__ == "__main__":
    main()
