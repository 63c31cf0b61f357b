#!/usr/bin/env python3
"""Golden vectors of the coefficient coder's rate from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_txb_cost.npz   warehouse_efficients_txb (av1/encoder/txb_rdopt.c:450-544: what av1_cost_coeffs_txb returns for eob > 0) with get_eob_cost /
                          get_br_cost / get_golomb_cost (txb_rdopt_utils.h:66-97), av1_get_eob_pos_token, av1_txb_init_levels_c, av1_get_nz_map_contexts_c
                          (encodetxb.c:100-130,222-267) and get_br_ctx[_eob] (av1/common/txb_common.h:90-135) under it, on random cost tables;
                          and warehouse_efficients_txb_laplacian + av1_cost_coeffs_txb_estimate (:546-601) on the same blocks and tables (costLUT, txb_rdopt_utils.h:31-37);
                          and av1_get_txb_entropy_context (encodetxb.c:451-467) of every block.

Supplied as inputs / adaptations:
  * get_tx_type_cost returns 0 (a table look-up on the block's mode: the caller's addend); get_scan returns the scan order of (tx_size, tx_type) built
    from the separately pinned scan tables;
  * MACROBLOCK / macroblock_plane as views with the members the function reads (coeff_costs.eob_costs, qcoeff); MACROBLOCKD opaque (only passed on);
  * TX_SIZE / TX_CLASS / TX_TYPE / PLANE_TYPE are UENUM1BYTE enums the evaluator skips: int, with the enumerators' declaration-order values.
"""
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save, REF  # noqa: E402

TXW = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TXH = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
N_COSTS = 944


def make_txb_evaluator(extra_macroblock_members="", extra_plane_members="", extra_xd_members="int unused;", pre_views=""):
    """The evaluator of the coefficient coder's rate (shared with gen_ref_eval_yrd.py, which widens the MACROBLOCK views)."""
    ev = evaluator([])
    for n in ("TX_SIZE", "TX_CLASS", "TX_TYPE", "PLANE_TYPE"):
        ev.define(n, "int")
    for i, n in enumerate(("TX_4X4", "TX_8X8", "TX_16X16", "TX_32X32", "TX_64X64", "TX_4X8", "TX_8X4", "TX_8X16", "TX_16X8", "TX_16X32", "TX_32X16", "TX_32X64",
                           "TX_64X32", "TX_4X16", "TX_16X4", "TX_8X32", "TX_32X8", "TX_16X64", "TX_64X16", "TX_SIZES_ALL")):
        ev.define(n, "(%d)" % i)
    ev.define("TX_SIZES", "(5)"); ev.define("PLANE_TYPES", "(2)")
    for i, n in enumerate(("TX_CLASS_2D", "TX_CLASS_HORIZ", "TX_CLASS_VERT")):
        ev.define(n, "(%d)" % i)
    for f in ("av1/common/common_data.h", "av1/common/common_data.c"):
        ev.load(REF + f)
    blockd = open(REF + "av1/common/blockd.h").read()
    ev.load_text(re.search(r"static INLINE TX_SIZE av1_get_adjusted_tx_size\(TX_SIZE tx_size\) \{.*?\n}\n", blockd, re.S).group(0), "blockd.h:av1_get_adjusted_tx_size")
    ev.load_text("#define BLOCK_OFFSET(i) ((i) << 4)\ntypedef int8_t ENTROPY_CONTEXT; typedef struct { int txb_skip_ctx; int dc_sign_ctx; } TXB_CTX;\n"
                 "typedef struct { const int16_t *scan; const int16_t *iscan; } SCAN_ORDER;\n", "blockd.h / entropymode.h: types")
    ent = open(REF + "av1/common/entropy.h").read()
    ev.load_text("\n".join(re.findall(r"#define (?:SIG_COEF_CONTEXTS\w*|COEFF_CONTEXT_\w+|TXB_SKIP_CONTEXTS|EOB_COEF_CONTEXTS|DC_SIGN_CONTEXTS|LEVEL_CONTEXTS|BR_CDF_SIZE|COEFF_BASE_RANGE|NUM_BASE_LEVELS) [^\n]*",
                                      ent)) + "\n", "entropy.h:context counts")
    ev.load(REF + "av1/common/txb_common.h")
    ev.load(REF + "av1/common/txb_common.c")
    ev.load(REF + "av1/encoder/cost.h")
    blk = open(REF + "av1/encoder/block.h").read()
    ev.load_text(re.search(r"typedef struct \{\s*//! Cost to skip txfm for the current txfm block\..*?\} LV_MAP_EOB_COST;", blk, re.S).group(0), "block.h:LV_MAP_*")
    if pre_views:
        ev.load_text(pre_views, "views: types the wider MACROBLOCK views need")
    ev.load_text("typedef struct { LV_MAP_COEFF_COST coeff_costs[TX_SIZES][PLANE_TYPES]; LV_MAP_EOB_COST eob_costs[7][2]; } CoeffCosts;\n"
                 "struct macroblock_plane { tran_low_t *qcoeff; uint16_t *eobs; " + extra_plane_members + " };\n" + extra_xd_members_decl(extra_xd_members) +
                 "typedef struct macroblock { CoeffCosts coeff_costs; struct macroblock_plane plane[3]; " + extra_macroblock_members + " } MACROBLOCK;\n", "block.h:views")
    text = open(REF + "av1/encoder/encodetxb.c").read()
    for pat in (r"static const int8_t eob_to_pos_small\[33\] = \{.*?\};", r"static const int8_t eob_to_pos_large\[17\] = \{.*?\};",
                r"int av1_get_eob_pos_token\([^;{]*\)\s*\{.*?\n}\n", r"static INLINE int get_nz_map_ctx\([^;{]*\)\s*\{.*?\n}\n",
                r"void av1_txb_init_levels_c\([^;{]*\)\s*\{.*?\n}\n", r"void av1_get_nz_map_contexts_c\([^;{]*\)\s*\{.*?\n}\n"):
        ev.load_text(re.search(pat, text, re.S).group(0), "encodetxb.c:" + pat[:30])
    ev.define("av1_txb_init_levels", "av1_txb_init_levels_c"); ev.define("av1_get_nz_map_contexts", "av1_get_nz_map_contexts_c")
    ev.load_text(re.search(r"uint8_t av1_get_txb_entropy_context\([^;{]*\)\s*\{.*?\n}\n", text, re.S).group(0), "encodetxb.c:av1_get_txb_entropy_context")
    utl = open(REF + "av1/encoder/txb_rdopt_utils.h").read()
    ev.load_text(re.search(r"static const int costLUT\[15\] = \{.*?\};", utl, re.S).group(0) + "\n" + re.search(r"static const int const_term = [^;]*;", utl).group(0) + "\n"
                 + re.search(r"static const int loge_par = [^;]*;", utl).group(0) + "\n", "txb_rdopt_utils.h:costLUT")
    for pat in (r"static int get_eob_cost\([^;{]*\)\s*\{.*?\n}\n", r"static INLINE int get_golomb_cost\([^;{]*\)\s*\{.*?\n}\n", r"static INLINE int get_br_cost\([^;{]*\)\s*\{.*?\n}\n"):
        ev.load_text(re.search(pat, utl, re.S).group(0), "txb_rdopt_utils.h:" + pat[:30])
    state = {}
    ev.interp.pycalls["get_tx_type_cost"] = lambda it, a: (0, R.I32)
    ev.interp.pycalls["get_scan"] = lambda it, a: (state["scan_order"], R.PTR)
    rd = open(REF + "av1/encoder/txb_rdopt.c").read()
    ev.load_text(re.search(r"static AOM_FORCE_INLINE int warehouse_efficients_txb\([^;{]*\)\s*\{.*?\n}\n", rd, re.S).group(0), "txb_rdopt.c:warehouse_efficients_txb")
    for pat in (r"int av1_cost_coeffs_txb_estimate\([^;{]*\)\s*\{.*?\n}\n", r"static AOM_FORCE_INLINE int warehouse_efficients_txb_laplacian\([^;{]*\)\s*\{.*?\n}\n"):
        ev.load_text(re.search(pat, rd, re.S).group(0), "txb_rdopt.c:" + pat[:40])
    bad = [s for s in ev.skipped if s[0].startswith(("txb_rdopt", "encodetxb.c", "block.h"))]
    assert not bad, bad
    return ev, state


def extra_xd_members_decl(members):
    return "typedef struct macroblockd { " + members + " } MACROBLOCKD;\n"


def main():
    import pyoracle as orc   # scan orders as INPUTS (pinned separately)
    ev, state = make_txb_evaluator()
    rng = np.random.default_rng(20261114)
    arrays, cases = {}, []
    k = 0
    for tx_size in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 13, 14, 15, 16, 17, 18):
        W, H = TXW[tx_size], TXH[tx_size]
        w, h = min(W, 32), min(H, 32)
        n = w * h
        for tx_type in ((0, 10, 11) if W <= 16 and H <= 16 else (0,)):
            tx_class = 0 if tx_type < 10 else (2 if tx_type == 10 else 1)
            scan, iscan = orc.get_scan(tx_size, tx_type)
            so = ev.new("SCAN_ORDER")
            ev.set(so, "scan", ev.array(scan, "int16_t")); ev.set(so, "iscan", ev.array(iscan, "int16_t"))
            state["scan_order"] = so
            for trial in range(3 if n <= 256 else 2):
                eob = [n, max(2, n // 6), 1][trial] if n > 16 or trial < 2 else 1
                costs = rng.integers(10, 4000, N_COSTS + 22)
                coeff = np.zeros(n, np.int64)
                mags = rng.choice([0, 0, 1, 1, 2, 3, 4, 9, 14, 15, 16, 40, 3000], n)
                coeff[scan[:eob]] = (mags * rng.choice([-1, 1], n))[:eob]
                coeff[scan[eob - 1]] = [-1, 3, 20][trial]
                if trial == 1:
                    coeff[0] = -17
                x = ev.new("MACROBLOCK")
                cc = ev.new("LV_MAP_COEFF_COST")
                names = (("txb_skip_cost", 13, 2), ("base_eob_cost", 4, 3), ("base_cost", 42, 8), ("eob_extra_cost", 9, 2), ("dc_sign_cost", 3, 2), ("lps_cost", 21, 26))
                o = 0
                for nm, a, b in names:
                    for i in range(a):
                        for j in range(b):
                            ev.set(cc, "%s[%d][%d]" % (nm, i, j), int(costs[o])); o += 1
                assert o == N_COSTS
                ems = int(ev.global_values("txsize_log2_minus4")[tx_size])
                plane_type = k % 2
                for i in range(2):
                    for j in range(11):
                        ev.set(x, "coeff_costs.eob_costs[%d][%d].eob_cost[%d][%d]" % (ems, plane_type, i, j), int(costs[N_COSTS + i * 11 + j]))
                p = ev.new("macroblock_plane") if "macroblock_plane" in ev.typedefs else ev.interp.alloc(ev.structs["macroblock_plane"], True)
                ev.set(p, "qcoeff", ev.array(coeff, "int32_t"))
                tc = ev.new("TXB_CTX")
                skip_ctx, dc_ctx = int(rng.integers(0, 13)), int(rng.integers(0, 3))
                ev.set(tc, "txb_skip_ctx", skip_ctx); ev.set(tc, "dc_sign_ctx", dc_ctx)
                xd = ev.new("MACROBLOCKD")
                cost = ev.call("warehouse_efficients_txb", x, plane_type, 0, tx_size, tc, p, eob, plane_type, cc, xd, tx_type, tx_class, 0)
                # the Laplacian form on the same block (plane 0: av1_cost_coeffs_txb_estimate asserts it): x->plane[0] carries the coefficients and the eob
                ev.set(x, "plane[0].qcoeff", ev.array(coeff, "int32_t")); ev.set(x, "plane[0].eobs", ev.array([eob], "uint16_t"))
                for i in range(2):
                    for j in range(11):
                        ev.set(x, "coeff_costs.eob_costs[%d][0].eob_cost[%d][%d]" % (ems, i, j), int(costs[N_COSTS + i * 11 + j]))
                lap = ev.call("warehouse_efficients_txb_laplacian", x, 0, 0, tx_size, tc, eob, 0, cc, xd, tx_type, tx_class, 0)
                arrays["c%d" % k], arrays["t%d" % k] = coeff.astype(np.int32), costs.astype(np.int32)
                cases.append({"k": k, "tx_size": tx_size, "tx_type": tx_type, "tx_class": tx_class, "eob": eob, "txb_skip_ctx": skip_ctx, "dc_sign_ctx": dc_ctx,
                              "cost": int(cost), "cost_laplacian": int(lap),
                              "entropy_ctx": int(ev.call("av1_get_txb_entropy_context", ev.array(coeff, "int32_t"), so, eob))})
                k += 1
        print(tx_size, k, flush=True)
    # av1_get_txb_entropy_context below its saturation: small sums, the three DC signs (TX_4X4, default scan)
    scan, iscan = orc.get_scan(0, 0)
    so = ev.new("SCAN_ORDER")
    ev.set(so, "scan", ev.array(scan, "int16_t")); ev.set(so, "iscan", ev.array(iscan, "int16_t"))
    small = []
    for vals, eob in (([1], 1), ([-1], 1), ([0, 1], 2), ([2, -1, 1], 3), ([-3, 0, 2, 1], 4), ([0, 0, 0, 5], 4), ([1, 1, 1, 1, 1, 1, 1], 7), ([-1, 0, 0, 0, 0, 0, 0, 7], 8),
                      ([6, 2], 1)):
        coeff = np.zeros(16, np.int64)
        coeff[scan[:len(vals)]] = vals
        small.append({"coeff": coeff.tolist(), "eob": eob, "entropy_ctx": int(ev.call("av1_get_txb_entropy_context", ev.array(coeff, "int32_t"), so, eob))})
    arrays["small_ctx"] = np.frombuffer(json.dumps(small).encode(), np.uint8)
    save("ref_eval_txb_cost.npz", arrays, cases)


if __name__ == "__main__":
    main()
