#!/usr/bin/env python3
"""Golden vectors of the RD path's luma rate / distortion estimate from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_yrd.npz   av1_estimate_txfm_yrd (av1/encoder/tx_search.c:3016-3139) AS IT IS WRITTEN -- the loop over the block's transform blocks with
                     get_txb_ctx on the running above / left contexts, av1_setup_xform / av1_setup_quant, av1_xform, av1_quant, cost_coeffs ->
                     av1_cost_coeffs_txb, dist_block_tx_domain, av1_merge_rd_stats, av1_set_txb_context, the header-rate tail and the forced-skip
                     check -- with its callees interpreted too: av1_fwd_txfm -> av1_highbd_fwd_txfm -> av1_fwd_txfm2d_WxH_c (hybrid_fwd_txfm.c,
                     av1_fwd_txfm2d.c), av1_[highbd_]quantize_b_facade -> aom_[highbd_]quantize_b*_c (av1_quantize.c, quantize.c),
                     av1_[highbd_]block_error_c (rdopt.c), av1_get_txb_entropy_context, av1_get_entropy_contexts (rd.c), txfm_partition_context,
                     av1_get_skip_txfm_context, the level-map coder's rate (gen_ref_eval_txb_cost.py's evaluator).
                     Cases: inter blocks 8x8 .. 32x32 and rectangles (one transform block), 8 / 10-bit, skipping and non-skipping quantisers,
                     random above / left contexts; and the 64- and 128-class blocks (1 / 2 / 4 transform blocks of 64x64: the context update between
                     them), 64x64 transforms interpreted like the rest (the generator takes about a minute).

Supplied as inputs / adaptations:
  * MACROBLOCK / MACROBLOCKD / MB_MODE_INFO / AV1_COMP as views holding the members the functions read; enums the evaluator skips are ints;
  * get_scan returns the scan order built from the separately pinned scan tables; get_tx_type_cost returns mode_costs' DCT_DCT entry as a given number;
  * the second half of the file pins the RD-based second-MV choice: RDCOST(x->rdmult, mv_rate + stats.rate, stats.dist) of
    av1/encoder/motion_search_facade.c:378-425 evaluated as written on given (rate, dist, mv_rate) pairs.
"""
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_txb_cost as T  # noqa: E402
from gen_ref_eval_golden import REF, save  # noqa: E402

TXW, TXH = T.TXW, T.TXH
BSIZES = ["BLOCK_4X4", "BLOCK_4X8", "BLOCK_8X4", "BLOCK_8X8", "BLOCK_8X16", "BLOCK_16X8", "BLOCK_16X16", "BLOCK_16X32", "BLOCK_32X16", "BLOCK_32X32", "BLOCK_32X64",
          "BLOCK_64X32", "BLOCK_64X64", "BLOCK_64X128", "BLOCK_128X64", "BLOCK_128X128", "BLOCK_4X16", "BLOCK_16X4", "BLOCK_8X32", "BLOCK_32X8", "BLOCK_16X64", "BLOCK_64X16"]


def cut(text, start):
    """The definition that begins with `start` (a function or an initialised table), by brace matching."""
    a = text.index(start)
    i = text.index("{", a)
    depth = 0
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                break
        i += 1
    j = i + 1
    if start.startswith("typedef") or "=" in text[a:text.index("{", a)]:   # `} NAME;` / `};`
        j = text.index(";", j) + 1
    return text[a:j] + "\n"


PRE_VIEWS = """
typedef uint8_t TXFM_CONTEXT;
typedef int BLOCK_SIZE; typedef int TX_MODE; typedef int TxSetType; typedef uint8_t qm_val_t;
typedef struct { int flags; } YV12_BUFFER_CONFIG;
#define YV12_FLAG_HIGHBITDEPTH 8
typedef struct MB_MODE_INFO { int bsize; int tx_size; int segment_id; int8_t ref_frame[2]; int use_intrabc; int skip_txfm; } MB_MODE_INFO;
struct macroblockd_plane { int subsampling_x; int subsampling_y; ENTROPY_CONTEXT *above_entropy_context; ENTROPY_CONTEXT *left_entropy_context; };
typedef struct { int tx_mode_search_type; int use_qm_dist_metric; } TxfmSearchParams;
typedef struct { int txfm_partition_cost[21][2]; int skip_txfm_cost[3][2]; } ModeCosts;
typedef struct { int rate; int zero_rate; int64_t dist; int64_t rdcost; int64_t sse; uint8_t skip_txfm; } RD_STATS;
"""
XD_MEMBERS = ("MB_MODE_INFO **mi; struct macroblockd_plane plane[3]; TXFM_CONTEXT *above_txfm_context; TXFM_CONTEXT *left_txfm_context; MB_MODE_INFO *above_mbmi; "
              "MB_MODE_INFO *left_mbmi; int lossless[8]; int bd; const YV12_BUFFER_CONFIG *cur_buf; int mb_to_right_edge; int mb_to_bottom_edge;")
PLANE_MEMBERS = ("int16_t *src_diff; tran_low_t *coeff; tran_low_t *dqcoeff; uint8_t *txb_entropy_ctx; const int16_t *zbin_QTX; const int16_t *round_QTX; "
                 "const int16_t *quant_QTX; const int16_t *quant_shift_QTX; const int16_t *dequant_QTX; const int16_t *quant_fp_QTX; const int16_t *round_fp_QTX;")
MB_MEMBERS = "MACROBLOCKD e_mbd; TxfmSearchParams txfm_search_params; ModeCosts mode_costs; int rdmult; int seg_skip_block;"


def make_evaluator():
    ev, state = T.make_txb_evaluator(MB_MEMBERS, PLANE_MEMBERS, XD_MEMBERS, PRE_VIEWS)
    # the transform-type enums are UENUM1BYTE enums the evaluator skipped, and with them the tables indexed by them: enumerators in declaration order
    # (av1/common/enums.h), then the three tables' own text
    for i, n in enumerate(("DCT_DCT", "ADST_DCT", "DCT_ADST", "ADST_ADST", "FLIPADST_DCT", "DCT_FLIPADST", "FLIPADST_FLIPADST", "ADST_FLIPADST", "FLIPADST_ADST", "IDTX",
                           "V_DCT", "H_DCT", "V_ADST", "H_ADST", "V_FLIPADST", "H_FLIPADST", "TX_TYPES")):
        ev.define(n, "(%d)" % i)
    for i, n in enumerate(("DCT_1D", "ADST_1D", "FLIPADST_1D", "IDTX_1D", "TX_TYPES_1D")):
        ev.define(n, "(%d)" % i)
    ev.define("TX_TYPE_1D", "int")
    cd_h = open(REF + "av1/common/common_data.h").read()
    ev.load_text(cut(cd_h, "static const TX_TYPE_1D vtx_tab[TX_TYPES] =") + cut(cd_h, "static const TX_TYPE_1D htx_tab[TX_TYPES] ="), "common_data.h:vtx_tab, htx_tab")
    ev.load_text(cut(open(REF + "av1/common/txb_common.h").read(), "static const TX_CLASS tx_type_to_class[TX_TYPES] ="), "txb_common.h:tx_type_to_class")
    ev.load_text("typedef struct { int reduced_tx_set_used; } FeatureFlags; typedef struct AV1Common { FeatureFlags features; } AV1_COMMON;\n"
                 "typedef struct AV1_COMP { AV1_COMMON common; } AV1_COMP;\n", "encoder.h: views")
    # ---- the transforms (as gen_ref_eval_more.py loads them) and hybrid_fwd_txfm.c
    for f in ("aom_dsp/txfm_common.h", "av1/common/common.h", "av1/common/av1_txfm.h", "av1/common/av1_txfm.c", "av1/encoder/av1_fwd_txfm1d.h",
              "av1/encoder/av1_fwd_txfm1d_cfg.h", "av1/encoder/av1_fwd_txfm1d.c", "av1/encoder/av1_fwd_txfm2d.c"):
        ev.load(REF + f)
    for w, h in zip(TXW, TXH):
        ev.define("av1_fwd_txfm2d_%dx%d" % (w, h), "av1_fwd_txfm2d_%dx%d_c" % (w, h))
    ev.define("av1_fwht4x4", "av1_fwht4x4_c"); ev.define("av1_highbd_fwht4x4", "av1_highbd_fwht4x4_c"); ev.define("av1_lowbd_fwd_txfm", "av1_lowbd_fwd_txfm_c")
    ev.load(REF + "av1/encoder/hybrid_fwd_txfm.c")
    # ---- the quantisers and their facades
    for f in ("aom_dsp/quantize.h", "aom_dsp/quantize.c"):
        ev.load(REF + f)
    for n in ("aom_quantize_b", "aom_quantize_b_32x32", "aom_quantize_b_64x64", "aom_highbd_quantize_b", "aom_highbd_quantize_b_32x32", "aom_highbd_quantize_b_64x64",
              "aom_quantize_b_adaptive", "aom_quantize_b_32x32_adaptive", "aom_quantize_b_64x64_adaptive", "aom_highbd_quantize_b_adaptive",
              "aom_highbd_quantize_b_32x32_adaptive", "aom_highbd_quantize_b_64x64_adaptive"):
        ev.define(n, n + "_c")
    aq_h = open(REF + "av1/encoder/av1_quantize.h").read()
    ev.load_text(cut(aq_h, "typedef struct QUANT_PARAM {"), "av1_quantize.h:QUANT_PARAM")
    ev.load_text("typedef struct { const int16_t *scan; const int16_t *iscan; } SCAN_ORDER_;\n", "unused")
    aq = open(REF + "av1/encoder/av1_quantize.c").read()
    for sig in ("void av1_quantize_skip(", "void av1_quantize_b_facade(", "void av1_highbd_quantize_b_facade("):
        ev.load_text(cut(aq, sig), "av1_quantize.c:" + sig)
    # ---- blockd.h / common helpers the function reads
    bd_h = open(REF + "av1/common/blockd.h").read()
    for sig in ("static INLINE int is_intrabc_block(", "static INLINE int is_inter_block(", "static INLINE int is_cur_buf_hbd(", "static INLINE int block_signals_txsize(",
                "static INLINE int av1_get_max_eob(", "static INLINE PLANE_TYPE get_plane_type("):
        ev.load_text(cut(bd_h, sig), "blockd.h:" + sig)
    ev.load_text(cut(bd_h, "static const int av1_ext_tx_used[EXT_TX_SET_TYPES][TX_TYPES] ="), "blockd.h:av1_ext_tx_used")
    ev.load_text("static INLINE TxSetType av1_get_ext_tx_set_type(TX_SIZE tx_size, int is_inter, int use_reduced_set) { (void)tx_size; (void)is_inter; (void)use_reduced_set; return 0; }\n",
                 "blockd.h: av1_get_ext_tx_set_type (the value only travels in TxfmParam; the forward transform does not read it)")
    ci = open(REF + "av1/common/av1_common_int.h").read()
    for sig in ("static INLINE int max_block_wide(", "static INLINE int max_block_high(", "static INLINE TX_SIZE get_sqr_tx_size("):
        ev.load_text(cut(ci, sig), "av1_common_int.h:" + sig)
    ev.load_text(cut(open(REF + "av1/common/entropy.h").read(), "static INLINE TX_SIZE get_txsize_entropy_ctx("), "entropy.h:get_txsize_entropy_ctx")
    ev.load_text(cut(open(REF + "av1/common/idct.c").read(), "int av1_get_tx_scale("), "idct.c:av1_get_tx_scale")
    ev.load_text(cut(ci, "static INLINE int txfm_partition_context("), "av1_common_int.h:txfm_partition_context")
    ev.load_text(cut(open(REF + "av1/common/pred_common.h").read(), "static INLINE int av1_get_skip_txfm_context("), "pred_common.h:av1_get_skip_txfm_context")
    rd_c = open(REF + "av1/encoder/rd.c").read()
    ev.load_text(cut(rd_c, "static void get_entropy_contexts_plane(") + cut(rd_c, "void av1_get_entropy_contexts("), "rd.c:av1_get_entropy_contexts")
    rd_h = open(REF + "av1/encoder/rd.h").read()
    ev.load_text("#define RDDIV_BITS 7\n" + re.search(r"#define RDCOST\(RM, R, D\).*?\n\n", rd_h, re.S).group(0), "rd.h:RDCOST")
    for sig in ("static INLINE void av1_init_rd_stats(", "static INLINE void av1_invalid_rd_stats(", "static INLINE void av1_merge_rd_stats("):
        ev.load_text(cut(rd_h, sig), "rd.h:" + sig)
    ro = open(REF + "av1/encoder/rdopt.c").read()
    ev.load_text(cut(ro, "int64_t av1_highbd_block_error_c(") + cut(ro, "int64_t av1_block_error_c("), "rdopt.c:block errors")
    ev.define("av1_block_error", "av1_block_error_c"); ev.define("av1_highbd_block_error", "av1_highbd_block_error_c")
    em = open(REF + "av1/encoder/encodemb.c").read()
    ev.load_text("enum { AV1_XFORM_QUANT_FP = 0, AV1_XFORM_QUANT_B = 1, AV1_XFORM_QUANT_DC = 2, AV1_XFORM_QUANT_SKIP_QUANT, AV1_XFORM_QUANT_TYPES };\n"
                 "#define MAX_TX_SCALE 1\n#define RIGHT_SIGNED_SHIFT(value, n) ((n) < 0 ? ((value) << (-(n))) : ((value) >> (n)))\n#define LIKELY(v) (v)\n",
                 "encodemb.h / tx_search.c: constants")
    for sig in ("void av1_xform(", "void av1_setup_xform(", "void av1_setup_quant("):
        ev.load_text(cut(em, sig), "encodemb.c:" + sig)
    # av1_quant as written, with its dispatch table reduced to the B quantiser of the depth (quant_func_list[AV1_XFORM_QUANT_B][is_hbd]: encodemb.c:268-293)
    q = cut(em, "void av1_quant(")
    q, n = re.subn(r"quant_func_list\[qparam->xform_quant_idx\]\[txfm_param->is_hbd\]\(", "(txfm_param->is_hbd ? av1_highbd_quantize_b_facade : av1_quantize_b_facade)(", q)
    assert n == 1
    q = re.sub(r"#else.*?#endif", "#endif", q, flags=re.S)
    ev.load_text(q, "encodemb.c:av1_quant")
    ev.load_text(cut(open(REF + "av1/encoder/encodemb.h").read(), "static INLINE void av1_set_txb_context("), "encodemb.h:av1_set_txb_context")
    tr = open(REF + "av1/encoder/txb_rdopt.c").read()
    ev.load_text(cut(tr, "int av1_cost_coeffs_txb("), "txb_rdopt.c:av1_cost_coeffs_txb")
    ts = open(REF + "av1/encoder/tx_search.c").read()
    cc = cut(ts, "static INLINE int cost_coeffs(")
    cc = re.sub(r"#if TXCOEFF_COST_TIMER.*?#endif", "", cc, flags=re.S)
    ev.load_text("struct rdcost_block_args { MACROBLOCK *x; const AV1_COMP *cpi; ENTROPY_CONTEXT t_above[32]; ENTROPY_CONTEXT t_left[32]; RD_STATS rd_stats; "
                 "int64_t current_rd; int64_t best_rd; int exit_early; int incomplete_exit; int ftxs_mode; int skip_trellis; };\n", "tx_search.c:rdcost_block_args")
    ev.load_text(cut(ts, "static INLINE void dist_block_tx_domain(") + cc, "tx_search.c:dist_block_tx_domain, cost_coeffs")
    body = cut(ts, "int64_t av1_estimate_txfm_yrd(")
    body = body.replace("av1_zero(args);", "args.exit_early = 0; args.incomplete_exit = 0; args.current_rd = 0; args.best_rd = 0; args.ftxs_mode = 0; args.skip_trellis = 0;")
    ev.load_text(body, "tx_search.c:av1_estimate_txfm_yrd")
    bad = [s for s in ev.skipped if s[0].startswith(("tx_search", "encodemb.c", "rd.", "txb_rdopt.c:av1_cost", "encodemb.h:av1_set", "av1_quantize", "blockd.h", "rdopt"))]
    assert not bad, bad
    return ev, state


N_COSTS = T.N_COSTS
YV12_FLAG_HIGHBITDEPTH = 8
COST_NAMES = (("txb_skip_cost", 13, 2), ("base_eob_cost", 4, 3), ("base_cost", 42, 8), ("eob_extra_cost", 9, 2), ("dc_sign_cost", 3, 2), ("lps_cost", 21, 26))


def run_case(ev, state, orc, rng, bw, bh, bd, qindex, amp, ctx_mode, tx_select, use_restated_kernels, rdmult_override=None):
    """One call of av1_estimate_txfm_yrd on an inter block of bw x bh luma pixels; returns (case dict, arrays)."""
    hbd = bd > 8
    bsize = BSIZES.index("BLOCK_%dX%d" % (bw, bh))
    txw, txh = min(bw, 64), min(bh, 64)
    tx_size = [i for i in range(19) if TXW[i] == txw and TXH[i] == txh][0]
    scan, iscan = orc.get_scan(tx_size, 0)
    so = ev.new("SCAN_ORDER")
    ev.set(so, "scan", ev.array(scan, "int16_t")); ev.set(so, "iscan", ev.array(iscan, "int16_t"))
    state["scan_order"] = so
    residual = rng.integers(-amp, amp + 1, (bh, bw)).astype(np.int16)
    if amp > 8:   # some structure, so that coefficients survive the quantiser
        residual += (np.add.outer(np.arange(bh), np.arange(bw)) % 7 * (amp // 4)).astype(np.int16)
    q = orc.build_quantizer_y(bd, qindex)
    x = ev.new("MACROBLOCK")
    n4w, n4h = bw // 4, bh // 4
    above = rng.integers(0, 7, n4w) | (rng.integers(0, 3, n4w) << 3)
    left = rng.integers(0, 7, n4h) | (rng.integers(0, 3, n4h) << 3)
    if ctx_mode == 0:
        above[:] = 0; left[:] = 0
    ev.set(x, "plane[0].src_diff", ev.array(residual.ravel(), "int16_t"))
    for nm in ("coeff", "qcoeff", "dqcoeff"):
        ev.set(x, "plane[0]." + nm, ev.array(np.zeros(bw * bh, np.int32), "int32_t"))
    ev.set(x, "plane[0].eobs", ev.array(np.zeros(bw * bh // 16, np.int64), "uint16_t"))
    ev.set(x, "plane[0].txb_entropy_ctx", ev.array(np.zeros(bw * bh // 16, np.int64), "uint8_t"))
    for nm, key in (("zbin_QTX", "zbin"), ("round_QTX", "round"), ("quant_QTX", "quant"), ("quant_shift_QTX", "quant_shift"), ("dequant_QTX", "dequant")):
        ev.set(x, "plane[0]." + nm, ev.array([int(v) for v in q[key]], "int16_t"))
    mbmi = ev.new("MB_MODE_INFO")
    ev.set(mbmi, "bsize", bsize); ev.set(mbmi, "segment_id", 0); ev.set(mbmi, "ref_frame[0]", 1); ev.set(mbmi, "ref_frame[1]", -1); ev.set(mbmi, "use_intrabc", 0)
    mi = R.Ptr([mbmi], 0, R.PTR)
    ev.set(x, "e_mbd.mi", mi)
    ev.set(x, "e_mbd.plane[0].above_entropy_context", ev.array(above, "int8_t")); ev.set(x, "e_mbd.plane[0].left_entropy_context", ev.array(left, "int8_t"))
    ev.set(x, "e_mbd.plane[0].subsampling_x", 0); ev.set(x, "e_mbd.plane[0].subsampling_y", 0)
    atx, ltx = int(rng.choice([4, 8, 16, 32, 64])), int(rng.choice([4, 8, 16, 32, 64]))
    ev.set(x, "e_mbd.above_txfm_context", ev.array([atx], "uint8_t")); ev.set(x, "e_mbd.left_txfm_context", ev.array([ltx], "uint8_t"))
    skips = [int(rng.integers(0, 3)) for _ in range(2)]   # 0: no neighbour, 1: neighbour coded, 2: neighbour skipped
    for nm, sk in zip(("above_mbmi", "left_mbmi"), skips):
        if sk:
            nb = ev.new("MB_MODE_INFO")
            ev.set(nb, "skip_txfm", sk - 1)
            ev.set(x, "e_mbd." + nm, nb)
        else:
            ev.set(x, "e_mbd." + nm, None)
    ev.set(x, "e_mbd.bd", bd)
    buf = ev.new("YV12_BUFFER_CONFIG")
    ev.set(buf, "flags", YV12_FLAG_HIGHBITDEPTH if hbd else 0)
    ev.set(x, "e_mbd.cur_buf", buf)
    ev.set(x, "e_mbd.mb_to_right_edge", 0); ev.set(x, "e_mbd.mb_to_bottom_edge", 0)
    ev.set(x, "txfm_search_params.tx_mode_search_type", 2 if tx_select else 1)   # TX_MODE_SELECT / TX_MODE_LARGEST (enums.h: ONLY_4X4, TX_MODE_LARGEST, TX_MODE_SELECT)
    ev.set(x, "txfm_search_params.use_qm_dist_metric", 0)
    mode = rng.integers(20, 3000, 21 * 2 + 3 * 2)
    for i in range(21):
        for j in range(2):
            ev.set(x, "mode_costs.txfm_partition_cost[%d][%d]" % (i, j), int(mode[2 * i + j]))
    for i in range(3):
        for j in range(2):
            ev.set(x, "mode_costs.skip_txfm_cost[%d][%d]" % (i, j), int(mode[42 + 2 * i + j]))
    rdmult = int(rng.integers(40, 4000))
    if rdmult_override is not None:
        rdmult = rdmult_override
    ev.set(x, "rdmult", rdmult); ev.set(x, "seg_skip_block", 0)
    txs_ctx = int(ev.call("get_txsize_entropy_ctx", tx_size))
    ems = int(ev.global_values("txsize_log2_minus4")[tx_size])
    costs = rng.integers(10, 4000, N_COSTS + 22)
    o = 0
    for nm, a_, b_ in COST_NAMES:
        for i in range(a_):
            for j in range(b_):
                ev.set(x, "coeff_costs.coeff_costs[%d][0].%s[%d][%d]" % (txs_ctx, nm, i, j), int(costs[o])); o += 1
    for i in range(2):
        for j in range(11):
            ev.set(x, "coeff_costs.eob_costs[%d][0].eob_cost[%d][%d]" % (ems, i, j), int(costs[N_COSTS + i * 11 + j]))
    tx_type_rate = int(rng.integers(0, 900)) if max(txw, txh) <= 32 else 0   # (64-point sizes have one transform type: get_tx_type_cost returns 0)
    state["tx_type_rate"] = tx_type_rate
    cpi = ev.new("AV1_COMP")
    ev.set(cpi, "common.features.reduced_tx_set_used", 0)
    stats = ev.new("RD_STATS")
    saved = {}
    if use_restated_kernels:
        # the 64x64 transform and its quantiser from the (separately pinned) restatement; everything around them stays the reference's text
        for nm in ("av1_fwd_txfm", "av1_quantize_b_facade", "av1_highbd_quantize_b_facade"):
            saved[nm] = ev.interp.funcs.pop(nm)

        def py_fwd(it, a):
            src, dst, stride, prm = a[0][0], a[1][0], int(a[2][0]), a[3][0]
            blk = np.array([[src.buf[src.off + r * stride + c] for c in range(txw)] for r in range(txh)], np.int16)
            out = orc.fwd_txfm2d(blk, tx_size, 0, bd).ravel()
            for k_, v in enumerate(out):
                dst.buf[dst.off + k_] = int(v)
            return None, R.VOID

        def py_quant(hb):
            def f(it, a):
                co, n = a[0][0], int(a[1][0])
                c = np.array([co.buf[co.off + k_] for k_ in range(n)], np.int32)
                qc, dq, eob = orc.quantize_b(c, q, scan, iscan, int(ev.call("av1_get_tx_scale", tx_size)), highbd=hb)
                for nm_, arr in ((3, qc), (4, dq)):
                    p_ = a[nm_][0]
                    for k_, v in enumerate(arr):
                        p_.buf[p_.off + k_] = int(v)
                a[5][0].buf[a[5][0].off] = int(eob)
                return None, R.VOID
            return f
        ev.interp.pycalls["av1_fwd_txfm"] = py_fwd
        ev.interp.pycalls["av1_quantize_b_facade"] = py_quant(False)
        ev.interp.pycalls["av1_highbd_quantize_b_facade"] = py_quant(True)
    try:
        rd = ev.call("av1_estimate_txfm_yrd", cpi, x, stats, 9223372036854775807, bsize, tx_size)
    finally:
        for nm, f in saved.items():
            ev.interp.funcs[nm] = f
            ev.interp.pycalls.pop(nm, None)
    # what the function looked up for the transform-size signalling (the caller's addend in the device form): block_signals_txsize (blockd.h:1031-1033)
    # and txfm_partition_context, both interpreted
    tx_size_rate = 0
    if tx_select and int(ev.call("block_signals_txsize", bsize)):
        pctx = int(ev.call("txfm_partition_context", ev.array([atx], "uint8_t"), ev.array([ltx], "uint8_t"), bsize, tx_size))
        tx_size_rate = int(mode[2 * pctx])
    n_txb = (bw // txw) * (bh // txh)
    step = (txw // 4) * (txh // 4)
    eobs = [int(ev.field(x, "plane[0].eobs").deref()[0].buf[k * step]) for k in range(n_txb)]
    skip_ctx = sum(1 for sk in skips if sk == 2)
    case = {"bw": bw, "bh": bh, "bd": bd, "qindex": qindex, "tx_size": tx_size, "tx_select": int(tx_select), "rdmult": rdmult, "tx_type_rate": tx_type_rate, "tx_size_rate": tx_size_rate,
            "above_txfm_context": atx, "left_txfm_context": ltx, "neighbour_skip": skips, "restated_kernels": int(bool(use_restated_kernels)),
            "no_skip_txfm_rate": int(mode[42 + 2 * skip_ctx]), "skip_txfm_rate": int(mode[42 + 2 * skip_ctx + 1]),
            "rd": str(int(rd)), "rate": int(ev.get(stats, "rate")), "dist": str(int(ev.get(stats, "dist"))), "sse": str(int(ev.get(stats, "sse"))),
            "skip_txfm": int(ev.get(stats, "skip_txfm")), "eobs": eobs, "mbmi_tx_size": int(ev.get(mbmi, "tx_size"))}
    arrays = {"res": residual, "above": above.astype(np.uint8), "left": left.astype(np.uint8), "costs": costs.astype(np.int32), "mode": mode.astype(np.int32)}
    return case, arrays


def main():
    import time
    import pyoracle as orc   # scan orders / quantiser tables as INPUTS, and the 64x64 kernels of the 128-class cases (all pinned separately)
    ev, state = make_evaluator()
    ev.interp.pycalls["get_tx_type_cost"] = lambda it, a: (state["tx_type_rate"], R.I32)
    rng = np.random.default_rng(20261205)
    arrays, cases = {}, []
    plan = []
    for (bw, bh) in ((8, 8), (16, 16), (16, 8), (8, 16), (4, 4), (32, 32), (32, 16), (4, 16)):
        for bd in (8, 10):
            for qindex, amp in ((40, 60), (160, 24), (255, 3)):
                if bw * bh > 512 and not (qindex == 160):
                    continue
                plan.append((bw, bh, bd, qindex, amp << (bd - 8), len(plan) % 3 != 0, len(plan) % 2 == 0, False))
    for (bw, bh) in ((128, 128), (128, 64), (64, 128), (64, 64)):
        for bd, qindex, amp in ((8, 120, 40), (10, 200, 30 << 2)):
            plan.append((bw, bh, bd, qindex, amp, True, len(plan) % 2 == 0, False))
    plan.append((128, 128, 8, 255, 2, True, True, False))   # every transform block skipped
    # a few coefficients survive a fine quantiser but cost more to code than the block's whole energy: the forced-skip check takes over
    for (bw, bh) in ((8, 8), (16, 16), (32, 32), (128, 64)):
        for bd in (8, 10):
            plan.append((bw, bh, bd, 24, 5 << (bd - 8), True, len(plan) % 2 == 0, False, 60000))
    t0 = time.time()
    for k, item in enumerate(plan):
        bw, bh, bd, qindex, amp, ctxm, txsel, restated = item[:8]
        c, a = run_case(ev, state, orc, rng, bw, bh, bd, qindex, amp, int(ctxm), txsel, restated, item[8] if len(item) > 8 else None)
        c["k"] = k
        cases.append(c)
        for nm, v in a.items():
            arrays["%s%d" % (nm, k)] = v
        print(k, bw, bh, bd, qindex, c["rate"], c["dist"], c["skip_txfm"], c["eobs"], "%.0f s" % (time.time() - t0), flush=True)
    save("ref_eval_yrd.npz", arrays, cases)


if __name__ == "__main__":
    main()
