#!/usr/bin/env python3
"""Golden vectors for TPL's mode_estimation AS IT IS WRITTEN (av1/encoder/tpl_model.c:438-1021), obtained by interpreting the WHOLE function for every
block of some rows of a frame in raster order, as av1_mc_flow_dispenser_row calls it (build container only; tests/golden/ref_c_eval.py, views of
gen_ref_eval_composites.py): the neighbours' TPL stats gathered into center_mvs with is_alike_mv (:652-683), the prune_starting_mv block on the
candidates' SADs (:706-731), motion_estimation (:248-301, the function itself, with av1_full_pixel_search and the sub-pel tree under it) from every
remaining candidate, the best candidate per reference (:733-746), the predictor and tpl_get_satd_cost per reference (:748-757), the best reference
(:759-765), the mode decision against the intra cost, tpl_stats' fields at the end, and tpl_model_store (:1164-1182) between blocks.
ref_eval_tpl.npz (round 5) holds slices of the same statements on synthetic inputs; here the function runs and its blocks feed each other.

Supplied as inputs / adaptations (frame plumbing, the intra leg and the evaluator's memory model):
  * the intra leg: av1_predict_intra_block does nothing and tpl_get_satd_cost returns a GIVEN cost while xd->mi[0]->ref_frame[0] == INTRA_FRAME and
    no reference has been searched yet (intra_cost of mode m for block b: meta["intra_costs"]); the intra predictors are out of scope;
  * get_rate_distortion (the final encode of the block and its rate, :317-436) writes fixed values: its callers' bookkeeping runs, its own work is
    aomhip_tpl_rate_distortion's subject (ref_eval_composites.npz);
  * av1_enc_build_one_inter_predictor writes what the oracle's predictor (pinned by ref_eval_convolve.npz) gives for the MV -- as in
    gen_ref_eval_joint.py; tpl_get_satd_cost on it is the reference's own (av1_subtract_block, av1_quick_txfm -> av1_fwd_txfm2d_16x16_c, aom_satd_c);
  * set_mode_info_offsets / set_plane_n4 do nothing, set_mi_row_col is replaced by the four assignments the function reads afterwards
    (up_available, left_available, mb_to_right_edge, mb_to_bottom_edge: av1_common_int.h:1362-1390), av1_num_planes is 1;
  * x->mv_limits per block is an input (av1_set_mv_row/col_limits in the dispenser); cpi->third_pass_ctx is NULL, use_ducky_encode 0,
    allow_compound_pred 0 (the compound loop does not run: aomhip_joint_motion_search_batch's subject);
  * aom_memalign / aom_free: typed allocations (the predictor buffer typed by bit depth); memset(tpl_stats, 0, ..) is a field-wise zeroing; qsort is a
    stable sort through the reference's comparator (as gen_ref_eval_tpl.py); int_mv is a struct holding as_mv with as_int operations written
    component-wise, `center_mv_t center_mvs[4] = { { { 0 }, INT_MAX }, .. }` and `int_mv best_rfidx_mv = { 0 }` become loops / assignments, the
    struct buf_2d / comp_ref_frames initialisers are kept.

Output: tests/golden/ref_eval_tpl_mode.npz.
"""
import json
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402
import gen_ref_eval_composites as C  # noqa: E402
import gen_ref_eval_compound_search as CS  # noqa: E402
import gen_ref_eval_single_caller as SC  # noqa: E402
import gen_ref_eval_yrd as Y  # noqa: E402  (cut())
import pyoracle as orc  # noqa: E402  (the predictor only)

REF = G.REF
W, H, BORDER = G.W, G.H, G.BORDER
I32, I64, U8, PTR = R.I32, R.I64, R.U8, R.PTR
BS = 16               # tpl_bsize_1d: BLOCK_16X16 / TX_16X16
N_REFS = 7
OPER = SC.OPER


def rewrite_as_int(t):
    """int_mv's as_int operations, component-wise on as_mv (mv.h:26-34: the same bits)"""
    inv = lambda a, acc: "(%s%sas_mv.row == INVALID_MV_ROW_COL && %s%sas_mv.col == INVALID_MV_ROW_COL)" % (a, acc, a, acc)
    setinv = lambda a, acc: "%s%sas_mv.row = INVALID_MV_ROW_COL; %s%sas_mv.col = INVALID_MV_ROW_COL" % (a, acc, a, acc)
    eq = lambda a1, c1, a2, c2: "(%s%sas_mv.row == %s%sas_mv.row && %s%sas_mv.col == %s%sas_mv.col)" % (a1, c1, a2, c2, a1, c1, a2, c2)
    t = re.sub(OPER + r" == INVALID_MV", lambda m: inv(m.group(1), m.group(2)), t)
    t = re.sub(OPER + r" != INVALID_MV", lambda m: "!" + inv(m.group(1), m.group(2)), t)
    t = re.sub(OPER + r" == " + OPER, lambda m: eq(m.group(1), m.group(2), m.group(3), m.group(4)), t)
    t = re.sub(OPER + r" != " + OPER, lambda m: "!" + eq(m.group(1), m.group(2), m.group(3), m.group(4)), t)
    t = re.sub(OPER + r" = INVALID_MV;", lambda m: "{ " + setinv(m.group(1), m.group(2)) + "; }", t)
    t = re.sub(OPER + r" =\s+" + OPER + ";", lambda m: "%s%sas_mv = %s%sas_mv;" % (m.group(1), m.group(2), m.group(3), m.group(4)), t)
    assert "as_int" not in t, re.findall(r".{40}as_int.{20}", t)
    return t


def adapt(fn):
    t = fn.replace("static AOM_INLINE void mode_estimation(", "void tpl_mode_est(")
    for old, new in (("uint8_t *predictor8 = aom_memalign(32, tpl_block_pels * 2 * sizeof(uint8_t));", "uint8_t *predictor8 = tpl_alloc_pred(tpl_block_pels);"),
                     ("int16_t *src_diff = aom_memalign(32, tpl_block_pels * sizeof(int16_t));", "int16_t *src_diff = tpl_alloc_i16(tpl_block_pels);"),
                     ("tran_low_t *coeff = aom_memalign(32, tpl_block_pels * sizeof(tran_low_t));", "tran_low_t *coeff = tpl_alloc_i32(tpl_block_pels);"),
                     ("tran_low_t *qcoeff = aom_memalign(32, tpl_block_pels * sizeof(tran_low_t));", "tran_low_t *qcoeff = tpl_alloc_i32(tpl_block_pels);"),
                     ("tran_low_t *dqcoeff = aom_memalign(32, tpl_block_pels * sizeof(tran_low_t));", "tran_low_t *dqcoeff = tpl_alloc_i32(tpl_block_pels);"),
                     ("memset(tpl_stats, 0, sizeof(*tpl_stats));", "tpl_zero_stats(tpl_stats);"),
                     ("int_mv best_rfidx_mv = { 0 };", "int_mv best_rfidx_mv; best_rfidx_mv.as_mv.row = 0; best_rfidx_mv.as_mv.col = 0;"),
                     ("sizeof(center_mvs[0])", "8")):
        assert t.count(old) == 1, old
        t = t.replace(old, new)
    t, n = re.subn(r"center_mv_t center_mvs\[4\] = \{ \{ \{ 0 \}, INT_MAX \},\s*\{ \{ 0 \}, INT_MAX \},\s*\{ \{ 0 \}, INT_MAX \},\s*\{ \{ 0 \}, INT_MAX \} \};",
                   "center_mv_t center_mvs[4]; for (int z_ = 0; z_ < 4; z_++) { center_mvs[z_].mv.as_mv.row = 0; center_mvs[z_].mv.as_mv.col = 0; center_mvs[z_].sad = INT_MAX; }", t)
    assert n == 1
    t, n = re.subn(r"set_mi_row_col\(xd, &xd->tile, mi_row, mi_height, mi_col, mi_width,\s*cm->mi_params\.mi_rows, cm->mi_params\.mi_cols\);",
                   "xd->up_available = (mi_row != 0); xd->left_available = (mi_col > xd->tile.mi_col_start);\n"
                   "  xd->mb_to_bottom_edge = GET_MV_SUBPEL((cm->mi_params.mi_rows - mi_height - mi_row) * MI_SIZE);\n"
                   "  xd->mb_to_right_edge = GET_MV_SUBPEL((cm->mi_params.mi_cols - mi_width - mi_col) * MI_SIZE);", t)
    assert n == 1
    return rewrite_as_int(t)


TX_SIZES = ("TX_4X4", "TX_8X8", "TX_16X16", "TX_32X32", "TX_64X64", "TX_4X8", "TX_8X4", "TX_8X16", "TX_16X8", "TX_16X32", "TX_32X16", "TX_32X64", "TX_64X32",
            "TX_4X16", "TX_16X4", "TX_8X32", "TX_32X8", "TX_16X64", "TX_64X16")
TX_TYPES = ("DCT_DCT", "ADST_DCT", "DCT_ADST", "ADST_ADST", "FLIPADST_DCT", "DCT_FLIPADST", "FLIPADST_FLIPADST", "ADST_FLIPADST", "FLIPADST_ADST", "IDTX", "V_DCT",
            "H_DCT", "V_ADST", "H_ADST", "V_FLIPADST", "H_FLIPADST", "TX_TYPES")


def setup():
    ev = CS.make_evaluator()
    C.view(ev, "int_mv", [("as_mv", ev.structs["mv"])])
    ev.define("as_fullmv", "as_mv")
    R.ALIASED_STRUCTS.add(frozenset(("mv", "fullpel_mv")))
    ev.load(REF + "av1/common/common_data.c")
    C.grab(ev, "av1/encoder/encoder.h", "cond_cost_list_const")
    C.grab(ev, "av1/encoder/encoder.h", "cond_cost_list")
    ev.load_text(re.search(r"enum \{\s*SS_CFG_SRC = 0,.*?\} UENUM1BYTE\(SS_CFG_OFFSET\);", open(REF + "av1/encoder/encoder.h").read(), re.S).group(0)
                 .replace("UENUM1BYTE(SS_CFG_OFFSET)", "SS_CFG_OFFSET_e"), "encoder.h:SS_CFG_OFFSET")
    # ---- the transform and the SATD under tpl_get_satd_cost
    ev.load(REF + "aom_dsp/txfm_common.h")
    # enumerators of UENUM1BYTE enums the evaluator skipped, in declaration order (av1/common/enums.h)
    for names in (TX_SIZES + ("TX_SIZES_ALL",), TX_TYPES):
        for i, n in enumerate(names):
            ev.define(n, "(%d)" % i)
    n_skipped = len(ev.skipped)
    pm = open(REF + "av1/common/enums.h").read()
    pm = re.search(r"enum \{\s*DC_PRED,.*?\} UENUM1BYTE\(PREDICTION_MODE\);", pm, re.S).group(0)
    ev.load_text(pm.replace("UENUM1BYTE(PREDICTION_MODE)", "PREDICTION_MODE_e"), "enums.h:PREDICTION_MODE")
    for n, v in (("TX_SIZE", "int"), ("TX_TYPE", "int"), ("PREDICTION_MODE", "int"), ("TxSetType", "int"), ("EXT_TX_SET_ALL16", "(5)"), ("FILTER_INTRA_MODES", "(5)"),
                 ("TPL_DEP_COST_SCALE_LOG2", "(4)"), ("INTER_REFS_PER_FRAME", "(7)")):
        if n not in ev.globs and n not in ev.typedefs:
            ev.define(n, v)
    for f in ("av1/common/common.h", "av1/common/av1_txfm.h", "av1/common/av1_txfm.c", "av1/encoder/av1_fwd_txfm1d.h", "av1/encoder/av1_fwd_txfm1d_cfg.h",
              "av1/encoder/av1_fwd_txfm1d.c", "av1/encoder/av1_fwd_txfm2d.c"):
        ev.load(REF + f)
    for w, h in zip(Y.TXW, Y.TXH):
        ev.define("av1_fwd_txfm2d_%dx%d" % (w, h), "av1_fwd_txfm2d_%dx%d_c" % (w, h))
    ev.define("av1_fwht4x4", "av1_fwht4x4_c"); ev.define("av1_highbd_fwht4x4", "av1_highbd_fwht4x4_c"); ev.define("av1_lowbd_fwd_txfm", "av1_lowbd_fwd_txfm_c")
    ev.load_text("typedef struct { int bit_depth; int use_highbitdepth_buf; } BitDepthInfo;\n", "blockd.h:BitDepthInfo (view)")
    ev.load(REF + "av1/encoder/hybrid_fwd_txfm.c")
    ev.load_text(Y.cut(open(REF + "aom_dsp/avg.c").read(), "int aom_satd_c("), "avg.c:aom_satd_c")
    ev.define("aom_satd", "aom_satd_c")
    sub = open(REF + "aom_dsp/subtract.c").read()
    ev.load_text(Y.cut(sub, "void aom_subtract_block_c(") + Y.cut(sub, "void aom_highbd_subtract_block_c("), "subtract.c")
    ev.define("aom_subtract_block", "aom_subtract_block_c"); ev.define("aom_highbd_subtract_block", "aom_highbd_subtract_block_c")
    ev.load_text(Y.cut(open(REF + "av1/encoder/encodemb.c").read(), "void av1_subtract_block("), "encodemb.c:av1_subtract_block")
    bd_h = open(REF + "av1/common/blockd.h").read()
    ev.load_text(Y.cut(bd_h, "static INLINE BitDepthInfo get_bit_depth_info("), "blockd.h:get_bit_depth_info")
    for i, n in enumerate(("DCT_1D", "ADST_1D", "FLIPADST_1D", "IDTX_1D", "TX_TYPES_1D")):
        ev.define(n, "(%d)" % i)
    ev.define("TX_TYPE_1D", "int")
    cd_h = open(REF + "av1/common/common_data.h").read()
    for tab in ("tx_size_wide_log2", "tx_size_high_log2", "tx_size_wide", "tx_size_high", "tx_size_wide_unit", "tx_size_high_unit"):   # (skipped while TX_SIZES_ALL was unknown)
        ev.load_text(Y.cut(cd_h, "static const int %s[TX_SIZES_ALL] =" % tab), "common_data.h:" + tab)
    ev.load_text(Y.cut(cd_h, "static const TX_TYPE_1D vtx_tab[TX_TYPES] =") + Y.cut(cd_h, "static const TX_TYPE_1D htx_tab[TX_TYPES] ="), "common_data.h:vtx_tab, htx_tab")
    ev.define("EXT_TX_SET_TYPES", "(6)")
    ev.load_text(Y.cut(bd_h, "static const int av1_ext_tx_used[EXT_TX_SET_TYPES][TX_TYPES] ="), "blockd.h:av1_ext_tx_used")
    assert not ev.skipped[n_skipped:], ev.skipped[n_skipped:]
    return ev


def build(ev):
    """views, supplied functions and the function's own text -> (enc, state)"""
    enc = C.Encoder(ev)
    S = ev.structs
    tpl_sf_t = C.view(ev, "TPL_SPEED_FEATURES", [(f, I32) for f in ("prune_intra_modes", "prune_starting_mv", "skip_alike_starting_mv", "reduce_first_step_size",
                                                                     "subpel_force_stop", "search_method", "allow_compound_pred", "use_y_only_rate_distortion")],
                      opaque=False)
    S["SPEED_FEATURES"].fields.append(("tpl_sf", tpl_sf_t))
    S["MV_SPEED_FEATURES"].fields.append(("disable_second_mv", I32))
    enc.yv12_t.fields += [("y_width", I32), ("y_height", I32), ("u_buffer", ("ptr", U8)), ("v_buffer", ("ptr", U8)), ("uv_stride", I32), ("subsampling_x", I32),
                          ("subsampling_y", I32)]
    imv = ev.typedefs["int_mv"]
    stats_t = C.view(ev, "TplDepStats", [("srcrf_sse", I64), ("srcrf_dist", I64), ("recrf_sse", I64), ("recrf_dist", I64), ("intra_sse", I64), ("intra_dist", I64),
                                          ("cmp_recrf_dist", ("arr", I64, 2)), ("mc_dep_rate", I64), ("mc_dep_dist", I64), ("pred_error", ("arr", I64, N_REFS)),
                                          ("intra_cost", I32), ("inter_cost", I32), ("srcrf_rate", I32), ("recrf_rate", I32), ("intra_rate", I32),
                                          ("cmp_recrf_rate", ("arr", I32, 2)), ("mv", ("arr", imv, N_REFS)), ("ref_frame_index", ("arr", R.I8, 2))], opaque=False)
    frame_t = C.view(ev, "TplDepFrame", [("tpl_stats_ptr", ("ptr", stats_t)), ("rec_picture", ("ptr", enc.yv12_t)), ("stride", I32)], opaque=False)
    params_t = C.view(ev, "TplParams", [("tpl_stats_block_mis_log2", U8), ("tpl_bsize_1d", U8), ("frame_idx", I32), ("tpl_frame", ("ptr", frame_t)),
                                         ("src_ref_frame", ("arr", ("ptr", enc.yv12_t), N_REFS)), ("ref_frame", ("arr", ("ptr", enc.yv12_t), N_REFS)),
                                         ("sf", S["scale_factors"])], opaque=False)
    gf_t = C.view(ev, "GF_GROUP", [("size", I32)], opaque=False)
    S["<opaque>AV1_PRIMARY"].fields += [("gf_group", gf_t), ("tpl_data", params_t)]
    seq_t = C.view(ev, "SequenceHeader", [("sb_size", I32), ("enable_intra_edge_filter", I32)], opaque=False)
    cm = S["<opaque>AV1_COMMON"] if "<opaque>AV1_COMMON" in S else S["AV1_COMMON"]
    cm.fields += [("seq_params", ("ptr", seq_t)), ("error", ("ptr", I32))]
    mip = S["<opaque>CommonModeInfoParams"] if "<opaque>CommonModeInfoParams" in S else S["CommonModeInfoParams"]
    mip.fields.append(("mi_stride", I32))
    enc.cpi_t.fields += [("gf_frame_index", I32), ("use_ducky_encode", I32), ("third_pass_ctx", ("ptr", I32)), ("mbmi_ext_info", I32)]
    tile_t = C.view(ev, "TileInfo", [("mi_row_start", I32), ("mi_row_end", I32), ("mi_col_start", I32), ("mi_col_end", I32)], opaque=False)
    enc.xd_t.fields += [("tile", tile_t), ("up_available", I32), ("left_available", I32), ("mb_to_right_edge", I32), ("mb_to_bottom_edge", I32)]
    S["macroblockd_plane"].fields += [("subsampling_x", I32), ("subsampling_y", I32)]
    enc.mbmi_t.fields += [("compound_idx", I32), ("ref_mv_idx", I32)]
    C.view(ev, "InterPredParams", [("conv_params", I32)], opaque=False)
    ev.define("int_interpfilters", "int")
    C.grab(ev, "av1/common/mv.h", "convert_fullmv_to_mv")
    state = dict(log=[])
    pyc = ev.interp.pycalls
    nothing = lambda it, a: (None, R.VOID)
    for f in ("set_mode_info_offsets", "set_plane_n4", "av1_predict_intra_block", "aom_free", "av1_setup_pre_planes"):
        pyc[f] = nothing
    pyc["av1_num_planes"] = lambda it, a: (1, I32)
    pyc["av1_broadcast_interp_filter"] = lambda it, a: (0, I32)
    pyc["get_conv_params"] = lambda it, a: (0, I32)
    pyc["use_fine_search_interval"] = lambda it, a: (0, I32)
    pyc["av1_get_scaled_ref_frame"] = lambda it, a: (None, PTR)
    ct = lambda: "uint8_t" if state["bd"] == 8 else "uint16_t"
    pyc["tpl_alloc_pred"] = lambda it, a: (ev.array([0] * int(a[0][0]), ct()), PTR)
    pyc["tpl_alloc_i16"] = lambda it, a: (ev.array([0] * int(a[0][0]), "int16_t"), PTR)
    pyc["tpl_alloc_i32"] = lambda it, a: (ev.array([0] * int(a[0][0]), "int32_t"), PTR)

    def init_inter_params(it, a):   # av1_init_inter_params(&params, bw, bh, pix_row, pix_col, ssx, ssy, bd, hbd, is_intrabc, sf, &ref_buf, kernel)
        ev.globs["g_tpl_phase_intra"].buf[0] = 0
        buf0 = a[11][0].deref()[0].f["buf0"].deref()[0]
        which = [r for r, p in state["ref_ptrs"].items() if p.buf is buf0.buf]
        assert len(which) == 1 and buf0.off == BORDER * buf0_stride(state) + BORDER
        state["pred_ref"], state["pred_pos"] = which[0], (int(a[4][0]), int(a[3][0]))
        assert (int(a[1][0]), int(a[2][0])) == (BS, BS)
        return (None, R.VOID)
    pyc["av1_init_inter_params"] = init_inter_params

    def one_inter_predictor(it, a):   # av1_enc_build_one_inter_predictor(dst, dst_stride, &mv, &params)
        dst, mv = a[0][0], a[2][0].deref()[0]
        row, col = int(mv.f["row"].deref()[0]), int(mv.f["col"].deref()[0])
        bx, by = state["pred_pos"]
        assert int(a[1][0]) == BS and (bx, by) == state["pos"]
        blk = np.zeros(1, [("bx", "<i2"), ("by", "<i2")])
        blk["bx"], blk["by"] = bx, by
        plane = orc.build_inter_pred(state["refs"][state["pred_ref"]], BORDER, W, H, BS, BS, blk, [(row, col)], 0, 0, bd=state["bd"])
        for i, v in enumerate(plane[by:by + BS, bx:bx + BS].ravel()):
            dst.add(i).store(int(v), I32)
        state["log"].append(["pred", state["pred_ref"], row, col])
        return (None, R.VOID)
    pyc["av1_enc_build_one_inter_predictor"] = one_inter_predictor

    def log_me(it, a):
        state["log"].append(["me"] + [int(v[0]) for v in a])
        return (None, R.VOID)
    pyc["tpl_log_me"] = log_me

    def rate_distortion(it, a):   # get_rate_distortion(&rate, &recon_error, &pred_error, .., ref_frame_ptr (9), .., best_mode (13), .., tpl_txfm_stats (17))
        k = len([e for e in state["log"] if e[0] == "rd"])
        refp = a[9][0]
        r0 = None if refp is None else refp.deref()[0]
        which = None if r0 is None else [r for r, p in state["yv12"].items() if p.buf is r0.buf and p.off == r0.off]
        vals = (100 + 10 * k + state["pos"][0], 1000 + 100 * k + state["pos"][1], 3000 + 7 * k)
        a[0][0].store(vals[0], I32); a[1][0].store(vals[1], I64); a[2][0].store(vals[2], I64)
        state["log"].append(["rd", int(a[13][0]), which and which[0], a[17][0] is not None, *vals])
        return (None, R.VOID)
    pyc["get_rate_distortion"] = rate_distortion
    text = open(REF + "av1/encoder/tpl_model.c").read()
    helpers = "".join(Y.cut(text, sig) for sig in ("static AOM_INLINE int32_t tpl_get_satd_cost(", "static uint32_t motion_estimation(", "static int compare_sad(",
                                                   "static int is_alike_mv(", "static AOM_INLINE void tpl_model_store(", "int av1_tpl_ptr_pos("))
    helpers = helpers.replace("static AOM_INLINE int32_t tpl_get_satd_cost(", "static int32_t tpl_get_satd_cost_ref(")
    helpers = helpers.replace("static uint32_t motion_estimation(", "static uint32_t motion_estimation_ref(")
    tdef = re.search(r"typedef struct \{\n  int_mv mv;\n  int sad;\n\} center_mv_t;\n", text).group(0)
    glue = """
int g_tpl_phase_intra[1];
int g_tpl_intra_calls[1];
int g_tpl_intra_cost[13];
static int32_t tpl_get_satd_cost(BitDepthInfo bd_info, int16_t *src_diff, int diff_stride, const uint8_t *src, int src_stride, const uint8_t *dst, int dst_stride,
                                 tran_low_t *coeff, int bw, int bh, TX_SIZE tx_size) {
  if (g_tpl_phase_intra[0]) return g_tpl_intra_cost[g_tpl_intra_calls[0]++];
  return tpl_get_satd_cost_ref(bd_info, src_diff, diff_stride, src, src_stride, dst, dst_stride, coeff, bw, bh, tx_size);
}
/* motion_estimation with its arguments and results logged */
static uint32_t motion_estimation(AV1_COMP *cpi, MACROBLOCK *x, uint8_t *cur_frame_buf, uint8_t *ref_frame_buf, int stride, int stride_ref, BLOCK_SIZE bsize,
                                  MV center_mv, int_mv *best_mv) {
  uint32_t sme = motion_estimation_ref(cpi, x, cur_frame_buf, ref_frame_buf, stride, stride_ref, bsize, center_mv, best_mv);
  tpl_log_me(center_mv.row, center_mv.col, sme, best_mv->as_mv.row, best_mv->as_mv.col);
  return sme;
}
/* libc's qsort for this element type: a stable sort through the reference's comparator */
static void qsort(center_mv_t *base, int n, int size, int (*cmp)(const void *, const void *)) {
  for (int a = 1; a < n; ++a) {
    center_mv_t t;
    t = base[a];
    int j = a - 1;
    while (j >= 0 && cmp(&base[j], &t) > 0) { base[j + 1] = base[j]; --j; }
    base[j + 1] = t;
  }
}
static void tpl_zero_stats(TplDepStats *s) {
  s->srcrf_sse = 0; s->srcrf_dist = 0; s->recrf_sse = 0; s->recrf_dist = 0; s->intra_sse = 0; s->intra_dist = 0; s->mc_dep_rate = 0; s->mc_dep_dist = 0;
  s->intra_cost = 0; s->inter_cost = 0; s->srcrf_rate = 0; s->recrf_rate = 0; s->intra_rate = 0;
  for (int i = 0; i < 2; ++i) { s->cmp_recrf_dist[i] = 0; s->cmp_recrf_rate[i] = 0; s->ref_frame_index[i] = 0; }
  for (int i = 0; i < INTER_REFS_PER_FRAME; ++i) { s->pred_error[i] = 0; s->mv[i].as_mv.row = 0; s->mv[i].as_mv.col = 0; }
}
"""
    ev.load_text(tdef + rewrite_as_int(helpers) + glue, "tpl_model.c:helpers")
    fn = Y.cut(text, "static AOM_INLINE void mode_estimation(")
    ev.load_text(adapt(fn), "tpl_model.c:mode_estimation")
    for f in list(pyc):
        ev.funcs.pop(f, None)
    bad = [s_ for s_ in ev.skipped if s_[0].startswith("tpl_model.c")]
    assert not bad, bad
    return enc, state


def buf0_stride(state):
    return state["S"]


CONFIGS = [dict(bd=8, prune_starting_mv=0, skip_alike_starting_mv=0, search_method="NSTEP", subpel_search_method="SUBPEL_TREE", reduce_first_step_size=0,
                subpel_force_stop=0, use_fullpel_costlist=0, prune_intra_modes=0, rows=3, clip=None),
           dict(bd=8, prune_starting_mv=2, skip_alike_starting_mv=0, search_method="DIAMOND", subpel_search_method="SUBPEL_TREE_PRUNED", reduce_first_step_size=4,
                subpel_force_stop=0, use_fullpel_costlist=1, prune_intra_modes=1, rows=3, clip=30),
           dict(bd=10, prune_starting_mv=1, skip_alike_starting_mv=1, search_method="NSTEP", subpel_search_method="SUBPEL_TREE_PRUNED_MORE", reduce_first_step_size=6,
                subpel_force_stop=2, use_fullpel_costlist=0, prune_intra_modes=1, rows=3, clip=None),
           dict(bd=10, prune_starting_mv=3, skip_alike_starting_mv=0, search_method="NSTEP", subpel_search_method="SUBPEL_TREE", reduce_first_step_size=2,
                subpel_force_stop=0, use_fullpel_costlist=1, prune_intra_modes=0, rows=3, clip=24)]
REFS = (0, 3)          # LAST_FRAME and GOLDEN_FRAME exist; the other five entries of tpl_data->ref_frame / src_ref_frame are NULL
STATS_SCALARS = ("srcrf_sse", "srcrf_dist", "recrf_sse", "recrf_dist", "intra_sse", "intra_dist", "mc_dep_rate", "mc_dep_dist", "intra_cost", "inter_cost",
                 "srcrf_rate", "recrf_rate", "intra_rate")


def read_stats(ev, st):
    out = {k: int(ev.get(st, k)) for k in STATS_SCALARS}
    out["pred_error"] = [int(ev.get(st, "pred_error[%d]" % r)) for r in range(N_REFS)]
    out["mv"] = [[int(ev.get(st, "mv[%d].as_mv.row" % r)), int(ev.get(st, "mv[%d].as_mv.col" % r))] for r in range(N_REFS)]
    out["ref_frame_index"] = [int(ev.get(st, "ref_frame_index[%d]" % r)) for r in range(2)]
    out["cmp_recrf_dist"] = [int(ev.get(st, "cmp_recrf_dist[%d]" % r)) for r in range(2)]
    out["cmp_recrf_rate"] = [int(ev.get(st, "cmp_recrf_rate[%d]" % r)) for r in range(2)]
    return out


def main():
    ev = setup()
    enc, state = build(ev)
    arrays, frames = {}, []
    mvc = G.synth_mv_costs(31)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    rng = np.random.default_rng(20261007)
    t0 = time.time()
    for ci, cfg in enumerate(CONFIGS):
        bd = cfg["bd"]
        src_b, ref0_b = G.synth_planes(bd, 700 + ci)
        # the second reference: the first one displaced by (2, -3); each reference is the noisy one in half of the columns
        amp = 6 << (bd - 8)
        n0, n1 = rng.integers(-amp, amp + 1, ref0_b.shape), rng.integers(-amp, amp + 1, ref0_b.shape)
        half = ref0_b.shape[1] // 2
        n0[:, :half] = 0; n1[:, half:] = 0
        ref1_b = np.clip(np.roll(ref0_b.astype(np.int32), (2, -3), (0, 1)) + n1, 0, (1 << bd) - 1).astype(ref0_b.dtype)
        ref0_b = np.clip(ref0_b.astype(np.int32) + n0, 0, (1 << bd) - 1).astype(ref0_b.dtype)
        # (borders are replications of the visible edge, as aom_extend_frame_borders leaves them)
        ref0_b, ref1_b = (np.pad(a[BORDER:BORDER + H, BORDER:BORDER + W], BORDER, mode="edge") for a in (ref0_b, ref1_b))
        arrays["src_%d" % ci], arrays["ref0_%d" % ci], arrays["ref1_%d" % ci] = src_b, ref0_b, ref1_b
        hs = G.Harness(ev, bd, src_b, ref0_b, mvc)
        S = hs.S
        ct = "uint8_t" if bd == 8 else "uint16_t"
        ref1p, recp = ev.array(ref1_b.ravel(), ct), ev.array(np.zeros(src_b.size, np.int64), ct)
        state.update(bd=bd, S=S, refs={REFS[0]: ref0_b, REFS[1]: ref1_b}, ref_ptrs={REFS[0]: hs.refp, REFS[1]: ref1p})
        sf = dict(search_method=cfg["search_method"], subpel_search_method=cfg["subpel_search_method"], use_accurate_subpel_search="USE_8_TAPS",
                  sadperbit=int(rng.integers(10, 40)), errorperbit=int(rng.integers(30, 100)), mesh=SC.MESH, use_fullpel_costlist=cfg["use_fullpel_costlist"],
                  exhaustive_searches_thresh=C.INT_MAX)
        cpi, x = enc.make(hs, bd, W, H, sf, 30, mvc, sizes=((16, 16),))
        for k in ("prune_intra_modes", "prune_starting_mv", "skip_alike_starting_mv", "reduce_first_step_size", "subpel_force_stop"):
            ev.set(cpi, "sf.tpl_sf." + k, cfg[k])
        ev.set(cpi, "sf.tpl_sf.search_method", hs.const(cfg["search_method"]))
        ev.set(cpi, "sf.tpl_sf.allow_compound_pred", 0); ev.set(cpi, "sf.tpl_sf.use_y_only_rate_distortion", 1)
        ev.set(cpi, "common.mi_params.mi_rows", H // 4); ev.set(cpi, "common.mi_params.mi_cols", W // 4); ev.set(cpi, "common.mi_params.mi_stride", W // 4)
        seq = ev.new("SequenceHeader")
        ev.set(cpi, "common.seq_params", seq)

        def yv12(plane_ptr):
            y = ev.interp.alloc(enc.yv12_t, True)
            ev.set(y, "y_buffer", plane_ptr.add(BORDER * S + BORDER)); ev.set(y, "y_stride", S); ev.set(y, "y_width", W); ev.set(y, "y_height", H)
            ev.set(y, "y_crop_width", W); ev.set(y, "y_crop_height", H); ev.set(y, "uv_stride", S); ev.set(y, "flags", 8 if bd > 8 else 0)
            return y
        ppi = ev.get(cpi, "ppi")
        tpl = ev.field(ppi, "tpl_data")
        state["yv12"] = {}
        for r, pp in ((REFS[0], hs.refp), (REFS[1], ref1p)):
            state["yv12"]["src%d" % r], state["yv12"]["rec%d" % r] = yv12(pp), yv12(pp)
            ev.set(tpl, "src_ref_frame[%d]" % r, state["yv12"]["src%d" % r]); ev.set(tpl, "ref_frame[%d]" % r, state["yv12"]["rec%d" % r])
        stride = 8
        n_stats = stride * (H // BS)
        stats_arr = ev.interp.alloc(("arr", ev.structs["TplDepStats"], n_stats), True)
        frame = ev.new("TplDepFrame")
        ev.set(frame, "tpl_stats_ptr", R.Ptr(stats_arr.buf, 0, stats_arr.t)); ev.set(frame, "rec_picture", yv12(recp)); ev.set(frame, "stride", stride)
        ev.set(tpl, "tpl_frame", frame); ev.set(tpl, "frame_idx", 0); ev.set(tpl, "tpl_stats_block_mis_log2", 2); ev.set(tpl, "tpl_bsize_1d", BS)
        xd = ev.field(x, "e_mbd")
        ev.set(xd, "cur_buf", yv12(hs.srcp))
        ev.set(xd, "tile.mi_row_start", 0); ev.set(xd, "tile.mi_row_end", H // 4); ev.set(xd, "tile.mi_col_start", 0); ev.set(xd, "tile.mi_col_end", W // 4)
        mis = ev.interp.alloc(("arr", ("ptr", enc.mbmi_t), 4 * (W // 4)), True)
        mis.buf[0] = enc.mi
        ev.set(xd, "mi", R.Ptr(mis.buf, 0, mis.t))
        txfm_stats = ev.array([0], "int")
        blocks = []
        n_modes = 3 if cfg["prune_intra_modes"] else 13
        for mi_row in range(0, cfg["rows"] * 4, 4):
            for mi_col in range(0, W // 4, 4):
                bx, by = mi_col * 4, mi_row * 4
                lim = G.limits(bx, by, BS, BS, cfg["clip"])
                for kk, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
                    ev.set(x, "mv_limits." + kk, v)
                base = int(rng.choice([400, 3000, 40000], p=[0.2, 0.2, 0.6])) << (bd - 8)
                intra = (base + rng.integers(0, 3000, n_modes)).tolist()
                g = ev.globs["g_tpl_intra_cost"]
                for m_, v in enumerate(intra):
                    g.buf[m_] = int(v)
                ev.globs["g_tpl_phase_intra"].buf[0] = 1; ev.globs["g_tpl_intra_calls"].buf[0] = 0
                state["pos"], state["log"] = (bx, by), []
                st = ev.new("TplDepStats")
                t1 = time.time()
                ev.call("tpl_mode_est", cpi, txfm_stats, x, mi_row, mi_col, hs.const("BLOCK_16X16"), 2, st)
                rec = dict(mi_row=mi_row, mi_col=mi_col, limits=list(lim), intra_costs=intra, intra_calls=int(ev.globs["g_tpl_intra_calls"].buf[0]),
                           stats=read_stats(ev, st), log=state["log"], mi_ref_frame=[int(ev.get(enc.mi, "ref_frame[%d]" % r)) for r in range(2)],
                           mi_mv=[int(ev.get(enc.mi, "mv[0].as_mv.row")), int(ev.get(enc.mi, "mv[0].as_mv.col"))])
                ev.call("tpl_model_store", R.Ptr(stats_arr.buf, 0, stats_arr.t), mi_row, mi_col, stride, st, 2)
                rec["stored"] = read_stats(ev, R.Ptr(stats_arr.buf, (mi_row >> 2) * stride + (mi_col >> 2), stats_arr.t))
                blocks.append(rec)
                print(ci, mi_row, mi_col, rec["stats"]["mv"][0], rec["stats"]["mv"][3], rec["stats"]["pred_error"][0], rec["stats"]["pred_error"][3],
                      rec["stats"]["intra_cost"], rec["stats"]["inter_cost"], rec["stats"]["ref_frame_index"], "%.0f s (%.0f)" % (time.time() - t1, time.time() - t0),
                      flush=True)
        frames.append(dict(config=cfg, sadperbit=sf["sadperbit"], errorperbit=sf["errorperbit"], stats_stride=stride, blocks=blocks))
    meta = dict(border=BORDER, width=W, height=H, bs=BS, refs=list(REFS), mesh=SC.MESH, generated_by="tests/golden/gen_ref_eval_tpl_mode.py", frames=frames)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_tpl_mode.npz"), **arrays)
    print("wrote ref_eval_tpl_mode.npz: %d blocks" % sum(len(f["blocks"]) for f in frames))


if __name__ == "__main__":
    main()
