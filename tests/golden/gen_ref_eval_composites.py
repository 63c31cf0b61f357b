#!/usr/bin/env python3
"""Golden vectors for the SEQUENCING of the search composites, obtained by interpreting the reference's caller functions themselves
(build container only; tests/golden/ref_c_eval.py):

  tf_motion_search                                       av1/encoder/temporal_filter.c:87-253
  first_pass_motion_search                               av1/encoder/firstpass.c:261-299
  av1_simple_motion_search + av1_simple_motion_sse_var   av1/encoder/motion_search_facade.c:925-1060

Until round 4 only the CALLEES of these functions (av1_full_pixel_search, the sub-pel trees, av1_set_mv_search_range, ...) were interpreted
and the callers were re-read into the generators and into oracle/pyoracle.py: a mis-reading of a caller would have been shared by the oracle
and the kernels.  Here the callers run as they are written, on VIEWS of the encoder's big objects -- AV1_COMP, AV1_COMMON, MACROBLOCK,
MACROBLOCKD, the speed-feature structs, YV12_BUFFER_CONFIG -- that carry exactly the members those functions (and the mcomp.c helpers they
call: av1_make_default_fullpel_ms_params, av1_make_default_subpel_ms_params, init_ms_buffers, init_mv_cost_params) read; a member that is
read but absent from a view is an interpreter error, so the views are also the list of what the composites depend on.  Helpers that live
in encoder.h are loaded from their own text (cond_cost_list[_const], av1_get_search_site_config, get_search_range's callers);
av1_get_q and the site tables (the reference's builders' output for the plane's stride) are inputs.

Output: tests/golden/ref_eval_composites.npz.
"""
import json
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402
import gen_ref_eval_tf as T  # noqa: E402

REF = G.REF
W, H, BORDER = G.W, G.H, G.BORDER
I32, U8, PTR = R.I32, R.U8, R.PTR
INT_MAX = 2147483647


def view(ev, name, fields, opaque=True):
    key = "<opaque>" + name
    st = ev.structs.get(key) if opaque and key in ev.structs else ev.structs.get(name)
    if st is None:
        st = ev.structs.setdefault(name, R.StructType(name))
    st.fields = fields
    if name not in ev.typedefs:
        ev.typedefs[name] = st
    return st


def grab(ev, path, fn, kind=r"static (?:INLINE|AOM_INLINE) "):
    text = open(REF + path).read()
    m = re.search(kind + r"[^;{]*?\b%s\([^;{]*\)\s*\{.*?\n}\n" % fn, text, re.S)
    assert m, (path, fn)
    ev.load_text(m.group(0), "%s:%s" % (path, fn))


class Encoder:
    """cpi / x objects: views filled with the inputs of the composites."""

    def __init__(self, ev):
        self.ev = ev
        S = ev.structs
        mesh_t = ev.typedefs.get("MESH_PATTERN") or S["MESH_PATTERN"]
        mv_sf_t = view(ev, "MV_SPEED_FEATURES", [
            ("search_method", I32), ("use_bsize_dependent_search_method", I32), ("use_downsampled_sad", I32), ("mesh_patterns", ("arr", mesh_t, 4)),
            ("intrabc_mesh_patterns", ("arr", mesh_t, 4)), ("exhaustive_searches_thresh", I32), ("prune_mesh_search", I32),
            ("obmc_full_pixel_search_level", I32), ("subpel_force_stop", I32), ("subpel_iters_per_step", I32), ("use_accurate_subpel_search", I32),
            ("subpel_search_method", I32), ("use_fullpel_costlist", I32), ("simple_motion_subpel_force_stop", I32)], opaque=False)
        fp_sf_t = view(ev, "FIRST_PASS_SPEED_FEATURES", [("reduce_mv_step_param", I32), ("skip_motion_search_threshold", I32), ("disable_recon", I32),
                                                         ("skip_zeromv_motion_search", I32)], opaque=False)
        part_sf_t = view(ev, "PARTITION_SPEED_FEATURES", [("simple_motion_search_reduce_search_steps", I32)], opaque=False)
        sf_t = view(ev, "SPEED_FEATURES", [("mv_sf", mv_sf_t), ("fp_sf", fp_sf_t), ("part_sf", part_sf_t)], opaque=False)
        feat_t = view(ev, "FeatureFlags", [("cur_frame_force_integer_mv", I32), ("allow_high_precision_mv", I32), ("allow_intrabc", I32)], opaque=False)
        mip_t = view(ev, "CommonModeInfoParams", [("mb_rows", I32), ("mb_cols", I32), ("mi_rows", I32), ("mi_cols", I32)])
        cf_t = view(ev, "CurrentFrame", [("frame_number", R.U32)], opaque=False)
        cm_t = view(ev, "AV1_COMMON", [("width", I32), ("height", I32), ("features", feat_t), ("mi_params", mip_t), ("current_frame", cf_t)])
        vfp_t = ev.typedefs["aom_variance_fn_ptr_t"]
        self.n_bs = ev.globs["BLOCK_SIZES_ALL"].buf[0]
        ppi_t = view(ev, "AV1_PRIMARY", [("fn_ptr", ("arr", vfp_t, self.n_bs))])
        site_t = ev.typedefs["search_site_config"]
        self.n_sm = ev.globs["NUM_DISTINCT_SEARCH_METHODS"].buf[0]
        mvsp_t = view(ev, "MotionVectorSearchParams", [("find_fractional_mv_step", ("ptr", I32)), ("search_site_cfg", ("arr", ("arr", site_t, self.n_sm), 3)),
                                                        ("max_mv_magnitude", I32), ("mv_step_param", I32)], opaque=False)
        dims_t = view(ev, "InitialDimensions", [("width", I32), ("height", I32)], opaque=False)
        self.yv12_t = view(ev, "YV12_BUFFER_CONFIG", [("y_crop_width", I32), ("y_crop_height", I32), ("y_stride", I32), ("y_buffer", ("ptr", U8)), ("flags", R.U32)])
        oxcf_t = view(ev, "AV1EncoderConfig", [("border_in_pixels", I32)])
        self.cpi_t = view(ev, "AV1_COMP", [("common", cm_t), ("ppi", ("ptr", ppi_t)), ("sf", sf_t), ("mv_search_params", mvsp_t),
                                           ("initial_dimensions", dims_t), ("is_screen_content_type", I32), ("oxcf", oxcf_t),
                                           ("unscaled_last_source", ("ptr", self.yv12_t))])
        buf2d = S["buf_2d"]
        mbp_t = view(ev, "macroblock_plane", [("src", buf2d)], opaque=False)
        pdp_t = view(ev, "macroblockd_plane", [("pre", ("arr", buf2d, 2))], opaque=False)
        mbmi = S["<opaque>MB_MODE_INFO"]
        mbmi.fields = [("use_intrabc", U8), ("bsize", I32), ("mode", I32), ("mv", ("arr", ev.structs["<opaque>int_mv"] if "<opaque>int_mv" in ev.structs else ev.structs["int_mv"], 2)),
                       ("tx_size", I32), ("ref_frame", ("arr", R.I8 if hasattr(R, "I8") else I32, 2)), ("motion_mode", I32), ("interp_filters", I32)]
        self.mbmi_t = mbmi
        self.xd_t = view(ev, "MACROBLOCKD", [("plane", ("arr", pdp_t, 3)), ("mi", ("ptr", ("ptr", mbmi))),
                                             ("block_ref_scale_factors", ("arr", ("ptr", S["scale_factors"]), 2)), ("mi_row", I32), ("mi_col", I32), ("bd", I32),
                                             ("cur_buf", ("ptr", self.yv12_t))])
        self.mvc_t = view(ev, "MvCosts", [("nmv_joint_cost", ("arr", I32, 4)), ("mv_cost_stack", ("arr", ("ptr", I32), 2))])
        obmc_t = view(ev, "OBMCBuffer", [("wsrc", ("ptr", I32)), ("mask", ("ptr", I32))])
        csb_t = view(ev, "CONTENT_STATE_SB", [("source_sad_nonrd", I32)], opaque=False)
        self.x_t = view(ev, "MACROBLOCK", [("pred_sse", ("arr", R.U32, 8)), ("plane", ("arr", mbp_t, 3)), ("e_mbd", self.xd_t), ("mv_limits", ev.typedefs["FullMvLimits"]),
                                           ("mv_costs", ("ptr", self.mvc_t)), ("errorperbit", I32), ("sadperbit", I32), ("qindex", I32),
                                           ("content_state_sb", csb_t), ("obmc_buffer", obmc_t), ("search_site_cfg_buf", ("arr", site_t, self.n_sm))])

    def make(self, hs, bd, width, height, sf, q, mvc, sizes=((32, 32), (16, 16)), fpf=False):
        """-> (cpi, x).  hs: the G.Harness of the (source, reference) plane pair (vtables, strides, planes)."""
        ev = self.ev
        cpi = ev.interp.alloc(self.cpi_t, True)
        ev.set(cpi, "common.width", width); ev.set(cpi, "common.height", height)
        ev.set(cpi, "common.features.cur_frame_force_integer_mv", sf.get("force_integer_mv", 0))
        ev.set(cpi, "common.features.allow_high_precision_mv", sf.get("allow_hp", 1))
        ev.set(cpi, "initial_dimensions.width", width); ev.set(cpi, "initial_dimensions.height", height)
        ppi = ev.interp.alloc(ev.structs["<opaque>AV1_PRIMARY"], True)
        # cpi->ppi->fn_ptr[bsize]: the vtable of every block size the composites touch (av1_init_motion_estimation / highbd_set_var_fns fill all)
        for (w, h) in sizes:
            name = G.BSIZE[(w, h)]
            vt = hs.vtable(w, h)
            dst = ev.field(ppi, "fn_ptr[%d]" % hs.const(name))
            dst.store(vt.deref()[0] if hasattr(vt, "deref") else vt.buf[vt.off], vt.t)
        ev.set(cpi, "ppi", ppi)
        mv = "sf.mv_sf."
        ev.set(cpi, mv + "search_method", hs.const(sf.get("search_method", "NSTEP")))
        for k in ("use_bsize_dependent_search_method", "use_downsampled_sad", "prune_mesh_search", "obmc_full_pixel_search_level", "subpel_force_stop",
                  "use_fullpel_costlist", "simple_motion_subpel_force_stop"):
            ev.set(cpi, mv + k, sf.get(k, 0))
        ev.set(cpi, mv + "subpel_iters_per_step", sf.get("subpel_iters_per_step", 2))
        ev.set(cpi, mv + "use_accurate_subpel_search", hs.const(sf.get("use_accurate_subpel_search", "USE_8_TAPS")))
        ev.set(cpi, mv + "subpel_search_method", hs.const(sf.get("subpel_search_method", "SUBPEL_TREE")))
        ev.set(cpi, mv + "exhaustive_searches_thresh", sf.get("exhaustive_searches_thresh", INT_MAX))
        for i, (rng_, itv) in enumerate(sf.get("mesh", T.GOOD_MESH)):
            ev.set(cpi, mv + "mesh_patterns[%d].range" % i, rng_); ev.set(cpi, mv + "mesh_patterns[%d].interval" % i, itv)
            ev.set(cpi, mv + "intrabc_mesh_patterns[%d].range" % i, rng_); ev.set(cpi, mv + "intrabc_mesh_patterns[%d].interval" % i, itv)
        ev.set(cpi, "sf.fp_sf.reduce_mv_step_param", sf.get("reduce_mv_step_param", 3))
        tree = {"SUBPEL_TREE": "av1_find_best_sub_pixel_tree", "SUBPEL_TREE_PRUNED": "av1_find_best_sub_pixel_tree_pruned",
                "SUBPEL_TREE_PRUNED_MORE": "av1_find_best_sub_pixel_tree_pruned_more"}[sf.get("subpel_search_method", "SUBPEL_TREE")]
        ev.set(cpi, "mv_search_params.find_fractional_mv_step", R.FuncRef(tree))   # av1_set_speed_features_framesize_independent's choice
        # search_site_cfg[SS_CFG_SRC][*]: every distinct method's table for this stride (init_motion_estimation, encoder.c)
        lookup = ev.global_values("search_method_lookup")
        table = ev.globs["av1_init_motion_compensation"]
        for i in range(self.n_sm):
            fn = table.buf[i]
            level = int(i in (lookup[hs.const("NSTEP_8PT")], lookup[hs.const("CLAMPED_DIAMOND")]))
            for k in range(2):   # SS_CFG_SRC, SS_CFG_LOOKAHEAD: one stride here
                ev.interp.call(fn.name, [(ev.field(cpi, "mv_search_params.search_site_cfg[%d][%d]" % (k, i)), PTR), (hs.S, I32), (level, I32)])
            if fpf:              # SS_CFG_FPF: av1_init_motion_fpf's table under every method (init_motion_estimation, encoder.c)
                ev.interp.call("av1_init_motion_fpf", [(ev.field(cpi, "mv_search_params.search_site_cfg[2][%d]" % i), PTR), (hs.S, I32)])
        self.q = q
        x = ev.interp.alloc(self.x_t, True)
        xd = ev.field(x, "e_mbd")
        cur = ev.interp.alloc(self.yv12_t, True)
        ev.set(cur, "flags", 8 if bd > 8 else 0)
        ev.set(xd, "cur_buf", cur); ev.set(xd, "bd", bd)
        mi = ev.interp.alloc(self.mbmi_t, True)
        mip = ev.interp.alloc(("ptr", self.mbmi_t), True)
        mip.store(mi, PTR)
        ev.set(xd, "mi", mip)
        self.mi = mi
        sfac = ev.interp.alloc(ev.structs["scale_factors"], True)
        no_scale = ev.interp.ev(R.Parser(ev.pp.expand(R.tokenize("REF_NO_SCALE")), ev.typedefs).expr())[0]
        ev.set(sfac, "x_scale_fp", no_scale); ev.set(sfac, "y_scale_fp", no_scale)
        ev.set(xd, "block_ref_scale_factors[0]", sfac)
        costs = ev.interp.alloc(self.mvc_t, True)
        joint, c0, c1 = mvc
        for i in range(4):
            ev.set(costs, "nmv_joint_cost[%d]" % i, int(joint[i]))
        ev.set(costs, "mv_cost_stack[0]", hs.comp[0].add(hs.mv_max)); ev.set(costs, "mv_cost_stack[1]", hs.comp[1].add(hs.mv_max))
        ev.set(x, "mv_costs", costs)
        ev.set(x, "errorperbit", sf.get("errorperbit", 60)); ev.set(x, "sadperbit", sf.get("sadperbit", 20)); ev.set(x, "qindex", sf.get("qindex", 100))
        return cpi, x

    def thread_data(self, x):
        td_t = view(self.ev, "ThreadData", [("mb", self.x_t)])
        td = self.ev.interp.alloc(td_t, True)
        self.ev.field(td, "mb").store(x.deref()[0], self.x_t)
        return td

    def yv12(self, hs, plane_ptr, width, height):
        ev = self.ev
        b = ev.interp.alloc(self.yv12_t, True)
        ev.set(b, "y_crop_width", width); ev.set(b, "y_crop_height", height); ev.set(b, "y_stride", hs.S)
        ev.set(b, "y_buffer", plane_ptr.add(BORDER * hs.S + BORDER))
        ev.set(b, "flags", 8 if hs.bd > 8 else 0)
        return b


def fp_cases(ev, enc, arrays, cases, mvc, t0):
    """firstpass_inter_prediction (firstpass.c:690-815) block after block along a row, best_ref_mv handed on as the raster loop does
    (:1175-1182: the same variable goes in by value and comes back through best_mv); first_pass_motion_search (:261-299) inside it."""
    # is_zero_mv reads the MV through a uint32_t pointer (av1/common/mv.h:297-299): the same test on the two members (memory-model adaptation)
    ev.funcs.pop("is_zero_mv", None)
    ev.interp.pycalls["is_zero_mv"] = lambda it, a: (int(a[0][0].buf[a[0][0].off].f["row"].deref()[0] == 0 and a[0][0].buf[a[0][0].off].f["col"].deref()[0] == 0), I32)
    def mv_of(p):
        sv = p.buf[p.off]
        return sv.f["row"].deref()[0], sv.f["col"].deref()[0]
    ev.interp.pycalls["is_equal_mv"] = lambda it, a: (int(mv_of(a[0][0]) == mv_of(a[1][0])), I32)   # (mv.h:301-303: the same punning)
    # enumerators of enums the evaluator skipped (UENUM1BYTE forms in av1/common/enums.h): only stored into the mode info here, never read back
    for name, val in (("TX_4X4", 0), ("NEWMV", 16), ("LAST_FRAME", 1), ("NONE_FRAME", -1)):
        if name not in ev.globs:
            ev.define(name, "(%d)" % val)
    ev.funcs.pop("av1_num_planes", None)
    ev.interp.pycalls["av1_num_planes"] = lambda it, a: (1, I32)      # monochrome view: the chroma pointer resets (:789-792) are not part of the search
    for f in ["av1/encoder/firstpass.h", "av1/encoder/firstpass.c"]:
        ev.load(REF + f)
    ev.funcs.pop("is_zero_mv", None); ev.funcs.pop("av1_num_planes", None); ev.funcs.pop("is_equal_mv", None)
    stats_t = ev.typedefs.get("FRAME_STATS") or ev.structs["FRAME_STATS"]
    assert stats_t.fields, "FRAME_STATS did not parse"
    for bd, spec in ((8, dict(thr=0, skip_zeromv=0, golden=1, bs=16)), (10, dict(thr=300, skip_zeromv=0, golden=1, bs=16)),
                     (8, dict(thr=0, skip_zeromv=1, golden=0, bs=16)), (8, dict(thr=0, skip_zeromv=0, golden=1, bs=8))):
        bs = spec["bs"]
        rng = np.random.default_rng(900 + bd + spec["thr"] + spec["skip_zeromv"] + bs)
        frames = T.window(bd, 77 + bd + bs, 3)              # 0: last recon, 1: source, 2: golden
        hi = (1 << bd) - 1
        lsrc = np.clip(frames[0].astype(np.int32) + rng.integers(-6, 7, frames[0].shape), 0, hi).astype(frames[0].dtype)
        lsrc[BORDER:BORDER + bs, :] = frames[1][BORDER:BORDER + bs, :]     # first block row: raw_motion_error 0
        tag = "fp%d_%d_%d_%d" % (bd, spec["thr"], spec["skip_zeromv"], bs)
        arrays[tag + "_src"], arrays[tag + "_last"], arrays[tag + "_golden"], arrays[tag + "_lastsrc"] = frames[1], frames[0], frames[2], lsrc
        hs = G.Harness(ev, bd, frames[1], frames[0], mvc)
        hg = G.Harness(ev, bd, frames[1], frames[2], mvc)
        hl = G.Harness(ev, bd, frames[1], lsrc, mvc)
        sf = dict(search_method="NSTEP", reduce_mv_step_param=3, sadperbit=20, errorperbit=60)
        cpi, x = enc.make(hs, bd, W, H, sf, 30, mvc, sizes=((16, 16), (8, 8)), fpf=True)
        ev.set(cpi, "sf.fp_sf.skip_motion_search_threshold", spec["thr"]); ev.set(cpi, "sf.fp_sf.skip_zeromv_motion_search", spec["skip_zeromv"])
        ev.set(cpi, "sf.fp_sf.disable_recon", 1)
        ev.set(cpi, "oxcf.border_in_pixels", BORDER)
        ev.set(cpi, "common.current_frame.frame_number", 5)
        ev.set(cpi, "common.mi_params.mi_rows", H // 4); ev.set(cpi, "common.mi_params.mi_cols", W // 4)
        ev.set(cpi, "common.mi_params.mb_rows", H // 16); ev.set(cpi, "common.mi_params.mb_cols", W // 16)
        ev.set(cpi, "unscaled_last_source", enc.yv12(hl, hl.refp, W, H))
        last_frame, golden_frame = enc.yv12(hs, hs.refp, W, H), (enc.yv12(hg, hg.refp, W, H) if spec["golden"] else None)
        td = enc.thread_data(x)
        xx = ev.field(td, "mb")
        fp_bsize = hs.const("BLOCK_16X16" if bs == 16 else "BLOCK_8X8")
        ev.set(enc.mi, "bsize", fp_bsize)
        unit_cols = W // bs
        for unit_row in (0, 2, (H // bs) - 1):
            ev.call("av1_set_mv_row_limits", ev.field(cpi, "common.mi_params"), ev.field(xx, "mv_limits"), unit_row * (bs // 4), bs // 4, BORDER)
            best_ref = hs.mv_struct("MV", 0, 0)
            last_nz = hs.mv_struct("MV", 0, 0)
            raw_list = ev.array([0] * unit_cols, "int")
            row = []
            for unit_col in range(unit_cols):
                off = unit_row * bs * hs.S + unit_col * bs
                ev.set(xx, "plane[0].src.buf", hs.srcp.add(BORDER * hs.S + BORDER + off)); ev.set(xx, "plane[0].src.stride", hs.S)
                ev.set(xx, "e_mbd.plane[0].pre[0].stride", hs.S)
                stats = ev.interp.alloc(stats_t, True)
                intra = int(rng.integers(200, 60000))
                err = ev.call("firstpass_inter_prediction", cpi, td, last_frame, golden_frame, unit_row, unit_col, off, 0, off, fp_bsize, intra, unit_col,
                              raw_list, best_ref.buf[0], best_ref, last_nz, stats)
                row.append(dict(intra=intra, inter=err, best_mv=[ev.get(best_ref, "row"), ev.get(best_ref, "col")], raw=raw_list.buf[unit_col],
                                sr_coded_error=float(ev.get(stats, "sr_coded_error")), inter_count=int(ev.get(stats, "inter_count")),
                                second_ref_count=int(ev.get(stats, "second_ref_count"))))
            cases.append(dict(kind="fp", tag=tag, bd=bd, bs=bs, unit_row=unit_row, spec=spec, row=row))
            print("fp", tag, unit_row, [r["best_mv"] for r in row][:6], "%.0f s" % (time.time() - t0), flush=True)


def sms_cases(ev, enc, arrays, cases, mvc, t0):
    """av1_simple_motion_search (motion_search_facade.c:925-1030): its sequencing -- step_param from mv_step_param + the partition speed feature,
    the default full-pel parameters around kZeroMv, cond_cost_list, the sub-pel decision and simple_motion_subpel_force_stop, or
    convert_fullmv_to_mv.  The frame plumbing around it is supplied as inputs: set_offsets_for_motion_search / av1_setup_pre_planes place the
    source and reference block pointers (done here), the reference is unscaled (av1_get_scaled_ref_frame -> NULL), and the predictor build at
    the end (av1_enc_build_inter_predictor; pinned by the convolve fixtures) is not part of the search."""
    state = {}
    ev.interp.pycalls["frame_is_intra_only"] = lambda it, a: (0, I32)      # (inside an assert: an inter frame)
    ev.interp.pycalls["set_offsets_for_motion_search"] = lambda it, a: (None, R.VOID)
    ev.interp.pycalls["get_ref_frame_yv12_buf"] = lambda it, a: (state["yv12"], PTR)
    ev.interp.pycalls["av1_get_scaled_ref_frame"] = lambda it, a: (None, PTR)
    ev.interp.pycalls["get_ref_scale_factors"] = lambda it, a: (None, PTR)
    ev.interp.pycalls["av1_setup_pre_planes"] = lambda it, a: (None, R.VOID)
    ev.interp.pycalls["set_ref_ptrs"] = lambda it, a: (None, R.VOID)
    ev.interp.pycalls["use_fine_search_interval"] = lambda it, a: (0, I32)
    ev.interp.pycalls["av1_broadcast_interp_filter"] = lambda it, a: (0, I32)
    ev.interp.pycalls["av1_enc_build_inter_predictor"] = lambda it, a: (None, R.VOID)
    for name, val in (("SIMPLE_TRANSLATION", 0), ("EIGHTTAP_REGULAR", 0), ("NONE_FRAME", -1), ("AOM_PLANE_Y", 0)):
        if name not in ev.globs:
            ev.define(name, "(%d)" % val)
    grab(ev, "av1/common/mv.h", "convert_fullmv_to_mv")   # (parsed again under the as_fullmv = as_mv adaptation)
    text = open(REF + "av1/encoder/motion_search_facade.c").read()
    m = re.search(r"int_mv av1_simple_motion_search\([^;{]*\)\s*\{.*?\n}\n", text, re.S)
    ev.load_text(m.group(0), "motion_search_facade.c:av1_simple_motion_search")
    for fn in list(ev.interp.pycalls):
        ev.funcs.pop(fn, None)
    rng = np.random.default_rng(4242)
    for bd in (8, 10):
        s_, r_ = G.synth_planes(bd, 500 + bd)
        arrays["sms_src%d" % bd], arrays["sms_ref%d" % bd] = s_, r_
        hs = G.Harness(ev, bd, s_, r_, mvc)
        for k, spec in enumerate((dict(search_method="NSTEP", mv_step_param=2, reduce=1, subpel=1, tree="SUBPEL_TREE", force_stop=0, costlist=0),
                                  dict(search_method="DIAMOND", mv_step_param=4, reduce=0, subpel=1, tree="SUBPEL_TREE_PRUNED", force_stop=1, costlist=1),
                                  dict(search_method="NSTEP", mv_step_param=9, reduce=2, subpel=0, tree="SUBPEL_TREE", force_stop=0, costlist=0),
                                  dict(search_method="BIGDIA", mv_step_param=3, reduce=0, subpel=1, tree="SUBPEL_TREE_PRUNED_MORE", force_stop=2, costlist=1))):
            sf = dict(search_method=spec["search_method"], subpel_search_method=spec["tree"], use_fullpel_costlist=spec["costlist"],
                      simple_motion_subpel_force_stop=spec["force_stop"], use_accurate_subpel_search="USE_2_TAPS", sadperbit=int(rng.integers(10, 40)),
                      errorperbit=int(rng.integers(30, 100)))
            cpi, x = enc.make(hs, bd, W, H, sf, 30, mvc, sizes=((32, 32), (16, 16), (8, 8)))
            ev.set(cpi, "mv_search_params.mv_step_param", spec["mv_step_param"])
            ev.set(cpi, "sf.part_sf.simple_motion_search_reduce_search_steps", spec["reduce"])
            state["yv12"] = enc.yv12(hs, hs.refp, W, H)
            for (w, h) in ((16, 16), (32, 32), (8, 8)):
                bx, by = int(rng.integers(0, (W - w) // 8 + 1)) * 8, int(rng.integers(0, (H - h) // 8 + 1)) * 8
                lim = G.limits(bx, by, w, h, 28)
                for kk, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
                    ev.set(x, "mv_limits." + kk, v)
                off = (BORDER + by) * hs.S + BORDER + bx
                ev.set(x, "plane[0].src.buf", hs.srcp.add(off)); ev.set(x, "plane[0].src.stride", hs.S)
                ev.set(x, "e_mbd.plane[0].pre[0].buf", hs.refp.add(off)); ev.set(x, "e_mbd.plane[0].pre[0].buf0", hs.refp.add(off))
                ev.set(x, "e_mbd.plane[0].pre[0].stride", hs.S); ev.set(x, "e_mbd.plane[0].pre[0].width", W); ev.set(x, "e_mbd.plane[0].pre[0].height", H)
                start = (int(rng.integers(-5, 6)), int(rng.integers(-5, 6)))
                st = hs.mv_struct("FULLPEL_MV", start[0], start[1])
                ev.set(x, "pred_sse[1]", 0)
                best = ev.call("av1_simple_motion_search", cpi, x, by // 4, bx // 4, hs.const(G.BSIZE[(w, h)]), 1, st.buf[0], 1, spec["subpel"])
                mv = [best.f["as_mv"].buf[0].f["row"].deref()[0], best.f["as_mv"].buf[0].f["col"].deref()[0]]
                cases.append(dict(kind="sms", bd=bd, w=w, h=h, bx=bx, by=by, limits=list(lim), start=list(start), spec=spec, sadperbit=sf["sadperbit"],
                                  errorperbit=sf["errorperbit"], mv=mv, pred_sse=int(ev.get(x, "pred_sse[1]"))))
                print("sms", bd, w, h, spec["search_method"], mv, "%.0f s" % (time.time() - t0), flush=True)


def main():
    which = sys.argv[1:] or ["tf", "fp", "sms"]
    ev = G.make_evaluator()
    # int_mv is a union of { uint32_t as_int; MV as_mv; FULLPEL_MV as_fullmv } (av1/common/mv.h:51-55); MV and FULLPEL_MV are both
    # { int16_t row; int16_t col }.  The evaluator has no unions: int_mv becomes a struct holding as_mv, and as_fullmv names the same member
    # (a third value-preserving adaptation of the memory model, next to the two of gen_ref_eval_mcomp.py; as_int is not used by these callers)
    view(ev, "int_mv", [("as_mv", ev.structs["mv"])])
    ev.define("as_fullmv", "as_mv")
    R.ALIASED_STRUCTS.add(frozenset(("mv", "fullpel_mv")))
    for f in ["av1/common/common_data.c", "av1/encoder/temporal_filter.h", "av1/encoder/temporal_filter.c"]:   # (common_data.c: av1_ss_size_lookup)
        ev.load(REF + f)
    grab(ev, "av1/encoder/encoder.h", "cond_cost_list_const")
    grab(ev, "av1/encoder/encoder.h", "cond_cost_list")
    ev.load_text(re.search(r"enum \{\s*SS_CFG_SRC = 0,.*?\} UENUM1BYTE\(SS_CFG_OFFSET\);", open(REF + "av1/encoder/encoder.h").read(), re.S).group(0)
                 .replace("UENUM1BYTE(SS_CFG_OFFSET)", "SS_CFG_OFFSET_e"), "encoder.h:SS_CFG_OFFSET")
    grab(ev, "av1/encoder/motion_search_facade.h", "av1_get_search_site_config")
    enc = Encoder(ev)
    ev.funcs.pop("av1_get_q", None)
    ev.interp.pycalls["av1_get_q"] = lambda it, a: (enc.q, I32)
    arrays, cases = {}, []
    mvc = G.synth_mv_costs(7)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    t0 = time.time()
    if "tf" in which:
        for bd, spec in ((8, dict(q=30, prune_mesh_search=1, subpel_search_method="SUBPEL_TREE")),
                         (10, dict(q=25, prune_mesh_search=1, subpel_search_method="SUBPEL_TREE")),
                         (8, dict(q=40, prune_mesh_search=2, subpel_search_method="SUBPEL_TREE_PRUNED_MORE", use_fullpel_costlist=1, allow_hp=0,
                                  subpel_iters_per_step=1)),
                         (10, dict(q=30, prune_mesh_search=1, force_integer_mv=1))):
            frames = T.window(bd, 40 + bd + spec["q"], 3)
            tag = "tf%d_q%d_%d" % (bd, spec["q"], spec.get("force_integer_mv", 0))
            for f, fr in enumerate(frames):
                arrays["%s_frame%d" % (tag, f)] = fr
            hs = G.Harness(ev, bd, frames[1], frames[0], mvc)     # frame 1 is filtered against frame 0, then (chained ref_mv) frame 2
            hs2 = G.Harness(ev, bd, frames[1], frames[2], mvc)
            cpi, x = enc.make(hs, bd, W, H, spec, spec["q"], mvc)
            to_filter = enc.yv12(hs, hs.srcp, W, H)
            for (mb_row, mb_col) in ((0, 0), (1, 2)) if spec['q'] != 25 else ((2, 0), (1, 1)):
                lim = T.block_limits(mb_row, mb_col)
                for k, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
                    ev.set(x, "mv_limits." + k, v)
                ref_mv = G.Harness.mv_struct(hs, "MV", 0, 0)
                chain = []
                for fi, hsx in enumerate((hs, hs2)):
                    if fi == 1:   # the frame loop of av1_tf_do_filtering_row passes the frame to filter between the two: ref_mv changes sign (:864-867)
                        ev.set(ref_mv, "row", -ev.get(ref_mv, "row")); ev.set(ref_mv, "col", -ev.get(ref_mv, "col"))
                    ref_frame = enc.yv12(hsx, hsx.refp, W, H)
                    mvs = ev.interp.alloc(("arr", ev.structs["mv"], 4), True)
                    mses = ev.array([INT_MAX] * 4, "int")          # the caller's initial values (:859-861): kZeroMv and INT_MAX
                    ev.call("tf_motion_search", cpi, x, to_filter, ref_frame, hs.const("BLOCK_32X32"), mb_row, mb_col, ref_mv, mvs.deref()[0], mses)
                    chain.append(dict(sub_mvs=[[ev.get(mvs, "[%d].row" % k), ev.get(mvs, "[%d].col" % k)] for k in range(4)], sub_mses=list(mses.buf),
                                      ref_mv=[ev.get(ref_mv, "row"), ev.get(ref_mv, "col")]))
                cases.append(dict(kind="tf", tag=tag, bd=bd, mb_row=mb_row, mb_col=mb_col, limits=lim, spec=spec, chain=chain))
                print("tf", tag, mb_row, mb_col, chain[-1]["ref_mv"], "%.0f s" % (time.time() - t0), flush=True)
    if "fp" in which:
        fp_cases(ev, enc, arrays, cases, mvc, t0)
    if "sms" in which:
        sms_cases(ev, enc, arrays, cases, mvc, t0)
    meta = dict(border=BORDER, width=W, height=H, generated_by="tests/golden/gen_ref_eval_composites.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.environ.get("COMPOSITES_OUT") or os.path.join(HERE, "ref_eval_composites.npz"), **arrays)
    print("wrote ref_eval_composites.npz: %d cases" % len(cases))


if __name__ == "__main__":
    main()
