"""ctypes binding of include/aomhip.h.  Fails loudly if libaomhip.so is missing."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AOMHIP_LIB") or os.path.join(_HERE, "lib", "libaomhip.so")  # (override: kernel A/B tools)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libaomhip.so not built (%s): run `make lib` / __graft_entry__.build(). "
        "There is no CPU fallback for the HIP path." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

OK, ERR_INVALID = 0, 2
SAD_SKIP_ROWS = 1


class Planes(C.Structure):
    _fields_ = [("base", C.c_void_p), ("frame_stride", C.c_int64), ("width", C.c_int32), ("height", C.c_int32),
                ("stride", C.c_int32), ("border", C.c_int32), ("bit_depth", C.c_int32), ("n_frames", C.c_int32)]


SEARCH_METHODS = ["DIAMOND", "NSTEP", "NSTEP_8PT", "CLAMPED_DIAMOND", "HEX", "BIGDIA", "SQUARE", "FAST_HEX", "FAST_DIAMOND",
                  "FAST_BIGDIA", "VFAST_DIAMOND", "NSTEP_FPF"]   # AOMHIP_SEARCH_* values


class SearchParams(C.Structure):
    """aomhip_search_params."""
    _fields_ = [(n, C.c_int32) for n in ("search_method", "step_param", "mv_cost_type", "sad_per_bit", "error_per_bit",
                                         "use_downsampled_sad", "run_mesh_search", "prune_mesh_search",
                                         "mesh_search_mv_diff_threshold", "force_mesh_thresh", "fine_search_interval")] + \
               [("mesh_patterns", C.c_int32 * 8)]

    @classmethod
    def make(cls, method, step_param, cost_type, sad_per_bit=0, error_per_bit=0, skip_sad=0, run_mesh=0, prune_mesh=0,
             mesh_diff_thr=0, force_mesh_thresh=2147483647, fine_interval=0, mesh=None):
        q = cls(method if isinstance(method, int) else SEARCH_METHODS.index(method), step_param, cost_type, sad_per_bit,
                error_per_bit, int(skip_sad), run_mesh, prune_mesh, mesh_diff_thr, force_mesh_thresh, fine_interval)
        for i, v in enumerate(np.asarray(mesh if mesh is not None else [[0, 0]] * 4).reshape(-1)):
            q.mesh_patterns[i] = int(v)
        return q


SUBPEL_TREES = {"pruned_more": 0, "pruned": 1, "tree": 2}


class SubpelParams(C.Structure):
    """aomhip_subpel_params."""
    _fields_ = [(n, C.c_int32) for n in ("tree", "mv_cost_type", "error_per_bit", "iters_per_step", "allow_hp", "forced_stop",
                                         "subpel_search_type")]


class TfParams(C.Structure):
    """aomhip_tf_params."""
    _fields_ = [("full", SearchParams), ("sub", SubpelParams), ("use_cost_list", C.c_int32), ("force_integer_mv", C.c_int32),
                ("mse_thresh", C.c_int32)]

    @classmethod
    def default(cls, width, height, bit_depth, q, prune_mesh_level, mesh, subpel_tree=2, iters_per_step=2, allow_hp=1, use_cost_list=0,
                use_downsampled_sad=0, force_integer_mv=0):
        """aomhip_tf_default_params (host/aomhip_tf.c)."""
        out = cls()
        pat = (C.c_int * 8)(*[int(v) for v in np.asarray(mesh).reshape(-1)])
        lib.aomhip_tf_default_params(width, height, bit_depth, q, prune_mesh_level, pat, SUBPEL_TREES.get(subpel_tree, subpel_tree), iters_per_step,
                                     allow_hp, use_cost_list, use_downsampled_sad, force_integer_mv, C.byref(out))
        return out


class FirstPassParams(C.Structure):
    _fields_ = [("unit_rows", C.c_int32), ("unit_cols", C.c_int32), ("skip_motion_search_threshold", C.c_int32), ("skip_zeromv_motion_search", C.c_int32)]


class TfApplyParams(C.Structure):
    _fields_ = [("noise_levels", C.c_double * 3), ("q_factor", C.c_int32), ("filter_strength", C.c_int32), ("num_planes", C.c_int32),
                ("ss_x", C.c_int32), ("ss_y", C.c_int32)]

    @classmethod
    def make(cls, noise_levels, q_factor, filter_strength, num_planes=1, ss_x=0, ss_y=0):
        o = cls()
        for i, v in enumerate(list(noise_levels)[:3]):
            o.noise_levels[i] = float(v)
        o.q_factor, o.filter_strength, o.num_planes, o.ss_x, o.ss_y = q_factor, filter_strength, num_planes, ss_x, ss_y
        return o


def tf_block_list(width, height, border):
    n = lib.aomhip_tf_block_list(width, height, border, None)
    b = np.zeros(n, search_block_dtype)
    assert lib.aomhip_tf_block_list(width, height, border, b.ctypes.data) == n
    return b


def search_sites(method):
    """aomhip_search_sites -> (num_search_steps, searches_per_step[22], radius[22], mv[22, 17, 2])."""
    ns = C.c_int()
    per, rad, mv = np.zeros(22, np.int32), np.zeros(22, np.int32), np.zeros((22, 17, 2), np.int16)
    check(lib.aomhip_search_sites(method if isinstance(method, int) else SEARCH_METHODS.index(method), C.byref(ns), per.ctypes.data,
                                  rad.ctypes.data, mv.ctypes.data), "aomhip_search_sites")
    return ns.value, per, rad, mv


class QuantParams(C.Structure):
    _fields_ = [(n, C.c_int16 * 2) for n in ("zbin", "round", "quant", "quant_shift", "dequant")]

    @classmethod
    def from_tables(cls, t):
        """t: mapping name -> 2 int16 (DC, AC), e.g. rows of av1_build_quantizer output."""
        q = cls()
        for n in ("zbin", "round", "quant", "quant_shift", "dequant"):
            getattr(q, n)[0], getattr(q, n)[1] = int(t[n][0]), int(t[n][1])
        return q


class SingleRdParams(C.Structure):
    """aomhip_single_rd_params."""
    _fields_ = [("pred", C.POINTER(Planes)), ("filter_x", C.c_int32), ("filter_y", C.c_int32), ("qparams", C.POINTER(QuantParams)), ("d_costs", C.c_void_p),
                ("tx_type_rate", C.c_int32), ("rdmult", C.c_int32), ("lossless", C.c_int32), ("d_yrd_blocks", C.c_void_p), ("d_stats_first", C.c_void_p),
                ("d_stats_second", C.c_void_p), ("d_candidate_mvs", C.c_void_p)]


txb_dtype = np.dtype([("x", "<i4"), ("y", "<i4"), ("out_offset", "<u4"), ("tx_type", "u1"), ("reserved", "u1", (3,))])
sad_cand_dtype = np.dtype([("sx", "<i2"), ("sy", "<i2"), ("rx", "<i2"), ("ry", "<i2")])
search_block_dtype = np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min",
                                                    "row_max", "col_min", "col_max")])
MV_COST_ENTROPY, MV_COST_L1_LOWRES, MV_COST_L1_MIDRES, MV_COST_L1_HDRES, MV_COST_NONE = 0, 1, 2, 3, 4
COMP_AVG, COMP_DIST_WTD, COMP_MASK, COMP_OBMC = 0, 1, 2, 3
blend_item_dtype = np.dtype([("x", "<i2"), ("y", "<i2"), ("w", "<i2"), ("h", "<i2"), ("mask_offset", "<u2"), ("vertical", "u1"), ("reserved", "u1")])
rect_dtype = np.dtype([("h_start", "<i4"), ("h_end", "<i4"), ("v_start", "<i4"), ("v_end", "<i4")])
scaled_block_dtype = np.dtype([(n, "<i4") for n in ("src_x", "src_y", "subpel_x_qn", "subpel_y_qn", "dst_x", "dst_y")])   # aomhip_scaled_block
txfm_yrd_block_dtype = np.dtype([("bx", "<i2"), ("by", "<i2"), ("tx_size_rate", "<i4"), ("no_skip_txfm_rate", "<i4"), ("skip_txfm_rate", "<i4"),
                                 ("above_ctx", "u1", (32,)), ("left_ctx", "u1", (32,))])   # aomhip_txfm_yrd_block
txfm_yrd_stats_dtype = np.dtype([("rd", "<i8"), ("dist", "<i8"), ("sse", "<i8"), ("rate", "<i4"), ("skip_txfm", "<i4")])   # aomhip_txfm_yrd_stats
warp_model_dtype = np.dtype([("mat", "<i4", (6,)), ("alpha", "<i2"), ("beta", "<i2"), ("gamma", "<i2"), ("delta", "<i2")])
warp_refine_block_dtype = np.dtype([(n_, "<i2") for n_ in ("bx", "by", "mv_row", "mv_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")] +
                                   [("total_samples", "<i4"), ("num_proj_ref", "<i4"), ("pts", "<i4", (16,)), ("pts_inref", "<i4", (16,)),
                                    ("model", warp_model_dtype)])   # aomhip_warp_refine_block (188 bytes)
warp_refine_result_dtype = np.dtype([("mv_row", "<i2"), ("mv_col", "<i2"), ("num_proj_ref", "<i4"), ("bestmse", "<u4"), ("model", warp_model_dtype)])
warp_block_dtype = np.dtype([("mat", "<i4", (6,)), ("alpha", "<i2"), ("beta", "<i2"), ("gamma", "<i2"), ("delta", "<i2"), ("p_col", "<i4"), ("p_row", "<i4"),
                             ("p_width", "<i4"), ("p_height", "<i4")])   # aomhip_warp_block (48 bytes)


class CompoundParams(C.Structure):
    """aomhip_compound_params (include/aomhip.h)."""
    _fields_ = [("kind", C.c_int32), ("subpel", C.c_int32), ("fwd_offset", C.c_int32), ("bck_offset", C.c_int32),
                ("mask_stride", C.c_int32), ("invert_mask", C.c_int32)]


var_cand_dtype = np.dtype([("sx", "<i2"), ("sy", "<i2"), ("rx", "<i2"), ("ry", "<i2"), ("xoff", "u1"), ("yoff", "u1"),
                           ("reserved", "u1", (2,))])
sad_x4d_dtype = np.dtype([("sx", "<i2"), ("sy", "<i2"), ("rx", "<i2", (4,)), ("ry", "<i2", (4,))])

_vp, _i, _i64, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_size_t
_PP = C.POINTER(Planes)

_protos = {
    "aomhip_abi_version": (C.c_int, []),
    "aomhip_device_count": (C.c_int, []),
    "aomhip_last_error": (C.c_char_p, []),
    "aomhip_ctx_create": (C.c_int, [_i, _vp, C.POINTER(_vp)]),
    "aomhip_ctx_destroy": (None, [_vp]),
    "aomhip_ctx_sync": (C.c_int, [_vp]),
    "aomhip_ctx_stream": (_vp, [_vp]),
    "aomhip_timer_begin": (C.c_int, [_vp]),
    "aomhip_timer_end": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "aomhip_malloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "aomhip_free": (C.c_int, [_vp, _vp]),
    "aomhip_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, _sz]),
    "aomhip_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, _sz]),
    "aomhip_memset": (C.c_int, [_vp, _vp, _i, _sz]),
    "aomhip_calc_stride": (C.c_int, [_i, _i]),
    "aomhip_planes_alloc": (C.c_int, [_vp, _i, _i, _i, _i, _i, _PP]),
    "aomhip_planes_free": (C.c_int, [_vp, _PP]),
    "aomhip_planes_upload": (C.c_int, [_vp, _PP, _i, _vp, _i]),
    "aomhip_planes_extend_borders": (C.c_int, [_vp, _PP, _i, _i]),
    "aomhip_planes_download": (C.c_int, [_vp, _PP, _i, _vp]),
    "aomhip_sad_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _vp, _i, _i64, _vp]),
    "aomhip_sad_x4d_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _vp, _i, _i64, _vp]),
    "aomhip_sad_avg_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _vp, _i, _i64, _vp, _vp, _i, _i, _vp]),
    "aomhip_compound_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_compound": (C.c_uint, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, C.POINTER(C.c_uint)]),
    "aomhip_sad_sb_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i64, _vp, _vp, _vp,
                                      _i, _i64, _vp]),
    "aomhip_variance_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _vp, _i, _i64, _vp, _vp]),
    "aomhip_estimate_txfm_yrd_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "aomhip_variance_sb_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "aomhip_sub_pixel_variance_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _vp, _i, _i64, _vp, _vp]),
    "aomhip_variance": (C.c_uint, [_vp, _i, _vp, _i, _i, _i, C.POINTER(C.c_uint)]),
    "aomhip_mse": (C.c_uint, [_vp, _i, _vp, _i, _i, _i, C.POINTER(C.c_uint)]),
    "aomhip_get_var": (None, [_vp, _i, _vp, _i, _i, _i, C.POINTER(C.c_uint), C.POINTER(C.c_int)]),
    "aomhip_get_mb_ss": (C.c_uint, [_vp]),
    "aomhip_mse_wxh_16bit": (C.c_uint64, [_vp, _i, _vp, _i, _i, _i]),
    "aomhip_mse_16xh_16bit": (C.c_uint64, [_vp, _i, _vp, _i, _i]),
    "aomhip_mse_wxh_16bit_highbd": (C.c_uint64, [_vp, _i, _vp, _i, _i, _i]),
    "aomhip_comp_mask_pred": (None, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _i]),
    "aomhip_highbd_comp_mask_pred": (None, [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _i]),
    "aomhip_get_var_sse_sum_8x8_quad": (None, [_vp, _i, _vp, _i, _vp, _vp, C.POINTER(C.c_uint), C.POINTER(C.c_int), _vp]),
    "aomhip_get_var_sse_sum_16x16_dual": (None, [_vp, _i, _vp, _i, _vp, C.POINTER(C.c_uint), C.POINTER(C.c_int), _vp]),
    "aomhip_sub_pixel_variance": (C.c_uint, [_vp, _i, _i, _i, _vp, _i, _i, _i, C.POINTER(C.c_uint)]),
    "aomhip_variance16x16": (C.c_uint, [_vp, _i, _vp, _i, C.POINTER(C.c_uint)]),
    "aomhip_highbd_variance": (C.c_uint, [_vp, _i, _vp, _i, _i, _i, _i, C.POINTER(C.c_uint)]),
    "aomhip_highbd_sub_pixel_variance": (C.c_uint, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, C.POINTER(C.c_uint)]),
    "aomhip_tx_size_wide": (C.c_int, [_i]),
    "aomhip_tx_size_high": (C.c_int, [_i]),
    "aomhip_tx_max_eob": (C.c_int, [_i]),
    "aomhip_xform_quant_batch": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _i, _i, C.POINTER(QuantParams), _i, _vp, _vp,
                                           _vp, _vp]),
    "aomhip_subtract_xform_quant_ex_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_xform_quant_ex_batch": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_subtract_xform_quant_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _vp, _i, _i, _i, C.POINTER(QuantParams),
                                                    _vp, _vp, _vp, _vp]),
    "aomhip_quantize_b_adaptive_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, C.POINTER(QuantParams), _i, _vp, _vp, _vp]),
    "aomhip_inv_txfm_add_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _PP, _i]),
    "aomhip_quantize_b_qm_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_quantize_fp_qm_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_quantize_b_adaptive_qm_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_int_pro_motion_estimation_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _i, _vp, _vp]),
    "aomhip_vbp_8x8_stats_plane": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _i, _vp, _i]),
    "aomhip_vbp_4x4_avg_plane": (C.c_int, [_vp, _PP, _i, _i, _i, _i, _vp, _i]),
    "aomhip_get_shear_params": (C.c_int, [_vp]),
    "aomhip_select_samples": (C.c_int, [_i, _i, _vp, _vp, _i, _i, _i]),
    "aomhip_refine_warped_mv_batch": (C.c_int, [_vp, _PP, _PP, _i, _PP, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "aomhip_find_projection": (C.c_int, [_i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i]),
    "aomhip_warp_error_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "aomhip_segmented_frame_error": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _i, _vp]),
    "aomhip_quantize_lp_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_xform_quant_qm_batch": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_subtract_xform_quant_qm_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_encode_inter_blocks_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _PP, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "aomhip_deblock_plane": (C.c_int, [_vp, _PP, _i, _vp, _i, _i, _i]),
    "aomhip_cdef_luma_plane": (C.c_int, [_vp, _PP, _i, _PP, _i, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    "aomhip_cdef_chroma_plane": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i]),
    "aomhip_refining_search_8p_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "aomhip_compound_full_pixel_search_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "aomhip_obmc_full_pixel_search_batch": (C.c_int, [_vp, _PP, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "aomhip_build_inter_pred_contiguous_batch": (C.c_int, [_vp, _PP, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i]),
    "aomhip_joint_motion_search_batch": (C.c_int, [_vp, _PP, _PP, _PP, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "aomhip_compound_single_motion_search_batch": (C.c_int, [_vp, _PP, _PP, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "aomhip_joint_motion_search_extensive_batch": (C.c_int, [_vp, _PP, _PP, _PP, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "aomhip_compound_subpel_tree_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "aomhip_obmc_subpel_tree_batch": (C.c_int, [_vp, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_strip_read_probe": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int64)]),
    "aomhip_valu_issue_probe": (C.c_int, [_vp, _i, _i, _i, _vp]),
    "aomhip_valu_issue_probe_name": (C.c_char_p, [_i]),
    "aomhip_fullpel_diamond_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    "aomhip_subpel_bilinear_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp]),
    "aomhip_mesh_search_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _i, C.POINTER(C.c_int), _i, _vp, _i, _vp, _vp]),
    "aomhip_full_pixel_search_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "aomhip_search_sites": (C.c_int, [_i, C.POINTER(C.c_int), _vp, _vp, _vp]),
    "aomhip_subpel_tree_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "aomhip_subpel_tree_list_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_single_motion_search_rd_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_single_motion_search_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_bind_variance_vtable": (C.c_int, [_vp, _i]),
    "aomhip_build_inter_pred_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _i, _i, _i]),
    "aomhip_sse_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _i, _vp]),
    "aomhip_hadamard_batch": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "aomhip_sum_sse_2d_i16_batch": (C.c_int, [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "aomhip_txb_init_levels_batch": (C.c_int, [_vp, _vp, _i, _i, _vp, _i, _vp, _i64]),
    "aomhip_cost_coeffs_txb_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "aomhip_txb_entropy_context_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "aomhip_cost_coeffs_txb_laplacian_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "aomhip_get_nz_map_contexts_batch": (C.c_int, [_vp, _vp, _i64, _i, _vp, _i, _i, _vp, _vp, _i64]),
    "aomhip_scaled_pred_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i]),
    "aomhip_scaled_pred_compound_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i]),
    "aomhip_warp_affine_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _i]),
    "aomhip_warp_affine_compound_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i]),
    "aomhip_selfguided_restoration_batch": (C.c_int, [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i64]),
    "aomhip_apply_selfguided_restoration_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _i, _i64]),
    "aomhip_wiener_convolve_add_src_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _i]),
    "aomhip_calc_proj_params_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i64, _vp, _vp, _vp]),
    "aomhip_pixel_proj_error_batch": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _i, _i64, _vp, _vp, _i, _vp]),
    "aomhip_wedge_sse_from_residuals_batch": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "aomhip_wedge_sign_from_residuals_batch": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "aomhip_wedge_compute_delta_squares_batch": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp]),
    "aomhip_cdef_search_sse_luma": (C.c_int, [_vp, _PP, _i, _PP, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "aomhip_cdef_search_sse_chroma": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "aomhip_lpf_search_sse": (C.c_int, [_vp, _PP, _i, _PP, _i, _PP, _i, _vp, _i64, _i, _i, _i, _i, _vp]),
    "aomhip_compute_stats_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "aomhip_plane_sse": (C.c_int, [_vp, _PP, _i, _PP, _i, _vp]),
    "aomhip_build_inter_pred_ex_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i]),
    "aomhip_build_compound_pred_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i]),
    "aomhip_highbd_sad_skip": (C.c_uint, [_vp, _i, _vp, _i, _i, _i, _i]),
    "aomhip_highbd_sad_x4d": (None, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i]),
    "aomhip_build_masked_compound_pred_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i]),
    "aomhip_build_diffwtd_compound_pred_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "aomhip_blend_a64_1d_batch": (C.c_int, [_vp, _PP, _i, _PP, _i, _vp, _i, _vp]),
    "aomhip_build_pred_fullpel": (C.c_int, [_vp, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _i]),
    "aomhip_sad": (C.c_uint, [_vp, _i, _vp, _i, _i, _i]),
    "aomhip_sad_skip": (C.c_uint, [_vp, _i, _vp, _i, _i, _i]),
    "aomhip_sad_x4d": (None, [_vp, _i, C.POINTER(_vp), _i, _vp, _i, _i]),
    "aomhip_sad_skip_x4d": (None, [_vp, _i, C.POINTER(_vp), _i, _vp, _i, _i]),
    "aomhip_sad16x16": (C.c_uint, [_vp, _i, _vp, _i]),
    "aomhip_sad16x16x4d": (None, [_vp, _i, C.POINTER(_vp), _i, _vp]),
    "aomhip_highbd_sad": (C.c_uint, [_vp, _i, _vp, _i, _i, _i, _i]),
    "aomhip_first_pass_motion_search_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "aomhip_graph_capture_begin": (C.c_int, [_vp]),
    "aomhip_graph_capture_end": (C.c_int, [_vp, C.POINTER(_vp)]),
    "aomhip_graph_launch": (C.c_int, [_vp, _vp]),
    "aomhip_graph_destroy": (C.c_int, [_vp]),
    "aomhip_first_pass_inter_frame": (C.c_int, [_vp, _PP, _i, _PP, _i, _PP, _i, _PP, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_tpl_inter_estimation_batch": (C.c_int, [_vp, _PP, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "aomhip_motion_estimation_batch": (C.c_int, [_vp, _PP, _PP, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "aomhip_tf_default_params": (None, [_i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "aomhip_tf_block_list": (C.c_int, [_i, _i, _i, _vp]),
    "aomhip_tf_motion_search_frames": (C.c_int, [_vp, _PP, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "aomhip_tile_column_bounds": (C.c_int, [_i, _i, _i, _vp]),
    "aomhip_tile_column_bounds_balanced": (C.c_int, [_i, _i, _i, _i, _vp]),
    "aomhip_tile_column_bounds_widths": (C.c_int, [_i, _i, _vp, _i, _i, _i, _vp]),
    "aomhip_recon_exchange_plan": (C.c_int, [_i, _i, _vp, _i, _i, _vp, _vp]),
    "aomhip_comm_unique_id": (C.c_int, [_vp]),
    "aomhip_deblock_plane_fused": (C.c_int, [_vp, _vp, _i, _vp, _i, _vp, _i, _i]),
    "aomhip_simple_motion_search_batch": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "aomhip_tf_apply_frames": (C.c_int, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "aomhip_comm_init": (C.c_int, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "aomhip_comm_destroy": (None, [_vp]),
    "aomhip_comm_info": (C.c_int, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "aomhip_allgather_recon": (C.c_int, [_vp, _vp, _PP, _i, _vp, _i]),
    "aomhip_exchange_loopback": (C.c_int, [_vp, _vp, _PP, _i, _i, _i, _i]),
}
for _name, (_res, _args) in _protos.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

# The rtcd-signature conformance surface (host pointers; the tests call these with explicit ctypes arguments).
RTCD_TX_SIZES = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64), (64, 32),
                 (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]  # TX_SIZE order (av1/common/enums.h)
RTCD_STAMPED = (["aomhip_%squantize_b%s%s" % (h, sz, a) for h in ("", "highbd_") for sz in ("", "_32x32", "_64x64") for a in ("", "_adaptive")] +
                ["aomhip_%s_%dx%d" % (k, w, hh) for k in ("fwd_txfm2d", "inv_txfm2d_add") for w, hh in RTCD_TX_SIZES] +
                ["aomhip_%slpf_%s_%d%s" % (h, d, n, k) for h in ("", "highbd_") for d in ("horizontal", "vertical") for n in (4, 6, 8, 14)
                 for k in (("", "_dual", "_quad") if not h else ("", "_dual"))] +
                ["aomhip_cdef_filter_%d_%d" % (b, v) for b in (8, 16) for v in range(4)])
for _name in RTCD_STAMPED + ["aomhip_fwd_txfm2d", "aomhip_inv_txfm2d_add", "aomhip_subtract_block", "aomhip_highbd_subtract_block",
                             "aomhip_cdef_find_dir_dual", "aomhip_status_clear"]:
    getattr(lib, _name).restype = None  # raises AttributeError if the library lacks the symbol
    _protos[_name] = (None, None)
for _name, _res in (("aomhip_lf_build_edge_params", C.c_int), ("aomhip_lf_level_table", None), ("aomhip_cdef_build_skip8x8", C.c_int),
                    ("aomhip_cdef_build_strengths", C.c_int), ("aomhip_cdef_find_dir", C.c_int), ("aomhip_rtcd", C.c_int), ("aomhip_status", C.c_int), ("aomhip_failure_count", C.c_long)):
    getattr(lib, _name).restype = _res
    _protos[_name] = (_res, None)

EXPORTED = sorted(_protos)


def get_shear_params(models):
    """av1_get_shear_params on a warp_model_dtype array in place (host, no GPU); returns the per-model verdicts"""
    return [int(lib.aomhip_get_shear_params(C.c_void_p(models[i:i + 1].ctypes.data))) for i in range(len(models))]


def select_samples(mv, pts, pts_inref, n, bw, bh):
    """av1_selectSamples in place on two int32 arrays of (x, y) pairs (host, no GPU); mv = (row, col); returns the number kept"""
    assert pts.dtype == np.int32 and pts_inref.dtype == np.int32 and pts.flags.c_contiguous and pts_inref.flags.c_contiguous
    return int(lib.aomhip_select_samples(int(mv[0]), int(mv[1]), C.c_void_p(pts.ctypes.data), C.c_void_p(pts_inref.ctypes.data), n, bw, bh))


def find_projection(n, pts, pts_inref, bw, bh, mv, model, mi_row, mi_col):
    """av1_find_projection into model (a 1-element warp_model_dtype array; host, no GPU); True = a usable model"""
    assert pts.dtype == np.int32 and pts_inref.dtype == np.int32
    return bool(lib.aomhip_find_projection(n, C.c_void_p(pts.ctypes.data), C.c_void_p(pts_inref.ctypes.data), bw, bh, int(mv[0]), int(mv[1]),
                                           C.c_void_p(model.ctypes.data), mi_row, mi_col))


class AomHipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != OK:
        raise AomHipError("%s failed (status %d): %s" % (what, rc, lib.aomhip_last_error().decode()))


class Context:
    """aomhip_ctx wrapper: one HIP stream on one device."""

    def __init__(self, device=0, stream=None):
        h = _vp()
        check(lib.aomhip_ctx_create(device, stream, C.byref(h)), "aomhip_ctx_create")
        self.h = h
        self.device = device
        self._allocs = []

    def close(self):
        if self.h:
            lib.aomhip_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(lib.aomhip_ctx_sync(self.h), "sync")

    # ---- raw device memory
    def malloc(self, nbytes):
        p = _vp()
        check(lib.aomhip_malloc(self.h, nbytes, C.byref(p)), "malloc")
        return p.value

    def free(self, ptr):
        check(lib.aomhip_free(self.h, ptr), "free")

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(max(arr.nbytes, 16))
        check(lib.aomhip_memcpy_h2d(self.h, p, arr.ctypes.data, arr.nbytes), "h2d")
        return p

    def capture(self, fn):
        """Record the batched calls fn() makes on this context into a graph (fn must have run once before); -> handle for graph_launch."""
        check(lib.aomhip_graph_capture_begin(self.h), "aomhip_graph_capture_begin")
        try:
            fn()
        finally:
            g = _vp()
            rc = lib.aomhip_graph_capture_end(self.h, C.byref(g))
        check(rc, "aomhip_graph_capture_end")
        return g.value

    def graph_launch(self, g):
        check(lib.aomhip_graph_launch(self.h, g), "aomhip_graph_launch")

    def graph_destroy(self, g):
        check(lib.aomhip_graph_destroy(g), "aomhip_graph_destroy")

    def memcpy_h2d(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        check(lib.aomhip_memcpy_h2d(self.h, ptr, arr.ctypes.data, arr.nbytes), "h2d")

    def from_device(self, ptr, shape, dtype):
        out = np.empty(shape, dtype)
        check(lib.aomhip_memcpy_d2h(self.h, out.ctypes.data, ptr, out.nbytes), "d2h")
        return out

    def memset(self, ptr, value, nbytes):
        check(lib.aomhip_memset(self.h, ptr, value, nbytes), "memset")

    def timer_begin(self):
        check(lib.aomhip_timer_begin(self.h), "timer_begin")

    def timer_end(self):
        ms = C.c_float()
        check(lib.aomhip_timer_end(self.h, C.byref(ms)), "timer_end")
        return ms.value

    # ---- planes
    def planes_alloc(self, width, height, border, bit_depth, n_frames):
        p = Planes()
        check(lib.aomhip_planes_alloc(self.h, width, height, border, bit_depth, n_frames, C.byref(p)), "planes_alloc")
        return p

    def planes_free(self, p):
        check(lib.aomhip_planes_free(self.h, C.byref(p)), "planes_free")

    def planes_upload(self, p, frame, pixels):
        dt = np.uint8 if p.bit_depth == 8 else np.uint16
        pixels = np.ascontiguousarray(pixels, dtype=dt)
        assert pixels.shape == (p.height, p.width)
        check(lib.aomhip_planes_upload(self.h, C.byref(p), frame, pixels.ctypes.data, pixels.shape[1]), "upload")

    def planes_download(self, p, frame):
        dt = np.uint8 if p.bit_depth == 8 else np.uint16
        out = np.empty((p.height + 2 * p.border, p.stride), dt)
        check(lib.aomhip_planes_download(self.h, C.byref(p), frame, out.ctypes.data), "download")
        return out

    def first_pass_motion_search_batch(self, src, ref, frame, bw, bh, params, d_blocks, n, d_mv, d_err, d_mvjcost=None, d_mvcost_row=None,
                                       d_mvcost_col=None):
        check(lib.aomhip_first_pass_motion_search_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(params), d_mvjcost, d_mvcost_row,
                                                        d_mvcost_col, d_blocks, n, d_mv, d_err), "aomhip_first_pass_motion_search_batch")

    def first_pass_inter_frame(self, src, src_frame, last, last_frame, golden, golden_frame, last_source, last_source_frame, bw, bh, params, fp,
                               d_blocks, d_intra_error, d_best_mv, d_motion_error, d_full_mv=None, d_gf_motion_error=None, d_raw_motion_error=None,
                               d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        check(lib.aomhip_first_pass_inter_frame(self.h, C.byref(src), src_frame, C.byref(last), last_frame, C.byref(golden) if golden is not None else None,
                                                golden_frame, C.byref(last_source), last_source_frame, bw, bh, C.byref(params), d_mvjcost, d_mvcost_row,
                                                d_mvcost_col, C.byref(fp), d_blocks, d_intra_error, d_best_mv, d_full_mv, d_motion_error, d_gf_motion_error,
                                                d_raw_motion_error), "aomhip_first_pass_inter_frame")

    def motion_estimation_batch(self, src, ref, frame, bw, bh, full, sub, use_cost_list, d_blocks, n, d_mv, d_err, d_dist, d_sse, d_full_mv=None,
                                d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        check(lib.aomhip_motion_estimation_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(full), C.byref(sub), use_cost_list, d_mvjcost,
                                                 d_mvcost_row, d_mvcost_col, d_blocks, n, d_mv, d_err, d_dist, d_sse, d_full_mv),
              "aomhip_motion_estimation_batch")

    def tpl_inter_estimation_batch(self, src, refs, frame, bw, full, sub, use_cost_list, prune_starting_mv, d_blocks, d_center_mvs, d_center_counts, n,
                                   d_best_mv, d_pred_error, d_best_rf, d_best_cost, d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        """The inter leg of tpl_model.c's mode_estimation for independent blocks (aomhip.h); refs: list of plane rings."""
        arr = (C.POINTER(Planes) * len(refs))(*[C.pointer(r) for r in refs])
        check(lib.aomhip_tpl_inter_estimation_batch(self.h, C.byref(src), arr, len(refs), frame, bw, C.byref(full), C.byref(sub), use_cost_list,
                                                    prune_starting_mv, d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, d_center_mvs, d_center_counts, n,
                                                    d_best_mv, d_pred_error, d_best_rf, d_best_cost), "aomhip_tpl_inter_estimation_batch")

    def tf_motion_search_frames(self, frames, filter_frame, params, d_blocks, n_blocks, d_mvs, d_mses, d_ref_mv=None, frame_present=None):
        fp = None if frame_present is None else np.ascontiguousarray(frame_present, np.uint8)
        check(lib.aomhip_tf_motion_search_frames(self.h, C.byref(frames), filter_frame, None if fp is None else fp.ctypes.data, C.byref(params),
                                                 d_blocks, n_blocks, d_mvs, d_mses, d_ref_mv), "aomhip_tf_motion_search_frames")

    def deblock_plane_fused(self, src, src_frame, dst, dst_frame, d_params, units_stride, sharpness=0):
        check(lib.aomhip_deblock_plane_fused(self.h, C.byref(src), src_frame, C.byref(dst), dst_frame, d_params, units_stride, sharpness),
              "aomhip_deblock_plane_fused")

    def simple_motion_search_batch(self, src, ref, frame, bw, bh, full, sub, use_cost_list, d_blocks, n, pred, pred_frame, d_mv, d_sse=None, d_var=None,
                                   d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        check(lib.aomhip_simple_motion_search_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(full), None if sub is None else C.byref(sub),
                                                    use_cost_list, d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, n,
                                                    None if pred is None else C.byref(pred), pred_frame, d_mv, d_sse, d_var),
              "aomhip_simple_motion_search_batch")

    def tf_apply_frames(self, frames, filter_frame, params, n_blocks, d_mvs, d_mses, outs, out_frame, frame_present=None, d_diff=None):
        """frames / outs: (y,) or (y, u, v) plane rings; params: TfApplyParams."""
        fp = None if frame_present is None else np.ascontiguousarray(frame_present, np.uint8)
        fr = [C.byref(x) for x in frames] + [None] * (3 - len(frames))
        ou = [C.byref(x) for x in outs] + [None] * (3 - len(outs))
        check(lib.aomhip_tf_apply_frames(self.h, fr[0], fr[1], fr[2], filter_frame, None if fp is None else fp.ctypes.data, C.byref(params), n_blocks,
                                         d_mvs, d_mses, ou[0], ou[1], ou[2], out_frame, d_diff), "aomhip_tf_apply_frames")

    # ---- multi-GPU: the per-frame exchange of the reconstruction (RCCL inside the library)
    def comm_init(self, unique_id, rank, n_ranks):
        """unique_id: the 128 bytes rank 0 got from comm_unique_id(), handed to every rank by the launcher."""
        uid = np.frombuffer(bytes(unique_id), np.uint8).copy()
        assert uid.size == 128
        h = _vp()
        check(lib.aomhip_comm_init(self.h, uid.ctypes.data, rank, n_ranks, C.byref(h)), "aomhip_comm_init")
        return h

    def comm_destroy(self, comm):
        lib.aomhip_comm_destroy(comm)

    def comm_info(self, comm):
        """(rank, n_ranks) as RCCL reports them for this communicator."""
        r, n = _i(), _i()
        check(lib.aomhip_comm_info(comm, C.byref(r), C.byref(n)), "aomhip_comm_info")
        return r.value, n.value

    def allgather_recon(self, comm, p, frame, col_bounds, halo=-1):
        b = np.ascontiguousarray(col_bounds, np.int32).reshape(-1, 2)
        check(lib.aomhip_allgather_recon(self.h, comm, C.byref(p), frame, b.ctypes.data, halo), "aomhip_allgather_recon")

    # ---- SAD
    def sad_batch(self, src, ref, first_frame, n_frames, bw, bh, flags, d_cands, n_cands, cand_frame_stride, d_out):
        check(lib.aomhip_sad_batch(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, bw, bh, flags, d_cands,
                                   n_cands, cand_frame_stride, d_out), "aomhip_sad_batch")

    def sad_x4d_batch(self, src, ref, first_frame, n_frames, bw, bh, flags, d_groups, n_groups, group_frame_stride,
                      d_out):
        check(lib.aomhip_sad_x4d_batch(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, bw, bh, flags,
                                       d_groups, n_groups, group_frame_stride, d_out), "aomhip_sad_x4d_batch")

    # ---- forward transform + quantise
    def xform_quant_batch(self, d_residual, stride, tx_size, d_blocks, n_blocks, grid_cols, tx_type, qp, is_hbd,
                          d_coeff, d_qcoeff, d_dqcoeff, d_eob):
        check(lib.aomhip_xform_quant_batch(self.h, d_residual, stride, tx_size, d_blocks, n_blocks, grid_cols, tx_type,
                                           C.byref(qp), int(is_hbd), d_coeff, d_qcoeff, d_dqcoeff, d_eob),
              "aomhip_xform_quant_batch")

    def subtract_xform_quant_batch(self, src, pred, frame, tx_size, d_blocks, n_blocks, grid_cols, tx_type, qp,
                                   d_coeff, d_qcoeff, d_dqcoeff, d_eob):
        check(lib.aomhip_subtract_xform_quant_batch(self.h, C.byref(src), C.byref(pred), frame, tx_size, d_blocks,
                                                    n_blocks, grid_cols, tx_type, C.byref(qp), d_coeff, d_qcoeff,
                                                    d_dqcoeff, d_eob), "aomhip_subtract_xform_quant_batch")

    def sad_avg_batch(self, src, ref, first_frame, n_frames, bw, bh, d_cands, n_cands, cand_frame_stride, d_second_pred,
                      d_pred_index, fwd_offset, bck_offset, d_out):
        check(lib.aomhip_sad_avg_batch(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, bw, bh, d_cands, n_cands,
                                       cand_frame_stride, d_second_pred, d_pred_index, fwd_offset, bck_offset, d_out),
              "aomhip_sad_avg_batch")

    def compound_batch(self, src, ref, first_frame, n_frames, bw, bh, d_cands, n_cands, cand_frame_stride, params, d_second_pred=None,
                       d_mask=None, d_obmc_wsrc=None, d_obmc_mask=None, d_pred_index=None, d_mask_offset=None, d_var=None, d_sse=None,
                       d_sad=None):
        check(lib.aomhip_compound_batch(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, bw, bh, d_cands, n_cands,
                                        cand_frame_stride, C.byref(params), d_second_pred, d_mask, d_obmc_wsrc, d_obmc_mask, d_pred_index,
                                        d_mask_offset, d_var, d_sse, d_sad), "aomhip_compound_batch")

    def sad_sb_batch(self, src, ref, first_frame, n_frames, bw, bh, flags, sb_w, sb_h, rng, n_buckets, d_groups=None,
                     d_group_off=None, n_groups=0, group_frame_stride=0, d_out_groups=None, d_cands=None, d_cand_off=None,
                     n_cands=0, cand_frame_stride=0, d_out_cands=None):
        check(lib.aomhip_sad_sb_batch(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, bw, bh, flags, sb_w, sb_h,
                                      rng, n_buckets, d_groups, d_group_off, n_groups, group_frame_stride, d_out_groups,
                                      d_cands, d_cand_off, n_cands, cand_frame_stride, d_out_cands), "aomhip_sad_sb_batch")

    def xform_quant_ex_batch(self, d_res, stride, tx_size, d_blocks, n, grid_cols, tx_type, qp, is_hbd, bit_depth, quant_kind, d_coeff,
                             d_q, d_dq, d_eob, d_err=None):
        check(lib.aomhip_xform_quant_ex_batch(self.h, d_res, stride, tx_size, d_blocks, n, grid_cols, tx_type, C.byref(qp), int(is_hbd),
                                              bit_depth, quant_kind, d_coeff, d_q, d_dq, d_eob, d_err), "aomhip_xform_quant_ex_batch")

    def subtract_xform_quant_ex_batch(self, src, pred, frame, tx_size, d_blocks, n, grid_cols, tx_type, qp, quant_kind, d_coeff, d_q, d_dq,
                                      d_eob, d_err=None):
        check(lib.aomhip_subtract_xform_quant_ex_batch(self.h, C.byref(src), C.byref(pred), frame, tx_size, d_blocks, n, grid_cols, tx_type,
                                                       C.byref(qp), quant_kind, d_coeff, d_q, d_dq, d_eob, d_err),
              "aomhip_subtract_xform_quant_ex_batch")

    def quantize_b_adaptive_batch(self, d_coeff, tx_size, d_blocks, n_blocks, tx_type, qp, is_hbd, d_qcoeff, d_dqcoeff, d_eob):
        check(lib.aomhip_quantize_b_adaptive_batch(self.h, d_coeff, tx_size, d_blocks, n_blocks, tx_type, C.byref(qp),
                                                   int(is_hbd), d_qcoeff, d_dqcoeff, d_eob), "aomhip_quantize_b_adaptive_batch")

    # ---- variance
    def estimate_txfm_yrd_batch(self, src, pred, frame, bw, bh, qparams, d_costs, tx_type_rate, rdmult, lossless, d_blocks, n_blocks, d_stats):
        check(lib.aomhip_estimate_txfm_yrd_batch(self.h, C.byref(src), C.byref(pred), frame, bw, bh, C.byref(qparams), d_costs, tx_type_rate, rdmult, lossless,
                                                 d_blocks, n_blocks, d_stats), "aomhip_estimate_txfm_yrd_batch")

    def variance_sb_batch(self, src, ref, first_frame, n_frames, bw, bh, sb_w, sb_h, rng, n_buckets, d_groups=None, d_group_off=None, n_groups=0,
                          group_frame_stride=0, d_var_groups=None, d_sse_groups=None, d_cands=None, d_cand_off=None, n_cands=0, cand_frame_stride=0,
                          d_var_cands=None, d_sse_cands=None):
        check(lib.aomhip_variance_sb_batch(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, bw, bh, sb_w, sb_h, rng, n_buckets, d_groups,
                                           d_group_off, n_groups, group_frame_stride, d_var_groups, d_sse_groups, d_cands, d_cand_off, n_cands,
                                           cand_frame_stride, d_var_cands, d_sse_cands), "aomhip_variance_sb_batch")

    def variance_batch(self, src, ref, first_frame, n_frames, bw, bh, d_cands, n, stride, d_var, d_sse, subpel=False):
        f = lib.aomhip_sub_pixel_variance_batch if subpel else lib.aomhip_variance_batch
        check(f(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, bw, bh, d_cands, n, stride, d_var, d_sse),
              "aomhip_variance_batch")

    def inv_txfm_add_batch(self, d_dqcoeff, tx_size, d_blocks, n_blocks, grid_cols, tx_type, d_eob, dst, frame):
        check(lib.aomhip_inv_txfm_add_batch(self.h, d_dqcoeff, tx_size, d_blocks, n_blocks, grid_cols, tx_type, d_eob,
                                            C.byref(dst), frame), "aomhip_inv_txfm_add_batch")

    def quantize_b_qm_batch(self, d_coeff, tx_size, d_blocks, n_blocks, tx_type, qparams, is_hbd, d_qm, d_iqm, d_qcoeff, d_dqcoeff, d_eob):
        check(lib.aomhip_quantize_b_qm_batch(self.h, d_coeff, tx_size, d_blocks, n_blocks, tx_type, C.byref(qparams), int(is_hbd), d_qm, d_iqm,
                                             d_qcoeff, d_dqcoeff, d_eob), "aomhip_quantize_b_qm_batch")

    def quantize_b_adaptive_qm_batch(self, d_coeff, tx_size, d_blocks, n_blocks, tx_type, qparams, is_hbd, d_qm, d_iqm, d_qcoeff, d_dqcoeff, d_eob):
        check(lib.aomhip_quantize_b_adaptive_qm_batch(self.h, d_coeff, tx_size, d_blocks, n_blocks, tx_type, C.byref(qparams), int(is_hbd), d_qm, d_iqm,
                                                      d_qcoeff, d_dqcoeff, d_eob), "aomhip_quantize_b_adaptive_qm_batch")

    def quantize_fp_qm_batch(self, d_coeff, tx_size, d_blocks, n_blocks, tx_type, qparams, is_hbd, d_qm, d_iqm, d_qcoeff, d_dqcoeff, d_eob):
        """the `fp` quantiser with matrices: qparams carries round_fp / quant_fp in its round / quant fields"""
        check(lib.aomhip_quantize_fp_qm_batch(self.h, d_coeff, tx_size, d_blocks, n_blocks, tx_type, C.byref(qparams), int(is_hbd), d_qm, d_iqm,
                                              d_qcoeff, d_dqcoeff, d_eob), "aomhip_quantize_fp_qm_batch")

    def int_pro_motion_estimation_batch(self, src, src_frame, ref, ref_frame, bw, bh, d_blocks, n_blocks, d_best_mv, d_best_sad):
        """av1_int_pro_motion_estimation for a batch of blocks: (row, col) in 1/8 pel and the SAD"""
        check(lib.aomhip_int_pro_motion_estimation_batch(self.h, C.byref(src), src_frame, C.byref(ref), ref_frame, bw, bh, d_blocks, n_blocks, d_best_mv,
                                                         d_best_sad), "aomhip_int_pro_motion_estimation_batch")

    def vbp_8x8_stats_plane(self, src, src_frame, ref, ref_frame, vis_w, vis_h, d_sum8, sum_stride, d_minmax16=None, minmax_stride=0):
        check(lib.aomhip_vbp_8x8_stats_plane(self.h, C.byref(src), src_frame, C.byref(ref), ref_frame, vis_w, vis_h, d_sum8, sum_stride, d_minmax16,
                                             minmax_stride), "aomhip_vbp_8x8_stats_plane")

    def vbp_4x4_avg_plane(self, src, src_frame, vis_w, vis_h, border_offset_4x4, d_sum4, sum_stride):
        check(lib.aomhip_vbp_4x4_avg_plane(self.h, C.byref(src), src_frame, vis_w, vis_h, border_offset_4x4, d_sum4, sum_stride), "aomhip_vbp_4x4_avg_plane")

    def warp_error_batch(self, ref, ref_frame, cur, cur_frame, ssx, ssy, d_models, n_models, p_col, p_row, p_width, p_height, d_seg, seg_stride, d_error):
        """av1_warp_error for n_models candidate models (shear values already in them: get_shear_params)"""
        check(lib.aomhip_warp_error_batch(self.h, C.byref(ref), ref_frame, C.byref(cur), cur_frame, ssx, ssy, d_models, n_models, p_col, p_row, p_width,
                                          p_height, d_seg, seg_stride, d_error), "aomhip_warp_error_batch")

    def segmented_frame_error(self, ref, ref_frame, cur, cur_frame, p_width, p_height, d_seg, seg_stride, d_error):
        check(lib.aomhip_segmented_frame_error(self.h, C.byref(ref), ref_frame, C.byref(cur), cur_frame, p_width, p_height, d_seg, seg_stride, d_error),
              "aomhip_segmented_frame_error")

    def quantize_lp_batch(self, d_coeff, tx_size, d_blocks, n_blocks, tx_type, qparams, d_qcoeff, d_dqcoeff, d_eob, d_err=None):
        """av1_quantize_lp on int16 coefficients (+ av1_block_error_lp into d_err): qparams carries round_fp / quant_fp in its round / quant fields"""
        check(lib.aomhip_quantize_lp_batch(self.h, d_coeff, tx_size, d_blocks, n_blocks, tx_type, C.byref(qparams), d_qcoeff, d_dqcoeff, d_eob, d_err),
              "aomhip_quantize_lp_batch")

    def xform_quant_qm_batch(self, d_residual, stride, tx_size, d_blocks, n_blocks, grid_cols, tx_type, qparams, is_hbd, d_qm, d_iqm, d_coeff, d_qcoeff,
                             d_dqcoeff, d_eob):
        check(lib.aomhip_xform_quant_qm_batch(self.h, d_residual, stride, tx_size, d_blocks, n_blocks, grid_cols, tx_type, C.byref(qparams), int(is_hbd),
                                              d_qm, d_iqm, d_coeff, d_qcoeff, d_dqcoeff, d_eob), "aomhip_xform_quant_qm_batch")

    def subtract_xform_quant_qm_batch(self, src, pred, frame, tx_size, d_blocks, n_blocks, grid_cols, tx_type, qparams, d_qm, d_iqm, d_coeff, d_qcoeff,
                                      d_dqcoeff, d_eob):
        check(lib.aomhip_subtract_xform_quant_qm_batch(self.h, C.byref(src), C.byref(pred), frame, tx_size, d_blocks, n_blocks, grid_cols, tx_type,
                                                       C.byref(qparams), d_qm, d_iqm, d_coeff, d_qcoeff, d_dqcoeff, d_eob),
              "aomhip_subtract_xform_quant_qm_batch")

    def encode_inter_blocks_batch(self, src, src_frame, ref, ref_frame, recon, recon_frame, bw, d_blocks, d_mv, n_blocks, qparams, d_qcoeff, d_dqcoeff,
                                  d_eob, filter_x=0, filter_y=0, tx_type=0):
        """prediction -> residual -> fwd_txfm2d + quantize_b -> inverse + add for square inter blocks, one kernel (csrc/encode_block.hip)."""
        check(lib.aomhip_encode_inter_blocks_batch(self.h, C.byref(src), src_frame, C.byref(ref), ref_frame, C.byref(recon), recon_frame, bw, d_blocks,
                                                   d_mv, n_blocks, filter_x, filter_y, tx_type, C.byref(qparams), d_qcoeff, d_dqcoeff, d_eob),
              "aomhip_encode_inter_blocks_batch")

    def mesh_search_batch(self, src, ref, frame, bw, bh, cost_type, patterns, fine, d_blocks, n_blocks, d_mv, d_cost):
        pat = (C.c_int * 8)(*[int(v) for pair in patterns for v in pair])
        check(lib.aomhip_mesh_search_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, cost_type, pat, fine, d_blocks,
                                           n_blocks, d_mv, d_cost), "aomhip_mesh_search_batch")

    def cdef_chroma_plane(self, src, src_frame, dst, dst_frame, xdec, ydec, d_luma_dir, d_pri, d_sec, fb_stride, d_skip,
                          damping):
        check(lib.aomhip_cdef_chroma_plane(self.h, C.byref(src), src_frame, C.byref(dst), dst_frame, xdec, ydec, d_luma_dir,
                                           d_pri, d_sec, fb_stride, d_skip, damping), "aomhip_cdef_chroma_plane")

    def deblock_plane(self, p, frame, d_params, units_stride, sharpness=0, passes=3):
        check(lib.aomhip_deblock_plane(self.h, C.byref(p), frame, d_params, units_stride, sharpness, passes),
              "aomhip_deblock_plane")

    def cdef_luma_plane(self, src, src_frame, dst, dst_frame, d_pri, d_sec, fb_stride, d_skip, damping, d_dir=None,
                        d_var=None):
        check(lib.aomhip_cdef_luma_plane(self.h, C.byref(src), src_frame, C.byref(dst), dst_frame, d_pri, d_sec,
                                         fb_stride, d_skip, damping, d_dir, d_var), "aomhip_cdef_luma_plane")

    # ---- motion search
    def refining_search_8p_batch(self, src, ref, frame, bw, bh, cost_type, sad_per_bit, error_per_bit, d_blocks, n, d_second_pred, d_mask, invert_mask,
                                 d_mv, d_sad, d_var, d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        check(lib.aomhip_refining_search_8p_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, cost_type, sad_per_bit, error_per_bit, d_mvjcost,
                                                  d_mvcost_row, d_mvcost_col, d_blocks, n, d_second_pred, d_mask, int(invert_mask), d_mv, d_sad, d_var),
              "aomhip_refining_search_8p_batch")

    def compound_full_pixel_search_batch(self, src, ref, frame, bw, bh, params, d_blocks, n, d_second_pred, d_mask, invert_mask, d_mv, d_cost, d_second,
                                         d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        """av1_full_pixel_search with a second predictor [and mask] (the extensive joint-search step); params: SearchParams."""
        check(lib.aomhip_compound_full_pixel_search_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(params), d_mvjcost, d_mvcost_row,
                                                          d_mvcost_col, d_blocks, n, d_second_pred, d_mask, int(invert_mask), d_mv, d_cost, d_second),
              "aomhip_compound_full_pixel_search_batch")

    def obmc_full_pixel_search_batch(self, ref, frame, bw, bh, method, step_param, fast, cost_type, sad_per_bit, error_per_bit, d_blocks, n, d_wsrc, d_mask,
                                     d_mv, d_cost, d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        check(lib.aomhip_obmc_full_pixel_search_batch(self.h, C.byref(ref), frame, bw, bh, method if isinstance(method, int) else SEARCH_METHODS.index(method),
                                                      step_param, int(fast), cost_type, sad_per_bit, error_per_bit, d_mvjcost, d_mvcost_row, d_mvcost_col,
                                                      d_blocks, n, d_wsrc, d_mask, d_mv, d_cost), "aomhip_obmc_full_pixel_search_batch")

    def build_inter_pred_contiguous_batch(self, ref, frame, d_pred, bw, bh, d_blocks, d_mv, n, fx=0, fy=0):
        check(lib.aomhip_build_inter_pred_contiguous_batch(self.h, C.byref(ref), frame, d_pred, bw, bh, d_blocks, d_mv, n, fx, fy),
              "aomhip_build_inter_pred_contiguous_batch")

    def compound_single_motion_search_batch(self, src, ref, ref_other, frame, bw, bh, full, sub, force_integer_mv, d_blocks, d_ref_mv, d_this_mv, d_other_mv,
                                            fx, fy, d_second_pred, d_mask, ref_idx, n, d_rate_mv, d_bestsme, d_mvjcost, d_mvcost_row, d_mvcost_col):
        """av1_compound_single_motion_search[_interinter] per block; ref_other None with d_second_pred given, or the other way round."""
        check(lib.aomhip_compound_single_motion_search_batch(self.h, C.byref(src), C.byref(ref), None if ref_other is None else C.addressof(ref_other), frame, bw,
                                                             bh, C.byref(full), C.byref(sub), int(force_integer_mv), d_mvjcost, d_mvcost_row, d_mvcost_col,
                                                             d_blocks, d_ref_mv, d_this_mv, d_other_mv, fx, fy, d_second_pred, d_mask, ref_idx, n, d_rate_mv,
                                                             d_bestsme), "aomhip_compound_single_motion_search_batch")

    def joint_motion_search_extensive_batch(self, src, ref0, ref1, frame, bw, bh, full, sub, allow_second_mv, force_integer_mv, d_blocks, d_ref_mv, d_cur_mv,
                                            d_mask, n, d_rate_mv, d_best_err, d_mvjcost, d_mvcost_row, d_mvcost_col):
        """av1_joint_motion_search per block (extensive branch: av1_full_pixel_search on the compound); full: SearchParams, sub: SubpelParams."""
        check(lib.aomhip_joint_motion_search_extensive_batch(self.h, C.byref(src), C.byref(ref0), C.byref(ref1), frame, bw, bh, C.byref(full), C.byref(sub),
                                                             int(allow_second_mv), int(force_integer_mv), d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks,
                                                             d_ref_mv, d_cur_mv, d_mask, n, d_rate_mv, d_best_err),
              "aomhip_joint_motion_search_extensive_batch")

    def joint_motion_search_batch(self, src, ref0, ref1, frame, bw, bh, cost_type, sad_per_bit, sub, force_integer_mv, d_blocks, d_ref_mv, d_cur_mv, d_mask, n,
                                  d_rate_mv, d_best_err, d_mvjcost, d_mvcost_row, d_mvcost_col):
        """av1_joint_motion_search per block (refining-search branch); sub: SubpelParams."""
        check(lib.aomhip_joint_motion_search_batch(self.h, C.byref(src), C.byref(ref0), C.byref(ref1), frame, bw, bh, cost_type, sad_per_bit, C.byref(sub),
                                                   int(force_integer_mv), d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, d_ref_mv, d_cur_mv, d_mask, n,
                                                   d_rate_mv, d_best_err), "aomhip_joint_motion_search_batch")

    def compound_subpel_tree_batch(self, src, ref, frame, bw, bh, params, d_blocks, n, d_second_pred, d_mask, invert_mask, d_mv, d_err, d_dist, d_sse,
                                   d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        """the sub-pel trees on a compound prediction (second_pred [/ mask]); params: SubpelParams."""
        check(lib.aomhip_compound_subpel_tree_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(params), d_mvjcost, d_mvcost_row, d_mvcost_col,
                                                    d_blocks, n, d_second_pred, d_mask, int(invert_mask), d_mv, d_err, d_dist, d_sse),
              "aomhip_compound_subpel_tree_batch")

    def obmc_subpel_tree_batch(self, ref, frame, bw, bh, params, d_blocks, n, d_wsrc, d_mask, d_mv, d_err, d_dist=None, d_sse=None, d_mvjcost=None,
                               d_mvcost_row=None, d_mvcost_col=None):
        """av1_find_best_obmc_sub_pixel_tree_up per block; params: SubpelParams (subpel_search_type 0 or 3)."""
        check(lib.aomhip_obmc_subpel_tree_batch(self.h, C.byref(ref), frame, bw, bh, C.byref(params), d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, n,
                                                d_wsrc, d_mask, d_mv, d_err, d_dist, d_sse), "aomhip_obmc_subpel_tree_batch")

    def strip_read_probe(self, src, ref, first_frame, n_frames, x0, x1, sb_w, sb_h, rng):
        """-> bytes requested (measurement support: the transport of sad_sb_batch alone)."""
        b = C.c_int64(0)
        check(lib.aomhip_strip_read_probe(self.h, C.byref(src), C.byref(ref), first_frame, n_frames, x0, x1, sb_w, sb_h, rng, C.byref(b)), "aomhip_strip_read_probe")
        return b.value

    def valu_issue_probe(self, op_class, waves_per_simd=8, iters=3000):
        """-> dict (measurement support: the issue rate of one VALU opcode class on this box, aomhip.h)."""
        r = ValuProbeResult()
        check(lib.aomhip_valu_issue_probe(self.h, op_class, waves_per_simd, iters, C.byref(r)), "aomhip_valu_issue_probe")
        return {"op": valu_issue_probe_names()[op_class], "wave_insts_per_s_per_simd": r.wave_insts_per_s_per_simd, "launch_ms": r.launch_ms,
                "memtime_ticks_per_wave_inst": r.memtime_ticks_per_wave_inst, "memtime_hz": r.memtime_hz, "waves_per_simd": r.waves_per_simd,
                "compute_units": r.compute_units}

    def fullpel_diamond_batch(self, src, ref, frame, bw, bh, clamped, step_param, cost_type, d_blocks, n, d_mv, d_cost):
        check(lib.aomhip_fullpel_diamond_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, clamped, step_param,
                                               cost_type, d_blocks, n, d_mv, d_cost), "aomhip_fullpel_diamond_batch")

    def subpel_bilinear_batch(self, src, ref, frame, bw, bh, cost_type, iters, allow_hp, forced_stop, d_blocks, n, d_mv,
                              d_err, d_dist, d_sse):
        check(lib.aomhip_subpel_bilinear_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, cost_type, iters,
                                               allow_hp, forced_stop, d_blocks, n, d_mv, d_err, d_dist, d_sse),
              "aomhip_subpel_bilinear_batch")

    def full_pixel_search_batch(self, src, ref, frame, bw, bh, params, d_blocks, n, d_mv, d_cost, d_cost_list=None, d_second=None,
                                d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None):
        """av1_full_pixel_search per block; params: SearchParams; d_mvcost_row/col point at the CENTRE of their tables."""
        check(lib.aomhip_full_pixel_search_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(params), d_mvjcost,
                                                 d_mvcost_row, d_mvcost_col, d_blocks, n, d_mv, d_cost, d_cost_list, d_second),
              "aomhip_full_pixel_search_batch")

    def subpel_tree_batch(self, src, ref, frame, bw, bh, params, d_blocks, n, d_mv, d_err, d_dist, d_sse, d_cost_list=None,
                          d_mvjcost=None, d_mvcost_row=None, d_mvcost_col=None, d_mv_lists=None):
        if d_mv_lists is not None:
            check(lib.aomhip_subpel_tree_list_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(params), d_mvjcost, d_mvcost_row, d_mvcost_col,
                                                    d_blocks, d_cost_list, n, d_mv, d_err, d_dist, d_sse, d_mv_lists), "aomhip_subpel_tree_list_batch")
            return
        check(lib.aomhip_subpel_tree_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(params), d_mvjcost,
                                           d_mvcost_row, d_mvcost_col, d_blocks, d_cost_list, n, d_mv, d_err, d_dist, d_sse),
              "aomhip_subpel_tree_batch")

    def single_motion_search_batch(self, src, ref, frame, bw, bh, full, sub, d_blocks, n, d_best_mv, d_bestsme, d_rate_mv, d_mvjcost, d_mvcost_row,
                                   d_mvcost_col, d_start2=None, use_cost_list=0, try_second_mv=0, force_integer_mv=0, d_pred_sse=None, d_full_mv=None,
                                   d_second_best=None):
        check(lib.aomhip_single_motion_search_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(full), None if sub is None else C.byref(sub),
                                                    use_cost_list, try_second_mv, force_integer_mv, d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, d_start2,
                                                    n, d_best_mv, d_bestsme, d_rate_mv, d_pred_sse, d_full_mv, d_second_best),
              "aomhip_single_motion_search_batch")

    def single_motion_search_rd_batch(self, src, ref, frame, bw, bh, full, sub, d_blocks, n, rd, d_best_mv, d_bestsme, d_rate_mv, d_mvjcost, d_mvcost_row,
                                      d_mvcost_col, d_start2=None, use_cost_list=0, force_integer_mv=0, d_pred_sse=None, d_full_mv=None, d_second_best=None):
        """rd: SingleRdParams (keep the objects it points at alive for the call)"""
        check(lib.aomhip_single_motion_search_rd_batch(self.h, C.byref(src), C.byref(ref), frame, bw, bh, C.byref(full), None if sub is None else C.byref(sub),
                                                       use_cost_list, force_integer_mv, d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, d_start2, n,
                                                       None if rd is None else C.byref(rd), d_best_mv, d_bestsme, d_rate_mv, d_pred_sse, d_full_mv, d_second_best),
              "aomhip_single_motion_search_rd_batch")

    def refine_warped_mv_batch(self, src, ref, frame, pred, bw, bh, allow_hp, mv_cost_type, error_per_bit, d_blocks, n, d_results, d_mvjcost=None,
                               d_mvcost_row=None, d_mvcost_col=None):
        check(lib.aomhip_refine_warped_mv_batch(self.h, C.byref(src), C.byref(ref), frame, C.byref(pred), bw, bh, allow_hp, mv_cost_type, error_per_bit,
                                                d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, n, d_results), "aomhip_refine_warped_mv_batch")

    def build_inter_pred_batch(self, ref, ref_frame, pred, pred_frame, bw, bh, d_blocks, d_mv, n_blocks, filter_x=0, filter_y=0, ss_x=0, ss_y=0):
        if ss_x or ss_y:
            check(lib.aomhip_build_inter_pred_ex_batch(self.h, C.byref(ref), ref_frame, C.byref(pred), pred_frame, bw, bh, d_blocks, d_mv,
                                                       n_blocks, filter_x, filter_y, ss_x, ss_y), "aomhip_build_inter_pred_ex_batch")
            return
        check(lib.aomhip_build_inter_pred_batch(self.h, C.byref(ref), ref_frame, C.byref(pred), pred_frame, bw, bh, d_blocks, d_mv, n_blocks,
                                                filter_x, filter_y), "aomhip_build_inter_pred_batch")

    def sse_batch(self, a, b, frame, w, h, d_cands, n, d_out):
        check(lib.aomhip_sse_batch(self.h, C.byref(a), C.byref(b), frame, w, h, d_cands, n, d_out), "aomhip_sse_batch")

    def hadamard_batch(self, d_res, stride, n, flavour, d_blocks, n_blocks, d_coeff=None, d_satd=None):
        check(lib.aomhip_hadamard_batch(self.h, d_res, stride, n, flavour, d_blocks, n_blocks, d_coeff, d_satd), "aomhip_hadamard_batch")

    def cost_coeffs_txb_batch(self, d_qcoeff, tx_size, d_blocks, n_blocks, tx_type, d_eob, d_txb_ctx, d_costs, d_cost, laplacian=False):
        """av1_cost_coeffs_txb[_laplacian] minus get_tx_type_cost: d_costs = LV_MAP_COEFF_COST (944 ints) + eob_cost[2][11]"""
        f = lib.aomhip_cost_coeffs_txb_laplacian_batch if laplacian else lib.aomhip_cost_coeffs_txb_batch
        check(f(self.h, d_qcoeff, tx_size, d_blocks, n_blocks, tx_type, d_eob, d_txb_ctx, d_costs, d_cost), "aomhip_cost_coeffs_txb_batch")

    def txb_entropy_context_batch(self, d_qcoeff, tx_size, d_blocks, n_blocks, tx_type, d_eob, d_out):
        check(lib.aomhip_txb_entropy_context_batch(self.h, d_qcoeff, tx_size, d_blocks, n_blocks, tx_type, d_eob, d_out), "aomhip_txb_entropy_context_batch")

    def get_nz_map_contexts_batch(self, d_levels, levels_pitch, tx_size, d_blocks, n_blocks, tx_type, d_eob, d_contexts, contexts_pitch):
        check(lib.aomhip_get_nz_map_contexts_batch(self.h, d_levels, levels_pitch, tx_size, d_blocks, n_blocks, tx_type, d_eob, d_contexts, contexts_pitch),
              "aomhip_get_nz_map_contexts_batch")

    def sum_sse_2d_i16_batch(self, d_residual, stride, width, height, d_blocks, n_blocks, d_sse, d_sum=None):
        check(lib.aomhip_sum_sse_2d_i16_batch(self.h, d_residual, stride, width, height, d_blocks, n_blocks, d_sse, d_sum), "aomhip_sum_sse_2d_i16_batch")

    def txb_init_levels_batch(self, d_coeff, w, h, d_off, n_blocks, d_levels, pitch):
        check(lib.aomhip_txb_init_levels_batch(self.h, d_coeff, w, h, d_off, n_blocks, d_levels, pitch), "aomhip_txb_init_levels_batch")

    def selfguided_restoration_batch(self, dgd, dgd_frame, d_units, h_units, n_units, d_idx, max_w, max_h, d_flt0, d_flt1, flt_stride, flt_pitch):
        """av1_selfguided_restoration per restoration unit (rect_dtype records, sgr_params index per unit): two int32 outputs each."""
        hu = None if h_units is None else np.ascontiguousarray(h_units).ctypes.data
        check(lib.aomhip_selfguided_restoration_batch(self.h, C.byref(dgd), dgd_frame, d_units, hu, n_units, d_idx, max_w, max_h, d_flt0, d_flt1, flt_stride, flt_pitch),
              "aomhip_selfguided_restoration_batch")

    def apply_selfguided_restoration_batch(self, dat, dat_frame, dst, dst_frame, d_units, h_units, n_units, d_idx, d_xqd, max_w, max_h, d_flt0, d_flt1, flt_stride,
                                           flt_pitch):
        hu = None if h_units is None else np.ascontiguousarray(h_units).ctypes.data
        check(lib.aomhip_apply_selfguided_restoration_batch(self.h, C.byref(dat), dat_frame, C.byref(dst), dst_frame, d_units, hu, n_units, d_idx, d_xqd, max_w, max_h,
                                                            d_flt0, d_flt1, flt_stride, flt_pitch), "aomhip_apply_selfguided_restoration_batch")

    def wiener_convolve_add_src_batch(self, dat, dat_frame, dst, dst_frame, d_units, h_units, n_units, d_filters, max_w, max_h):
        hu = None if h_units is None else np.ascontiguousarray(h_units).ctypes.data
        check(lib.aomhip_wiener_convolve_add_src_batch(self.h, C.byref(dat), dat_frame, C.byref(dst), dst_frame, d_units, hu, n_units, d_filters, max_w, max_h),
              "aomhip_wiener_convolve_add_src_batch")

    def calc_proj_params_batch(self, src, src_frame, dat, dat_frame, d_units, n_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, d_H, d_C):
        """av1_calc_proj_params[_high_bd] per restoration unit: H (4 int64) and C (2 int64) each."""
        check(lib.aomhip_calc_proj_params_batch(self.h, C.byref(src), src_frame, C.byref(dat), dat_frame, d_units, n_units, d_flt0, d_flt1, flt_stride, flt_pitch,
                                                d_radii, d_H, d_C), "aomhip_calc_proj_params_batch")

    def pixel_proj_error_batch(self, src, src_frame, dat, dat_frame, d_units, n_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, d_xq, n_xq, d_err):
        """av1_[lowbd|highbd]_pixel_proj_error per (unit, xq): int64 each."""
        check(lib.aomhip_pixel_proj_error_batch(self.h, C.byref(src), src_frame, C.byref(dat), dat_frame, d_units, n_units, d_flt0, d_flt1, flt_stride, flt_pitch,
                                                d_radii, d_xq, n_xq, d_err), "aomhip_pixel_proj_error_batch")

    def scaled_pred_batch(self, ref, ref_frame, pred, pred_frame, bw, bh, fx, fy, x_step_qn, y_step_qn, d_blocks, n_blocks):
        """av1_[highbd_]convolve_2d_scale for a batch of blocks (scaled_block_dtype records), single reference."""
        check(lib.aomhip_scaled_pred_batch(self.h, C.byref(ref), ref_frame, C.byref(pred), pred_frame, bw, bh, fx, fy, x_step_qn, y_step_qn, d_blocks, n_blocks),
              "aomhip_scaled_pred_batch")

    def scaled_pred_compound_batch(self, ref, ref_frame, pred, pred_frame, bw, bh, fx, fy, x_step_qn, y_step_qn, d_blocks, n_blocks, d_conv, conv_stride, do_average,
                                   weights=None):
        check(lib.aomhip_scaled_pred_compound_batch(self.h, C.byref(ref), ref_frame, None if pred is None else C.byref(pred), pred_frame, bw, bh, fx, fy, x_step_qn,
                                                    y_step_qn, d_blocks, n_blocks, d_conv, conv_stride, do_average, int(weights is not None),
                                                    weights[0] if weights else 0, weights[1] if weights else 0), "aomhip_scaled_pred_compound_batch")

    def warp_affine_batch(self, ref, ref_frame, pred, pred_frame, ssx, ssy, d_blocks, n_blocks, max_w, max_h):
        """av1_[highbd_]warp_affine for a batch of blocks (warp_block_dtype records), single reference, not compound."""
        check(lib.aomhip_warp_affine_batch(self.h, C.byref(ref), ref_frame, C.byref(pred), pred_frame, ssx, ssy, d_blocks, n_blocks, max_w, max_h),
              "aomhip_warp_affine_batch")

    def warp_affine_compound_batch(self, ref, ref_frame, pred, pred_frame, ssx, ssy, d_blocks, n_blocks, max_w, max_h, d_conv, conv_stride, do_average,
                                   weights=None):
        """av1_[highbd_]warp_affine with is_compound: do_average 0 fills the CONV_BUF, 1 blends the second reference in (weights = (fwd, bck) or None)."""
        check(lib.aomhip_warp_affine_compound_batch(self.h, C.byref(ref), ref_frame, None if pred is None else C.byref(pred), pred_frame, ssx, ssy, d_blocks, n_blocks,
                                                    max_w, max_h, d_conv, conv_stride, do_average, int(weights is not None), weights[0] if weights else 0,
                                                    weights[1] if weights else 0), "aomhip_warp_affine_compound_batch")

    def wedge_sse_from_residuals_batch(self, d_r1, d_d, d_masks, n, n_blocks, n_masks, d_sse):
        """av1_wedge_sse_from_residuals for every (block, mask): d_sse[i * n_masks + k] (uint64)."""
        check(lib.aomhip_wedge_sse_from_residuals_batch(self.h, d_r1, d_d, d_masks, n, n_blocks, n_masks, d_sse), "aomhip_wedge_sse_from_residuals_batch")

    def wedge_sign_from_residuals_batch(self, d_ds, d_masks, n, n_blocks, n_masks, d_limits, d_sign):
        """av1_wedge_sign_from_residuals for every (block, mask): d_sign[i * n_masks + k] (int8), limits per block (int64)."""
        check(lib.aomhip_wedge_sign_from_residuals_batch(self.h, d_ds, d_masks, n, n_blocks, n_masks, d_limits, d_sign), "aomhip_wedge_sign_from_residuals_batch")

    def wedge_compute_delta_squares_batch(self, d_a, d_b, n, n_blocks, d_d):
        check(lib.aomhip_wedge_compute_delta_squares_batch(self.h, d_a, d_b, n, n_blocks, d_d), "aomhip_wedge_compute_delta_squares_batch")

    def cdef_search_sse_luma(self, recon, recon_frame, source, source_frame, d_strengths, n, d_skip, damping, fb_stride, d_sse,
                             d_dir=None, d_var=None):
        check(lib.aomhip_cdef_search_sse_luma(self.h, C.byref(recon), recon_frame, C.byref(source), source_frame, d_strengths, n, d_skip,
                                              damping, fb_stride, d_sse, d_dir, d_var), "aomhip_cdef_search_sse_luma")

    def cdef_search_sse_chroma(self, recon, recon_frame, source, source_frame, xdec, ydec, d_luma_dir, d_strengths, n, d_skip, damping,
                               fb_stride, d_sse):
        check(lib.aomhip_cdef_search_sse_chroma(self.h, C.byref(recon), recon_frame, C.byref(source), source_frame, xdec, ydec, d_luma_dir,
                                                d_strengths, n, d_skip, damping, fb_stride, d_sse), "aomhip_cdef_search_sse_chroma")

    def lpf_search_sse(self, recon, recon_frame, scratch, scratch_frame, source, source_frame, d_params, trial_stride, n_trials, units_stride,
                       sharpness, passes, d_sse):
        check(lib.aomhip_lpf_search_sse(self.h, C.byref(recon), recon_frame, C.byref(scratch), scratch_frame, C.byref(source), source_frame,
                                        d_params, trial_stride, n_trials, units_stride, sharpness, passes, d_sse), "aomhip_lpf_search_sse")

    def compute_stats_batch(self, dgd, dgd_frame, src, src_frame, win, d_units, h_units, n, downsample, d_M, d_H):
        hp = h_units.ctypes.data if h_units is not None else None
        check(lib.aomhip_compute_stats_batch(self.h, C.byref(dgd), dgd_frame, C.byref(src), src_frame, win, d_units, hp, n, downsample, d_M, d_H),
              "aomhip_compute_stats_batch")

    def plane_sse(self, a, a_frame, b, b_frame, d_sse):
        check(lib.aomhip_plane_sse(self.h, C.byref(a), a_frame, C.byref(b), b_frame, d_sse), "aomhip_plane_sse")

    def build_compound_pred_batch(self, ref0, f0, ref1, f1, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n, filter_x=0, filter_y=0, fwd=0,
                                  bck=0, ss_x=0, ss_y=0):
        check(lib.aomhip_build_compound_pred_batch(self.h, C.byref(ref0), f0, C.byref(ref1), f1, C.byref(pred), pred_frame, bw, bh, d_blocks,
                                                   d_mv0, d_mv1, n, filter_x, filter_y, fwd, bck, ss_x, ss_y), "aomhip_build_compound_pred_batch")

    def build_masked_compound_pred_batch(self, ref0, f0, ref1, f1, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n, filter_x, filter_y, d_mask,
                                         d_mask_offset, mask_stride, subw=0, subh=0, ss_x=0, ss_y=0):
        check(lib.aomhip_build_masked_compound_pred_batch(self.h, C.byref(ref0), f0, C.byref(ref1), f1, C.byref(pred), pred_frame, bw, bh, d_blocks,
                                                          d_mv0, d_mv1, n, filter_x, filter_y, d_mask, d_mask_offset, mask_stride, subw, subh,
                                                          ss_x, ss_y), "aomhip_build_masked_compound_pred_batch")

    def build_diffwtd_compound_pred_batch(self, ref0, f0, ref1, f1, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n, filter_x, filter_y,
                                          mask_type, d_mask_out=None):
        check(lib.aomhip_build_diffwtd_compound_pred_batch(self.h, C.byref(ref0), f0, C.byref(ref1), f1, C.byref(pred), pred_frame, bw, bh, d_blocks,
                                                           d_mv0, d_mv1, n, filter_x, filter_y, mask_type, d_mask_out),
              "aomhip_build_diffwtd_compound_pred_batch")

    def blend_a64_1d_batch(self, pred, pred_frame, adjacent, adjacent_frame, d_items, n, d_masks):
        check(lib.aomhip_blend_a64_1d_batch(self.h, C.byref(pred), pred_frame, C.byref(adjacent), adjacent_frame, d_items, n, d_masks),
              "aomhip_blend_a64_1d_batch")

    def build_pred_fullpel(self, ref, ref_frame, pred, pred_frame, bw, bh, d_blocks, d_mv, n):
        check(lib.aomhip_build_pred_fullpel(self.h, C.byref(ref), ref_frame, C.byref(pred), pred_frame, bw, bh, d_blocks,
                                            d_mv, n), "aomhip_build_pred_fullpel")


def comm_unique_id():
    uid = np.zeros(128, np.uint8)
    check(lib.aomhip_comm_unique_id(uid.ctypes.data), "aomhip_comm_unique_id")
    return uid


class ValuProbeResult(C.Structure):
    _fields_ = [("wave_insts_per_s_per_simd", C.c_double), ("launch_ms", C.c_double), ("memtime_ticks_per_wave_inst", C.c_double),
                ("memtime_hz", C.c_double), ("waves_per_simd", C.c_int32), ("compute_units", C.c_int32)]


def valu_issue_probe_names():
    out, i = [], 0
    while True:
        n = lib.aomhip_valu_issue_probe_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def tile_column_bounds(width, n_cols, sb_size=64):
    """[n_cols, 2] int32 pixel bounds of the uniform tile columns; (0, 0) for ranks beyond the last column."""
    b = np.zeros((n_cols, 2), np.int32)
    n = lib.aomhip_tile_column_bounds(width, n_cols, sb_size, b.ctypes.data)
    return b, n


def tile_column_bounds_balanced(width, n_cols, sb_size=64, max_width_sb=0):
    """auto_tile_size_balancing's columns for n_cols = 2^k ranks: ([n_cols, 2] int32 pixel bounds, number of columns that exist)."""
    k = int(n_cols).bit_length() - 1
    assert 1 << k == n_cols, "auto_tile_size_balancing takes a power-of-two column count"
    b = np.zeros((n_cols, 2), np.int32)
    n = lib.aomhip_tile_column_bounds_balanced(width, k, sb_size, max_width_sb, b.ctypes.data)
    return b, n


def tile_column_bounds_widths(width, widths_sb, n_cols, sb_size=64, max_width_sb=0):
    w = np.ascontiguousarray(widths_sb, np.int32)
    b = np.zeros((n_cols, 2), np.int32)
    n = lib.aomhip_tile_column_bounds_widths(width, sb_size, w.ctypes.data, len(w), max_width_sb, n_cols, b.ctypes.data)
    return b, n


def recon_exchange_plan(n_ranks, rank, col_bounds, width, halo=-1):
    """(send, recv): [n_ranks, 2] int32 pixel column ranges per peer (aomhip_recon_exchange_plan; host only)."""
    b = np.ascontiguousarray(col_bounds, np.int32).reshape(-1, 2)
    send, recv = np.zeros((n_ranks, 2), np.int32), np.zeros((n_ranks, 2), np.int32)
    check(lib.aomhip_recon_exchange_plan(n_ranks, rank, b.ctypes.data, width, halo, send.ctypes.data, recv.ctypes.data), "aomhip_recon_exchange_plan")
    return send, recv


def planes_from_tensor(t, width, height, border, bit_depth, n_frames=1):
    """Wrap caller-owned device memory (e.g. a torch tensor that RCCL collectives operate on) as an aomhip_planes
    ring: t must hold n_frames * frame_stride elements laid out like aomhip_planes_alloc does.  Returns
    (Planes, frame_stride_elements, rows)."""
    stride = lib.aomhip_calc_stride(width, border)
    rows = ((height + 7) & ~7) + 2 * border
    frame_elems = (rows * stride + 255) & ~255
    assert t.numel() >= frame_elems * n_frames
    p = Planes()
    p.base, p.frame_stride, p.width, p.height = t.data_ptr(), frame_elems, width, height
    p.stride, p.border, p.bit_depth, p.n_frames = stride, border, bit_depth, n_frames
    return p, frame_elems, rows
