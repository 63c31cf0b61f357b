#!/usr/bin/env python3
"""Golden vectors for the producers of the in-loop filter parameter planes, from the interpreted reference (build
container only; see ref_c_eval.py):

  ref_eval_filtermaps.npz
      set_lpf_parameters (av1/common/av1_loopfilter.c:223-328) with get_transform_size (:197-217), av1_get_filter_level
      (:68-111) and av1_loop_filter_frame_init (:126-195) at every 4x4 unit of random mode-info grids, both edge
      directions, luma and 4:2:0 chroma: filter_length and the level index of the chosen thresholds;
      av1_cdef_compute_sb_list (av1/common/cdef.c:36-68) over the same grids.

AV1_COMMON, MB_MODE_INFO, MACROBLOCKD and macroblockd_plane contain dozens of unrelated members (and types outside the
evaluator's subset), so the evaluator sees them as opaque parameter types with views of just the members these functions
read; loop_filter_info_n, struct loopfilter, struct segmentation and buf_2d are used as the reference declares them."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save  # noqa: E402

BW = [4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 4, 16, 8, 32, 16, 64]     # Block_Width / Block_Height (AV1 spec 9.3)
BH = [4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 64, 32, 64, 128, 64, 128, 16, 4, 32, 8, 64, 16]
TXW = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TXH = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


def make_evaluator():
    ev = evaluator([])
    for nm, v in (("AOM_PLANE_Y", "0"), ("AOM_PLANE_U", "1"), ("AOM_PLANE_V", "2")):   # aom/aom_image.h (#defines inside a struct body)
        ev.define(nm, v)
    for nm in ("TX_SIZE", "TX_TYPE", "PREDICTION_MODE", "MV_REFERENCE_FRAME", "BLOCK_SIZE", "PARTITION_TYPE", "PLANE_TYPE", "EDGE_DIR"):
        ev.typedefs.setdefault(nm, R.U8 if nm not in ("MV_REFERENCE_FRAME",) else R.I8)
    for f in ["aom_dsp/txfm_common.h", "aom_dsp/aom_dsp_common.h", "av1/common/common.h", "av1/common/enums.h", "av1/common/common_data.h", "av1/common/common_data.c", "av1/common/seg_common.h",
              "av1/common/mv.h", "aom_scale/yv12config.h", "av1/common/blockd.h", "av1/common/av1_loopfilter.h", "av1/common/av1_loopfilter.c",
              "av1/common/cdef_block.h", "av1/common/cdef.c"]:
        ev.load("/root/reference/" + f)
    mbmi = ev.structs["<opaque>MB_MODE_INFO"]
    mbmi.fields = [("bsize", R.U8), ("tx_size", R.U8), ("inter_tx_size", ("arr", R.U8, 16)), ("skip_txfm", R.I8), ("ref_frame", ("arr", R.I8, 2)),
                   ("mode", R.U8), ("segment_id", R.U8), ("delta_lf_from_base", R.I8), ("delta_lf", ("arr", R.I8, 4)), ("use_intrabc", R.U8),
                   ("cdef_strength", R.I8)]
    mip = R.StructType("CommonModeInfoParams")
    mip.fields = [("mi_grid_base", ("ptr", ("ptr", mbmi))), ("mi_stride", R.I32), ("mi_rows", R.I32), ("mi_cols", R.I32)]
    ev.structs["CommonModeInfoParams"] = mip
    ev.typedefs["CommonModeInfoParams"] = mip
    dq = R.StructType("DeltaQInfo")
    dq.fields = [("delta_lf_present_flag", R.I32), ("delta_lf_multi", R.I32)]
    cm = ev.structs["<opaque>AV1_COMMON"]
    cm.fields = [("mi_params", mip), ("lf_info", ev.typedefs["loop_filter_info_n"]), ("delta_q_info", dq), ("lf", ev.structs["loopfilter"]),
                 ("seg", ev.structs["segmentation"])]
    pd = ev.structs.get("<opaque>macroblockd_plane") or ev.structs.setdefault("macroblockd_plane", R.StructType("macroblockd_plane"))
    pd.fields = [("subsampling_x", R.I32), ("subsampling_y", R.I32), ("dst", ev.structs["buf_2d"])]
    return ev, mbmi, cm, pd


def random_grid(rng, mi_rows, mi_cols):
    """A random tiling of the mode-info grid by AV1 block sizes; returns per-block records and the per-mi block index."""
    owner = -np.ones((mi_rows, mi_cols), np.int32)
    blocks = []
    sizes = [b for b in range(22) if BW[b] <= 64 and BH[b] <= 64]
    for r in range(mi_rows):
        for c in range(mi_cols):
            if owner[r, c] >= 0:
                continue
            cand = [b for b in sizes if r % (BH[b] // 4) == 0 and c % (BW[b] // 4) == 0 and r + BH[b] // 4 <= mi_rows and c + BW[b] // 4 <= mi_cols
                    and np.all(owner[r:r + BH[b] // 4, c:c + BW[b] // 4] < 0)]
            b = int(rng.choice(cand))
            owner[r:r + BH[b] // 4, c:c + BW[b] // 4] = len(blocks)
            inter = int(rng.integers(0, 2))
            # transform sizes that tile the block (the largest rectangular one, or a split of it)
            fits = [t for t in range(19) if BW[b] % TXW[t] == 0 and BH[b] % TXH[t] == 0 and TXW[t] <= 64 and TXH[t] <= 64]
            big = max(fits, key=lambda t: TXW[t] * TXH[t])
            tx = big if rng.integers(0, 2) else int(rng.choice(fits))
            blocks.append(dict(bsize=b, row=r, col=c, inter=inter, skip=int(rng.integers(0, 3) == 0), tx_size=tx,
                               inter_tx=[int(rng.choice([t for t in fits if TXW[t] * TXH[t] <= TXW[tx] * TXH[tx]] or [tx])) for _ in range(16)],
                               ref=int(rng.integers(1, 8)) if inter else 0, mode=int(rng.integers(13, 25)) if inter else int(rng.integers(0, 13)),
                               seg=int(rng.integers(0, 8)), cdef=int(rng.integers(-1, 4))))
    return blocks, owner


def main():
    ev, mbmi_t, cm_t, pd_t = make_evaluator()
    rng = np.random.default_rng(20261101)
    arrays, cases = {}, []
    k = 0
    for (mi_rows, mi_cols, delta_lf, mode_ref, seg_on, sharp) in ((16, 24, 0, 1, 1, 0), (18, 16, 0, 0, 0, 3), (16, 16, 1, 1, 1, 5), (32, 20, 0, 1, 0, 0)):
        blocks, owner = random_grid(rng, mi_rows, mi_cols)
        objs = []
        for b in blocks:
            o = ev.interp.alloc(mbmi_t, True)
            ev.set(o, "bsize", b["bsize"]); ev.set(o, "tx_size", b["tx_size"]); ev.set(o, "skip_txfm", b["skip"])
            for i, t in enumerate(b["inter_tx"]):
                ev.set(o, "inter_tx_size[%d]" % i, t)
            ev.set(o, "ref_frame[0]", b["ref"]); ev.set(o, "ref_frame[1]", -1); ev.set(o, "mode", b["mode"]); ev.set(o, "segment_id", b["seg"])
            b["dlf_base"] = int(rng.integers(-20, 21)); b["dlf"] = [int(v) for v in rng.integers(-20, 21, 4)]
            ev.set(o, "delta_lf_from_base", b["dlf_base"])
            for i in range(4):
                ev.set(o, "delta_lf[%d]" % i, b["dlf"][i])
            ev.set(o, "cdef_strength", b["cdef"])
            objs.append(o)
        grid = ev.interp.alloc(("arr", ("ptr", mbmi_t), mi_rows * mi_cols + 2 * mi_cols + 2), True)
        for r in range(mi_rows):
            for c in range(mi_cols):
                ev.set(grid, "[%d]" % (r * mi_cols + c), objs[owner[r, c]])
        cm = ev.interp.alloc(cm_t, True)
        ev.set(cm, "mi_params.mi_grid_base", R.Ptr(grid.buf, grid.off, grid.t, ()));   # (the array decays to a pointer to its first element)
        ev.set(cm, "mi_params.mi_stride", mi_cols)
        ev.set(cm, "mi_params.mi_rows", mi_rows); ev.set(cm, "mi_params.mi_cols", mi_cols)
        fl = [int(rng.integers(1, 64)), int(rng.integers(1, 64)), int(rng.integers(0, 64)), int(rng.integers(1, 64))]
        ev.set(cm, "lf.filter_level[0]", fl[0]); ev.set(cm, "lf.filter_level[1]", fl[1]); ev.set(cm, "lf.filter_level_u", fl[2]); ev.set(cm, "lf.filter_level_v", fl[3])
        ev.set(cm, "lf.sharpness_level", sharp); ev.set(cm, "lf.mode_ref_delta_enabled", mode_ref)
        ref_d = [1, 0, 0, 0, -1, 0, -1, -1] if rng.integers(0, 2) else [int(v) for v in rng.integers(-6, 7, 8)]
        mode_d = [int(v) for v in rng.integers(-4, 5, 2)]
        for i in range(8):
            ev.set(cm, "lf.ref_deltas[%d]" % i, ref_d[i])
        for i in range(2):
            ev.set(cm, "lf.mode_deltas[%d]" % i, mode_d[i])
        ev.set(cm, "delta_q_info.delta_lf_present_flag", delta_lf); ev.set(cm, "delta_q_info.delta_lf_multi", int(rng.integers(0, 2)) if delta_lf else 0)
        seg_mask = np.zeros(8, np.int64); seg_data = np.zeros((8, 8), np.int64)
        ev.set(cm, "seg.enabled", seg_on)
        if seg_on:
            for s in range(8):
                for f in range(1, 5):
                    if rng.integers(0, 2):
                        seg_mask[s] |= 1 << f
                        seg_data[s, f] = int(rng.integers(-30, 31))
                        ev.set(cm, "seg.feature_data[%d][%d]" % (s, f), int(seg_data[s, f]))
                ev.set(cm, "seg.feature_mask[%d]" % s, int(seg_mask[s]))
        ev.call("av1_loop_filter_frame_init", cm, 0, 3)
        lvl_tab = np.array([[[[[ev.get(cm, "lf_info.lvl[%d][%d][%d][%d][%d]" % (p, s, d, r, m)) for m in range(2)] for r in range(8)] for d in range(2)]
                             for s in range(8)] for p in range(3)], np.uint8)
        dmulti = ev.get(cm, "delta_q_info.delta_lf_multi")
        rec = dict(k=k, mi_rows=mi_rows, mi_cols=mi_cols, delta_lf=delta_lf, delta_lf_multi=int(dmulti), mode_ref=mode_ref, seg_on=seg_on, sharpness=sharp,
                   filter_level=fl, ref_deltas=ref_d, mode_deltas=mode_d, blocks=blocks)
        arrays["owner%d" % k] = owner
        arrays["lvl%d" % k] = lvl_tab
        arrays["segmask%d" % k], arrays["segdata%d" % k] = seg_mask.astype(np.uint8), seg_data.astype(np.int16)
        params_t = ev.typedefs["AV1_DEBLOCKING_PARAMETERS"]
        for plane, (ssx, ssy) in ((0, (0, 0)), (1, (1, 1)), (2, (1, 1))):
            pw, ph = (mi_cols * 4) >> ssx, (mi_rows * 4) >> ssy
            pd = ev.interp.alloc(pd_t, True)
            ev.set(pd, "subsampling_x", ssx); ev.set(pd, "subsampling_y", ssy); ev.set(pd, "dst.width", pw); ev.set(pd, "dst.height", ph)
            out = np.zeros((ph // 4, pw // 4, 5), np.int16)   # len_v, lvl_v, len_h, lvl_h, ts
            thr0 = ev.field(cm, "lf_info.lfthr[0]")
            for uy in range(ph // 4):
                for ux in range(pw // 4):
                    for d in range(2):
                        prm = ev.interp.alloc(params_t, True)
                        mode_step = (1 << ssx) if d == 0 else (mi_cols << ssy)
                        ts = ev.call("set_lpf_parameters", prm, mode_step, cm, None, d, 4 * ux, 4 * uy, plane, pd)
                        fl_ = ev.get(prm, "filter_length")
                        lv = 0
                        if fl_:
                            lf = ev.get(prm, "lfthr")
                            lv = (lf.off - thr0.off) // max(1, (ev.field(cm, "lf_info.lfthr[1]").off - thr0.off))
                        out[uy, ux, 2 * d], out[uy, ux, 2 * d + 1] = fl_, lv
                        out[uy, ux, 4] = ts
            arrays["edge%d_p%d" % (k, plane)] = out
        # av1_cdef_compute_sb_list for every 64x64 of the grid
        dl_t = ev.typedefs["cdef_list"]
        nfb_r, nfb_c = (mi_rows + 15) // 16, (mi_cols + 15) // 16
        skipmap = np.ones((mi_rows // 2, mi_cols // 2), np.uint8)
        for fr in range(nfb_r):
            for fc in range(nfb_c):
                dlist = ev.interp.alloc(("arr", dl_t, 256), True)
                cnt = ev.call("av1_cdef_compute_sb_list", ev.field(cm, "mi_params"), fr * 16, fc * 16, R.Ptr(dlist.buf, dlist.off, dlist.t, ()), 12)
                for i in range(cnt):
                    by, bx = ev.get(dlist, "[%d].by" % i), ev.get(dlist, "[%d].bx" % i)
                    skipmap[fr * 8 + by, fc * 8 + bx] = 0
        arrays["cdefskip%d" % k] = skipmap
        cases.append(rec)
        print("case", k, "done", flush=True)
        k += 1
    save("ref_eval_filtermaps.npz", arrays, cases)


if __name__ == "__main__":
    main()
