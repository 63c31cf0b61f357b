#!/usr/bin/env python3
"""Golden vectors for the temporal filter AFTER its motion search (build container only; output tests/golden/ref_eval_tf_apply.npz).

Interpreted where they lie (tests/golden/ref_c_eval.py):
  * av1_[highbd_]convolve_2d_facade with MULTITAP_SHARP2 (the 12-tap set) on both axes -- what tf_build_predictor reaches through
    av1_enc_build_one_inter_predictor for every sub-block (av1/encoder/temporal_filter.c:331-392); the position arithmetic of
    init_subpel_params (av1/common/reconinter.h:130-165, unscaled) is the one the other prediction fixtures already use;
  * tf_apply_temporal_filter_self, av1_apply_temporal_filter_c with compute_square_diff / compute_luma_sq_error_sum,
    tf_normalize_filtered_frame (OD_DIVU through the reference's own OD_DIVU_SMALL_CONSTS table), on whole blocks, luma only and 4:2:0.
YV12_BUFFER_CONFIG and MACROBLOCKD are seen as opaque parameter types with views of just the members these functions read
(y_crop_width / y_crop_height / strides / buffers / flags; plane[].subsampling_x / _y, bd).  libm's pow / log / sqrt / exp are the
host's (python's math module calls the same libm)."""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save  # noqa: E402


def make_evaluator():
    ev = evaluator(["av1/common/filter.h", "av1/common/convolve.h", "aom_dsp/aom_convolve.c", "av1/common/convolve.c", "aom_dsp/odintrin.h",
                    "aom_dsp/odintrin.c"])
    for nm, v in (("AOM_PLANE_Y", "0"), ("AOM_PLANE_U", "1"), ("AOM_PLANE_V", "2"), ("YV12_FLAG_HIGHBITDEPTH", "8"), ("MAX_MB_PLANE", "3")):
        ev.define(nm, v)
    for nm in ("BLOCK_SIZE",):
        ev.typedefs.setdefault(nm, R.U8)
    yv = R.StructType("YV12_BUFFER_CONFIG")
    yv.fields = [("y_crop_width", R.I32), ("y_crop_height", R.I32), ("strides", ("arr", R.I32, 2)), ("buffers", ("arr", ("ptr", R.U8), 3)),
                 ("flags", R.U32)]
    ev.structs["YV12_BUFFER_CONFIG"] = yv
    ev.typedefs["YV12_BUFFER_CONFIG"] = yv
    pd = R.StructType("macroblockd_plane")
    pd.fields = [("subsampling_x", R.I32), ("subsampling_y", R.I32)]
    ev.structs["macroblockd_plane"] = pd
    mbd = R.StructType("MACROBLOCKD")
    mbd.fields = [("plane", ("arr", pd, 3)), ("bd", R.I32), ("error_info", ("ptr", R.I32))]
    ev.structs["MACROBLOCKD"] = mbd
    ev.typedefs["MACROBLOCKD"] = mbd
    mv = R.StructType("mv")
    mv.fields = [("row", R.I16), ("col", R.I16)]
    ev.structs["mv"] = mv
    ev.typedefs["MV"] = mv
    ev.globs["block_size_wide"] = ev.array([4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 4, 16, 8, 32, 16, 64], "uint8_t", (22,))
    ev.globs["block_size_high"] = ev.array([4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 64, 32, 64, 128, 64, 128, 16, 4, 32, 8, 64, 16], "uint8_t", (22,))
    it = ev.interp
    it.pycalls["pow"] = lambda i, a: (math.pow(float(a[0][0]), float(a[1][0])), R.F64)
    it.pycalls["log"] = lambda i, a: (math.log(float(a[0][0])), R.F64)
    it.pycalls["sqrt"] = lambda i, a: (math.sqrt(float(a[0][0])), R.F64)
    it.pycalls["exp"] = lambda i, a: (math.exp(float(a[0][0])), R.F64)
    it.pycalls["aom_memalign"] = lambda i, a: (R.Ptr([0] * (int(a[1][0]) // 4), 0, R.U32), R.PTR)
    it.pycalls["aom_free"] = lambda i, a: (None, R.VOID)
    it.pycalls["is_cur_buf_hbd"] = lambda i, a: (1 if ev.get(a[0][0], "bd") > 8 else 0, R.I32)
    # the functions themselves: their text out of temporal_filter.c (the file as a whole needs the encoder's headers)
    src = open("/root/reference/av1/encoder/temporal_filter.c").read()
    hdr = open("/root/reference/av1/encoder/temporal_filter.h").read()
    defs = "\n".join(l for l in hdr.split("\n") if l.startswith("#define TF_") or l.startswith("#define BH") or l.startswith("#define BW"))

    def func_text(name):
        k = src.index(name + "(")
        start = src.rindex("\n", 0, src.rindex("\n", 0, k)) if src[src.rindex("\n", 0, k) + 1:k].strip() in ("", "void", "static void") else src.rindex("\n", 0, k)
        # back up to the start of the declaration line(s): the return type sits on the line of the name or the one before
        start = src.rindex("\n", 0, k)
        line = src[start + 1:k]
        if line.strip() == "":
            start = src.rindex("\n", 0, start)
        depth, i = 0, src.index("{", k)
        while True:
            if src[i] == "{":
                depth += 1
            elif src[i] == "}":
                depth -= 1
                if depth == 0:
                    break
            i += 1
        return src[start + 1:i + 1]

    text = defs + "\n#define CLIP(v, lo, hi) ((v) < (lo) ? (lo) : (v) > (hi) ? (hi) : (v))\n"
    text += "static INLINE int is_frame_high_bitdepth(const YV12_BUFFER_CONFIG *frame) { return (frame->flags & YV12_FLAG_HIGHBITDEPTH) ? 1 : 0; }\n"
    for name in ("tf_apply_temporal_filter_self", "compute_square_diff", "compute_luma_sq_error_sum", "av1_apply_temporal_filter_c",
                 "tf_normalize_filtered_frame"):
        text += func_text(name) + "\n"
    ev.load_text(text, "temporal_filter.c (five functions)")
    missing = [n for n in ("tf_apply_temporal_filter_self", "compute_square_diff", "compute_luma_sq_error_sum", "av1_apply_temporal_filter_c",
                           "tf_normalize_filtered_frame") if n not in ev.funcs]
    assert not missing, (missing, ev.skipped[-5:])
    return ev, yv, mbd, mv


def smooth_frames(rng, bd, n, h, w, pad):
    base = rng.integers(0, 1 << bd, (h + 2 * pad + 16, w + 2 * pad + 16)).astype(np.float64)
    for _ in range(2):
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, (1, 1), (0, 1))) / 4
    out = []
    for f in range(n):
        img = base[f:f + h + 2 * pad, 2 * f:2 * f + w + 2 * pad] + rng.normal(0, (1 << bd) / 64.0, (h + 2 * pad, w + 2 * pad))
        out.append(np.clip(np.rint(img), 0, (1 << bd) - 1).astype(np.int64))
    return out


def main():
    ev, yv_t, mbd_t, mv_t = make_evaluator()
    rng = np.random.default_rng(20261201)
    arrays, cases = {}, []
    k = 0
    # ---- 1. the 12-tap predictor: facade with MULTITAP_SHARP2 (MULTITAP_SHARP2 = 4 in InterpFilter, filter.h:30-43)
    S, ROWS = 96, 80
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        base = rng.integers(0, mx + 1, (ROWS, S))
        base[:24] = np.where(rng.integers(0, 2, (24, S)) > 0, mx, 0)
        arrays["p%d" % bd] = base.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        P = ev.array(base.ravel(), ct)
        cpv = ev.call("get_conv_params", 0, 0, bd)
        cp = R.Ptr([cpv], 0, cpv.st)
        for (w, h) in ((16, 16), (8, 8)):
            fp = [ev.call("av1_get_interp_filter_params_with_block_size", 4, w), ev.call("av1_get_interp_filter_params_with_block_size", 4, h)]
            assert ev.get(fp[0], "taps") == 12
            filt = R.Ptr(fp, 0, ("ptr", ev.structs["InterpFilterParams"]))
            for trial in range(5):
                x0, y0 = int(rng.integers(8, S - w - 8)), int(rng.integers(8, 14) if trial % 2 == 0 else rng.integers(8, ROWS - h - 8))
                sx, sy = [(0, 0), (int(rng.integers(1, 16)), 0), (0, int(rng.integers(1, 16))), (int(rng.integers(1, 16)), int(rng.integers(1, 16))),
                          (2 * int(rng.integers(1, 8)), 2 * int(rng.integers(1, 8)))][trial]
                dst = ev.array([0] * (w * h), ct)
                args = [P.add(y0 * S + x0), S, dst, w, w, h, filt, sx, 16, sy, 16, 0, cp]
                if bd > 8:
                    args.append(bd)
                ev.call("av1_convolve_2d_facade" if bd == 8 else "av1_highbd_convolve_2d_facade", *args)
                arrays["c%d" % k] = np.asarray(dst.buf, np.uint16)
                cases.append({"kind": "convolve12", "k": k, "bd": bd, "w": w, "h": h, "x0": x0, "y0": y0, "sx": sx, "sy": sy})
                k += 1
    # ---- 2. whole blocks: self + two reference frames -> accum / count -> normalised pixels
    W, H, PAD = 64, 64, 0
    for (bd, planes, ssx, ssy, q, strength, noise) in ((8, 3, 1, 1, 40, 5, (1.8, 0.9, 1.1)), (10, 3, 1, 1, 160, 2, (3.0, 2.0, 2.5)),
                                                        (10, 1, 0, 0, 12, 4, (0.4, 0, 0)), (12, 3, 0, 0, 64, 6, (2.0, 2.0, 2.0)),
                                                        (8, 1, 0, 0, 255, 1, (6.0, 0, 0))):
        ct = "uint8_t" if bd == 8 else "uint16_t"
        fr = [smooth_frames(rng, bd, 3, H >> (ssy if p else 0), W >> (ssx if p else 0), PAD) for p in range(planes)]   # [plane][frame]
        for p in range(planes):
            for f in range(3):
                arrays["f%d_%d_%d" % (k, p, f)] = fr[p][f].astype(np.uint16)
        strides = [W, W >> ssx]
        mbd = ev.interp.alloc(mbd_t, True)
        ev.set(mbd, "bd", bd)
        for p in range(3):
            ev.set(mbd, "plane[%d].subsampling_x" % p, ssx if p else 0); ev.set(mbd, "plane[%d].subsampling_y" % p, ssy if p else 0)
        bufs = [[ev.array(fr[p][f].ravel(), ct) for p in range(planes)] for f in range(3)]
        yvs = []
        for f in range(3):
            y = ev.interp.alloc(yv_t, True)
            ev.set(y, "y_crop_width", W); ev.set(y, "y_crop_height", H); ev.set(y, "strides[0]", strides[0]); ev.set(y, "strides[1]", strides[1])
            ev.set(y, "flags", 8 if bd > 8 else 0)
            for p in range(planes):
                ev.set(y, "buffers[%d]" % p, bufs[f][p])
            yvs.append(y)
        outb = [ev.array([0] * fr[p][0].size, ct) for p in range(planes)]
        yo = ev.interp.alloc(yv_t, True)
        ev.set(yo, "strides[0]", strides[0]); ev.set(yo, "strides[1]", strides[1]); ev.set(yo, "flags", 8 if bd > 8 else 0)
        for p in range(planes):
            ev.set(yo, "buffers[%d]" % p, outb[p])
        noise_p = R.Ptr([float(v) for v in noise], 0, R.F64)
        pels = 1024 + (2 * (32 >> ssx) * (32 >> ssy) if planes == 3 else 0)
        blocks = []
        for (mb_row, mb_col) in ((0, 0), (1, 1)):
            accum = ev.array([0] * pels, "uint32_t")
            count = ev.array([0] * pels, "uint16_t")
            ev.call("tf_apply_temporal_filter_self", yvs[1], mbd, 9, mb_row, mb_col, planes, accum, count)   # BLOCK_32X32 = 9
            rec = {"mb_row": mb_row, "mb_col": mb_col, "refs": []}
            for f in (0, 2):
                # predictors come from the pinned oracle-independent source: random pixels near the frame (the predictor itself is pinned above)
                pred_np = [np.clip(fr[p][1][(mb_row * 32 >> (ssy if p else 0)):((mb_row * 32 >> (ssy if p else 0)) + (32 >> (ssy if p else 0))),
                                            (mb_col * 32 >> (ssx if p else 0)):((mb_col * 32 >> (ssx if p else 0)) + (32 >> (ssx if p else 0)))]
                                   + rng.integers(-(6 << (bd - 8)), (6 << (bd - 8)) + 1, (32 >> (ssy if p else 0), 32 >> (ssx if p else 0)))
                                   * (rng.integers(0, 3, (32 >> (ssy if p else 0), 32 >> (ssx if p else 0))) > 0), 0, (1 << bd) - 1) for p in range(planes)]
                pred = ev.array(np.concatenate([a.ravel() for a in pred_np]), ct)
                mvs_np = rng.integers(-40, 41, (4, 2)) if f == 0 else np.array([[0, 0], [200, -150], [3, 2], [-90, 7]])
                mses_np = [int(v) for v in (rng.integers(0, 60, 4) << (bd - 8))] if f == 0 else [0, 5000, 17, 300]
                mvs = ev.interp.alloc(("arr", mv_t, 4), True)
                for s in range(4):
                    ev.set(mvs, "[%d].row" % s, int(mvs_np[s][0])); ev.set(mvs, "[%d].col" % s, int(mvs_np[s][1]))
                mses = ev.array(mses_np, "int")
                ev.call("av1_apply_temporal_filter_c", yvs[1], mbd, 9, mb_row, mb_col, planes, noise_p, R.Ptr(mvs.buf, mvs.off, mvs.t, ()), mses, q,
                        strength, pred, accum, count)
                rec["refs"].append({"mvs": [[int(a), int(b)] for a, b in mvs_np], "mses": mses_np, "pred": len(arrays)})
                arrays["pred%d_%d_%d_%d" % (k, mb_row, mb_col, f)] = np.concatenate([a.ravel() for a in pred_np]).astype(np.uint16)
                arrays["accum%d_%d_%d_%d" % (k, mb_row, mb_col, f)] = np.asarray(accum.buf, np.uint32)
                arrays["count%d_%d_%d_%d" % (k, mb_row, mb_col, f)] = np.asarray(count.buf, np.uint16)
            ev.call("tf_normalize_filtered_frame", mbd, 9, mb_row, mb_col, planes, accum, count, yo)
            blocks.append(rec)
        for p in range(planes):
            arrays["out%d_%d" % (k, p)] = np.asarray([v if v is not None else 0 for v in outb[p].buf], np.uint16).reshape(fr[p][0].shape)
        cases.append({"kind": "apply", "k": k, "bd": bd, "planes": planes, "ss_x": ssx, "ss_y": ssy, "q": q, "strength": strength, "noise": list(noise),
                      "w": W, "h": H, "blocks": blocks})
        k += 1
        print("apply case", k, "done", flush=True)
    save("ref_eval_tf_apply.npz", arrays, cases)


if __name__ == "__main__":
    main()
