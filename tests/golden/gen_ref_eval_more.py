#!/usr/bin/env python3
"""More golden vectors from the interpreted reference (build container only; see ref_c_eval.py):

  ref_eval_txfm2d.npz   av1_fwd_txfm2d_{W}x{H}_c (all 19 sizes, valid tx types) and av1_inv_txfm2d_add_{W}x{H}_c
                        (8/10/12-bit) -- av1/encoder/av1_fwd_txfm2d.c, av1/common/av1_inv_txfm2d.c and the 1-D networks
  ref_eval_tables.npz   update_sharpness + the hev rule of av1_loop_filter_init (av1/common/av1_loopfilter.c:47-66,
                        118-120) for every level x sharpness; av1_build_quantizer (av1/encoder/av1_quantize.c:580-674)
                        for 8/10/12-bit, all 256 qindex, Y / U / V with delta_q
  ref_eval_cdef_fb.npz  av1_cdef_filter_fb (av1/common/cdef_block.c:323-426) on whole 64x64 filter blocks: luma
                        (direction search, variance-adjusted strength) and chroma for 4:2:0 / 4:2:2 / 4:4:0 / 4:4:4
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_c_eval as R  # noqa: E402
from gen_ref_eval_golden import evaluator, save  # noqa: E402

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]


def gen_txfm2d():
    import pyoracle as orc      # only av1_tx_valid (which (size, type) pairs exist) is taken from the oracle
    ev = evaluator(["aom_dsp/txfm_common.h", "av1/common/common.h", "av1/common/common_data.h", "av1/common/av1_txfm.h", "av1/common/av1_txfm.c",
                    "av1/encoder/av1_fwd_txfm1d.h", "av1/encoder/av1_fwd_txfm1d_cfg.h", "av1/encoder/av1_fwd_txfm1d.c", "av1/encoder/av1_fwd_txfm2d.c",
                    "av1/common/av1_inv_txfm1d.h", "av1/common/av1_inv_txfm1d_cfg.h", "av1/common/av1_inv_txfm1d.c", "av1/common/av1_inv_txfm2d.c",
                    "av1/encoder/hybrid_fwd_txfm.c"])
    rng = np.random.default_rng(20261007)
    arrays, cases = {}, []
    k = 0
    for tx_size in range(19):
        w, h = TX_W[tx_size], TX_H[tx_size]
        n = w * h
        types = [t for t in range(16) if orc.av1_tx_valid(tx_size, t)]
        if n >= 1024:
            types = types[:1] if n >= 2048 else types[:2]
        elif n >= 256:
            types = types[::3]
        for tx_type in types:
            kinds = ("max", "min", "rand9", "rand11") if tx_type == types[0] else ("rand9", "rand11")
            if n >= 2048:
                kinds = ("max", "rand9")
            for kind in kinds:
                bd = 10 if kind == "rand11" else 8
                lim = (1 << (bd + 1)) - 1 if kind != "rand11" else 2047
                if kind == "max":
                    x = np.full(n, 255)
                elif kind == "min":
                    x = np.full(n, -255)
                else:
                    x = rng.integers(-(lim >> 1), (lim >> 1) + 1, n)
                S = w + 3                                        # a stride that is not the width
                buf = np.zeros(h * S, np.int64)
                buf.reshape(h, S)[:, :w] = x.reshape(h, w)
                out = ev.array([0] * n, "int32_t")
                ev.call("av1_fwd_txfm2d_%dx%d_c" % (w, h), ev.array(buf, "int16_t"), out, S, tx_type, bd)
                coeff = np.asarray(out.buf, np.int32)
                arrays["x%d" % k], arrays["c%d" % k] = x.astype(np.int16), coeff
                rec = {"tx_size": tx_size, "tx_type": tx_type, "w": w, "h": h, "bd": bd, "kind": kind}
                # inverse: a sparsified / quantised version of the coefficients added to a random prediction
                if kind.startswith("rand"):
                    dq = (coeff // 8) * 8
                    dq[np.abs(coeff) < 24] = 0
                    if n > 1024:                                  # 64-point sizes keep only the 32 low frequencies
                        dq = dq.reshape(-1)
                    ibd = bd if kind == "rand11" else int(rng.choice([8, 12]))
                    scale = 1 << (ibd - bd) if ibd >= bd else 1
                    dst = rng.integers(0, 1 << ibd, n)
                    dbuf = np.zeros(h * S, np.int64)
                    dbuf.reshape(h, S)[:, :w] = dst.reshape(h, w)
                    dp = ev.array(dbuf, "uint16_t")
                    ev.call("av1_inv_txfm2d_add_%dx%d_c" % (w, h), ev.array(dq * scale, "int32_t"), dp, S, tx_type, ibd)
                    arrays["dq%d" % k] = (dq * scale).astype(np.int32)
                    arrays["p%d" % k] = dst.astype(np.uint16)
                    arrays["r%d" % k] = np.asarray(dp.buf, np.uint16).reshape(h, S)[:, :w].copy()
                    rec["inv_bd"] = ibd
                cases.append(rec)
                k += 1
    # lossless 4x4: av1_fwht4x4_c and av1_highbd_iwht4x4_{16,1}_add_c (own generator: the cases above keep their values)
    rng2 = np.random.default_rng(20261012)
    for trial in range(24):
        bd = (8, 10, 12)[trial % 3]
        lim = (1 << bd) - 1
        x = rng2.integers(-lim, lim + 1, 16) if trial >= 3 else np.full(16, (lim, -lim, 0)[trial])
        S = 7
        buf = np.zeros(4 * S, np.int64)
        buf.reshape(4, S)[:, :4] = x.reshape(4, 4)
        out = ev.array([0] * 16, "int32_t")
        ev.call("av1_fwht4x4_c", ev.array(buf, "int16_t"), out, S)
        coeff = np.asarray(out.buf, np.int32)
        dq = coeff.copy()
        if trial % 4 == 1:
            dq[1:] = 0                                   # DC only: the _1_add form
        dst = rng2.integers(0, 1 << bd, 16)
        dbuf = np.zeros(4 * S, np.int64)
        dbuf.reshape(4, S)[:, :4] = dst.reshape(4, 4)
        dp = ev.array(dbuf, "uint16_t")
        eob = 1 if trial % 4 == 1 else 16
        ev.call("av1_highbd_iwht4x4_16_add_c" if eob > 1 else "av1_highbd_iwht4x4_1_add_c", ev.array(dq, "int32_t"), dp, S, bd)
        arrays["x%d" % k], arrays["c%d" % k] = x.astype(np.int16), coeff
        arrays["dq%d" % k], arrays["p%d" % k] = dq.astype(np.int32), dst.astype(np.uint16)
        arrays["r%d" % k] = np.asarray(dp.buf, np.uint16).reshape(4, S)[:, :4].copy()
        cases.append({"wht": 1, "tx_size": 0, "tx_type": 16, "w": 4, "h": 4, "bd": bd, "inv_bd": bd, "eob": eob})
        k += 1
    save("ref_eval_txfm2d.npz", arrays, cases)


def gen_tables():
    ev = evaluator(["aom/aom_codec.h", "av1/common/seg_common.h", "av1/common/blockd.h", "av1/common/av1_loopfilter.h", "av1/common/av1_loopfilter.c", "av1/common/quant_common.h", "av1/common/quant_common.c",
                    "av1/encoder/av1_quantize.h", "av1/encoder/av1_quantize.c"])
    arrays = {}
    # loop-filter thresholds: lfthr[lvl].{mblim, lim} from update_sharpness, hev_thr = lvl >> 4 (av1_loop_filter_init)
    th = np.zeros((8, 64, 3), np.int32)
    for sharp in range(8):
        lfi = ev.new("loop_filter_info_n")
        ev.interp.call("update_sharpness", [(lfi, R.PTR), (sharp, R.I32)])
        for lvl in range(64):
            th[sharp, lvl] = (ev.get(lfi, "lfthr[%d].mblim[0]" % lvl), ev.get(lfi, "lfthr[%d].lim[0]" % lvl), lvl >> 4)
            assert ev.get(lfi, "lfthr[%d].lim[15]" % lvl) == th[sharp, lvl, 1]
    arrays["lpf_thresholds"] = th
    # quantiser tables
    bits = {8: "AOM_BITS_8", 10: "AOM_BITS_10", 12: "AOM_BITS_12"}
    for bd in (8, 10, 12):
        for (ydc, udc, uac, vdc, vac) in ((0, 0, 0, 0, 0), (-7, 5, -3, 9, 12)):
            if bd != 8 and ydc:
                continue
            qs, dq = ev.new("QUANTS"), ev.new("Dequants")
            t0 = time.time()
            ev.call("av1_build_quantizer", ev.globs[bits[bd]].buf[0], ydc, udc, uac, vdc, vac, qs, dq)
            for plane in "yuv":
                tab = np.zeros((256, 5, 8), np.int32)
                for f, name in enumerate(("%s_zbin", "%s_round", "%s_quant", "%s_quant_shift")):
                    tab[:, f, :] = np.asarray(ev.field(qs, name % plane).buf, np.int64).reshape(256, 8)
                tab[:, 4, :] = np.asarray(ev.field(dq, "%s_dequant_QTX" % plane).buf, np.int64).reshape(256, 8)
                arrays["quant_%s_bd%d_%s" % (plane, bd, "d" if ydc else "0")] = tab.astype(np.int16)
            print("  build_quantizer bd %d: %.1f s" % (bd, time.time() - t0))
    save("ref_eval_tables.npz", arrays, [{"deltas": [-7, 5, -3, 9, 12]}])


def gen_cdef_fb():
    ev = evaluator(["aom/aom_image.h", "av1/common/cdef_block.h", "av1/common/cdef.h", "av1/common/cdef_block.c"])
    bstride = ev.interp.ev(R.Parser(ev.pp.expand(R.tokenize("CDEF_BSTRIDE")), ev.typedefs).expr())[0]
    VL = 0x4000
    rng = np.random.default_rng(20261008)
    arrays, cases = {}, []
    k = 0
    P = 192                                          # luma plane; the filter block under test is the centre 64x64
    for bd in (8, 10):
        mx = (1 << bd) - 1
        yy, xx = np.mgrid[0:P, 0:P]
        luma = np.clip(mx // 2 + (mx // 5) * np.sin(xx / 9.0 + yy / 17.0) + (mx // 7) * ((xx // 23 + yy // 29) % 2)
                       + rng.integers(-(6 << (bd - 8)), (6 << (bd - 8)) + 1, (P, P)), 0, mx).astype(np.int64)
        arrays["luma%d" % bd] = luma.astype(np.uint16)
        for (xdec, ydec, pli) in ((0, 0, 0), (1, 1, 1), (1, 0, 1), (0, 1, 1), (0, 0, 1)):
            for (level, sec, damping, at_edge) in ((4, 2, 6, 0), (9, 0, 5, 0), (0, 4, 3, 0), (15, 1, 4, 1), (1, 2, 6, 1)):
                if pli and (level, sec) == (9, 0) and bd == 10:
                    continue
                plane = luma if not pli else np.ascontiguousarray(luma[::(1 << ydec), ::(1 << xdec)])
                pw, ph = 64 >> xdec, 64 >> ydec             # the filter block in this plane
                y0, x0 = (0, 0) if at_edge else (ph, pw)     # at_edge: the top-left filter block (frame edges above / left)
                tile = np.full((ph + 4 + 2, bstride), VL, np.int64)      # 2 border rows above and below (+ spare)
                # cdef_prepare_fb semantics: available neighbours are copied, frame edges are CDEF_VERY_LARGE
                for r in range(-2, ph + 2):
                    for c0, c1 in ((-8, pw + 8),):
                        ys = y0 + r
                        if ys < 0 or ys >= plane.shape[0]:
                            continue
                        xs0, xs1 = max(x0 + c0, 0), min(x0 + c1, plane.shape[1])
                        tile[r + 2, 8 + (xs0 - x0):8 + (xs1 - x0)] = plane[ys, xs0:xs1]
                skip = rng.random((8, 8)) < 0.2
                dl = [(by, bx) for by in range(8) for bx in range(8) if not skip[by, bx]]
                dlist = ev.interp.alloc(("arr", ev.typedefs["cdef_list"], 64), True)
                for i, (by, bx) in enumerate(dl):
                    ev.set(dlist, "[%d].by" % i, by); ev.set(dlist, "[%d].bx" % i, bx)
                dirs = ev.interp.alloc(("arr", ("arr", R.I32, 16), 16), True)
                var = ev.interp.alloc(("arr", ("arr", R.I32, 16), 16), True)
                if pli:                                     # chroma reuses the luma directions: give it a full table
                    ldir = rng.integers(0, 8, (16, 16))
                    for i, v in enumerate(ldir.ravel()):
                        dirs.buf[i] = int(v)
                    arrays["ld%d" % k] = ldir[:8, :8].astype(np.uint8)
                inp = ev.array(tile.ravel(), "uint16_t")
                use8 = bd == 8
                dst = ev.array(plane[y0:y0 + ph, x0:x0 + pw].ravel(), "uint8_t" if use8 else "uint16_t")
                ev.call("av1_cdef_filter_fb", dst if use8 else None, None if use8 else dst, pw, inp.add(2 * bstride + 8), xdec, ydec,
                        dirs.deref()[0], None, var.deref()[0], pli, dlist.deref()[0], len(dl), level, sec, damping, bd - 8)
                arrays["o%d" % k] = np.asarray(dst.buf, np.uint16).reshape(ph, pw)
                arrays["s%d" % k] = skip.astype(np.uint8)
                if not pli:
                    arrays["d%d" % k] = np.asarray(dirs.buf, np.int32).reshape(16, 16)[:8, :8]
                    arrays["v%d" % k] = np.asarray(var.buf, np.int32).reshape(16, 16)[:8, :8]
                cases.append({"bd": bd, "xdec": xdec, "ydec": ydec, "pli": pli, "level": level, "sec": sec, "damping": damping, "at_edge": at_edge,
                              "y0": y0, "x0": x0, "pw": pw, "ph": ph})
                k += 1
    save("ref_eval_cdef_fb.npz", arrays, cases)



def gen_compound():
    """The compound / masked / OBMC members of aom_variance_fn_ptr_t (svaf, jsvaf, msdf, msvf, osdf, ovf, osvf), 8/10/12-bit."""
    ev = evaluator(["aom_dsp/variance.h", "aom_dsp/blend.h", "av1/common/mv.h", "aom_scale/yv12config.h", "av1/common/blockd.h", "aom_dsp/sad.c",
                    "aom_dsp/variance.c", "aom_dsp/sad_av1.c"])
    rng = np.random.default_rng(20261013)
    arrays, cases = {}, []
    S, ROWS = 160, 150
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        a = rng.integers(0, mx + 1, (ROWS, S))
        b = np.clip(a + rng.integers(-(24 << (bd - 8)), (24 << (bd - 8)) + 1, a.shape), 0, mx)
        arrays["a%d" % bd], arrays["b%d" % bd] = a.astype(np.uint16), b.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        pa, pb = ev.array(a.ravel(), ct), ev.array(b.ravel(), ct)
        hb = "" if bd == 8 else "highbd_"
        hbn = "" if bd == 8 else "highbd_%d_" % bd
        hbo = "" if bd == 8 else ("highbd_" if bd == 8 else "highbd_%d_" % bd)
        sizes = {8: [(4, 4), (8, 8), (16, 16), (16, 8), (8, 16), (4, 16), (16, 4), (32, 16), (32, 32), (64, 32), (128, 128)],
                 10: [(4, 4), (8, 8), (16, 16), (8, 4), (16, 32), (32, 8), (64, 16), (64, 64)],
                 12: [(4, 8), (8, 8), (16, 16), (8, 32), (16, 64), (32, 64)]}[bd]
        for (w, h) in sizes:
            ax, ay = int(rng.integers(0, S - w - 1)), int(rng.integers(0, ROWS - h - 1))
            bx, by = int(rng.integers(0, S - w - 1)), int(rng.integers(0, ROWS - h - 1))
            A, B = pa.add(ay * S + ax), pb.add(by * S + bx)
            sp = rng.integers(0, mx + 1, w * h)
            ms = w + 3                                                            # a mask with a stride of its own
            mask = rng.integers(0, 65, (h, ms))
            mask[0, 0], mask[h - 1, w - 1] = 0, 64
            om = rng.integers(0, 4097, w * h)                                      # OBMC weights (scaled by 4096)
            ws = rng.integers(0, mx + 1, w * h) * 4096 - rng.integers(0, mx + 1, w * h) * (4096 - om)
            arrays["sp%d" % k], arrays["mask%d" % k] = sp.astype(np.uint16), mask.astype(np.uint8)
            arrays["om%d" % k], arrays["ws%d" % k] = om.astype(np.int32), ws.astype(np.int32)
            SP, M, OM, WS = ev.array(sp, ct), ev.array(mask.ravel(), "uint8_t"), ev.array(om, "int32_t"), ev.array(ws, "int32_t")
            sse = ev.array([0], "uint32_t")
            rec = {"k": k, "bd": bd, "w": w, "h": h, "ax": ax, "ay": ay, "bx": bx, "by": by, "mask_stride": ms}
            rec["svaf"] = []
            for (xo, yo) in ((3, 5), (0, 4), (0, 0)):
                v = ev.call("aom_%ssub_pixel_avg_variance%dx%d_c" % (hbn if bd > 8 else "", w, h), A, S, xo, yo, B, S, sse, SP)
                rec["svaf"].append([xo, yo, v, sse.buf[0]])
            rec["jsvaf"] = []
            for (xo, yo, fwd, bck) in ((5, 2, 9, 7), (1, 0, 13, 3), (6, 7, 4, 12)):
                jcp = ev.new("DIST_WTD_COMP_PARAMS")
                ev.set(jcp, "use_dist_wtd_comp_avg", 1); ev.set(jcp, "fwd_offset", fwd); ev.set(jcp, "bck_offset", bck)
                v = ev.call("aom_%sdist_wtd_sub_pixel_avg_variance%dx%d_c" % (hbn if bd > 8 else "", w, h), A, S, xo, yo, B, S, sse, SP, jcp)
                rec["jsvaf"].append([xo, yo, fwd, bck, v, sse.buf[0]])
            rec["msvf"] = []
            for (xo, yo, inv) in ((7, 1, 0), (2, 3, 1), (0, 0, 0)):
                fn = "aom_masked_sub_pixel_variance%dx%d_c" % (w, h) if bd == 8 else "aom_highbd_%d_masked_sub_pixel_variance%dx%d_c" % (bd, w, h)
                v = ev.call(fn, A, S, xo, yo, B, S, SP, M, ms, inv, sse)
                rec["msvf"].append([xo, yo, inv, v, sse.buf[0]])
            # msdf(src, ref, second_pred, ...): the source block is b, the reference a (the raw kernel value; the
            # _bits10 / _bits12 wrappers of encoder_utils.h:363-387 shift it by 2 / 4)
            rec["msdf"] = [[inv, ev.call("aom_%smasked_sad%dx%d_c" % (hb, w, h), B, S, A, S, SP, M, ms, inv)] for inv in (0, 1)]
            rec["osdf"] = ev.call("aom_%sobmc_sad%dx%d_c" % (hb, w, h), A, S, WS, OM)
            on = "aom_obmc_" if bd == 8 else ("aom_highbd_obmc_" if bd == 8 else "aom_highbd_%d_obmc_" % bd)
            v = ev.call(on + "variance%dx%d_c" % (w, h), A, S, WS, OM, sse)
            rec["ovf"] = [v, sse.buf[0]]
            rec["osvf"] = []
            for (xo, yo) in ((2, 6), (0, 0)):
                v = ev.call(on + "sub_pixel_variance%dx%d_c" % (w, h), A, S, xo, yo, WS, OM, sse)
                rec["osvf"].append([xo, yo, v, sse.buf[0]])
            cases.append(rec)
            k += 1
    # 8-bit content through the highbd_8 / un-numbered highbd symbols (what a high-bit-depth build uses for 8-bit video)
    a, b = arrays["a8"].astype(np.int64), arrays["b8"].astype(np.int64)
    pa, pb = ev.array(a.ravel(), "uint16_t"), ev.array(b.ravel(), "uint16_t")
    for (w, h) in ((8, 8), (16, 16)):
        ax, ay, bx, by = 5, 7, 11, 3
        A, B = pa.add(ay * S + ax), pb.add(by * S + bx)
        sp = rng.integers(0, 256, w * h)
        om = rng.integers(0, 4097, w * h)
        ws = rng.integers(0, 256, w * h) * 4096 - rng.integers(0, 256, w * h) * (4096 - om)
        mask = rng.integers(0, 65, (h, w))
        arrays["sp%d" % k], arrays["mask%d" % k] = sp.astype(np.uint16), mask.astype(np.uint8)
        arrays["om%d" % k], arrays["ws%d" % k] = om.astype(np.int32), ws.astype(np.int32)
        SP, M, OM, WS = ev.array(sp, "uint16_t"), ev.array(mask.ravel(), "uint8_t"), ev.array(om, "int32_t"), ev.array(ws, "int32_t")
        sse = ev.array([0], "uint32_t")
        rec = {"k": k, "bd": 8, "hbd8": 1, "w": w, "h": h, "ax": ax, "ay": ay, "bx": bx, "by": by, "mask_stride": w}
        v = ev.call("aom_highbd_8_sub_pixel_avg_variance%dx%d_c" % (w, h), A, S, 3, 5, B, S, sse, SP)
        rec["svaf"] = [[3, 5, v, sse.buf[0]]]
        v = ev.call("aom_highbd_8_masked_sub_pixel_variance%dx%d_c" % (w, h), A, S, 7, 1, B, S, SP, M, w, 1, sse)
        rec["msvf"] = [[7, 1, 1, v, sse.buf[0]]]
        v = ev.call("aom_highbd_obmc_variance%dx%d_c" % (w, h), A, S, WS, OM, sse)
        rec["ovf"] = [v, sse.buf[0]]
        v = ev.call("aom_highbd_obmc_sub_pixel_variance%dx%d_c" % (w, h), A, S, 2, 6, WS, OM, sse)
        rec["osvf"] = [[2, 6, v, sse.buf[0]]]
        cases.append(rec)
        k += 1
    save("ref_eval_compound.npz", arrays, cases)


def gen_convolve():
    """av1_[highbd_]convolve_2d_facade (single reference, unscaled): the copy / x_sr / y_sr / 2d_sr choice, the 8-tap and (for a
    dimension <= 4) 4-tap kernel sets, round_0 / round_1 from get_conv_params."""
    ev = evaluator(["av1/common/filter.h", "av1/common/convolve.h", "aom_dsp/aom_convolve.c", "av1/common/convolve.c"])
    rng = np.random.default_rng(20261015)
    arrays, cases = {}, []
    S, ROWS = 112, 100
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        base = rng.integers(0, mx + 1, (ROWS, S))
        base[:20] = np.where(rng.integers(0, 2, (20, S)) > 0, mx, 0)       # a band of extreme pixels: the clips and the offsets must hold
        arrays["p%d" % bd] = base.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        P = ev.array(base.ravel(), ct)
        cpv = ev.call("get_conv_params", 0, 0, bd)                        # a struct by value; give it storage to point at
        cp = R.Ptr([cpv], 0, cpv.st)
        sizes = {8: [(4, 4), (8, 8), (16, 16), (4, 16), (16, 4), (32, 16), (8, 32), (64, 32)],
                 10: [(4, 8), (8, 4), (16, 16), (16, 8), (32, 32), (64, 16)],
                 12: [(4, 4), (8, 16), (16, 16), (16, 32)]}[bd]
        for (w, h) in sizes:
            x0, y0 = int(rng.integers(4, S - w - 5)), int(rng.integers(4, ROWS - h - 5))
            if k % 3 == 0:
                y0 = int(rng.integers(4, 14))                               # inside the extreme band
            for (fxi, fyi) in ((0, 0), (1, 2), (2, 1), (3, 3)):
                fp = [ev.call("av1_get_interp_filter_params_with_block_size", fxi, w), ev.call("av1_get_interp_filter_params_with_block_size", fyi, h)]
                filt = R.Ptr(fp, 0, ("ptr", ev.structs["InterpFilterParams"]))
                subs = [(int(rng.integers(1, 16)), 0), (0, int(rng.integers(1, 16))), (int(rng.integers(1, 16)), int(rng.integers(1, 16)))]
                if (fxi, fyi) == (0, 0):
                    subs.append((0, 0))
                for (sx, sy) in subs:
                    dst = ev.array([0] * (w * h), ct)
                    fn = "av1_convolve_2d_facade" if bd == 8 else "av1_highbd_convolve_2d_facade"
                    args = [P.add(y0 * S + x0), S, dst, w, w, h, filt, sx, 16, sy, 16, 0, cp]
                    if bd > 8:
                        args.append(bd)
                    ev.call(fn, *args)
                    arrays["d%d" % k] = np.asarray(dst.buf, np.uint16)
                    cases.append({"k": k, "bd": bd, "w": w, "h": h, "x0": x0, "y0": y0, "fx": fxi, "fy": fyi, "sx": sx, "sy": sy})
                    k += 1
    save("ref_eval_convolve.npz", arrays, cases)


def gen_rdhelp():
    """aom_[highbd_]sse_c, the Hadamard family + aom_satd[_lp]_c (aom_dsp/avg.c), av1_txb_init_levels_c (av1/encoder/encodetxb.c)."""
    ev = evaluator(["aom_dsp/sse.c", "aom_dsp/avg.c", "av1/common/txb_common.h", "av1/encoder/encodetxb.c"])
    rng = np.random.default_rng(20261017)
    arrays, cases = {}, []
    k = 0
    # sse: odd sizes too (the function takes any width x height)
    for bd in (8, 12):
        mx = (1 << bd) - 1
        S = 80
        a, b = rng.integers(0, mx + 1, (70, S)), rng.integers(0, mx + 1, (70, S))
        arrays["sa%d" % bd], arrays["sb%d" % bd] = a.astype(np.uint16), b.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        pa, pb = ev.array(a.ravel(), ct), ev.array(b.ravel(), ct)
        for (w, h) in ((4, 4), (8, 8), (16, 16), (64, 64), (5, 7), (12, 3), (33, 17), (64, 1)):
            ox, oy = int(rng.integers(0, S - w)), int(rng.integers(0, 70 - h))
            v = ev.call("aom_sse_c" if bd == 8 else "aom_highbd_sse_c", pa.add(oy * S + ox), S, pb.add(3 * S + 2), S, w, h)
            cases.append({"kind": "sse", "bd": bd, "w": w, "h": h, "ox": ox, "oy": oy, "value": int(v)})
    # hadamard: residuals of 8-bit range for the plain / lp forms (their int16 arithmetic is sized for that), 12-bit range for highbd,
    # plus one full-range int16 input per flavour so that the wrap-around of the 16-bit intermediates is pinned too
    S = 40
    for flavour, name, sizes, rngbits in ((0, "aom_hadamard_%dx%d_c", (4, 8, 16, 32), 8), (1, "aom_hadamard_lp_%dx%d_c", (8, 16), 8),
                                          (2, "aom_highbd_hadamard_%dx%d_c", (8, 16, 32), 12)):
        for n in sizes:
            for trial in range(3):
                lim = (1 << rngbits) - 1 if trial < 2 else 32767
                res = rng.integers(-lim, lim + 1, (n + 3, S))
                if trial == 1:
                    res[:] = np.where(rng.integers(0, 2, res.shape) > 0, lim, -lim)
                arrays["r%d" % k] = res.astype(np.int16)
                R_ = ev.array(res.ravel(), "int16_t")
                out = ev.array([0] * (n * n), "int16_t" if flavour == 1 else "tran_low_t")
                ev.call(name % (n, n), R_.add(2 * S + 3), S, out)
                satd = ev.call("aom_satd_lp_c" if flavour == 1 else "aom_satd_c", out, n * n)
                arrays["c%d" % k] = np.asarray(out.buf, np.int32)
                cases.append({"kind": "hadamard", "k": k, "n": n, "flavour": flavour, "x": 3, "y": 2, "satd": int(satd), "wide_input": int(trial == 2)})
                k += 1
    # txb_init_levels
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (4, 16), (16, 4), (8, 32), (32, 8), (16, 32)):
        coeff = rng.integers(-300, 301, w * h)
        coeff[rng.integers(0, w * h, 5)] = (-70000, 127, -128, 128, 0)
        Cf = ev.array(coeff, "tran_low_t")
        size = (h + 4) * (w + 4) + 16
        lv = ev.array([0xAA] * size, "uint8_t")
        ev.call("av1_txb_init_levels_c", Cf, w, h, lv)
        arrays["tc%d" % k], arrays["tl%d" % k] = coeff.astype(np.int32), np.asarray(lv.buf, np.uint8)
        cases.append({"kind": "levels", "k": k, "w": w, "h": h})
        k += 1
    save("ref_eval_rdhelp.npz", arrays, cases)


def gen_cdef_search():
    """get_filt_error (av1/encoder/pickcdef.c:401-501): the per-filter-block, per-strength distortion of av1_cdef_search --
    av1_cdef_filter_fb + aom_sse (8-bit build path) / compute_cdef_dist_highbd (high-bit-depth path) -- for full and partly
    skipped 64x64 luma filter blocks.  CdefSearchCtx and macroblockd_plane contain dozens of unrelated members, so the
    evaluator sees them as opaque parameter types with views of just the members this function reads."""
    ev = evaluator([])
    ev.define("AOM_PLANE_Y", "0")      # aom/aom_image.h:208 (a #define inside a struct body, which the loader does not pick up)
    for f in ["av1/common/common.h", "av1/common/common_data.h", "av1/common/cdef_block.h", "av1/common/cdef.h", "av1/common/cdef_block.c", "aom_dsp/sse.c",
              "aom_dsp/variance.c", "av1/common/mv.h", "aom_scale/yv12config.h", "av1/common/blockd.h", "av1/encoder/mcomp_structs.h",
              "av1/encoder/mcomp.h", "av1/encoder/pickcdef.h", "av1/encoder/pickcdef.c"]:
        ev.load("/root/reference/" + f)
    # CdefSearchCtx parses as the reference declares it.  `struct macroblockd_plane` does not (its entropy-context and
    # quantiser-matrix members are outside the evaluator's subset), so it gets a view with the members get_filt_error reads.
    ctx_t = ev.typedefs["CdefSearchCtx"]
    pd_t = ev.structs["macroblockd_plane"]
    pd_t.fields = [("subsampling_x", R.I32), ("subsampling_y", R.I32), ("dst", ev.structs["buf_2d"])]
    rng = np.random.default_rng(20261019)
    arrays, cases = {}, []
    BS, VB, HB = 144, 2, 8                       # CDEF_BSTRIDE, CDEF_VBORDER, CDEF_HBORDER (cdef_block.h:24-31)
    BLOCK_8X8, BLOCK_64X64 = 3, 12
    k = 0
    BLOCK_4X4 = 0
    for bd, variant, pli in ((8, "full", 0), (8, "partial", 0), (10, "full", 0), (10, "partial", 0), (8, "partial", 1), (10, "partial", 1)):
        hbd = int(bd > 8)
        cs = bd - 8
        dec = 1 if pli else 0                   # 4:2:0 chroma: 32 x 32 filter block of 4 x 4 units
        N = 64 >> dec
        mx = (1 << bd) - 1
        smooth = np.add.outer(np.arange(N + 2 * VB) * (3 << cs), np.arange(N + 2 * HB) * (2 << cs)) % (mx + 1)
        img = np.clip(smooth + rng.integers(-(12 << cs), (12 << cs) + 1, smooth.shape), 0, mx)
        src = np.clip(img[VB:VB + N, HB:HB + N] + rng.integers(-(6 << cs), (6 << cs) + 1, (N, N)), 0, mx)
        inbuf = np.full(((N + 2 * VB) * BS + 64,), 0x4000, np.int64)     # CDEF_VERY_LARGE
        for r in range(N + 2 * VB):
            inbuf[r * BS:r * BS + N + 2 * HB] = img[r]
        if variant == "partial":            # a filter block at the frame's top-left corner: outside = CDEF_VERY_LARGE
            for r in range(N + 2 * VB):
                inbuf[r * BS:r * BS + HB] = 0x4000
            inbuf[:VB * BS] = 0x4000
        IN = ev.array(inbuf, "uint16_t")
        ct = "uint16_t" if hbd else "uint8_t"
        # the reconstruction plane (pd->dst) = the footprint's pixels, the source plane (ref_buffer) = src; both N wide at (0, 0)
        rec_plane = ev.array(img[VB:VB + N, HB:HB + N].ravel(), ct)
        src_plane = ev.array(src.ravel(), ct)
        skip = np.zeros((8, 8), np.uint8)
        if variant == "partial":
            skip = (rng.integers(0, 3, (8, 8)) == 0).astype(np.uint8)
            skip[0, :4] = 0                  # a run of four unskipped neighbours: the 4-unit error path (pickcdef.c:383-389)
        ys, xs = np.nonzero(skip == 0)
        count = len(ys)
        dl_t = ev.typedefs["cdef_list"]
        dlist = ev.interp.alloc(("arr", dl_t, 64), True)
        for i, (by, bx) in enumerate(zip(ys, xs)):
            ev.set(dlist, "[%d].by" % i, int(by)); ev.set(dlist, "[%d].bx" % i, int(bx))
        cctx = ev.interp.alloc(ctx_t, True)
        ev.set(cctx, "damping", 5); ev.set(cctx, "use_highbitdepth", hbd)
        ev.set(cctx, "bsize[0]", BLOCK_8X8); ev.set(cctx, "bsize[1]", BLOCK_4X4); ev.set(cctx, "bsize[2]", BLOCK_4X4)
        for i in (1, 2):
            ev.set(cctx, "xdec[%d]" % i, 1); ev.set(cctx, "ydec[%d]" % i, 1)
        ev.set(cctx, "compute_cdef_dist_fn", R.FuncRef("compute_cdef_dist_highbd" if hbd else "compute_cdef_dist"))
        pd = ev.interp.alloc(pd_t, True)
        ev.set(pd, "dst.buf", rec_plane); ev.set(pd, "dst.stride", N)
        ev.set(pd, "subsampling_x", dec); ev.set(pd, "subsampling_y", dec)
        dirs = ev.interp.alloc(("arr", ("arr", R.I32, 16), 16), True)
        var = ev.interp.alloc(("arr", ("arr", R.I32, 16), 16), True)
        dirinit = ev.array([0], "int")
        if pli:                              # chroma reuses the directions the luma pass left in dir[][] (dirinit = 1)
            ldir = rng.integers(0, 8, (16, 16))
            for i, v in enumerate(ldir.ravel()):
                dirs.buf[i] = int(v)
            dirinit = ev.array([1], "int")
            arrays["ld%d" % k] = ldir[:8, :8].astype(np.uint8)
        errs = []
        strengths = [(0, 0), (1, 0), (0, 2), (3, 1), (7, 3), (15, 2), (4, 3), (9, 0)]
        for (pri, sec) in strengths:
            e = ev.call("get_filt_error", cctx, pd, dlist.deref()[0], dirs.deref()[0], dirinit, var.deref()[0], IN.add(VB * BS + HB), src_plane, N, 0, 0,
                        pri, sec, count, pli, cs, BLOCK_64X64)
            errs.append(int(e))
        arrays["in%d" % k], arrays["src%d" % k], arrays["skip%d" % k] = inbuf.astype(np.uint16), src.astype(np.uint16), skip
        cases.append({"k": k, "bd": bd, "variant": variant, "pli": pli, "count": count, "damping": 5, "strengths": strengths, "errors": errs})
        k += 1
    save("ref_eval_cdef_search.npz", arrays, cases)


def gen_lrstats():
    """av1_compute_stats_c (with and without the down-sampled rows) and av1_compute_stats_highbd_c (av1/encoder/pickrst.c)."""
    ev = evaluator([])
    for f in ["aom/aom_codec.h", "av1/common/restoration.h", "av1/encoder/pickrst.h", "av1/encoder/pickrst.c"]:
        ev.load("/root/reference/" + f)
    rng = np.random.default_rng(20261021)
    arrays, cases = {}, []
    S, ROWS = 40, 32
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        dgd = rng.integers(0, mx + 1, (ROWS, S))
        dgd[:8] = np.where(rng.integers(0, 2, (8, S)) > 0, mx, 0)
        src = np.clip(dgd + rng.integers(-(9 << (bd - 8)), (9 << (bd - 8)) + 1, dgd.shape), 0, mx)
        arrays["dgd%d" % bd], arrays["src%d" % bd] = dgd.astype(np.uint16), src.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        D, Sx = ev.array(dgd.ravel(), ct), ev.array(src.ravel(), ct)
        for (win, rect, ds) in ((7, (4, 17, 3, 13), 0), (5, (5, 14, 4, 15), 0), (7, (4, 15, 3, 14), 1), (5, (8, 13, 10, 12), 1)):
            if bd > 8 and ds:
                continue
            hs, he, vs, ve = rect
            win2 = win * win
            M, Hm = ev.array([0] * win2, "int64_t"), ev.array([0] * (win2 * win2), "int64_t")
            if bd == 8:
                ev.call("av1_compute_stats_c", win, D, Sx, hs, he, vs, ve, S, S, M, Hm, ds)
            else:
                ev.call("av1_compute_stats_highbd_c", win, D, Sx, hs, he, vs, ve, S, S, M, Hm, bd)
            arrays["M%d" % k], arrays["H%d" % k] = np.asarray(M.buf, np.int64), np.asarray(Hm.buf, np.int64)
            cases.append({"k": k, "bd": bd, "win": win, "rect": list(rect), "downsample": ds})
            k += 1
    save("ref_eval_lrstats.npz", arrays, cases)


def gen_convolve_compound():
    """Compound prediction through av1_[highbd_]convolve_2d_facade with is_compound = 1: first reference into the CONV_BUF
    (do_average 0), second reference averaged in (do_average 1), plain average and the distance weights; the copy / x / y / 2-D
    compound kernels av1_[highbd_]dist_wtd_convolve_{2d_copy,x,y,2d}_c (av1/common/convolve.c:176-370,670-868)."""
    ev = evaluator(["av1/common/filter.h", "av1/common/convolve.h", "aom_dsp/aom_convolve.c", "av1/common/convolve.c"])
    rng = np.random.default_rng(20261023)
    arrays, cases = {}, []
    S, ROWS = 96, 80
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        planes = []
        for r in range(2):
            base = rng.integers(0, mx + 1, (ROWS, S))
            base[:16] = np.where(rng.integers(0, 2, (16, S)) > 0, mx, 0)
            planes.append(base)
            arrays["p%d_%d" % (bd, r)] = base.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        P = [ev.array(pl.ravel(), ct) for pl in planes]
        sizes = {8: [(4, 4), (8, 8), (16, 16), (4, 16), (32, 8)], 10: [(8, 4), (16, 16), (16, 32)], 12: [(4, 8), (8, 8), (16, 16)]}[bd]
        for (w, h) in sizes:
            for (fxi, fyi, wts) in ((0, 0, None), (2, 1, (9, 7)), (1, 3, (4, 12)), (0, 2, None)):
                pos = [(int(rng.integers(4, S - w - 5)), int(rng.integers(4, ROWS - h - 5))) for _ in range(2)]
                if k % 2 == 0:
                    pos[0] = (pos[0][0], int(rng.integers(4, 10)))            # extreme band
                subs = [(int(rng.integers(0, 16)) * int(rng.integers(0, 2)), int(rng.integers(0, 16)) * int(rng.integers(0, 2))) for _ in range(2)]
                fp = [ev.call("av1_get_interp_filter_params_with_block_size", fxi, w), ev.call("av1_get_interp_filter_params_with_block_size", fyi, h)]
                filt = R.Ptr(fp, 0, ("ptr", ev.structs["InterpFilterParams"]))
                buf16 = ev.array([0] * (w * h), "uint16_t")
                dst = ev.array([0] * (w * h), ct)
                for r in range(2):
                    cpv = ev.call("get_conv_params_no_round", r, 0, buf16, w, 1, bd)
                    cp = R.Ptr([cpv], 0, cpv.st)
                    if wts:
                        ev.set(cp, "use_dist_wtd_comp_avg", 1); ev.set(cp, "fwd_offset", wts[0]); ev.set(cp, "bck_offset", wts[1])
                    args = [P[r].add(pos[r][1] * S + pos[r][0]), S, dst, w, w, h, filt, subs[r][0], 16, subs[r][1], 16, 0, cp]
                    if bd > 8:
                        args.append(bd)
                    ev.call("av1_convolve_2d_facade" if bd == 8 else "av1_highbd_convolve_2d_facade", *args)
                arrays["d%d" % k] = np.asarray(dst.buf, np.uint16)
                cases.append({"k": k, "bd": bd, "w": w, "h": h, "pos": pos, "subs": subs, "fx": fxi, "fy": fyi, "weights": wts})
                k += 1
    save("ref_eval_convolve_compound.npz", arrays, cases)


def gen_convolve_masked():
    """Masked compound prediction (COMPOUND_WEDGE / COMPOUND_DIFFWTD given the mask): both references through the compound
    convolve into 16-bit CONV_BUFs (do_average 0), then aom_lowbd_blend_a64_d16_mask_c / aom_highbd_blend_a64_d16_mask_c
    (aom_dsp/blend_a64_mask.c) -- what av1_make_masked_inter_predictor -> build_masked_compound_no_round do
    (av1/common/reconinter.c) -- with the mask at plane resolution and at 2x resolution (4:2:0 / 4:2:2 / 4:4:0 chroma)."""
    ev = evaluator(["av1/common/filter.h", "av1/common/convolve.h", "aom_dsp/aom_convolve.c", "av1/common/convolve.c", "aom_dsp/blend.h",
                    "aom_dsp/blend_a64_mask.c"])
    rng = np.random.default_rng(20261025)
    arrays, cases = {}, []
    S, ROWS = 96, 80
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        planes = []
        for r in range(2):
            base = rng.integers(0, mx + 1, (ROWS, S))
            base[:16] = np.where(rng.integers(0, 2, (16, S)) > 0, mx, 0)
            planes.append(base)
            arrays["p%d_%d" % (bd, r)] = base.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        P = [ev.array(pl.ravel(), ct) for pl in planes]
        for (w, h, subw, subh) in ((8, 8, 0, 0), (16, 16, 0, 0), (4, 8, 0, 0), (8, 8, 1, 1), (16, 8, 1, 0), (8, 16, 0, 1), (32, 16, 0, 0), (4, 4, 1, 1)):
            if bd == 12 and w * h > 128:
                continue
            fxi, fyi = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            pos = [(int(rng.integers(4, S - w - 5)), int(rng.integers(4, ROWS - h - 5))) for _ in range(2)]
            if k % 2 == 0:
                pos[1] = (pos[1][0], int(rng.integers(4, 10)))
            subs = [(int(rng.integers(0, 16)) * int(rng.integers(0, 2)), int(rng.integers(0, 16)) * int(rng.integers(0, 2))) for _ in range(2)]
            fp = [ev.call("av1_get_interp_filter_params_with_block_size", fxi, w), ev.call("av1_get_interp_filter_params_with_block_size", fyi, h)]
            filt = R.Ptr(fp, 0, ("ptr", ev.structs["InterpFilterParams"]))
            bufs = [ev.array([0] * (w * h), "uint16_t") for _ in range(2)]
            dst = ev.array([0] * (w * h), ct)
            cps = []
            for r in range(2):
                cpv = ev.call("get_conv_params_no_round", 0, 0, bufs[r], w, 1, bd)
                cp = R.Ptr([cpv], 0, cpv.st)
                cps.append(cp)
                args = [P[r].add(pos[r][1] * S + pos[r][0]), S, dst, w, w, h, filt, subs[r][0], 16, subs[r][1], 16, 0, cp]
                if bd > 8:
                    args.append(bd)
                ev.call("av1_convolve_2d_facade" if bd == 8 else "av1_highbd_convolve_2d_facade", *args)
            mw, mh = w << subw, h << subh
            ms = mw + 4
            mask = rng.integers(0, 65, (mh, ms))
            mask[0, :mw // 2] = 64
            mask[-1, :mw // 2] = 0
            M = ev.array(mask.ravel(), "uint8_t")
            args = [dst, w, bufs[0], w, bufs[1], w, M, ms, w, h, subw, subh, cps[0]]
            if bd > 8:
                args.append(bd)
            ev.call("aom_lowbd_blend_a64_d16_mask_c" if bd == 8 else "aom_highbd_blend_a64_d16_mask_c", *args)
            arrays["d%d" % k], arrays["m%d" % k] = np.asarray(dst.buf, np.uint16), mask.astype(np.uint8)
            cases.append({"k": k, "bd": bd, "w": w, "h": h, "subw": subw, "subh": subh, "pos": pos, "subs": subs, "fx": fxi, "fy": fyi, "mask_stride": ms})
            k += 1
    # COMPOUND_DIFFWTD: the mask comes from the two CONV_BUFs themselves (av1_build_compound_diffwtd_mask_d16_c,
    # av1/common/reconinter.c:296-328), then the same blend
    ev2 = evaluator([])
    for f in ["av1/common/filter.h", "av1/common/convolve.h", "aom_dsp/aom_convolve.c", "av1/common/convolve.c", "aom_dsp/blend.h",
              "aom_dsp/blend_a64_mask.c", "av1/common/mv.h", "aom_scale/yv12config.h", "av1/common/blockd.h", "av1/common/reconinter.h",
              "av1/common/reconinter.c"]:
        ev2.load("/root/reference/" + f)
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        P = [ev2.array(arrays["p%d_%d" % (bd, r)].astype(np.int64).ravel(), ct) for r in range(2)]
        for (w, h, mtype) in ((8, 8, 0), (16, 16, 1), (4, 16, 0), (32, 8, 1)):
            fxi, fyi = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            pos = [(int(rng.integers(4, S - w - 5)), int(rng.integers(4, ROWS - h - 5))) for _ in range(2)]
            if k % 2 == 0:
                pos[0] = (pos[0][0], int(rng.integers(4, 10)))
            subs = [(int(rng.integers(0, 16)) * int(rng.integers(0, 2)), int(rng.integers(0, 16)) * int(rng.integers(0, 2))) for _ in range(2)]
            fp = [ev2.call("av1_get_interp_filter_params_with_block_size", fxi, w), ev2.call("av1_get_interp_filter_params_with_block_size", fyi, h)]
            filt = R.Ptr(fp, 0, ("ptr", ev2.structs["InterpFilterParams"]))
            bufs = [ev2.array([0] * (w * h), "uint16_t") for _ in range(2)]
            dst = ev2.array([0] * (w * h), ct)
            cps = []
            for r in range(2):
                cpv = ev2.call("get_conv_params_no_round", 0, 0, bufs[r], w, 1, bd)
                cp = R.Ptr([cpv], 0, cpv.st)
                cps.append(cp)
                args = [P[r].add(pos[r][1] * S + pos[r][0]), S, dst, w, w, h, filt, subs[r][0], 16, subs[r][1], 16, 0, cp]
                if bd > 8:
                    args.append(bd)
                ev2.call("av1_convolve_2d_facade" if bd == 8 else "av1_highbd_convolve_2d_facade", *args)
            M = ev2.array([0] * (w * h), "uint8_t")
            ev2.call("av1_build_compound_diffwtd_mask_d16_c", M, mtype, bufs[0], w, bufs[1], w, h, w, cps[0], bd)
            args = [dst, w, bufs[0], w, bufs[1], w, M, w, w, h, 0, 0, cps[0]]
            if bd > 8:
                args.append(bd)
            ev2.call("aom_lowbd_blend_a64_d16_mask_c" if bd == 8 else "aom_highbd_blend_a64_d16_mask_c", *args)
            arrays["d%d" % k], arrays["m%d" % k] = np.asarray(dst.buf, np.uint16), np.asarray(M.buf, np.uint8).reshape(h, w)
            cases.append({"k": k, "bd": bd, "w": w, "h": h, "subw": 0, "subh": 0, "pos": pos, "subs": subs, "fx": fxi, "fy": fyi, "mask_stride": w,
                          "diffwtd": mtype + 1})
            k += 1
    save("ref_eval_convolve_masked.npz", arrays, cases)


def gen_obmc_blend():
    """The OBMC blends: aom_[highbd_]blend_a64_vmask_c / _hmask_c (aom_dsp/blend_a64_vmask.c, blend_a64_hmask.c) in place on the
    prediction, with av1_get_obmc_mask's tables (av1/common/reconinter.c:744-777) -- build_obmc_inter_pred_above / _left (:844-920)."""
    ev = evaluator([])
    for f in ["aom_dsp/blend.h", "aom_dsp/blend_a64_vmask.c", "aom_dsp/blend_a64_hmask.c", "av1/common/mv.h", "aom_scale/yv12config.h",
              "av1/common/blockd.h", "av1/common/reconinter.h", "av1/common/reconinter.c"]:
        ev.load("/root/reference/" + f)
    rng = np.random.default_rng(20261027)
    arrays, cases = {}, []
    for n in (1, 2, 4, 8, 16, 32, 64):
        m = ev.call("av1_get_obmc_mask", n)
        arrays["obmc_mask_%d" % n] = np.asarray([m.buf[m.off + i] for i in range(n)], np.uint8)
    S, ROWS = 80, 72
    k = 0
    for bd in (8, 10, 12):
        mx = (1 << bd) - 1
        pred, adj = rng.integers(0, mx + 1, (ROWS, S)), rng.integers(0, mx + 1, (ROWS, S))
        pred[:8], adj[:8] = mx, 0
        arrays["pred%d" % bd], arrays["adj%d" % bd] = pred.astype(np.uint16), adj.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        for (w, h, vertical) in ((16, 8, 1), (8, 4, 1), (32, 32, 1), (4, 2, 1), (8, 16, 0), (4, 8, 0), (32, 64, 0), (2, 4, 0), (64, 1, 1), (16, 16, 0)):
            x, y = int(rng.integers(0, S - w)), int(rng.integers(0, ROWS - h))
            D = ev.array(pred.ravel(), ct)
            A = ev.array(adj.ravel(), ct)
            mask = ev.call("av1_get_obmc_mask", h if vertical else w)
            fn = "aom_%sblend_a64_%smask_c" % ("highbd_" if bd > 8 else "", "v" if vertical else "h")
            args = [D.add(y * S + x), S, D.add(y * S + x), S, A.add(y * S + x), S, mask, w, h]
            if bd > 8:
                args.append(bd)
            ev.call(fn, *args)
            arrays["o%d" % k] = np.asarray(D.buf, np.uint16).reshape(ROWS, S)[y:y + h, x:x + w].copy()
            cases.append({"k": k, "bd": bd, "x": x, "y": y, "w": w, "h": h, "vertical": vertical})
            k += 1
    save("ref_eval_obmc_blend.npz", arrays, cases)

if __name__ == "__main__":
    for w in sys.argv[1:] or ["txfm2d", "tables", "cdef_fb", "compound", "convolve", "rdhelp", "cdef_search", "lrstats", "convolve_compound", "convolve_masked", "obmc_blend"]:
        t = time.time()
        globals()["gen_" + w]()
        print("  (%s: %.1f s)" % (w, time.time() - t))
