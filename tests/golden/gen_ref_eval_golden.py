#!/usr/bin/env python3
"""Golden vectors obtained by INTERPRETING THE REFERENCE'S OWN C FUNCTIONS (build container only).

`ref_c_eval.CEval` loads the reference's sources where they lie under /root/reference and evaluates the
named functions with C integer semantics; this script feeds them seeded inputs and stores inputs + outputs
as committed fixtures (data only, no reference text):

  ref_eval_quant.npz    aom_[highbd_]quantize_b{,_32x32,_64x64}[_adaptive]_c        (aom_dsp/quantize.c)
  ref_eval_lpf.npz      aom_[highbd_]lpf_{horizontal,vertical}_{4,6,8,14}_c         (aom_dsp/loopfilter.c)
  ref_eval_cdef.npz     cdef_find_dir_c, cdef_filter_{8,16}_{0..3}_c                (av1/common/cdef_block.c)
  ref_eval_sadvar.npz   aom_[highbd_]sad*_c (+skip, x4d, avg), aom_[highbd_N_]variance*_c,
                        aom_[highbd_N_]sub_pixel_variance*_c, aom_[highbd_]subtract_block_c
  ref_eval_mcomp.npz    full_pixel_diamond / av1_full_pixel_search / full_pixel_exhaustive /
                        av1_find_best_sub_pixel_tree_pruned_more                    (av1/encoder/mcomp.c; gen_mcomp)

Usage: python tests/golden/gen_ref_eval_golden.py [quant lpf cdef sadvar mcomp ...]   (default: all)
The configuration values passed to the miniature preprocessor are those of the reference's `generic` target
(SURVEY.md 8c): CONFIG_AV1_HIGHBITDEPTH=1, CONFIG_REALTIME_ONLY=0.
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

REF = "/root/reference/"
CONFIG = {"CONFIG_AV1_HIGHBITDEPTH": 1, "CONFIG_REALTIME_ONLY": 0}
COMMON = ["aom_ports/mem.h", "aom_ports/bitops.h", "aom_dsp/aom_dsp_common.h", "av1/common/enums.h", "aom_dsp/aom_filter.h"]


def evaluator(files):
    ev = R.CEval(CONFIG)
    for f in COMMON:
        ev.load(REF + f)
    # the byte-pointer encoding of high-bit-depth planes (aom_ports/mem.h:79-80) is an address trick; in the
    # evaluator's (buffer, element) pointer model it is the identity
    ev.define("CONVERT_TO_SHORTPTR", "(x)", ["x"])
    ev.define("CONVERT_TO_BYTEPTR", "(x)", ["x"])
    for f in files:
        ev.load(REF + f)
    return ev


def save(name, arrays, cases):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, cases=np.frombuffer(json.dumps(cases).encode(), np.uint8), **arrays)
    print("%s: %d cases, %d arrays, %.1f KB" % (name, len(cases), len(arrays), os.path.getsize(path) / 1024))


# ------------------------------------------------------------------------------------------------- quantize

def gen_quant():
    import pyoracle as orc      # only for the (separately pinned) quantiser tables and scan orders used as INPUTS
    ev = evaluator(["aom_dsp/quantize.h", "aom_dsp/quantize.c"])
    rng = np.random.default_rng(20261002)
    arrays, cases = {}, []
    # (tx_size id, n, wrapper suffix): aom_quantize_b for <= 16x16, _32x32 (log_scale 1), _64x64 (log_scale 2)
    sizes = [(0, 16, ""), (1, 64, ""), (2, 256, ""), (3, 1024, "_32x32"), (4, 1024, "_64x64"), (7, 128, "")]
    k = 0
    for tx_size, n, suf in sizes:
        for tx_type in ((0, 10, 11) if n <= 256 else (0,)):      # default zig-zag, row (V_DCT) and column (H_DCT) scans
            if n > 64 and tx_type != 0 and tx_size != 2:
                continue
            scan, iscan = orc.get_scan(tx_size, tx_type)
            for hbd in (0, 1):
                for adaptive in (0, 1):
                    for qindex in ((0, 20, 100, 200, 255) if n <= 256 else (20, 130, 255)):
                        bd = 10 if hbd else 8
                        q = orc.build_quantizer_y(bd, qindex)
                        kinds = ("zero", "dc", "neg_dc", "const16", "random", "random_small") if qindex in (20, 100, 130) else ("random",)
                        for kind in kinds:
                            if n == 1024 and kind in ("zero", "const16") and adaptive:
                                continue
                            c = np.zeros(n, np.int64)
                            span = (1 << (bd + 7)) - 1
                            if kind == "dc":
                                c[0] = rng.integers(1, span)
                            elif kind == "neg_dc":
                                c[0] = -8191 if not hbd else -span
                            elif kind == "const16":
                                c[:] = 16
                            elif kind == "random":
                                c = rng.integers(-span, span + 1, n)
                                c[rng.random(n) < 0.5] //= 64            # a mix of large and near-threshold values
                            elif kind == "random_small":
                                c = rng.integers(-3 * int(q["zbin"][1]), 3 * int(q["zbin"][1]) + 1, n)
                            fn = "aom_%squantize_b%s%s_c" % ("highbd_" if hbd else "", suf, "_adaptive" if adaptive else "")
                            cp = ev.array(c, "int32_t")
                            qc, dq, eob = ev.array([0x55] * n, "int32_t"), ev.array([0x55] * n, "int32_t"), ev.array([77], "uint16_t")
                            t = {m: ev.array(q[m], "int16_t") for m in q}
                            ev.call(fn, cp, n, t["zbin"], t["round"], t["quant"], t["quant_shift"], qc, dq, t["dequant"], eob,
                                    ev.array(scan, "int16_t"), ev.array(iscan, "int16_t"))
                            arrays["c%d" % k] = np.asarray(c, np.int32)
                            arrays["q%d" % k] = np.asarray(qc.buf, np.int32)
                            arrays["d%d" % k] = np.asarray(dq.buf, np.int32)
                            cases.append({"fn": fn, "tx_size": tx_size, "tx_type": tx_type, "n": n, "hbd": hbd, "adaptive": adaptive,
                                          "qindex": qindex, "bd": bd, "kind": kind, "eob": int(eob.buf[0]),
                                          "log_scale": {"": 0, "_32x32": 1, "_64x64": 2}[suf],
                                          "tables": {m: [int(q[m][0]), int(q[m][1])] for m in q}})
                            k += 1
    # av1_block_error_c / av1_highbd_block_error_c (av1/encoder/rdopt.c:635-682) on (coeff, dqcoeff) pairs of the cases above:
    # the transform-domain distortion the fused transform + quantise + distortion entry point returns
    ev2 = evaluator(["aom_dsp/quantize.h", "av1/encoder/rdopt.c"])
    berr = []
    for kk in range(0, k, 7):
        c, dq = arrays["c%d" % kk], arrays["d%d" % kk]
        ssz = ev2.array([0], "int64_t")
        e8 = ev2.call("av1_block_error_c", ev2.array(c, "int32_t"), ev2.array(dq, "int32_t"), c.size, ssz)
        row = [kk, e8, ssz.buf[0]]
        for bd in (8, 10, 12):
            e = ev2.call("av1_highbd_block_error_c", ev2.array(c, "int32_t"), ev2.array(dq, "int32_t"), c.size, ssz, bd)
            row += [e, ssz.buf[0]]
        berr.append(row)
    arrays["block_error"] = np.asarray(berr, np.int64)
    # av1_quantize_fp{,_32x32,_64x64}_c and av1_highbd_quantize_fp_c (av1/encoder/av1_quantize.c) on the coefficient vectors of
    # every 5th case above; round_fp / quant_fp as av1_build_quantizer derives them from the dequantiser
    ev3 = evaluator(["aom_dsp/quantize.h", "av1/common/quant_common.h", "av1/encoder/av1_quantize.h", "av1/encoder/av1_quantize.c"])
    fp_rows = []
    for kk in range(0, k, 5):
        c0 = cases[kk]
        n, ls, hbd = c0["n"], c0["log_scale"], c0["hbd"]
        scan, iscan = orc.get_scan(c0["tx_size"], c0["tx_type"])
        dq2 = c0["tables"]["dequant"]
        rfp, qfp = [(64 * d) >> 7 for d in dq2], [(1 << 16) // d for d in dq2]
        c = arrays["c%d" % kk]
        qc, dqc, eob = ev3.array([0x55] * n, "int32_t"), ev3.array([0x55] * n, "int32_t"), ev3.array([77], "uint16_t")
        t = {m: ev3.array(c0["tables"][m], "int16_t") for m in c0["tables"]}
        args = [ev3.array(c, "int32_t"), n, t["zbin"], ev3.array(rfp, "int16_t"), ev3.array(qfp, "int16_t"), t["quant_shift"], qc, dqc,
                t["dequant"], eob, ev3.array(scan, "int16_t"), ev3.array(iscan, "int16_t")]
        if hbd:
            ev3.call("av1_highbd_quantize_fp_c", *args, ls)
        else:
            ev3.call("av1_quantize_fp%s_c" % {0: "", 1: "_32x32", 2: "_64x64"}[ls], *args)
        arrays["fq%d" % kk], arrays["fd%d" % kk] = np.asarray(qc.buf, np.int32), np.asarray(dqc.buf, np.int32)
        fp_rows.append([kk, int(eob.buf[0]), rfp[0], rfp[1], qfp[0], qfp[1]])
    arrays["quantize_fp"] = np.asarray(fp_rows, np.int64)
    save("ref_eval_quant.npz", arrays, cases)


# --------------------------------------------------------------------------------------------------- deblock

def lpf_thresholds(level, sharp):
    # inputs only; the formula (av1_loopfilter.c:47-66,118-120) is checked by tests/test_oracle_lpf.py
    shift = (sharp > 0) + (sharp > 4)
    lim = level >> shift
    if sharp > 0:
        lim = min(lim, 9 - sharp)
    lim = max(lim, 1)
    return 2 * (level + 2) + lim, lim, level >> 4


def gen_lpf():
    ev = evaluator(["aom_dsp/loopfilter.c"])
    rng = np.random.default_rng(20261003)
    arrays, cases = {}, []
    k = 0
    N = 24
    for bd in (8, 10, 12):
        for length in (4, 6, 8, 14):
            for vertical in (0, 1):
                for level in (1, 8, 24, 40, 63):
                    for sharp in (0, 5):
                        for kind in ("flat", "step", "ramp", "noise"):
                            if sharp and kind not in ("step", "noise"):
                                continue
                            mx = (1 << bd) - 1
                            sc = 1 << (bd - 8)
                            base = int(rng.integers(40, 200)) * sc
                            yy, xx = np.mgrid[0:N, 0:N]
                            a = xx if vertical else yy
                            if kind == "flat":
                                p = np.full((N, N), base) + rng.integers(-1, 2, (N, N)) * sc
                            elif kind == "step":
                                p = base + np.where(a >= N // 2, int(rng.integers(2, 30)) * sc, 0) + rng.integers(-2, 3, (N, N)) * sc
                            elif kind == "ramp":
                                p = base + (a - N // 2) * int(rng.integers(1, 6)) * sc + rng.integers(-1, 2, (N, N)) * sc
                            else:
                                p = base + rng.integers(-level // 2 - 2, level // 2 + 3, (N, N)) * sc
                            p = np.clip(p, 0, mx)
                            mbl, lim, hev = lpf_thresholds(level, sharp)
                            buf = ev.array(p.ravel(), "uint8_t" if bd == 8 else "uint16_t")
                            fn = "aom_%slpf_%s_%d_c" % ("highbd_" if bd > 8 else "", "vertical" if vertical else "horizontal", length)
                            y0, x0 = (8, N // 2) if vertical else (N // 2, 8)      # 4-pixel unit starting at (y0, x0)
                            args = [buf.add(y0 * N + x0), N, ev.array([mbl], "uint8_t"), ev.array([lim], "uint8_t"), ev.array([hev], "uint8_t")]
                            if bd > 8:
                                args.append(bd)
                            ev.call(fn, *args)
                            arrays["i%d" % k] = p.astype(np.uint16)
                            arrays["o%d" % k] = np.asarray(buf.buf, np.uint16).reshape(N, N)
                            cases.append({"fn": fn, "bd": bd, "len": length, "vertical": vertical, "level": level, "sharp": sharp,
                                          "blimit": mbl, "limit": lim, "thresh": hev, "y": y0, "x": x0, "kind": kind})
                            k += 1
    save("ref_eval_lpf.npz", arrays, cases)


# ------------------------------------------------------------------------------------------------------ CDEF

def gen_cdef():
    ev = evaluator(["av1/common/cdef_block.h", "av1/common/cdef.h", "av1/common/cdef_block.c"])
    bstride = ev.interp.ev(R.Parser(ev.pp.expand(R.tokenize("CDEF_BSTRIDE")), ev.typedefs).expr())[0]
    very_large = 0x4000
    rng = np.random.default_rng(20261004)
    arrays, cases = {}, []
    k = 0
    # -- direction search
    dirs_in, dirs_out = [], []
    for bd in (8, 10, 12):
        for kind in range(12):
            yy, xx = np.mgrid[0:8, 0:8]
            mx = (1 << bd) - 1
            if kind < 8:      # an edge along one of the 8 directions + noise
                ang = [(1, -1), (1, -2), (1, 0), (1, 2), (1, 1), (2, 1), (0, 1), (-2, 1)][kind]   # normal-ish vectors
                v = (xx * ang[0] + yy * ang[1])
                img = (v > v.mean()) * (mx // 2) + rng.integers(0, mx // 8 + 1, (8, 8)) + mx // 8
            elif kind == 8:
                img = np.full((8, 8), mx // 3)
            else:
                img = rng.integers(0, mx + 1, (8, 8))
            img = np.clip(img, 0, mx)
            var = ev.array([0], "int32_t")
            d = ev.call("cdef_find_dir_c", ev.array(img.ravel(), "uint16_t"), 8, var, bd - 8)
            dirs_in.append(img.astype(np.uint16))
            dirs_out.append((bd, d, var.buf[0]))
    arrays["find_dir_in"] = np.stack(dirs_in)
    arrays["find_dir_out"] = np.asarray(dirs_out, np.int64)
    # -- block filter: (8|16)-bit destination, the four enable combinations, 8x8 / 4x4 / 4x8 / 8x4 blocks
    for bd in (8, 10, 12):
        cs = bd - 8
        for (bw, bh) in ((8, 8), (4, 4), (4, 8), (8, 4)):
            for variant in range(4):
                for trial in range(4 if (bw, bh) == (8, 8) else 2):
                    mx = (1 << bd) - 1
                    tile = np.clip(rng.integers(0, mx // 3 + 1) + rng.integers(-(8 << cs), (8 << cs) + 1, (bh + 6, bstride))
                                   + (np.arange(bstride)[None, :] % 16) * (2 << cs), 0, mx).astype(np.int64)
                    if trial % 2 == 1:      # frame edge: CDEF_VERY_LARGE outside
                        tile[:3 if rng.random() < 0.5 else 0, :] = very_large
                        tile[:, :8] = very_large
                        if rng.random() < 0.5:
                            tile[bh + 3:, :] = very_large
                    pri = int(rng.integers(1, 16)) << cs if variant in (0, 1) else 0
                    sec = int(rng.choice([1, 2, 4])) << cs if variant in (0, 2) else 0
                    d = int(rng.integers(0, 8))
                    damping = int(rng.integers(3, 7)) + cs
                    use16 = bd > 8 or trial >= 2
                    fn = "cdef_filter_%d_%d_c" % (16 if use16 else 8, variant)
                    dst = ev.array([0] * (bw * bh), "uint16_t" if use16 else "uint8_t")
                    inp = ev.array(tile.ravel(), "uint16_t")
                    ev.call(fn, dst, bw, inp.add(3 * bstride + 8), pri, sec, d, damping, damping - (variant == 0 and 0), cs, bw, bh)
                    arrays["t%d" % k] = tile.astype(np.uint16)
                    arrays["f%d" % k] = np.asarray(dst.buf, np.uint16).reshape(bh, bw)
                    cases.append({"fn": fn, "bd": bd, "bw": bw, "bh": bh, "pri": pri, "sec": sec, "dir": d, "pri_damping": damping,
                                  "sec_damping": damping, "coeff_shift": cs, "variant": variant, "bstride": int(bstride)})
                    k += 1
    save("ref_eval_cdef.npz", arrays, cases)


# ------------------------------------------------------------------------------------------ SAD and variance

SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64),
         (64, 128), (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def gen_sadvar():
    ev = evaluator(["aom_dsp/variance.h", "aom_dsp/sad.c", "aom_dsp/variance.c", "aom_dsp/subtract.c"])
    rng = np.random.default_rng(20261005)
    rows = []
    planes = {}
    for bd in (8, 10, 12):
        S = 160
        mx = (1 << bd) - 1
        a = np.clip(rng.integers(0, mx + 1, (150, S)), 0, mx)
        b = np.clip(a + rng.integers(-(20 << (bd - 8)), (20 << (bd - 8)) + 1, a.shape), 0, mx)
        planes["a%d" % bd], planes["b%d" % bd] = a.astype(np.uint16), b.astype(np.uint16)
        ct = "uint8_t" if bd == 8 else "uint16_t"
        pa, pb = ev.array(a.ravel(), ct), ev.array(b.ravel(), ct)
        for (w, h) in SIZES:
            if max(w, h) > 64 and bd == 12:
                continue
            big = w * h >= 64 * 64
            ox, oy = int(rng.integers(0, S - w - 1)), int(rng.integers(0, 150 - h - 1))
            rx, ry = int(rng.integers(0, S - w - 1)), int(rng.integers(0, 150 - h - 1))
            A, Bp = pa.add(oy * S + ox), pb.add(ry * S + rx)
            hb = "highbd_" if bd > 8 else ""
            sad = ev.call("aom_%ssad%dx%d_c" % (hb, w, h), A, S, Bp, S)
            skip = ev.call("aom_%ssad_skip_%dx%d_c" % (hb, w, h), A, S, Bp, S)
            rec = {"bd": bd, "w": w, "h": h, "ox": ox, "oy": oy, "rx": rx, "ry": ry, "sad": sad, "sad_skip": skip}
            if not big:
                offs = [(int(rng.integers(0, S - w - 1)), int(rng.integers(0, 150 - h - 1))) for _ in range(4)]
                ptrs = R.Ptr([pb.add(y * S + x) for (x, y) in offs], 0, ("ptr", ev.ctype(ct)))
                out = ev.array([0] * 4, "uint32_t")
                ev.call("aom_%ssad%dx%dx4d_c" % (hb, w, h), A, S, ptrs, S, out)
                rec["x4d_offs"], rec["x4d"] = offs, list(out.buf)
            sse = ev.array([0], "uint32_t")
            vfn = "aom_variance%dx%d_c" % (w, h) if bd == 8 else "aom_highbd_%d_variance%dx%d_c" % (bd, w, h)
            rec["var"] = ev.call(vfn, A, S, Bp, S, sse)
            rec["sse"] = sse.buf[0]
            if not big:
                sub = []
                for (xo, yo) in ((0, 0), (4, 0), (0, 4), (3, 5), (7, 7), (1, 6)):
                    sfn = ("aom_sub_pixel_variance%dx%d_c" % (w, h)) if bd == 8 else ("aom_highbd_%d_sub_pixel_variance%dx%d_c" % (bd, w, h))
                    v = ev.call(sfn, A, S, xo, yo, Bp, S, sse)
                    sub.append([xo, yo, v, sse.buf[0]])
                rec["subpel"] = sub
            if (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (16, 8)):
                diff = ev.array([0] * (w * h), "int16_t")
                ev.call("aom_%ssubtract_block_c" % hb, h, w, diff, w, A, S, Bp, S)
                rec["subtract_sum"] = int(np.sum(np.asarray(diff.buf, np.int64) * (np.arange(w * h) % 251 + 1)))
                # compound average against a second predictor (A4)
                sp = np.clip(rng.integers(0, mx + 1, w * h), 0, mx)
                rec["second_pred_seed"] = int(rng.integers(0, 1 << 30))
                sp = np.random.default_rng(rec["second_pred_seed"]).integers(0, mx + 1, w * h)
                rec["sad_avg"] = ev.call("aom_%ssad%dx%d_avg_c" % (hb, w, h), A, S, Bp, S, ev.array(sp, ct))
            rows.append(rec)
    # the other forms of variance() (variance.c:200-262), 8-bit: own generator so that the rows above keep their values
    rng2 = np.random.default_rng(20261011)
    a, b = planes["a8"].astype(np.int64), planes["b8"].astype(np.int64)
    S = a.shape[1]
    pa, pb = ev.array(a.ravel(), "uint8_t"), ev.array(b.ravel(), "uint8_t")
    extras = []
    for trial in range(6):
        ox, oy, rx, ry = (int(rng2.integers(0, 100)) for _ in range(4))
        A, Bp = pa.add(oy * S + ox), pb.add(ry * S + rx)
        rec = {"extra": 1, "ox": ox, "oy": oy, "rx": rx, "ry": ry}
        sse, sm = ev.array([0], "unsigned int"), ev.array([0], "int")
        for (w, h) in ((16, 16), (16, 8), (8, 16), (8, 8)):
            rec["mse%dx%d" % (w, h)] = [ev.call("aom_mse%dx%d_c" % (w, h), A, S, Bp, S, sse), sse.buf[0]]
        for n in (8, 16):
            ev.call("aom_get%dx%dvar_c" % (n, n), A, S, Bp, S, sse, sm)
            rec["get%dvar" % n] = [sse.buf[0], sm.buf[0]]
        s8, m8, v8 = ev.array([0] * 4, "uint32_t"), ev.array([0] * 4, "int"), ev.array([0] * 4, "uint32_t")
        ts, tm = ev.array([1000 + trial], "unsigned int"), ev.array([-50 * trial], "int")       # the totals accumulate
        ev.call("aom_get_var_sse_sum_8x8_quad_c", A, S, Bp, S, s8, m8, ts, tm, v8)
        rec["quad"] = [list(s8.buf), list(m8.buf), ts.buf[0], tm.buf[0], list(v8.buf), 1000 + trial, -50 * trial]
        s16, v16 = ev.array([0] * 2, "uint32_t"), ev.array([0] * 2, "uint32_t")
        ts, tm = ev.array([7], "unsigned int"), ev.array([3], "int")
        ev.call("aom_get_var_sse_sum_16x16_dual_c", A, S, Bp, S, s16, ts, tm, v16)
        rec["dual"] = [list(s16.buf), ts.buf[0], tm.buf[0], list(v16.buf), 7, 3]
        extras.append(rec)
    save("ref_eval_sadvar.npz", planes, rows + extras)


if __name__ == "__main__":
    which = sys.argv[1:] or ["quant", "lpf", "cdef", "sadvar", "mcomp"]
    for w in which:
        t = time.time()
        if w == "mcomp":
            import gen_ref_eval_mcomp
            gen_ref_eval_mcomp.main()
        else:
            globals()["gen_" + w]()
        print("  (%s: %.1f s)" % (w, time.time() - t))
