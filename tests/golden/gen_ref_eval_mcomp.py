#!/usr/bin/env python3
"""Golden vectors for the motion search, obtained by interpreting av1/encoder/mcomp.c itself (build container only).

The reference's search functions are driven exactly as SURVEY.md 8(c) describes for a compiled harness, but through
tests/golden/ref_c_eval.py: a FULLPEL_MOTION_SEARCH_PARAMS / SUBPEL_MOTION_SEARCH_PARAMS object is filled field by
field, `search_sites` comes from the reference's own av1_init_motion_compensation[...] builders, the vtable entries
are the reference's aom_sad* / aom_variance* / aom_sub_pixel_variance* functions (10-bit: the `_bits10` wrappers of
av1/encoder/encoder_utils.h), and av1_full_pixel_search / full_pixel_diamond / full_pixel_exhaustive /
av1_find_best_sub_pixel_tree_pruned_more are interpreted statement by statement.

Two adaptations to the evaluator's memory model (both value-preserving): the byte-pointer encoding macros are the
identity, and MARK_MV_INVALID writes row = col = INVALID_MV_ROW_COL instead of storing 0x80008000 through a union
(av1/common/mv.h:26-34: the same bits on a little-endian target).

Output: tests/golden/ref_eval_mcomp.npz (planes, per-case block descriptors, expected outputs).
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402

REF = "/root/reference/"
W = H = 96
BORDER = 48
INT_MAX = 2147483647

METHODS = ["DIAMOND", "NSTEP", "NSTEP_8PT", "CLAMPED_DIAMOND", "HEX", "BIGDIA", "SQUARE", "FAST_HEX", "FAST_DIAMOND",
           "FAST_BIGDIA", "VFAST_DIAMOND"]
COST_TYPES = {"ENTROPY": 0, "L1_LOWRES": 1, "L1_MIDRES": 2, "L1_HDRES": 3, "NONE": 4}
BSIZE = {(4, 4): "BLOCK_4X4", (8, 8): "BLOCK_8X8", (16, 16): "BLOCK_16X16", (32, 16): "BLOCK_32X16", (16, 8): "BLOCK_16X8",
         (32, 32): "BLOCK_32X32", (8, 16): "BLOCK_8X16"}


def make_evaluator(with_compound=False):
    """with_compound: also aom_dsp/blend.h (before the files that use AOM_BLEND_A64) and aom_dsp/sad_av1.c -- the masked / OBMC vtable
    members of gen_ref_eval_compound_search.py; the default is what produced ref_eval_mcomp.npz."""
    ev = R.CEval({"CONFIG_AV1_HIGHBITDEPTH": 1, "CONFIG_REALTIME_ONLY": 0})
    for f in ["aom_ports/mem.h", "aom_ports/bitops.h", "aom_dsp/aom_dsp_common.h", "av1/common/enums.h", "aom_dsp/aom_filter.h",
              "av1/common/filter.h", "aom_dsp/variance.h", "av1/common/common_data.h"]:
        ev.load(REF + f)
    ev.define("CONVERT_TO_SHORTPTR", "(x)", ["x"])
    ev.define("CONVERT_TO_BYTEPTR", "(x)", ["x"])
    for f in ["av1/common/mv.h"]:
        ev.load(REF + f)
    ev.define("MARK_MV_INVALID", "do { (mv)->row = INVALID_MV_ROW_COL; (mv)->col = INVALID_MV_ROW_COL; } while (0)", ["mv"])
    files = ["av1/common/entropymv.h", "aom_scale/yv12config.h", "av1/common/blockd.h", "av1/encoder/speed_features.h", "av1/encoder/cost.h", "av1/encoder/rd.h", "av1/encoder/encodemv.h", "av1/encoder/mcomp_structs.h", "av1/encoder/mcomp.h",
             "av1/common/scale.h", "aom_dsp/sad.c", "aom_dsp/variance.c", "av1/encoder/encoder_utils.h", "aom_dsp/aom_convolve.c",
             "av1/encoder/reconinter_enc.c", "av1/encoder/mcomp.c"]
    if with_compound:
        files.insert(files.index("aom_dsp/sad.c"), "aom_dsp/blend.h")
        files.insert(files.index("aom_dsp/variance.c"), "aom_dsp/sad_av1.c")
    for f in files:
        ev.load(REF + f)
    # aom_convolve8_* recover the kernel table and the phase from the kernel POINTER (get_filter_base masks the address
    # with ~0xFF, relying on the table's 256-byte alignment; get_filter_offset is a pointer difference in kernels).  In
    # the evaluator's (buffer, element) pointer model the same two values are the buffer start and element / 8.
    for fn in ("get_filter_base", "get_filter_offset"):
        ev.funcs.pop(fn)
    ev.interp.pycalls["get_filter_base"] = lambda it, a: (R.Ptr(a[0][0].buf, 0, a[0][0].t, (8,)), R.PTR)
    ev.interp.pycalls["get_filter_offset"] = lambda it, a: (a[0][0].off // 8, R.I32)
    # MACROBLOCKD / MB_MODE_INFO contain unions and dozens of unrelated members, so the evaluator knows them only as
    # opaque parameter types.  The sub-pel entry points read exactly three things through `xd` (mcomp.c:2864-2867):
    # xd->mi[0]->use_intrabc and xd->block_ref_scale_factors[0] (a real `struct scale_factors`, av1/common/scale.h).
    # SURVEY 8(c)'s compiled harness passes zero-initialised objects for them; here the opaque types get views
    # with just those members.
    mbmi = ev.structs["<opaque>MB_MODE_INFO"]
    mbmi.fields = [("use_intrabc", R.U8)]
    xd = ev.structs["<opaque>MACROBLOCKD"]
    # upsampled_pref_error additionally reads xd->mi_row / mi_col (passed on, unused for an unscaled reference), xd->bd and
    # xd->cur_buf->flags (is_cur_buf_hbd, blockd.h:936-943)
    yv12 = ev.structs.setdefault("<opaque>YV12_BUFFER_CONFIG", R.StructType("YV12_BUFFER_CONFIG"))
    yv12.fields = [("flags", R.U32)]
    xd.fields = [("mi", ("ptr", ("ptr", mbmi))), ("block_ref_scale_factors", ("arr", ("ptr", ev.structs["scale_factors"]), 2)),
                 ("mi_row", R.I32), ("mi_col", R.I32), ("bd", R.I32), ("cur_buf", ("ptr", yv12))]
    return ev


def make_xd(ev, bd=8):
    xd = ev.interp.alloc(ev.structs["<opaque>MACROBLOCKD"], True)
    cur = ev.interp.alloc(ev.structs["<opaque>YV12_BUFFER_CONFIG"], True)
    ev.set(cur, "flags", 8 if bd > 8 else 0)              # YV12_FLAG_HIGHBITDEPTH (aom_scale/yv12config.h:127)
    ev.set(xd, "cur_buf", cur); ev.set(xd, "bd", bd)
    mi = ev.interp.alloc(ev.structs["<opaque>MB_MODE_INFO"], True)
    mip = ev.interp.alloc(("ptr", ev.structs["<opaque>MB_MODE_INFO"]), True)
    mip.store(mi, R.PTR)
    ev.set(xd, "mi", mip)
    sf = ev.interp.alloc(ev.structs["scale_factors"], True)
    no_scale = ev.interp.ev(R.Parser(ev.pp.expand(R.tokenize("REF_NO_SCALE")), ev.typedefs).expr())[0]
    ev.set(sf, "x_scale_fp", no_scale); ev.set(sf, "y_scale_fp", no_scale)      # what av1_setup_scale_factors_for_frame gives for equal sizes
    ev.set(xd, "block_ref_scale_factors[0]", sf)
    return xd


def synth_planes(bd, seed):
    """Smooth random content; src = ref displaced by a per-quadrant shift + noise.  Visible W x H, replicated border."""
    rng = np.random.default_rng(seed)
    big = rng.integers(0, 1 << bd, (H + 64, W + 64)).astype(np.float64)
    for _ in range(3):                                    # separable box blur (radius 2), numpy only
        c = np.cumsum(np.pad(big, ((3, 2), (0, 0)), mode="edge"), axis=0)
        big = (c[5:] - c[:-5]) / 5.0
        c = np.cumsum(np.pad(big, ((0, 0), (3, 2)), mode="edge"), axis=1)
        big = (c[:, 5:] - c[:, :-5]) / 5.0
    big = (big - big.min()) / (big.max() - big.min()) * ((1 << bd) - 1)
    ref = big[32:32 + H, 32:32 + W]
    src = np.empty_like(ref)
    shifts = {(0, 0): (3, 5), (0, 1): (-6, 2), (1, 0): (1, -9), (1, 1): (-2, -3)}       # (row, col): src(y,x) = ref(y+r, x+c)
    for (qy, qx), (r, c) in shifts.items():
        ys, xs = slice(qy * H // 2, (qy + 1) * H // 2), slice(qx * W // 2, (qx + 1) * W // 2)
        src[ys, xs] = big[32 + r + qy * H // 2:32 + r + (qy + 1) * H // 2, 32 + c + qx * W // 2:32 + c + (qx + 1) * W // 2]
    src = src + rng.normal(0, (1 << bd) / 128.0, src.shape)
    dt = np.uint8 if bd == 8 else np.uint16
    clip = lambda a: np.clip(np.rint(a), 0, (1 << bd) - 1).astype(dt)
    pad = lambda a: np.pad(clip(a), BORDER, mode="edge")
    return pad(src), pad(ref)


def synth_mv_costs(seed):
    """Entropy-style cost tables (inputs): joint[4] and two component tables indexed -MV_MAX..MV_MAX."""
    rng = np.random.default_rng(seed)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    comp = []
    for k in range(2):
        comp.append((120 + 50 * k + bits * (300 + 40 * k) + ((v * 2654435761 >> 9) & 63) + (np.arange(v.size) < mv_max) * 37).astype(np.int32))
    joint = np.asarray([180, 700 + int(rng.integers(0, 64)), 690 + int(rng.integers(0, 64)), 1100], np.int32)
    return joint, comp[0], comp[1]


class Harness:
    def __init__(self, ev, bd, src_b, ref_b, mvcosts):
        self.ev, self.bd = ev, bd
        self.S = src_b.shape[1]
        ct = "uint8_t" if bd == 8 else "uint16_t"
        self.srcp, self.refp = ev.array(src_b.ravel(), ct), ev.array(ref_b.ravel(), ct)
        self.site_cfgs = {}
        joint, c0, c1 = mvcosts
        self.mv_max = (c0.size - 1) // 2
        self.joint = ev.array(joint, "int")
        self.comp = [ev.array(c0, "int"), ev.array(c1, "int")]

    def const(self, name):
        return self.ev.globs[name].buf[0]

    def buf2d(self, p, y, x):
        ev = self.ev
        b = ev.new("buf_2d")
        o = (BORDER + y) * self.S + BORDER + x
        ev.set(b, "buf", p.add(o)); ev.set(b, "buf0", p.add(o))
        ev.set(b, "width", W); ev.set(b, "height", H); ev.set(b, "stride", self.S)
        return b

    def sites(self, method):
        if method not in self.site_cfgs:
            ev = self.ev
            cfg = ev.new("search_site_config")
            # av1_init_motion_compensation[] (mcomp.h:186-195) and the level rule of its callers (e.g. encoder.c:
            # level = search method is NSTEP_8PT or CLAMPED_DIAMOND)
            if method == "NSTEP_FPF":        # the first-pass table (firstpass.c:261-299 uses it with search_method NSTEP)
                ev.interp.call("av1_init_motion_fpf", [(cfg, R.PTR), (self.S, R.I32)])
            else:
                lookup = ev.global_values("search_method_lookup")
                table = ev.globs["av1_init_motion_compensation"]
                fn = table.buf[lookup[self.const(method)]]
                level = int(method in ("NSTEP_8PT", "CLAMPED_DIAMOND"))
                ev.interp.call(fn.name, [(cfg, R.PTR), (self.S, R.I32), (level, R.I32)])
            self.site_cfgs[method] = cfg
        return self.site_cfgs[method]

    def vtable(self, w, h):
        ev, bd = self.ev, self.bd
        vfp = ev.new("aom_variance_fn_ptr_t")
        if bd == 8:
            names = dict(sdf="aom_sad%dx%d_c", sdsf="aom_sad_skip_%dx%d_c", vf="aom_variance%dx%d_c", svf="aom_sub_pixel_variance%dx%d_c",
                         sdx4df="aom_sad%dx%dx4d_c", sdx3df="aom_sad%dx%dx3d_c", sdsx4df="aom_sad_skip_%dx%dx4d_c")
        else:
            b = "_bits%d" % bd
            names = dict(sdf="aom_highbd_sad%dx%d" + b, sdsf="aom_highbd_sad_skip_%dx%d" + b, vf="aom_highbd_%d_variance%%dx%%d_c" % bd,
                         svf="aom_highbd_%d_sub_pixel_variance%%dx%%d_c" % bd, sdx4df="aom_highbd_sad%dx%dx4d" + b,
                         sdx3df="aom_highbd_sad%dx%dx3d" + b, sdsx4df="aom_highbd_sad_skip_%dx%dx4d" + b)
        for k, pat in names.items():
            fn = pat % (w, h)
            assert fn in ev.funcs, fn
            ev.set(vfp, k, R.FuncRef(fn))
        return vfp

    def cost_params(self, obj, prefix, cost_type, ref_row, ref_col, sad_per_bit, error_per_bit):
        ev = self.ev
        refmv = ev.new("MV")
        ev.set(refmv, "row", ref_row); ev.set(refmv, "col", ref_col)
        ev.set(obj, prefix + "ref_mv", refmv)
        # full_ref_mv = get_fullmv_from_mv(ref_mv) (mcomp.c init_mv_cost_params): evaluated by the reference's helper
        full = ev.interp.call("get_fullmv_from_mv", [(refmv, R.PTR)])[0]
        ev.field(obj, prefix + "full_ref_mv").store(full, full.st)
        ev.set(obj, prefix + "mv_cost_type", COST_TYPES[cost_type])
        ev.set(obj, prefix + "mvjcost", self.joint)
        ev.set(obj, prefix + "mvcost[0]", self.comp[0].add(self.mv_max))
        ev.set(obj, prefix + "mvcost[1]", self.comp[1].add(self.mv_max))
        ev.set(obj, prefix + "sad_per_bit", sad_per_bit)
        ev.set(obj, prefix + "error_per_bit", error_per_bit)

    def fullpel_params(self, blk, w, h, method, cost_type, sad_per_bit=20, error_per_bit=60, skip_sad=False, mesh=None,
                       run_mesh=0, force_mesh_thresh=INT_MAX, prune_mesh=0, mesh_diff_thr=0, fine_interval=0):
        ev = self.ev
        bx, by, srow, scol, rrow, rcol, rmin, rmax, cmin, cmax = blk
        ms = ev.new("FULLPEL_MOTION_SEARCH_PARAMS")
        vfp = self.vtable(w, h)
        ev.set(ms, "bsize", self.const(BSIZE[(w, h)])); ev.set(ms, "vfp", vfp)
        ev.set(ms, "ms_buffers.ref", self.buf2d(self.refp, by, bx)); ev.set(ms, "ms_buffers.src", self.buf2d(self.srcp, by, bx))
        ev.set(ms, "search_method", self.const("NSTEP" if method == "NSTEP_FPF" else method)); ev.set(ms, "search_sites", self.sites(method))
        for k, v in dict(row_min=rmin, row_max=rmax, col_min=cmin, col_max=cmax).items():
            ev.set(ms, "mv_limits." + k, v)
        self.cost_params(ms, "mv_cost_params.", cost_type, rrow, rcol, sad_per_bit, error_per_bit)
        for k in ("sdf", "sdx4df", "sdx3df"):
            src_k = {"sdf": "sdsf", "sdx4df": "sdsx4df", "sdx3df": "sdsx4df"}[k] if skip_sad else k
            ev.set(ms, k, ev.get(vfp, src_k))
        ev.set(ms, "run_mesh_search", run_mesh); ev.set(ms, "force_mesh_thresh", force_mesh_thresh)
        ev.set(ms, "prune_mesh_search", prune_mesh); ev.set(ms, "mesh_search_mv_diff_threshold", mesh_diff_thr)
        ev.set(ms, "fine_search_interval", fine_interval)
        self.mesh_obj = None
        if mesh is not None:
            pat = ev.interp.alloc(("arr", ev.structs["MESH_PATTERN"], 4), True)
            for i, (rng_, itv) in enumerate(mesh):
                ev.set(pat, "[%d].range" % i, rng_); ev.set(pat, "[%d].interval" % i, itv)
            self.mesh_obj = pat.deref()[0]
            ev.set(ms, "mesh_patterns[0]", self.mesh_obj); ev.set(ms, "mesh_patterns[1]", self.mesh_obj)
        return ms

    def mv_struct(self, tname, row, col):
        m = self.ev.new(tname)
        self.ev.set(m, "row", row); self.ev.set(m, "col", col)
        return m


def limits(bx, by, w, h, clip=None):
    ext = BORDER - 8
    rmin, rmax, cmin, cmax = -(by + ext), (H - by - h) + ext, -(bx + ext), (W - bx - w) + ext
    if clip is not None:
        rmin, rmax, cmin, cmax = max(rmin, -clip), min(rmax, clip), max(cmin, -clip), min(cmax, clip)
    return rmin, rmax, cmin, cmax


def main():
    ev = make_evaluator()
    arrays, cases = {}, []
    rng = np.random.default_rng(20261006)
    mvc = synth_mv_costs(7)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    harness = {}
    for bd in (8, 10):
        s, r = synth_planes(bd, 100 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s, r
        harness[bd] = Harness(ev, bd, s, r, mvc)

    def block(w, h, edge=False, clip=None, start=None, refmv=None):
        if edge:
            bx, by = int(rng.choice([0, W - w])), int(rng.choice([0, H - h]))
        else:
            bx, by = int(rng.integers(0, (W - w) // 4 + 1)) * 4, int(rng.integers(0, (H - h) // 4 + 1)) * 4
        lim = limits(bx, by, w, h, clip)
        st = start if start is not None else (0, 0)
        rm = refmv if refmv is not None else (0, 0)
        return (bx, by, st[0], st[1], rm[0], rm[1]) + lim

    def run_fullpel(kind, bd, w, h, blk, method, step_param, cost_type, **kw):
        hs = harness[bd]
        ms = hs.fullpel_params(blk, w, h, method, cost_type, **kw)
        start = hs.mv_struct("FULLPEL_MV", blk[2], blk[3])
        best, second = ev.new("FULLPEL_MV"), ev.new("FULLPEL_MV")
        cl = ev.array([0] * 5, "int")
        t0 = time.time()
        if kind == "diamond":
            cost = ev.call("full_pixel_diamond", start.buf[0], ms, step_param, cl, best, second)
        elif kind == "mesh":
            cost = ev.call("full_pixel_exhaustive", start.buf[0], ms, hs.mesh_obj, cl, best, second)
        else:
            cost = ev.call("av1_full_pixel_search", start.buf[0], ms, step_param, cl, best, second)
        rec = dict(kind=kind, bd=bd, w=w, h=h, block=list(blk), method=method, step_param=step_param, cost_type=COST_TYPES[cost_type],
                   mv=[ev.get(best, "row"), ev.get(best, "col")], cost=cost, cost_list=list(cl.buf))
        if kind != "mesh":
            try:
                rec["second_best"] = [ev.get(second, "row"), ev.get(second, "col")]
            except R.CError:
                rec["second_best"] = None
        rec.update({k: v for k, v in kw.items() if k != "mesh"})
        if kw.get("mesh") is not None:
            rec["mesh"] = [list(p) for p in kw["mesh"]]
        cases.append(rec)
        return rec

    t0 = time.time()
    # 1. full_pixel_diamond (DIAMOND / CLAMPED_DIAMOND sites), the configuration of the device kernel
    for bd in (8, 10):
        for method in ("DIAMOND", "CLAMPED_DIAMOND"):
            for cost_type in ("NONE", "L1_LOWRES", "L1_MIDRES", "L1_HDRES"):
                for step_param in ((4, 6, 2) if cost_type == "L1_HDRES" else (4,)):
                    for (w, h) in ((16, 16), (8, 8)) if bd == 8 else ((16, 16),):
                        for trial in range(2):
                            edge = trial == 1
                            start = (int(rng.integers(-4, 5)), int(rng.integers(-4, 5))) if trial else None
                            refmv = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41))) if trial else None
                            run_fullpel("diamond", bd, w, h, block(w, h, edge=edge, start=start, refmv=refmv, clip=20 if edge else None),
                                        method, step_param, cost_type)
    print("diamond: %d cases, %.0f s" % (len(cases), time.time() - t0))
    # 2. av1_full_pixel_search: every search method, cost list, entropy costs
    n0 = len(cases)
    for method in METHODS:
        for cost_type in ("ENTROPY", "L1_HDRES", "NONE"):
            for trial in range(3):
                w, h = ((16, 16), (8, 8), (16, 8))[trial]
                bd = 10 if (trial == 0 and cost_type == "L1_HDRES") else 8
                edge = trial == 2
                start = (int(rng.integers(-6, 7)), int(rng.integers(-6, 7))) if trial else None
                refmv = (int(rng.integers(-60, 61)), int(rng.integers(-60, 61))) if trial != 1 else None
                step_param = int(rng.integers(0, 7)) if trial else 3
                run_fullpel("search", bd, w, h, block(w, h, edge=edge, start=start, refmv=refmv, clip=16 if edge else 32), method,
                            step_param, cost_type, sad_per_bit=int(rng.integers(8, 40)), error_per_bit=int(rng.integers(20, 120)))
    # mesh follow-up, forced mesh, pruned mesh, downsampled-SAD re-check
    mesh0 = [(16, 4), (8, 2), (4, 1), (3, 1)]
    for method, kw in (("NSTEP", dict(run_mesh=1, mesh=mesh0)), ("NSTEP", dict(force_mesh_thresh=0, mesh=mesh0)),
                       ("NSTEP_8PT", dict(force_mesh_thresh=1 << 20, mesh=mesh0)),
                       ("DIAMOND", dict(run_mesh=1, prune_mesh=1, mesh_diff_thr=2, mesh=mesh0)),
                       ("DIAMOND", dict(run_mesh=1, prune_mesh=1, mesh_diff_thr=64, mesh=mesh0)),
                       ("NSTEP", dict(skip_sad=True)), ("HEX", dict(skip_sad=True)), ("DIAMOND", dict(skip_sad=True, run_mesh=1, mesh=mesh0))):
        for trial in range(2):
            run_fullpel("search", 8, 16, 16, block(16, 16, clip=24, start=(trial, -trial)), method, 2 + trial, "L1_HDRES", **kw)
    print("search: %d cases, %.0f s" % (len(cases) - n0, time.time() - t0))
    # 3. full_pixel_exhaustive: pattern rows, fine_search_interval
    n0 = len(cases)
    for mesh, fine in (([(8, 2), (4, 1), (2, 1), (1, 1)], 0), ([(12, 4), (6, 2), (3, 1), (3, 1)], 0), ([(16, 8), (8, 4), (4, 1), (2, 1)], 1),
                       ([(6, 1), (3, 1), (2, 1), (1, 1)], 0)):
        for (bd, w, h, ct) in ((8, 8, 8, "L1_HDRES"), (8, 16, 16, "NONE"), (10, 8, 8, "L1_LOWRES")):
            run_fullpel("mesh", bd, w, h, block(w, h, clip=14, start=(int(rng.integers(-2, 3)), int(rng.integers(-2, 3)))), "DIAMOND", 0, ct,
                        mesh=mesh, fine_interval=fine)
    print("mesh: %d cases, %.0f s" % (len(cases) - n0, time.time() - t0))
    # 4. bilinear sub-pel tree (pruned_more / pruned / tree), starting from the full-pel optimum of a diamond search
    n0 = len(cases)
    for bd in (8, 10):
        hs = harness[bd]
        for fn in ("av1_find_best_sub_pixel_tree_pruned_more", "av1_find_best_sub_pixel_tree_pruned", "av1_find_best_sub_pixel_tree"):
            for cost_type in ("L1_HDRES", "NONE", "ENTROPY"):
                for trial in range(2 if fn.endswith("more") else 1):
                    w, h = (16, 16) if trial == 0 else (8, 8)
                    blk = block(w, h, clip=24, refmv=(int(rng.integers(-30, 31)), int(rng.integers(-30, 31))))
                    fp = run_fullpel("diamond", bd, w, h, blk, "DIAMOND", 4, cost_type)
                    cases.pop()                                                    # (only its MV is needed)
                    allow_hp, forced_stop, iters = int(rng.integers(0, 2)), int(rng.integers(0, 3)), int(rng.integers(1, 3))
                    sp = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
                    ev.set(sp, "allow_hp", allow_hp); ev.set(sp, "forced_stop", forced_stop); ev.set(sp, "iters_per_step", iters)
                    bx, by = blk[0], blk[1]
                    # av1_set_subpel_mv_search_range (mcomp.h:345-368) evaluated by the reference itself
                    fl = ev.new("FullMvLimits")
                    for k, v in zip(("row_min", "row_max", "col_min", "col_max"), blk[6:]):
                        ev.set(fl, k, v)
                    refmv = hs.mv_struct("MV", blk[4], blk[5])
                    ev.interp.call("av1_set_subpel_mv_search_range", [(ev.field(sp, "mv_limits"), R.PTR), (fl, R.PTR), (refmv, R.PTR)])
                    hs.cost_params(sp, "mv_cost_params.", cost_type, blk[4], blk[5], 20, 60)
                    ev.set(sp, "var_params.vfp", hs.vtable(w, h))
                    ev.set(sp, "var_params.subpel_search_type", hs.const("USE_2_TAPS_ORIG"))
                    ev.set(sp, "var_params.ms_buffers.ref", hs.buf2d(hs.refp, by, bx)); ev.set(sp, "var_params.ms_buffers.src", hs.buf2d(hs.srcp, by, bx))
                    ev.set(sp, "var_params.w", w); ev.set(sp, "var_params.h", h)
                    start = hs.mv_struct("MV", fp["mv"][0] * 8, fp["mv"][1] * 8)
                    best = ev.new("MV")
                    dist, sse = ev.array([0], "int"), ev.array([0], "unsigned int")
                    t1 = time.time()
                    err = ev.call(fn, make_xd(ev), None, sp, start.buf[0], best, dist, sse, None)
                    lim = [ev.get(sp, "mv_limits." + k) for k in ("row_min", "row_max", "col_min", "col_max")]
                    cases.append(dict(kind="subpel", fn=fn, bd=bd, w=w, h=h, block=list(blk), fullpel_mv=fp["mv"], cost_type=COST_TYPES[cost_type],
                                      allow_hp=allow_hp, forced_stop=forced_stop, iters=iters, subpel_limits=lim, error_per_bit=60,
                                      mv=[ev.get(best, "row"), ev.get(best, "col")], err=err, distortion=dist.buf[0], sse=sse.buf[0]))
    print("subpel: %d cases, %.0f s" % (len(cases) - n0, time.time() - t0))
    # 5. the sub-pel trees driven by a cost list, as the encoder does: cost_list = what av1_full_pixel_search returned
    #    (appended after the sections above with its own generator so that the earlier cases keep their values)
    n0 = len(cases)
    rng2 = np.random.default_rng(20261009)
    for bd in (8, 10):
        hs = harness[bd]
        for fn in ("av1_find_best_sub_pixel_tree_pruned_more", "av1_find_best_sub_pixel_tree_pruned"):
            for cost_type in ("L1_HDRES", "ENTROPY", "NONE"):
                for trial in range(3 if bd == 8 else 1):
                    w, h = (16, 16) if trial != 1 else (8, 8)
                    bx, by = int(rng2.integers(0, (W - w) // 4 + 1)) * 4, int(rng2.integers(0, (H - h) // 4 + 1)) * 4
                    blk = (bx, by, 0, 0, int(rng2.integers(-30, 31)), int(rng2.integers(-30, 31))) + limits(bx, by, w, h, 24)
                    method = ("NSTEP", "BIGDIA", "HEX")[trial]
                    fp = run_fullpel("search", bd, w, h, blk, method, 2, cost_type, sad_per_bit=25, error_per_bit=70)
                    cases.pop()
                    allow_hp, forced_stop, iters = int(rng2.integers(0, 2)), int(rng2.integers(0, 2)), int(rng2.integers(1, 3))
                    sp = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
                    ev.set(sp, "allow_hp", allow_hp); ev.set(sp, "forced_stop", forced_stop); ev.set(sp, "iters_per_step", iters)
                    ev.set(sp, "cost_list", ev.array(fp["cost_list"], "int"))
                    fl = ev.new("FullMvLimits")
                    for k, v in zip(("row_min", "row_max", "col_min", "col_max"), blk[6:]):
                        ev.set(fl, k, v)
                    refmv = hs.mv_struct("MV", blk[4], blk[5])
                    ev.interp.call("av1_set_subpel_mv_search_range", [(ev.field(sp, "mv_limits"), R.PTR), (fl, R.PTR), (refmv, R.PTR)])
                    hs.cost_params(sp, "mv_cost_params.", cost_type, blk[4], blk[5], 25, 70)
                    ev.set(sp, "var_params.vfp", hs.vtable(w, h))
                    ev.set(sp, "var_params.subpel_search_type", hs.const("USE_2_TAPS_ORIG"))
                    ev.set(sp, "var_params.ms_buffers.ref", hs.buf2d(hs.refp, by, bx)); ev.set(sp, "var_params.ms_buffers.src", hs.buf2d(hs.srcp, by, bx))
                    ev.set(sp, "var_params.w", w); ev.set(sp, "var_params.h", h)
                    start = hs.mv_struct("MV", fp["mv"][0] * 8, fp["mv"][1] * 8)
                    best = ev.new("MV")
                    dist, sse = ev.array([0], "int"), ev.array([0], "unsigned int")
                    err = ev.call(fn, make_xd(ev), None, sp, start.buf[0], best, dist, sse, None)
                    lim = [ev.get(sp, "mv_limits." + k) for k in ("row_min", "row_max", "col_min", "col_max")]
                    cases.append(dict(kind="subpel", fn=fn, bd=bd, w=w, h=h, block=list(blk), fullpel_mv=fp["mv"], cost_type=COST_TYPES[cost_type],
                                      allow_hp=allow_hp, forced_stop=forced_stop, iters=iters, subpel_limits=lim, cost_list=fp["cost_list"],
                                      error_per_bit=70, mv=[ev.get(best, "row"), ev.get(best, "col")], err=err, distortion=dist.buf[0],
                                      sse=sse.buf[0]))
    print("subpel with cost list: %d cases, %.0f s" % (len(cases) - n0, time.time() - t0))
    # 6. av1_find_best_sub_pixel_tree with the up-sampled prediction error (subpel_search_type USE_8_TAPS, the encoder's
    #    default): aom_[highbd_]upsampled_pred_c -> aom_[highbd_]convolve8_{horiz,vert}_c -> vfp->vf
    n0 = len(cases)
    rng3 = np.random.default_rng(20261010)
    for bd in (8, 10):
        hs = harness[bd]
        for (w, h, cost_type) in ((8, 8, "L1_HDRES"), (16, 16, "NONE"), (16, 8, "ENTROPY"), (8, 16, "L1_LOWRES")):
            if bd == 10 and (w, h) != (8, 8):
                continue
            for trial in range(2):
                bx, by = int(rng3.integers(0, (W - w) // 4 + 1)) * 4, int(rng3.integers(0, (H - h) // 4 + 1)) * 4
                blk = (bx, by, 0, 0, int(rng3.integers(-30, 31)), int(rng3.integers(-30, 31))) + limits(bx, by, w, h, 24)
                fp = run_fullpel("diamond", bd, w, h, blk, "DIAMOND", 4, cost_type)
                cases.pop()
                allow_hp, forced_stop, iters = int(rng3.integers(0, 2)), int(rng3.integers(0, 2)), 2 - (trial & 1)
                sp = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
                ev.set(sp, "allow_hp", allow_hp); ev.set(sp, "forced_stop", forced_stop); ev.set(sp, "iters_per_step", iters)
                fl = ev.new("FullMvLimits")
                for k, v in zip(("row_min", "row_max", "col_min", "col_max"), blk[6:]):
                    ev.set(fl, k, v)
                refmv = hs.mv_struct("MV", blk[4], blk[5])
                ev.interp.call("av1_set_subpel_mv_search_range", [(ev.field(sp, "mv_limits"), R.PTR), (fl, R.PTR), (refmv, R.PTR)])
                hs.cost_params(sp, "mv_cost_params.", cost_type, blk[4], blk[5], 25, 70)
                ev.set(sp, "var_params.vfp", hs.vtable(w, h))
                ev.set(sp, "var_params.subpel_search_type", hs.const("USE_8_TAPS"))
                ev.set(sp, "var_params.ms_buffers.ref", hs.buf2d(hs.refp, by, bx)); ev.set(sp, "var_params.ms_buffers.src", hs.buf2d(hs.srcp, by, bx))
                ev.set(sp, "var_params.w", w); ev.set(sp, "var_params.h", h)
                start = hs.mv_struct("MV", fp["mv"][0] * 8, fp["mv"][1] * 8)
                best = ev.new("MV")
                dist, sse = ev.array([0], "int"), ev.array([0], "unsigned int")
                t1 = time.time()
                err = ev.call("av1_find_best_sub_pixel_tree", make_xd(ev, bd), None, sp, start.buf[0], best, dist, sse, None)
                lim = [ev.get(sp, "mv_limits." + k) for k in ("row_min", "row_max", "col_min", "col_max")]
                cases.append(dict(kind="subpel", fn="av1_find_best_sub_pixel_tree", subpel_search_type=3, bd=bd, w=w, h=h, block=list(blk),
                                  fullpel_mv=fp["mv"], cost_type=COST_TYPES[cost_type], allow_hp=allow_hp, forced_stop=forced_stop, iters=iters,
                                  subpel_limits=lim, error_per_bit=70, mv=[ev.get(best, "row"), ev.get(best, "col")], err=err,
                                  distortion=dist.buf[0], sse=sse.buf[0]))
    print("subpel tree, 8-tap up-sampled error: %d cases, %.0f s" % (len(cases) - n0, time.time() - t0))
    # 7. NSTEP on the first-pass site table av1_init_motion_fpf (own generator)
    n0 = len(cases)
    rng4 = np.random.default_rng(20261013)
    for trial in range(6):
        w, h = ((16, 16), (8, 8), (16, 8))[trial % 3]
        bx, by = int(rng4.integers(0, (W - w) // 4 + 1)) * 4, int(rng4.integers(0, (H - h) // 4 + 1)) * 4
        blk = (bx, by, int(rng4.integers(-3, 4)), int(rng4.integers(-3, 4)), int(rng4.integers(-20, 21)), int(rng4.integers(-20, 21))) + limits(bx, by, w, h, 28)
        run_fullpel("search", 8 if trial < 4 else 10, w, h, blk, "NSTEP_FPF", int(rng4.integers(0, 5)), ("L1_HDRES", "NONE", "ENTROPY")[trial % 3],
                    sad_per_bit=22, error_per_bit=66)
    print("first-pass table: %d cases, %.0f s" % (len(cases) - n0, time.time() - t0))
    # the site tables themselves (G1): every builder, as (stage, index) -> (row, col), searches_per_step, radius
    sites = {}
    for m in METHODS + ["NSTEP_FPF"]:
        cfg = harness[8].sites(m)
        n = ev.get(cfg, "num_search_steps")
        sp = [ev.get(cfg, "searches_per_step[%d]" % i) for i in range(n)]
        rad = [ev.get(cfg, "radius[%d]" % i) for i in range(n)]
        first = 0 if m not in ("DIAMOND", "CLAMPED_DIAMOND", "NSTEP_FPF") else 11 - n
        mv = []
        for i in range(n):
            st = i + first
            lo = 1 if m in ("DIAMOND", "CLAMPED_DIAMOND", "NSTEP", "NSTEP_8PT", "NSTEP_FPF") else 0
            cnt = sp[i + first] if first else sp[i]
            mv.append([[ev.get(cfg, "site[%d][%d].mv.row" % (st, j)), ev.get(cfg, "site[%d][%d].mv.col" % (st, j))] for j in range(lo, lo + cnt)])
        sites[m] = dict(num_search_steps=n, searches_per_step=sp if not first else [ev.get(cfg, "searches_per_step[%d]" % (i + first)) for i in range(n)],
                        radius=rad if not first else [ev.get(cfg, "radius[%d]" % (i + first)) for i in range(n)], first_stage=first, mv=mv)
    path = os.path.join(HERE, "ref_eval_mcomp.npz")
    np.savez_compressed(path, cases=np.frombuffer(json.dumps({"cases": cases, "sites": sites, "W": W, "H": H, "border": BORDER}).encode(), np.uint8),
                        **arrays)
    print("ref_eval_mcomp.npz: %d cases, %.1f KB" % (len(cases), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
