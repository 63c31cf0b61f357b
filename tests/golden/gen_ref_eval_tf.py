#!/usr/bin/env python3
"""Golden vectors for the temporal filter's motion search (build container only; output tests/golden/ref_eval_tf.npz).

tf_motion_search (av1/encoder/temporal_filter.c:87-253) takes the whole encoder instance (AV1_COMP, MACROBLOCK), which the
evaluator (tests/golden/ref_c_eval.py) cannot build.  So its BODY is driven statement by statement from this script, with every
computation done by the interpreted reference:
  * av1_set_mv_search_range / av1_set_subpel_mv_search_range, get_fullmv_from_mv / get_mv_from_fullmv, av1_init_search_range,
  * av1_full_pixel_search (NSTEP sites from av1_init_motion_compensation[], run_mesh_search = 1, the prune rule, L1 MV cost),
  * av1_find_best_sub_pixel_tree[_pruned[_more]] with USE_8_TAPS (aom_[highbd_]upsampled_pred_c + vfp->vf), MV_COST_NONE,
  * the vtable's vf for the force_integer_mv branch,
  * tf_determine_block_partition itself (a static function: its text is read from temporal_filter.c at generation time),
and only the sequencing (which result feeds which call, the DIVIDE_AND_ROUND of the errors, the ref_mv hand-over and the
frame loop of av1_tf_do_filtering_row, :849-867) written here after the reference.  The block limits come from
av1_set_mv_row_limits / av1_set_mv_col_limits' formulas (mcomp.h:216-240; they need CommonModeInfoParams, so they are
restated here and pinned by the reference's arithmetic being plain)."""
import json
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402  (the evaluator set-up and parameter objects of the search fixtures)

REF = G.REF
W, H, BORDER = G.W, G.H, G.BORDER
INT_MAX = 2147483647
GOOD_MESH = [(64, 8), (28, 4), (15, 1), (7, 1)]          # good_quality_mesh_patterns[0] (speed_features.c:25-33) -- an input here
TREES = {2: "av1_find_best_sub_pixel_tree", 1: "av1_find_best_sub_pixel_tree_pruned", 0: "av1_find_best_sub_pixel_tree_pruned_more"}


def window(bd, seed, n_frames):
    """Frames of a smooth field moving with a per-quadrant velocity + noise; one noisy patch (large block_mse -> ref_mv reset)."""
    rng = np.random.default_rng(seed)
    big = rng.integers(0, 1 << bd, (H + 96, W + 96)).astype(np.float64)
    for _ in range(3):
        c = np.cumsum(np.pad(big, ((3, 2), (0, 0)), mode="edge"), axis=0)
        big = (c[5:] - c[:-5]) / 5.0
        c = np.cumsum(np.pad(big, ((0, 0), (3, 2)), mode="edge"), axis=1)
        big = (c[:, 5:] - c[:, :-5]) / 5.0
    big = (big - big.min()) / (big.max() - big.min()) * ((1 << bd) - 1)
    vel = {(0, 0): (1.5, -2.0), (0, 1): (-2.0, 1.0), (1, 0): (0.5, 3.0), (1, 1): (-1.0, -1.5)}
    dt = np.uint8 if bd == 8 else np.uint16
    frames = []
    for f in range(n_frames):
        img = np.empty((H, W))
        for (qy, qx), (vr, vc) in vel.items():
            r, c = int(round(vr * f)), int(round(vc * f))
            ys, xs = slice(qy * H // 2, (qy + 1) * H // 2), slice(qx * W // 2, (qx + 1) * W // 2)
            img[ys, xs] = big[48 + r + ys.start:48 + r + ys.stop, 48 + c + xs.start:48 + c + xs.stop]
        img = img + rng.normal(0, (1 << bd) / 420.0, img.shape)
        if f == n_frames - 1:
            img[64:96, 0:32] = rng.integers(0, 1 << bd, (32, 32))
        frames.append(np.pad(np.clip(np.rint(img), 0, (1 << bd) - 1).astype(dt), BORDER, mode="edge"))
    return frames


def block_limits(mb_row, mb_col):
    mi_rows, mi_cols = ((H + 7) & ~7) // 4, ((W + 7) & ~7) // 4
    out = []
    for pos, mi_n in ((mb_row * 8, mi_rows), (mb_col * 8, mi_cols)):
        lo = max(-(pos * 4 + BORDER - 8), -((pos + 8) * 4 + 8))
        hi = min((mi_n - pos - 8) * 4 + BORDER - 8, (mi_n - pos) * 4 + 8)
        out += [lo, hi]
    return out  # row_min, row_max, col_min, col_max


class Tf:
    def __init__(self, ev, bd, frames):
        self.ev, self.bd, self.frames = ev, bd, frames
        self.mvc = G.synth_mv_costs(7)                   # unused by the L1 / NONE cost types, the objects just need tables
        self.h = {}

    def harness(self, filter_frame, f):
        if (filter_frame, f) not in self.h:
            self.h[(filter_frame, f)] = G.Harness(self.ev, self.bd, self.frames[filter_frame], self.frames[f], self.mvc)
        return self.h[(filter_frame, f)]

    def full_limits(self, lim):
        ev = self.ev
        fl = ev.new("FullMvLimits")
        for k, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
            ev.set(fl, k, v)
        zero = ev.new("MV")
        ev.interp.call("av1_set_mv_search_range", [(fl, R.PTR), (zero, R.PTR)])      # mcomp.c:148-149 with the baseline MV
        return [ev.get(fl, k) for k in ("row_min", "row_max", "col_min", "col_max")]

    def fullpel(self, hs, bx, by, w, lim, start, p):
        ev = self.ev
        blk = (bx, by, start[0], start[1], 0, 0) + tuple(self.full_limits(lim))
        cost_type = {3: "L1_HDRES", 2: "L1_MIDRES", 1: "L1_LOWRES"}[p["cost_type"]]
        ms = hs.fullpel_params(blk, w, w, "NSTEP", cost_type, skip_sad=bool(p["skip_sad"]), mesh=p["mesh"], run_mesh=1,
                               prune_mesh=p["prune"], mesh_diff_thr=p["thr"])
        st = hs.mv_struct("FULLPEL_MV", start[0], start[1])
        best = ev.new("FULLPEL_MV")
        cl = ev.array([0] * 5, "int") if p["use_cost_list"] else None
        ev.call("av1_full_pixel_search", st.buf[0], ms, p["step_param"], cl, best, None)
        return [ev.get(best, "row"), ev.get(best, "col")], (list(cl.buf) if cl is not None else None)

    def subpel(self, hs, bx, by, w, lim, full_mv, cost_list, p):
        ev = self.ev
        sp = ev.new("SUBPEL_MOTION_SEARCH_PARAMS")
        ev.set(sp, "allow_hp", p["allow_hp"]); ev.set(sp, "forced_stop", 0); ev.set(sp, "iters_per_step", p["iters"])   # EIGHTH_PEL (:181)
        if cost_list is not None:
            ev.set(sp, "cost_list", ev.array(cost_list, "int"))
        fl = ev.new("FullMvLimits")
        for k, v in zip(("row_min", "row_max", "col_min", "col_max"), lim):
            ev.set(fl, k, v)
        zero = hs.mv_struct("MV", 0, 0)
        ev.interp.call("av1_set_subpel_mv_search_range", [(ev.field(sp, "mv_limits"), R.PTR), (fl, R.PTR), (zero, R.PTR)])
        hs.cost_params(sp, "mv_cost_params.", "NONE", 0, 0, 20, 60)                      # MV_COST_NONE (:183-185)
        ev.set(sp, "var_params.vfp", hs.vtable(w, w))
        ev.set(sp, "var_params.subpel_search_type", hs.const("USE_8_TAPS"))
        ev.set(sp, "var_params.ms_buffers.ref", hs.buf2d(hs.refp, by, bx)); ev.set(sp, "var_params.ms_buffers.src", hs.buf2d(hs.srcp, by, bx))
        ev.set(sp, "var_params.w", w); ev.set(sp, "var_params.h", w)
        fm = hs.mv_struct("FULLPEL_MV", full_mv[0], full_mv[1])
        start = ev.interp.call("get_mv_from_fullmv", [(fm, R.PTR)])[0]                   # :187
        best = ev.new("MV")
        dist, sse = ev.array([0], "int"), ev.array([0], "unsigned int")
        err = ev.call(TREES[p["tree"]], G.make_xd(ev, self.bd), None, sp, start, best, dist, sse, None)
        return [ev.get(best, "row"), ev.get(best, "col")], err

    def full_from_mv(self, hs, mv):
        m = hs.mv_struct("MV", mv[0], mv[1])
        full = self.ev.interp.call("get_fullmv_from_mv", [(m, R.PTR)])[0]
        f = self.ev.new("FULLPEL_MV")
        f.store(full, full.st)
        return [self.ev.get(f, "row"), self.ev.get(f, "col")]

    def divide_and_round(self, x, y):
        return (x + (y >> 1)) // y                                                        # DIVIDE_AND_ROUND, unsigned operands (aom_ports/mem.h:77)

    def motion_search(self, filter_frame, f, mb_row, mb_col, ref_mv, p):
        """tf_motion_search for one block and one reference frame -> (sub_mvs, sub_mses, ref_mv)."""
        ev = self.ev
        hs = self.harness(filter_frame, f)
        bx, by = mb_col * 32, mb_row * 32
        lim = block_limits(mb_row, mb_col)
        sub_mvs, sub_mses = [[0, 0] for _ in range(4)], [INT_MAX] * 4                    # :861-862
        best, cl = self.fullpel(hs, bx, by, 32, lim, self.full_from_mv(hs, ref_mv), p)
        if p["force_integer_mv"]:                                                        # :158-168
            block_mv = [best[0] * 8, best[1] * 8]
            vf = ("aom_variance32x32_c" if self.bd == 8 else "aom_highbd_%d_variance32x32_c" % self.bd)
            o = (BORDER + by) * hs.S + BORDER + bx
            sse = ev.array([0], "unsigned int")
            error = ev.call(vf, hs.refp.add(o + best[0] * hs.S + best[1]), hs.S, hs.srcp.add(o), hs.S, sse)
            block_mse = self.divide_and_round(error, 1024)
        else:
            block_mv, error = self.subpel(hs, bx, by, 32, lim, best, cl, p)
            block_mse = self.divide_and_round(error, 1024)
            ref_mv = list(block_mv)                                                      # :192
            start = self.full_from_mv(hs, ref_mv)                                        # :198
            k = 0
            for i in (0, 16):
                for j in (0, 16):
                    b16, cl16 = self.fullpel(hs, bx + j, by + i, 16, lim, start, p)      # the BLOCK's mv_limits (mb->mv_limits is not changed)
                    mv16, e16 = self.subpel(hs, bx + j, by + i, 16, lim, b16, cl16, p)
                    sub_mses[k] = self.divide_and_round(e16, 256)
                    sub_mvs[k] = mv16
                    k += 1
        # tf_determine_block_partition, interpreted
        mvs = ev.interp.alloc(("arr", ev.structs["mv"], 4), True)
        for k in range(4):
            ev.set(mvs, "[%d].row" % k, sub_mvs[k][0]); ev.set(mvs, "[%d].col" % k, sub_mvs[k][1])
        mses = ev.array(sub_mses, "int")
        bm = hs.mv_struct("MV", block_mv[0], block_mv[1])
        ev.call("tf_determine_block_partition", bm.buf[0], block_mse, mvs.deref()[0], mses)
        sub_mvs = [[ev.get(mvs, "[%d].row" % k), ev.get(mvs, "[%d].col" % k)] for k in range(4)]
        sub_mses = list(mses.buf)
        if block_mse > p["mse_thresh"]:                                                  # :249-252
            ref_mv = [0, 0]
        return sub_mvs, sub_mses, ref_mv, dict(block_mv=block_mv, block_mse=block_mse, full32=best)


def main():
    ev = G.make_evaluator()
    text = open(REF + "av1/encoder/temporal_filter.c").read()
    m = re.search(r"static void tf_determine_block_partition\([^;{]*\)\s*\{.*?\n}\n", text, re.S)
    ev.load_text(m.group(0), "temporal_filter.c:tf_determine_block_partition")
    cases, arrays = [], {}
    t0 = time.time()
    specs = [
        # name, bd, frames, filter idx, blocks (mb_row, mb_col), params
        dict(name="tree_prune_lvl1", bd=8, n_frames=4, filter_frame=1, blocks=[(0, 0), (1, 1), (2, 0), (1, 2)], q=30, prune_level=1, tree=2,
             use_cost_list=0, force_integer_mv=0, allow_hp=1, iters=2, skip_sad=0),
        dict(name="tree_mesh_10bit", bd=10, n_frames=3, filter_frame=1, blocks=[(1, 1), (2, 0)], q=12, prune_level=1, tree=2,
             use_cost_list=0, force_integer_mv=0, allow_hp=1, iters=2, skip_sad=0),
        dict(name="pruned_more_cost_list", bd=8, n_frames=3, filter_frame=2, blocks=[(0, 1), (1, 1)], q=40, prune_level=2, tree=0,
             use_cost_list=1, force_integer_mv=0, allow_hp=0, iters=1, skip_sad=1),
        dict(name="force_integer_mv", bd=10, n_frames=3, filter_frame=0, blocks=[(1, 1), (2, 0), (0, 2)], q=30, prune_level=1, tree=2,
             use_cost_list=0, force_integer_mv=1, allow_hp=1, iters=2, skip_sad=0),
    ]
    for ci, s in enumerate(specs):
        frames = window(s["bd"], 500 + ci, s["n_frames"])
        arrays["frames%d" % ci] = np.stack(frames)
        tf = Tf(ev, s["bd"], frames)
        step_param = ev.call("av1_init_search_range", max(W, H))                         # :121-122
        mn = min(W, H)
        prune, thr = int(s["prune_level"] == 2), 4                                       # mcomp.c:138-140
        if s["prune_level"] == 1:
            prune, thr = int(s["q"] > 20), 2                                             # :163-167
        p = dict(step_param=step_param, cost_type=3 if mn >= 720 else (2 if mn >= 480 else 1), prune=prune, thr=thr, mesh=GOOD_MESH, tree=s["tree"],
                 iters=s["iters"], allow_hp=s["allow_hp"], use_cost_list=s["use_cost_list"], skip_sad=s["skip_sad"],
                 force_integer_mv=s["force_integer_mv"], mse_thresh=(12 if mn >= 720 else 3) << (s["bd"] - 8))
        out = []
        for (mb_row, mb_col) in s["blocks"]:
            ref_mv = [0, 0]                                                              # :855
            per_frame = []
            for f in range(s["n_frames"]):
                if f == s["filter_frame"]:
                    ref_mv = [-ref_mv[0], -ref_mv[1]]                                    # :864-867
                    per_frame.append(None)
                    continue
                t1 = time.time()
                mvs, mses, ref_mv, info = tf.motion_search(s["filter_frame"], f, mb_row, mb_col, ref_mv, p)
                per_frame.append(dict(mvs=mvs, mses=mses, ref_mv_after=list(ref_mv), **info))
                print(s["name"], (mb_row, mb_col), f, per_frame[-1], flush=True)
            out.append(dict(mb_row=mb_row, mb_col=mb_col, frames=per_frame, ref_mv_final=list(ref_mv)))
        cases.append(dict(spec={k: v for k, v in s.items() if k != "blocks"}, params={k: (v if k != "mesh" else [list(x) for x in v]) for k, v in p.items()},
                          blocks=out))
        print("case %s done, %.0f s" % (s["name"], time.time() - t0), flush=True)
    path = os.path.join(HERE, "ref_eval_tf.npz")
    np.savez_compressed(path, cases=np.frombuffer(json.dumps({"cases": cases, "W": W, "H": H, "border": BORDER}).encode(), np.uint8), **arrays)
    print("ref_eval_tf.npz: %d cases, %.1f KB" % (len(cases), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
