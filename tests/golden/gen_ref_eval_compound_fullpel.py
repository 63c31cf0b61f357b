#!/usr/bin/env python3
"""Golden vectors for av1_full_pixel_search on a COMPOUND prediction, obtained by interpreting av1/encoder/mcomp.c itself (build container
only; tests/golden/ref_c_eval.py, harness of gen_ref_eval_mcomp.py / gen_ref_eval_compound_search.py):

  av1_full_pixel_search (mcomp.c:1693-1832) with ms_buffers.second_pred [/ mask, inv_mask] set as av1_set_ms_compound_refs does
  (mcomp.h:152-166), cost_list NULL, second_best_mv kept -- the full-pel step of av1_joint_motion_search when
  disable_extensive_joint_motion_search is 0 (motion_search_facade.c:613-619: speed 0; step_param 5).

What this pins beyond the single-reference cases of ref_eval_mcomp.npz: diamond_search_sad takes its per-site branch with
get_mvpred_compound_sad (vfp->sdaf / msdf) and full_pixel_diamond measures get_mvpred_compound_var_cost (svaf / msvf), while the follow-up
mesh passes (full_pixel_exhaustive) and the variance after them stay on the PLAIN sdf / vf even on a compound -- reproduced as is.

Output: tests/golden/ref_eval_compound_fullpel.npz.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as M  # noqa: E402
import gen_ref_eval_compound_search as CS  # noqa: E402


def main():
    ev = CS.make_evaluator()
    arrays, cases = {}, []
    rng = np.random.default_rng(20261201)
    mvc = M.synth_mv_costs(13)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    harness = {}
    for bd in (8, 10):
        s, r = M.synth_planes(bd, 500 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s, r
        harness[bd] = M.Harness(ev, bd, s, r, mvc)
    W, H, B = M.W, M.H, M.BORDER
    mesh0 = [(12, 4), (6, 2), (4, 1), (3, 1)]
    # (method, step_param, extra search params)
    configs = [("NSTEP", 5, {}), ("NSTEP", 5, dict(force_mesh_thresh=0, mesh=mesh0)), ("DIAMOND", 6, {}),
               ("NSTEP_8PT", 7, dict(run_mesh=1, prune_mesh=1, mesh_diff_thr=1, mesh=mesh0)), ("CLAMPED_DIAMOND", 5, {}), ("NSTEP", 4, {})]
    sizes = [(16, 16), (8, 8), (16, 8)]
    t0 = time.time()
    k = 0
    for bd in (8, 10):
        hs = harness[bd]
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        for ci, (method, step_param, kw) in enumerate(configs):
            for trial in range(3 if bd == 8 else 2):
                w, h = sizes[(ci + trial) % 3]
                masked = trial >= 1
                inv = trial == 2 or (bd == 10 and trial == 1 and ci % 2 == 1)
                cost_type = ["ENTROPY", "L1_HDRES", "NONE", "L1_LOWRES", "ENTROPY", "L1_MIDRES"][(ci + trial) % 6]
                edge = (ci + trial) % 4 == 3
                bx = int(rng.choice([0, W - w])) if edge else int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = int(rng.choice([0, H - h])) if edge else int(rng.integers(0, (H - h) // 4 + 1)) * 4
                lim = M.limits(bx, by, w, h, 9 if edge else 28)
                start = (int(rng.integers(-5, 6)), int(rng.integers(-5, 6)))
                refmv = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                blk = (bx, by, start[0], start[1], refmv[0], refmv[1]) + lim
                ms = hs.fullpel_params(blk, w, h, method, cost_type, sad_per_bit=int(rng.integers(8, 40)), error_per_bit=int(rng.integers(20, 120)), **kw)
                vfp = ev.get(ms, "vfp")
                CS.extend_vtable(ev, vfp, bd, w, h)
                # the other reference's predictor: the reference block some pixels off the start position, plus noise -- so that the compound
                # optimum differs from the single-reference one
                oy, ox = by + start[0] + int(rng.integers(-3, 4)), bx + start[1] + int(rng.integers(-3, 4))
                refpl = arrays["ref%d" % bd]
                sp = refpl[B + oy:B + oy + h, B + ox:B + ox + w].astype(np.int32) + rng.integers(-(8 << (bd - 8)), (8 << (bd - 8)) + 1, (h, w))
                sp = np.clip(sp, 0, mx).astype(np.uint16)
                ev.set(ms, "ms_buffers.second_pred", ev.array(sp.ravel(), ct))
                mask = None
                if masked:
                    ramp = np.clip((np.arange(w)[None, :] * 2 - np.arange(h)[:, None] + (h - w) // 2) * 5 + 32 + rng.integers(-3, 4, (h, w)), 0, 64)
                    mask = ramp.astype(np.uint8)
                    ev.set(ms, "ms_buffers.mask", ev.array(mask.ravel(), "uint8_t"))
                    ev.set(ms, "ms_buffers.mask_stride", w)
                    ev.set(ms, "ms_buffers.inv_mask", int(inv))
                startmv = hs.mv_struct("FULLPEL_MV", start[0], start[1])
                best, second = ev.new("FULLPEL_MV"), ev.new("FULLPEL_MV")
                t1 = time.time()
                cost = ev.call("av1_full_pixel_search", startmv.buf[0], ms, step_param, None, best, second)
                arrays["sp%d" % k] = sp
                if masked:
                    arrays["mask%d" % k] = mask
                rec = dict(k=k, bd=bd, w=w, h=h, block=list(blk), method=method, step_param=step_param, cost_type=M.COST_TYPES[cost_type],
                           masked=int(masked), inv=int(inv), sad_per_bit=ev.get(ms, "mv_cost_params.sad_per_bit"),
                           error_per_bit=ev.get(ms, "mv_cost_params.error_per_bit"), mv=[ev.get(best, "row"), ev.get(best, "col")], cost=cost,
                           second_best=[ev.get(second, "row"), ev.get(second, "col")])
                rec.update({kk: v for kk, v in kw.items() if kk != "mesh"})
                if kw.get("mesh") is not None:
                    rec["mesh"] = [list(p) for p in kw["mesh"]]
                cases.append(rec)
                print("case %d (%s %dx%d bd %d): %.0f s" % (k, method, w, h, bd, time.time() - t1), flush=True)
                k += 1
    meta = dict(border=B, width=W, height=H, generated_by="tests/golden/gen_ref_eval_compound_fullpel.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_compound_fullpel.npz"), **arrays)
    print("wrote ref_eval_compound_fullpel.npz: %d cases, %.0f s" % (len(cases), time.time() - t0))


if __name__ == "__main__":
    main()
