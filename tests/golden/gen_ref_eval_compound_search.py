#!/usr/bin/env python3
"""Golden vectors for the compound-reference and OBMC full-pel searches, obtained by interpreting av1/encoder/mcomp.c itself
(build container only; tests/golden/ref_c_eval.py, harness of gen_ref_eval_mcomp.py):

  av1_refining_search_8p_c (mcomp.c:1621-1691) with ms_buffers.second_pred [/ mask, inv_mask] set as av1_set_ms_compound_refs does
      (mcomp.h:152-166), followed by av1_get_mvpred_compound_var (:3679-3693) at the MV it returns -- the full-pel half of
      av1_joint_motion_search / av1_compound_single_motion_search (motion_search_facade.c:496-870);
  av1_obmc_full_pixel_search (mcomp.c:2272-2285), both forms: obmc_full_pixel_diamond (fast_obmc_search = 0) and
      obmc_refining_search_sad (1), with ms_buffers.wsrc / obmc_mask.

The vtable members are the reference's own functions: sdaf = aom_sad{W}x{H}_avg_c, msdf = aom_masked_sad{W}x{H}_c, svaf =
aom_sub_pixel_avg_variance{W}x{H}_c, msvf = aom_masked_sub_pixel_variance{W}x{H}_c, osdf = aom_obmc_sad{W}x{H}_c, ovf =
aom_obmc_variance{W}x{H}_c; 10-bit: the _bits10 wrappers of av1/encoder/encoder_utils.h and the aom_highbd_10_* variance forms.

Output: tests/golden/ref_eval_compound_search.npz.
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

REF = "/root/reference/"


def make_evaluator():
    return M.make_evaluator(with_compound=True)


def extend_vtable(ev, vfp, bd, w, h):
    if bd == 8:
        names = dict(sdaf="aom_sad%dx%d_avg_c", msdf="aom_masked_sad%dx%d_c", svaf="aom_sub_pixel_avg_variance%dx%d_c",
                     msvf="aom_masked_sub_pixel_variance%dx%d_c", osdf="aom_obmc_sad%dx%d_c", ovf="aom_obmc_variance%dx%d_c")
    else:
        names = dict(sdaf="aom_highbd_sad%dx%d_avg_bits10", msdf="aom_highbd_masked_sad%dx%d_bits10",
                     svaf="aom_highbd_10_sub_pixel_avg_variance%dx%d_c", msvf="aom_highbd_10_masked_sub_pixel_variance%dx%d_c",
                     osdf="aom_highbd_obmc_sad%dx%d_bits10", ovf="aom_highbd_10_obmc_variance%dx%d_c")
    for k, pat in names.items():
        fn = pat % (w, h)
        assert fn in ev.funcs, fn
        ev.set(vfp, k, R.FuncRef(fn))


def main():
    ev = make_evaluator()
    arrays, cases = {}, []
    rng = np.random.default_rng(20261101)
    mvc = M.synth_mv_costs(11)
    arrays["mvjcost"], arrays["mvcost0"], arrays["mvcost1"] = mvc
    harness = {}
    for bd in (8, 10):
        s, r = M.synth_planes(bd, 300 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s, r
        harness[bd] = M.Harness(ev, bd, s, r, mvc)
    W, H, B = M.W, M.H, M.BORDER
    t0 = time.time()
    k = 0
    sizes = [(16, 16), (8, 8), (32, 16), (16, 8), (8, 16)]
    # ---- av1_refining_search_8p_c + av1_get_mvpred_compound_var
    for bd in (8, 10):
        hs = harness[bd]
        mx = (1 << bd) - 1
        ct = "uint8_t" if bd == 8 else "uint16_t"
        for (w, h) in sizes:
            for trial in range(4):
                masked = trial >= 2
                inv = trial == 3
                cost_type = ["ENTROPY", "L1_HDRES", "NONE", "L1_LOWRES"][trial]
                edge = trial == 1
                bx = int(rng.choice([0, W - w])) if edge else int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = int(rng.choice([0, H - h])) if edge else int(rng.integers(0, (H - h) // 4 + 1)) * 4
                lim = M.limits(bx, by, w, h, 6 if edge else None)
                start = (int(rng.integers(-5, 6)), int(rng.integers(-5, 6)))
                refmv = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                blk = (bx, by, start[0], start[1], refmv[0], refmv[1]) + lim
                ms = hs.fullpel_params(blk, w, h, "NSTEP", cost_type, sad_per_bit=int(rng.integers(8, 40)), error_per_bit=int(rng.integers(20, 120)))
                vfp = ev.get(ms, "vfp")
                extend_vtable(ev, vfp, bd, w, h)
                # the other reference's predictor: the reference block near (start + a small offset) plus noise (what
                # av1_enc_build_one_inter_predictor would have produced is an input here)
                oy, ox = by + start[0] + int(rng.integers(-2, 3)), bx + start[1] + int(rng.integers(-2, 3))
                refpl = arrays["ref%d" % bd]
                sp = refpl[B + oy:B + oy + h, B + ox:B + ox + w].astype(np.int32) + rng.integers(-(6 << (bd - 8)), (6 << (bd - 8)) + 1, (h, w))
                sp = np.clip(sp, 0, mx).astype(np.uint16)
                SP = ev.array(sp.ravel(), ct)
                ev.set(ms, "ms_buffers.second_pred", SP)
                mask = None
                if masked:
                    # a wedge-like ramp with noise, values 0..64, stride = w (the stride av1_joint_motion_search passes for its masks is the block's)
                    ramp = np.clip((np.arange(w)[None, :] * 2 + np.arange(h)[:, None] - (w + h) // 2) * 4 + 32 + rng.integers(-3, 4, (h, w)), 0, 64)
                    mask = ramp.astype(np.uint8)
                    ev.set(ms, "ms_buffers.mask", ev.array(mask.ravel(), "uint8_t"))
                    ev.set(ms, "ms_buffers.mask_stride", w)
                    ev.set(ms, "ms_buffers.inv_mask", int(inv))
                startmv = hs.mv_struct("FULLPEL_MV", start[0], start[1])
                best = ev.new("FULLPEL_MV")
                sad = ev.call("av1_refining_search_8p_c", ms, startmv.buf[0], best)
                mv = [ev.get(best, "row"), ev.get(best, "col")]
                var = ev.call("av1_get_mvpred_compound_var", ev.field(ms, "mv_cost_params"),
                              best.buf[0], SP, ev.get(ms, "ms_buffers.mask") if masked else None, w if masked else 0, int(inv), vfp,
                              ev.get(ms, "ms_buffers.src"), ev.get(ms, "ms_buffers.ref"))
                arrays["sp%d" % k] = sp
                if masked:
                    arrays["mask%d" % k] = mask
                cases.append(dict(kind="refine8p", k=k, bd=bd, w=w, h=h, block=list(blk), cost_type=M.COST_TYPES[cost_type], masked=int(masked), inv=int(inv),
                                  sad_per_bit=ev.get(ms, "mv_cost_params.sad_per_bit"), error_per_bit=ev.get(ms, "mv_cost_params.error_per_bit"),
                                  mv=mv, sad=sad, var=var))
                k += 1
    print("refine8p: %d cases, %.0f s" % (len(cases), time.time() - t0))
    n0 = len(cases)
    # ---- av1_obmc_full_pixel_search
    for bd in (8, 10):
        hs = harness[bd]
        mx = (1 << bd) - 1
        for (w, h) in sizes[:4]:
            for trial in range(4):
                fast = trial % 2
                method = ["NSTEP", "DIAMOND", "NSTEP", "CLAMPED_DIAMOND"][trial]
                cost_type = ["ENTROPY", "L1_HDRES", "NONE", "L1_MIDRES"][trial]
                step_param = [4, 5, 3, 6][trial] if not fast else 0
                edge = trial == 3
                bx = int(rng.choice([0, W - w])) if edge else int(rng.integers(0, (W - w) // 4 + 1)) * 4
                by = int(rng.choice([0, H - h])) if edge else int(rng.integers(0, (H - h) // 4 + 1)) * 4
                lim = M.limits(bx, by, w, h, 10 if edge else 24)
                start = (int(rng.integers(-4, 5)), int(rng.integers(-4, 5)))
                refmv = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                blk = (bx, by, start[0], start[1], refmv[0], refmv[1]) + lim
                ms = hs.fullpel_params(blk, w, h, method, cost_type, sad_per_bit=int(rng.integers(8, 40)), error_per_bit=int(rng.integers(20, 120)))
                vfp = ev.get(ms, "vfp")
                extend_vtable(ev, vfp, bd, w, h)
                # calc_target_weighted_pred's outputs (reconinter_enc / rdopt): mask = the block's own weight (x 64 x 64 = 4096 where no neighbour
                # overlaps), wsrc = src * 4096 - neighbours' predictions * (4096 - mask)
                srcpl = arrays["src%d" % bd]
                sblk = srcpl[B + by:B + by + h, B + bx:B + bx + w].astype(np.int64)
                om = np.full((h, w), 4096, np.int64)
                om[:h // 2, :] = (np.linspace(36, 64, h // 2).astype(np.int64)[:, None]) * 64
                om[:, :w // 2] = np.minimum(om[:, :w // 2], (np.linspace(34, 64, w // 2).astype(np.int64)[None, :]) * 64)
                nb = np.clip(sblk + rng.integers(-(10 << (bd - 8)), (10 << (bd - 8)) + 1, (h, w)), 0, mx)
                ws = sblk * 4096 - nb * (4096 - om)
                ev.set(ms, "ms_buffers.wsrc", ev.array(ws.ravel().astype(np.int64), "int32_t"))
                ev.set(ms, "ms_buffers.obmc_mask", ev.array(om.ravel(), "int32_t"))
                ev.set(ms, "fast_obmc_search", fast)
                startmv = hs.mv_struct("FULLPEL_MV", start[0], start[1])
                best = ev.new("FULLPEL_MV")
                cost = ev.call("av1_obmc_full_pixel_search", startmv.buf[0], ms, step_param, best)
                arrays["ws%d" % k], arrays["om%d" % k] = ws.astype(np.int32), om.astype(np.int32)
                cases.append(dict(kind="obmc", k=k, bd=bd, w=w, h=h, block=list(blk), method=method, step_param=step_param, fast=fast,
                                  cost_type=M.COST_TYPES[cost_type], sad_per_bit=ev.get(ms, "mv_cost_params.sad_per_bit"),
                                  error_per_bit=ev.get(ms, "mv_cost_params.error_per_bit"), mv=[ev.get(best, "row"), ev.get(best, "col")], cost=cost))
                k += 1
    print("obmc: %d cases, %.0f s" % (len(cases) - n0, time.time() - t0))
    meta = dict(border=B, width=W, height=H, generated_by="tests/golden/gen_ref_eval_compound_search.py", cases=cases)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "ref_eval_compound_search.npz"), **arrays)
    print("wrote ref_eval_compound_search.npz: %d cases" % len(cases))


if __name__ == "__main__":
    main()
