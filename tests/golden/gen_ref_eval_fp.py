#!/usr/bin/env python3
"""Golden vectors for the first pass's motion-search leg (build container only; output tests/golden/ref_eval_fp.npz).

first_pass_motion_search (av1/encoder/firstpass.c:261-299) takes the encoder instance, so -- as for the temporal filter -- its body is
driven from here with every computation done by the interpreted reference: av1_init_motion_fpf builds the site table, av1_full_pixel_search
runs NSTEP on it from get_fullmv_from_mv(ref_mv) with the default MV_COST_ENTROPY parameters (init_mv_cost_params, mcomp.c:35-52), and
av1_get_mvpred_sse (mcomp.c:3637-3649) gives the error the function adds NEW_MV_MODE_PENALTY to.  Planes and cost tables are those of
tests/golden/ref_eval_mcomp.npz (same generator functions, same seeds)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402

INT_MAX = 2147483647


def main():
    ev = G.make_evaluator()
    rng = np.random.default_rng(20261101)
    mvc = G.synth_mv_costs(7)
    arrays = {"mvjcost": mvc[0], "mvcost0": mvc[1], "mvcost1": mvc[2]}
    cases = []
    penalty = 32                                                    # NEW_MV_MODE_PENALTY (firstpass.c:52)
    for bd in (8, 10):
        s, r = G.synth_planes(bd, 100 + bd)
        arrays["src%d" % bd], arrays["ref%d" % bd] = s, r
        hs = G.Harness(ev, bd, s, r, mvc)
        for trial in range(7):
            w, h = ((16, 16), (16, 16), (8, 8), (16, 8), (8, 16), (16, 16), (16, 16))[trial]
            bx, by = int(rng.integers(0, (G.W - w) // 4 + 1)) * 4, int(rng.integers(0, (G.H - h) // 4 + 1)) * 4
            ref_mv = (0, 0) if trial in (0, 5) else (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))   # the zero-MV legs and the chained one
            refmv = hs.mv_struct("MV", ref_mv[0], ref_mv[1])
            full = ev.interp.call("get_fullmv_from_mv", [(refmv, R.PTR)])[0]
            st = ev.new("FULLPEL_MV"); st.store(full, full.st)
            start = (ev.get(st, "row"), ev.get(st, "col"))
            blk = (bx, by, start[0], start[1], ref_mv[0], ref_mv[1]) + G.limits(bx, by, w, h, 30)
            step_param = int(rng.integers(0, 4))
            spb, epb = int(rng.integers(8, 40)), int(rng.integers(20, 120))
            ms = hs.fullpel_params(blk, w, h, "NSTEP_FPF", "ENTROPY", sad_per_bit=spb, error_per_bit=epb)
            best = ev.new("FULLPEL_MV")
            cost = ev.call("av1_full_pixel_search", st.buf[0], ms, step_param, None, best, None)
            err = INT_MAX
            if cost < INT_MAX:
                sse = ev.call("av1_get_mvpred_sse", ev.field(ms, "mv_cost_params"), best.buf[0], ev.get(ms, "vfp"), ev.get(ms, "ms_buffers.src"),
                              ev.get(ms, "ms_buffers.ref"))
                err = sse + penalty
            cases.append(dict(bd=bd, w=w, h=h, block=list(blk), step_param=step_param, sad_per_bit=spb, error_per_bit=epb,
                              mv=[ev.get(best, "row"), ev.get(best, "col")], search_cost=cost, err=err))
            print(cases[-1], flush=True)
    path = os.path.join(HERE, "ref_eval_fp.npz")
    np.savez_compressed(path, cases=np.frombuffer(json.dumps({"cases": cases, "W": G.W, "H": G.H, "border": G.BORDER}).encode(), np.uint8), **arrays)
    print("ref_eval_fp.npz: %d cases, %.1f KB" % (len(cases), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
