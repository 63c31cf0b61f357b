#!/usr/bin/env python3
"""av1_set_mv_search_range (av1/encoder/mcomp.c:196-215) and av1_set_subpel_mv_search_range (mcomp.h:344-361) evaluated by the interpreted
reference on random x->mv_limits and reference MVs (build container only) -> tests/golden/ref_eval_mvlimits.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_c_eval as R  # noqa: E402
import gen_ref_eval_mcomp as G  # noqa: E402


def main():
    ev = G.make_evaluator()
    rng = np.random.default_rng(20261102)
    n = 300
    raw = np.zeros((n, 4), np.int32)
    raw[:, 0] = -rng.integers(0, 2600, n); raw[:, 1] = rng.integers(0, 2600, n)
    raw[:, 2] = -rng.integers(0, 2600, n); raw[:, 3] = rng.integers(0, 2600, n)
    ref = rng.integers(-16000, 16001, (n, 2)).astype(np.int32)
    ref[:40] = rng.integers(-64, 65, (40, 2))
    ref[40:50] = 0
    full, sub = np.zeros((n, 4), np.int32), np.zeros((n, 4), np.int32)
    keys = ("row_min", "row_max", "col_min", "col_max")
    for i in range(n):
        fl = ev.new("FullMvLimits")
        for k, v in zip(keys, raw[i]):
            ev.set(fl, k, int(v))
        mv = ev.new("MV")
        ev.set(mv, "row", int(ref[i, 0])); ev.set(mv, "col", int(ref[i, 1]))
        sl = ev.new("SubpelMvLimits")
        ev.interp.call("av1_set_subpel_mv_search_range", [(sl, R.PTR), (fl, R.PTR), (mv, R.PTR)])   # reads the RAW limits
        ev.interp.call("av1_set_mv_search_range", [(fl, R.PTR), (mv, R.PTR)])                       # narrows fl in place
        full[i] = [ev.get(fl, k) for k in keys]
        sub[i] = [ev.get(sl, k) for k in keys]
    np.savez_compressed(os.path.join(HERE, "ref_eval_mvlimits.npz"), raw=raw, ref=ref, full=full, sub=sub)
    print("ref_eval_mvlimits.npz:", n, "cases")


if __name__ == "__main__":
    main()
