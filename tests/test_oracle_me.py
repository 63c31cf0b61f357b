"""The MV-limit derivations a search around a non-zero reference MV needs -- av1_set_mv_search_range (mcomp.c:196-215) and
av1_set_subpel_mv_search_range (mcomp.h:344-361) -- in the oracle against the interpreted reference (tests/golden/ref_eval_mvlimits.npz,
gen_ref_eval_mvlimits.py); they feed oracle.motion_estimation_batch, the restatement of tpl_model.c's motion_estimation."""
import os

import numpy as np

from conftest import ROOT


def test_limit_derivations_reproduce_the_interpreted_reference(oracle):
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_eval_mvlimits.npz"))
    assert len(z["raw"]) >= 300
    for raw, ref, full, sub in zip(z["raw"], z["ref"], z["full"], z["sub"]):
        assert oracle.set_mv_search_range(raw, ref[0], ref[1]) == full.tolist(), (raw, ref)
        assert oracle.set_subpel_mv_search_range(raw, ref[0], ref[1]) == sub.tolist(), (raw, ref)
    assert (z["full"] != np.clip(z["raw"], -1023, 1023)).any()     # the reference MV moves the window in some cases
