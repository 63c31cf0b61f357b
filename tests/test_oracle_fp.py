"""first_pass_motion_search's search leg (second call site; av1/encoder/firstpass.c:261-299): the oracle's composition
against the vectors tests/golden/gen_ref_eval_fp.py produced with the interpreted reference (av1_init_motion_fpf, av1_full_pixel_search with
the default MV_COST_ENTROPY parameters, av1_get_mvpred_sse)."""
import json
import os

import numpy as np

from conftest import ROOT

BLOCK_FIELDS = ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")


def fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_eval_fp.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def block_of(case, dtype):
    b = np.zeros(1, dtype)
    for k, v in zip(BLOCK_FIELDS, case["block"]):
        b[k] = v
    return b


def test_oracle_composition_reproduces_the_interpreted_reference(oracle):
    z, meta = fixture()
    dtype = np.dtype([(n, "<i2") for n in BLOCK_FIELDS])
    n_chained = 0
    for c in meta["cases"]:
        q = oracle.search_params("NSTEP_FPF", c["step_param"], 0, sad_per_bit=c["sad_per_bit"], error_per_bit=c["error_per_bit"], no_cost_list=1)
        mv, err = oracle.first_pass_motion_search_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], block_of(c, dtype), q,
                                                        z["mvjcost"], z["mvcost0"], z["mvcost1"], bd=c["bd"], threads=1)
        assert mv[0].tolist() == c["mv"] and int(err[0]) == c["err"], c
        n_chained += c["block"][4] != 0 or c["block"][5] != 0
    assert len(meta["cases"]) >= 12 and n_chained >= 6
