"""What av1_single_motion_search adds to the plain searches, oracle against tests/golden/ref_eval_single.npz (the reference's own functions
interpreted, tests/golden/gen_ref_eval_single.py): av1_full_pixel_search's second_best_mv, the three sub-pel trees called repeatedly on one
last_mv_search_list (check_repeated_mv_and_update: where a search stops with INT_MAX and what it leaves in bestmv / distortion / sse1 / the
list), and av1_mv_bit_cost."""
import json
import os

import numpy as np

from conftest import ROOT
from test_oracle_fp import BLOCK_FIELDS, block_of

INT_MAX = 2147483647
TREE = {"av1_find_best_sub_pixel_tree_pruned_more": "pruned_more", "av1_find_best_sub_pixel_tree_pruned": "pruned", "av1_find_best_sub_pixel_tree": "tree"}


def fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_eval_single.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def subpel_block(c, start, dtype):
    b = block_of(c, dtype)
    b["start_row"], b["start_col"] = start
    b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["subpel_limits"]
    return b


def test_oracle_equals_the_interpreted_reference(oracle):
    z, meta = fixture()
    dtype = np.dtype([(n, "<i2") for n in BLOCK_FIELDS])
    stopped = {0: 0, 1: 0, 2: 0}
    for c in meta["cases"]:
        src, ref, bd = z["src%d" % c["bd"]], z["ref%d" % c["bd"]], c["bd"]
        q = oracle.search_params("NSTEP", c["step_param"], c["cost_type"], sad_per_bit=c["sad_per_bit"], error_per_bit=c["error_per_bit"])
        mv, cost, cl, sec = oracle.full_pixel_search_batch(src, ref, meta["border"], c["w"], c["h"], block_of(c, dtype), q, z["mvjcost"], z["mvcost0"],
                                                           z["mvcost1"], bd=bd, threads=1)
        assert mv[0].tolist() == c["full_mv"] and int(cost[0]) == c["full_cost"] and sec[0].tolist() == c["second_best"] and cl[0].tolist() == c["cost_list"], c
        lst = np.full((1, 3, 2), -32768, np.int16)
        for call in c["calls"]:
            lst[0] = np.asarray(call["list_before"], np.int16)
            m, err, dist, sse = oracle.subpel_tree_batch(src, ref, meta["border"], c["w"], c["h"], subpel_block(c, call["start"], dtype), tree=TREE[c["fn"]],
                                                         cost_type=c["cost_type"], error_per_bit=c["error_per_bit"], mvjcost=z["mvjcost"], mvcost0=z["mvcost0"],
                                                         mvcost1=z["mvcost1"], iters=c["iters"], allow_hp=c["allow_hp"], forced_stop=c["forced_stop"],
                                                         cost_lists=np.asarray([c["cost_list"]], np.int32) if c["use_cost_list"] else None, bd=bd, threads=1,
                                                         mv_lists=lst)
            got = (int(np.int32(err[0])), m[0].tolist(), int(dist[0]), int(sse[0]), lst[0].tolist())
            want = (call["err"], call["mv"], call["distortion"], call["sse"], call["list_after"])
            assert got == want, (c["fn"], c["bd"], call, got)
            if call["err"] == INT_MAX:
                k = next(i for i in range(3) if call["list_before"][i] == call["list_after"][i] and call["list_after"][i][0] != -32768
                         and (i == 2 or call["list_before"][i + 1] == call["list_after"][i + 1]))
                stopped[k] += 1
        for call, want in zip(c["calls"], c["mv_bit_cost"]):
            assert oracle.mv_bit_cost(call["mv"][0], call["mv"][1], c["block"][4], c["block"][5], z["mvjcost"], z["mvcost0"], z["mvcost1"]) == want
    assert len(meta["cases"]) == 66 and sum(stopped.values()) >= 50 and stopped[1] >= 10 and stopped[2] >= 5, stopped
