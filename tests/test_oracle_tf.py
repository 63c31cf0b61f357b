"""tf_motion_search over a filter window (av1/encoder/temporal_filter.c:87-293, 849-867): the oracle's composition
(oracle/pyoracle.py tf_motion_search_frames) against the vectors tests/golden/gen_ref_eval_tf.py produced by driving the interpreted
reference (ref_eval_tf.npz), and the product's host helpers (aom-av1-psy_amd/host/aomhip_tf.c, no GPU) against the oracle's."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT

GOOD_MESH = [(64, 8), (28, 4), (15, 1), (7, 1)]


def _fixture():
    z = np.load(os.path.join(ROOT, "tests", "golden", "ref_eval_tf.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def fixture_cases():
    z, meta = _fixture()
    for ci, case in enumerate(meta["cases"]):
        yield ci, case, [np.ascontiguousarray(f) for f in z["frames%d" % ci]], meta


def oracle_params(oracle, case, meta):
    s = case["spec"]
    p = oracle.tf_params(meta["W"], meta["H"], s["bd"], s["q"], s["prune_level"], GOOD_MESH, subpel_tree=s["tree"], iters_per_step=s["iters"],
                         allow_hp=s["allow_hp"], use_cost_list=s["use_cost_list"], use_downsampled_sad=s["skip_sad"],
                         force_integer_mv=s["force_integer_mv"])
    want = case["params"]                      # what the generator derived with the reference's av1_init_search_range etc.
    for k in ("step_param", "cost_type", "prune", "thr", "mse_thresh", "tree", "iters", "allow_hp", "use_cost_list", "skip_sad", "force_integer_mv"):
        assert p[k] == want[k], k
    return p


def check_against_fixture(case, meta, mvs, mses, ref_mv):
    mb_cols = -(-meta["W"] // 32)
    n_checked = 0
    for blk in case["blocks"]:
        i = blk["mb_row"] * mb_cols + blk["mb_col"]
        for f, rec in enumerate(blk["frames"]):
            if rec is None:
                assert not mvs[f, i].any() and (mses[f, i] == 2147483647).all()
                continue
            assert mvs[f, i].tolist() == rec["mvs"], (case["spec"]["name"], blk["mb_row"], blk["mb_col"], f)
            assert mses[f, i].tolist() == rec["mses"], (case["spec"]["name"], blk["mb_row"], blk["mb_col"], f)
            n_checked += 1
        assert ref_mv[i].tolist() == blk["ref_mv_final"]
    return n_checked


def test_oracle_composition_reproduces_the_interpreted_reference(oracle):
    total, split, handed = 0, 0, 0
    for ci, case, frames, meta in fixture_cases():
        p = oracle_params(oracle, case, meta)
        blocks = oracle.tf_block_list(meta["W"], meta["H"], meta["border"])
        mvs, mses, ref_mv = oracle.tf_motion_search_frames(frames, case["spec"]["filter_frame"], meta["border"], blocks, p, threads=8)
        total += check_against_fixture(case, meta, mvs, mses, ref_mv)
        for blk in case["blocks"]:
            for rec in blk["frames"]:
                if rec is not None:
                    split += len({tuple(m) for m in rec["mvs"]}) > 1
                    handed += rec["ref_mv_after"] != [0, 0]
    assert total >= 20 and split >= 2 and handed >= 2      # the vectors exercise the split decision and the ref_mv hand-over


@pytest.mark.parametrize("w,h,border", [(96, 96, 48), (352, 288, 64), (1920, 1080, 160), (3840, 2160, 160), (100, 70, 32), (31, 33, 32)])
def test_host_block_list_matches_the_oracle(hip, oracle, w, h, border):
    got = hip.capi.tf_block_list(w, h, border)
    want = oracle.tf_block_list(w, h, border)
    assert got.dtype == want.dtype and np.array_equal(got, want)
    assert len(got) == -(-w // 32) * -(-h // 32)
    # every block + MV stays inside the bordered plane with 2 * AOM_INTERP_EXTEND to spare
    ah, aw = (h + 7) & ~7, (w + 7) & ~7
    assert (got["by"] + got["row_min"] >= -border + 8).all() and (got["by"] + 32 + got["row_max"] <= ah + border - 8).all()
    assert (got["bx"] + got["col_min"] >= -border + 8).all() and (got["bx"] + 32 + got["col_max"] <= aw + border - 8).all()


@pytest.mark.parametrize("w,h,bd,q,level,tree,ucl,fi", [(96, 96, 8, 30, 1, 2, 0, 0), (1920, 1080, 10, 12, 1, 0, 1, 0), (3840, 2160, 12, 60, 2, 1, 1, 0),
                                                       (640, 480, 8, 20, 0, 2, 0, 1), (1280, 720, 10, 21, 1, 2, 0, 0)])
def test_host_default_params_match_the_oracle(hip, oracle, w, h, bd, q, level, tree, ucl, fi):
    t = hip.capi.TfParams.default(w, h, bd, q, level, GOOD_MESH, subpel_tree=tree, iters_per_step=2, allow_hp=1, use_cost_list=ucl,
                                  use_downsampled_sad=1, force_integer_mv=fi)
    p = oracle.tf_params(w, h, bd, q, level, GOOD_MESH, subpel_tree=tree, use_cost_list=ucl, use_downsampled_sad=1, force_integer_mv=fi)
    assert (t.full.search_method, t.full.step_param, t.full.mv_cost_type, t.full.run_mesh_search) == (1, p["step_param"], p["cost_type"], 1)
    assert (t.full.prune_mesh_search, t.full.mesh_search_mv_diff_threshold, t.full.use_downsampled_sad) == (p["prune"], p["thr"], 1)
    assert [t.full.mesh_patterns[i] for i in range(8)] == [v for pr in GOOD_MESH for v in pr]
    assert (t.sub.tree, t.sub.mv_cost_type, t.sub.forced_stop, t.sub.subpel_search_type, t.sub.iters_per_step, t.sub.allow_hp) == (tree, 4, 0, 3, 2, 1)
    assert (t.use_cost_list, t.force_integer_mv, t.mse_thresh) == (ucl, fi, p["mse_thresh"])
