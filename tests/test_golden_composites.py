"""The SEQUENCING of the search composites, pinned: the oracle's compositions (oracle/pyoracle.py: tf_motion_search_frames,
first_pass_inter_frame, simple_motion_search_batch) against values obtained by interpreting the reference's CALLER functions themselves --
tf_motion_search (temporal_filter.c:87-253), firstpass_inter_prediction + first_pass_motion_search (firstpass.c:261-299, :690-815),
av1_simple_motion_search (motion_search_facade.c:925-1030) -- on views of the encoder's objects
(tests/golden/ref_eval_composites.npz, generator tests/golden/gen_ref_eval_composites.py).  The GPU tests check the kernels against the same
oracle compositions (tests/test_gpu_tf.py, test_gpu_fp_frame.py, test_gpu_simple_motion.py) and, for these cases, against the fixture itself
(tests/test_gpu_composites.py)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOOD_MESH = [(64, 8), (28, 4), (15, 1), (7, 1)]
TREES = {"SUBPEL_TREE": "tree", "SUBPEL_TREE_PRUNED": "pruned", "SUBPEL_TREE_PRUNED_MORE": "pruned_more"}
BLOCK_DT = np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")])


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_composites.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def tf_case_inputs(oracle, z, meta, c):
    s = c["spec"]
    frames = [np.ascontiguousarray(z["%s_frame%d" % (c["tag"], f)]) for f in range(3)]
    # window order of av1_tf_do_filtering_row: frame 0, the frame to filter (index 1), frame 2
    p = oracle.tf_params(meta["width"], meta["height"], c["bd"], s["q"], s["prune_mesh_search"], GOOD_MESH, subpel_tree=TREES[s.get("subpel_search_method", "SUBPEL_TREE")],
                         iters_per_step=s.get("subpel_iters_per_step", 2), allow_hp=s.get("allow_hp", 1), use_cost_list=s.get("use_fullpel_costlist", 0),
                         use_downsampled_sad=0, force_integer_mv=s.get("force_integer_mv", 0))
    b = np.zeros(1, BLOCK_DT)
    b["bx"], b["by"] = c["mb_col"] * 32, c["mb_row"] * 32
    b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
    return frames, p, b


def check_tf(c, mvs, mses, ref_mv):
    for f, rec in zip((0, 2), c["chain"]):
        assert mvs[f, 0].tolist() == rec["sub_mvs"], (c["tag"], c["mb_row"], c["mb_col"], f)
        if not c["spec"].get("force_integer_mv", 0):
            assert mses[f, 0].tolist() == rec["sub_mses"], (c["tag"], c["mb_row"], c["mb_col"], f)
    assert ref_mv[0].tolist() == c["chain"][-1]["ref_mv"]


def test_tf_motion_search_sequencing(oracle):
    z, meta = load()
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "tf":
            continue
        frames, p, b = tf_case_inputs(oracle, z, meta, c)
        mvs, mses, ref_mv = oracle.tf_motion_search_frames(frames, 1, meta["border"], b, p, threads=4)
        check_tf(c, mvs, mses, ref_mv)
        n += 1
    assert n >= 6


def fp_case_inputs(oracle, z, meta, c):
    tag, bs, W, H, B = c["tag"], c["bs"], meta["width"], meta["height"], meta["border"]
    cols = W // bs
    blocks = np.zeros(cols, BLOCK_DT)
    blocks["bx"], blocks["by"] = np.arange(cols) * bs, c["unit_row"] * bs
    for i in range(cols):
        blocks["row_min"][i], blocks["row_max"][i], blocks["col_min"][i], blocks["col_max"][i] = oracle.mv_limits_for_block(int(blocks["bx"][i]), int(blocks["by"][i]), bs, bs, W, H, B)
    planes = [np.ascontiguousarray(z[tag + k]) for k in ("_src", "_last", "_golden", "_lastsrc")]
    return planes, blocks, cols


def test_firstpass_inter_prediction_sequencing(oracle):
    z, meta = load()
    n = moved = 0
    for c in meta["cases"]:
        if c["kind"] != "fp":
            continue
        (src, last, golden, lsrc), blocks, cols = fp_case_inputs(oracle, z, meta, c)
        s = c["spec"]
        # first_pass_motion_search: step_param = reduce_mv_step_param (3) + get_search_range(min(W, H) = 96) (firstpass.c:252-259, :270-271)
        sr = 0
        while (min(meta["width"], meta["height"]) << sr) < 1023:
            sr += 1
        q = oracle.search_params("NSTEP_FPF", 3 + sr, 0, sad_per_bit=20, error_per_bit=60, no_cost_list=1)
        intra = np.array([r["intra"] for r in c["row"]], np.int32)
        best_mv, full_mv, err, gf, raw = oracle.first_pass_inter_frame(src, last, golden if s["golden"] else None, lsrc, meta["border"], c["bs"], blocks, 1, cols, q,
                                                                       intra, s["thr"], s["skip_zeromv"], z["mvjcost"], z["mvcost0"], z["mvcost1"], bd=c["bd"])
        for i, r in enumerate(c["row"]):
            assert best_mv[i].tolist() == r["best_mv"], (c["tag"], c["unit_row"], i)
            assert int(raw[i]) == r["raw"], (c["tag"], c["unit_row"], i)
            this_inter = int(err[i]) if int(err[i]) <= r["intra"] else r["intra"]     # the function returns this_inter_error (:694, :770)
            assert this_inter == r["inter"], (c["tag"], c["unit_row"], i)
            moved += r["best_mv"] != [0, 0]
        n += 1
    assert n >= 12 and moved >= 30


def test_simple_motion_search_sequencing(oracle):
    z, meta = load()
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "sms":
            continue
        s = c["spec"]
        b = np.zeros(1, BLOCK_DT)
        b["bx"], b["by"], b["start_row"], b["start_col"] = c["bx"], c["by"], c["start"][0], c["start"][1]
        b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
        step_param = min(s["mv_step_param"] + s["reduce"], 11 - 2)     # AOMMIN(.., MAX_MVSEARCH_STEPS - 2) (:955-958)
        q = oracle.search_params(s["search_method"], step_param, 0, sad_per_bit=c["sadperbit"], error_per_bit=c["errorperbit"],
                                 no_cost_list=int(not (s["costlist"] and s["tree"] != "SUBPEL_TREE")))
        sub = None
        if s["subpel"]:
            sub = dict(tree=TREES[s["tree"]], cost_type=0, error_per_bit=c["errorperbit"], iters=2, allow_hp=1, forced_stop=s["force_stop"], subpel_search_type=0)
        src, ref = np.ascontiguousarray(z["sms_src%d" % c["bd"]]), np.ascontiguousarray(z["sms_ref%d" % c["bd"]])
        mv, _, _, _ = oracle.simple_motion_search_batch(src, ref, meta["border"], meta["width"], meta["height"], c["w"], c["h"], b, q, sub,
                                                        use_cost_list=int(s["costlist"] and s["tree"] != "SUBPEL_TREE"), mvjcost=z["mvjcost"], mvcost0=z["mvcost0"],
                                                        mvcost1=z["mvcost1"], bd=c["bd"], threads=1)
        assert mv[0].tolist() == c["mv"], c
        n += 1
    assert n >= 24
