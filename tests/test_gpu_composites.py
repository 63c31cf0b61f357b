"""The device composites straight against the values obtained by interpreting the reference's CALLER functions
(tests/golden/ref_eval_composites.npz): aomhip_tf_motion_search_frames vs tf_motion_search, aomhip_first_pass_inter_frame vs
firstpass_inter_prediction + first_pass_motion_search along block rows, aomhip_simple_motion_search_batch vs av1_simple_motion_search."""
import numpy as np
import pytest

from test_golden_composites import GOOD_MESH, TREES, check_tf, load

pytestmark = pytest.mark.gpu


def test_tf_frames_call_reproduces_the_interpreted_tf_motion_search(hip, ctx):
    capi = hip.capi
    z, meta = load()
    W, H, B = meta["width"], meta["height"], meta["border"]
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "tf":
            continue
        s = c["spec"]
        frames = [np.ascontiguousarray(z["%s_frame%d" % (c["tag"], f)][B:B + H, B:B + W]) for f in range(3)]
        planes = ctx.planes_alloc(W, H, B, c["bd"], 3)
        for f, fr in enumerate(frames):
            ctx.planes_upload(planes, f, fr)
        tp = capi.TfParams.default(W, H, c["bd"], s["q"], s["prune_mesh_search"], GOOD_MESH, subpel_tree=TREES[s.get("subpel_search_method", "SUBPEL_TREE")],
                                   iters_per_step=s.get("subpel_iters_per_step", 2), allow_hp=s.get("allow_hp", 1), use_cost_list=s.get("use_fullpel_costlist", 0),
                                   use_downsampled_sad=0, force_integer_mv=s.get("force_integer_mv", 0))
        b = np.zeros(1, capi.search_block_dtype)
        b["bx"], b["by"] = c["mb_col"] * 32, c["mb_row"] * 32
        b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
        d_b = ctx.to_device(b)
        d_mv, d_mse, d_ref = ctx.malloc(3 * 16), ctx.malloc(3 * 16), ctx.malloc(4)
        ctx.tf_motion_search_frames(planes, 1, tp, d_b, 1, d_mv, d_mse, d_ref, None)
        check_tf(c, ctx.from_device(d_mv, (3, 1, 4, 2), np.int16), ctx.from_device(d_mse, (3, 1, 4), np.int32), ctx.from_device(d_ref, (1, 2), np.int16))
        for d in (d_b, d_mv, d_mse, d_ref):
            ctx.free(d)
        ctx.planes_free(planes)
        n += 1
    assert n >= 8


@pytest.mark.parametrize("form", ["rows", "columns"])
def test_first_pass_frame_call_reproduces_the_interpreted_firstpass_inter_prediction(hip, oracle, ctx, form, monkeypatch):
    monkeypatch.setenv("AOMHIP_FP_COLUMNS", "1" if form == "columns" else "0")
    capi = hip.capi
    z, meta = load()
    W, H, B = meta["width"], meta["height"], meta["border"]
    mv_max = z["mvcost0"].size // 2
    d_j, d_c0, d_c1 = ctx.to_device(z["mvjcost"].astype(np.int32)), ctx.to_device(z["mvcost0"].astype(np.int32)), ctx.to_device(z["mvcost1"].astype(np.int32))
    sr = 0
    while (min(W, H) << sr) < 1023:       # get_search_range (firstpass.c:252-259)
        sr += 1
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "fp":
            continue
        s, bs, tag = c["spec"], c["bs"], c["tag"]
        cols = W // bs
        rings = [ctx.planes_alloc(W, H, B, c["bd"], 1) for _ in range(4)]
        for ring, k in zip(rings, ("_src", "_last", "_golden", "_lastsrc")):
            ctx.planes_upload(ring, 0, np.ascontiguousarray(z[tag + k][B:B + H, B:B + W]))
        ps, pl, pg, pls = rings
        blocks = np.zeros(cols, capi.search_block_dtype)
        blocks["bx"], blocks["by"] = np.arange(cols) * bs, c["unit_row"] * bs
        for i in range(cols):
            blocks["row_min"][i], blocks["row_max"][i], blocks["col_min"][i], blocks["col_max"][i] = oracle.mv_limits_for_block(int(blocks["bx"][i]), int(blocks["by"][i]),
                                                                                                                          bs, bs, W, H, B)
        intra = np.array([r["intra"] for r in c["row"]], np.int32)
        q = capi.SearchParams.make("NSTEP_FPF", 3 + sr, capi.MV_COST_ENTROPY, sad_per_bit=20, error_per_bit=60)
        fp = capi.FirstPassParams(1, cols, s["thr"], s["skip_zeromv"])
        d_b, d_i = ctx.to_device(blocks), ctx.to_device(intra)
        outs = [ctx.malloc(cols * 4) for _ in range(5)]
        ctx.first_pass_inter_frame(ps, 0, pl, 0, pg if s["golden"] else None, 0, pls, 0, bs, bs, q, fp, d_b, d_i, outs[0], outs[2], outs[1], outs[3], outs[4], d_j,
                                   d_c0 + mv_max * 4, d_c1 + mv_max * 4)
        best, err, raw = ctx.from_device(outs[0], (cols, 2), np.int16), ctx.from_device(outs[2], (cols,), np.int32), ctx.from_device(outs[4], (cols,), np.int32)
        for i, r in enumerate(c["row"]):
            assert best[i].tolist() == r["best_mv"] and int(raw[i]) == r["raw"], (tag, c["unit_row"], i)
            assert (int(err[i]) if int(err[i]) <= r["intra"] else r["intra"]) == r["inter"], (tag, c["unit_row"], i)
        for d in [d_b, d_i] + outs:
            ctx.free(d)
        for ring in rings:
            ctx.planes_free(ring)
        n += 1
    assert n >= 12
    for d in (d_j, d_c0, d_c1):
        ctx.free(d)


def test_simple_motion_search_batch_reproduces_the_interpreted_av1_simple_motion_search(hip, ctx):
    capi = hip.capi
    z, meta = load()
    W, H, B = meta["width"], meta["height"], meta["border"]
    mv_max = z["mvcost0"].size // 2
    d_j, d_c0, d_c1 = ctx.to_device(z["mvjcost"].astype(np.int32)), ctx.to_device(z["mvcost0"].astype(np.int32)), ctx.to_device(z["mvcost1"].astype(np.int32))
    planes = {}
    for bd in (8, 10):
        ps, pr, pp = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(ps, 0, np.ascontiguousarray(z["sms_src%d" % bd][B:B + H, B:B + W])); ctx.planes_upload(pr, 0, np.ascontiguousarray(z["sms_ref%d" % bd][B:B + H, B:B + W]))
        planes[bd] = (ps, pr, pp)
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "sms":
            continue
        s = c["spec"]
        ps, pr, pp = planes[c["bd"]]
        b = np.zeros(1, capi.search_block_dtype)
        b["bx"], b["by"], b["start_row"], b["start_col"] = c["bx"], c["by"], c["start"][0], c["start"][1]
        b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
        ucl = int(s["costlist"] and s["tree"] != "SUBPEL_TREE")      # cond_cost_list (encoder.h:3936-3940)
        full = capi.SearchParams.make(s["search_method"], min(s["mv_step_param"] + s["reduce"], 9), capi.MV_COST_ENTROPY, sad_per_bit=c["sadperbit"],
                                      error_per_bit=c["errorperbit"])
        sub = capi.SubpelParams(capi.SUBPEL_TREES[TREES[s["tree"]]], capi.MV_COST_ENTROPY, c["errorperbit"], 2, 1, s["force_stop"], 0) if s["subpel"] else None
        d_b, d_mv = ctx.to_device(b), ctx.malloc(16)
        ctx.simple_motion_search_batch(ps, pr, 0, c["w"], c["h"], full, sub, ucl, d_b, 1, pp, 0, d_mv, None, None, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
        assert ctx.from_device(d_mv, (2,), np.int16).tolist() == c["mv"], c
        ctx.free(d_b); ctx.free(d_mv)
        n += 1
    assert n >= 24
    for d in (d_j, d_c0, d_c1):
        ctx.free(d)
    for trio in planes.values():
        for p in trio:
            ctx.planes_free(p)
