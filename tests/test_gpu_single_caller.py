"""The device against av1_single_motion_search interpreted AS IT IS WRITTEN (tests/golden/ref_eval_single_caller.npz): the SIMPLE_TRANSLATION cases
through aomhip_single_motion_search_batch (the caller's part -- step_param narrowing, the weight rule -- computed as INTEGRATION.md shows), the
OBMC_CAUSAL cases through aomhip_obmc_full_pixel_search_batch + aomhip_obmc_subpel_tree_batch.  No oracle in between."""
import numpy as np
import pytest

from test_golden_joint import TAPS, TREES
from test_golden_single_caller import load, rawpel, second_candidate, step_param_of

pytestmark = pytest.mark.gpu
INT_MAX = 2147483647


def test_device_matches_the_interpreted_caller(hip, oracle, ctx):
    capi = hip.capi
    z, meta = load()
    B, W, H = meta["border"], meta["width"], meta["height"]
    j, c0, c1 = z["mvjcost"].astype(np.int32), z["mvcost0"].astype(np.int32), z["mvcost1"].astype(np.int32)
    mv_max = c0.size // 2
    d_j, d_c0, d_c1 = ctx.to_device(j), ctx.to_device(c0), ctx.to_device(c1)
    tabs = (d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    planes = {}
    for bd in (8, 10):
        planes[bd] = [ctx.planes_alloc(W, H, B, bd, 1) for _ in range(2)]
        for p_, name in zip(planes[bd], ("src%d", "ref%d")):
            ctx.planes_upload(p_, 0, np.ascontiguousarray(z[name % bd][B:B + H, B:B + W]))
    n = 0
    for c in meta["cases"]:
        bd, w, h, k = c["bd"], c["w"], c["h"], c["k"]
        ps, pr = planes[bd]
        step = step_param_of(oracle, c)      # (site-table radii only: the caller's arithmetic, motion_search_facade.c:229-243)
        sst = TAPS[c["taps"]] if c["accurate"] else 0
        b = np.zeros(1, capi.search_block_dtype)
        b["bx"], b["by"] = c["bx"], c["by"]
        b["ref_row"], b["ref_col"] = c["ref_mv"]
        b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
        outs = [ctx.malloc(16) for _ in range(4)]
        extra = []
        if c["mode"] == "SIMPLE":
            b["start_row"], b["start_col"] = rawpel(c["ref_mv"][0]), rawpel(c["ref_mv"][1])
            kw = {} if "mesh_thr" not in c else dict(force_mesh_thresh=c["mesh_thr"])
            full = capi.SearchParams.make(c["method"], step, 0, c["sadperbit"], c["errorperbit"], mesh_diff_thr=4, mesh=meta["mesh"], **kw)
            sub = capi.SubpelParams(TREES[c["tree"]], 0, c["errorperbit"], 2, 1, 0, sst)
            c2 = second_candidate(c)
            d_b = ctx.to_device(b)
            d_s2 = ctx.to_device(np.array(c2, np.int16)) if c2 is not None else None
            ctx.single_motion_search_batch(ps, pr, 0, w, h, full, sub, d_b, 1, outs[0], outs[1], outs[2], *tabs, d_start2=d_s2, use_cost_list=c["costlist"],
                                           try_second_mv=int(bool(c["accurate"])), force_integer_mv=c.get("force_int", 0), d_pred_sse=outs[3])
            got = dict(best_mv=ctx.from_device(outs[0], (2,), np.int16).tolist(), rate_mv=int(ctx.from_device(outs[2], (1,), np.int32)[0]),
                       pred_sse=int(ctx.from_device(outs[3], (1,), np.uint32)[0]))
            extra += [d_b] + ([d_s2] if d_s2 is not None else [])
        else:
            b["start_row"], b["start_col"] = rawpel(c["mi_mv"][0]), rawpel(c["mi_mv"][1])
            fl = b.copy()
            fl["row_min"], fl["row_max"], fl["col_min"], fl["col_max"] = oracle.set_mv_search_range(tuple(c["limits"]), *c["ref_mv"])   # (host arithmetic)
            d_fl = ctx.to_device(fl)
            d_ws, d_om = ctx.to_device(np.ascontiguousarray(z["ws%d" % k])), ctx.to_device(np.ascontiguousarray(z["om%d" % k]))
            ctx.obmc_full_pixel_search_batch(pr, 0, w, h, c["method"], step, c.get("fast_obmc", 0), 0, c["sadperbit"], c["errorperbit"], d_fl, 1, d_ws, d_om, outs[0],
                                             outs[1], *tabs)
            fmv = ctx.from_device(outs[0], (2,), np.int16)
            sl = b.copy()
            sl["row_min"], sl["row_max"], sl["col_min"], sl["col_max"] = oracle.set_subpel_mv_search_range(tuple(c["limits"]), *c["ref_mv"])
            sl["start_row"], sl["start_col"] = int(fmv[0]) * 8, int(fmv[1]) * 8
            d_sl = ctx.to_device(sl)
            sub = capi.SubpelParams(2, 0, c["errorperbit"], 2, 1, 0, sst)
            ctx.obmc_subpel_tree_batch(pr, 0, w, h, sub, d_sl, 1, d_ws, d_om, outs[0], outs[1], outs[2], outs[3], *tabs)
            mv = ctx.from_device(outs[0], (2,), np.int16)
            rate = oracle.mv_bit_cost(int(mv[0]), int(mv[1]), c["ref_mv"][0], c["ref_mv"][1], j, c0, c1)     # (av1_mv_bit_cost: two table reads)
            got = dict(best_mv=mv.tolist(), rate_mv=int(rate), pred_sse=int(ctx.from_device(outs[3], (1,), np.uint32)[0]))
            extra += [d_fl, d_sl, d_ws, d_om]
        assert got == {kk: c[kk] for kk in got}, (c, got)
        n += 1
        for d in outs + extra:
            ctx.free(d)
    assert n >= 14
    for d in (d_j, d_c0, d_c1):
        ctx.free(d)
    for ps_ in planes.values():
        for p_ in ps_:
            ctx.planes_free(p_)
