"""aomhip_single_motion_search_rd_batch on the cases of tests/golden/ref_eval_single_rd.npz (av1_single_motion_search interpreted with the RD branch of
the second-MV decision kept, the rd measurement itself scripted): the two MVs the device measures are the two the reference measured, and its decision
is `tmp_rd < rd` on RDCOST(rdmult, mv rate + rate, dist) of ITS OWN RD_STATS for them (which tests/test_gpu_yrd.py and test_gpu_single_motion.py pin)."""
import ctypes as C

import numpy as np
import pytest

from test_golden_joint import TAPS, TREES
from test_golden_single_caller import rawpel, second_candidate
from test_golden_single_rd import load, measured_mvs

pytestmark = pytest.mark.gpu
INT_MAX = 2147483647


def rdcost(rdmult, rate, dist):
    return ((rate * rdmult + 256) >> 9) + dist * 128


def test_device_measures_the_reference_candidates_and_decides_by_rd(hip, oracle, ctx):
    capi = hip.capi
    z, meta = load()
    B, W, H = meta["border"], meta["width"], meta["height"]
    j, c0, c1 = z["mvjcost"].astype(np.int32), z["mvcost0"].astype(np.int32), z["mvcost1"].astype(np.int32)
    mv_max = c0.size // 2
    d_j, d_c0, d_c1 = ctx.to_device(j), ctx.to_device(c0), ctx.to_device(c1)
    tabs = (d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    rng = np.random.default_rng(5)
    costs = rng.integers(10, 3000, 966).astype(np.int32)
    d_costs = ctx.to_device(costs)
    planes = {}
    for bd in (8, 10):
        planes[bd] = [ctx.planes_alloc(W, H, B, bd, 1) for _ in range(3)]
        for p_, name in zip(planes[bd], ("src%d", "ref%d")):
            ctx.planes_upload(p_, 0, np.ascontiguousarray(z[name % bd][B:B + H, B:B + W]))
    took = kept = 0
    for c in meta["cases"]:
        if c.get("disable_second_mv", 0) > 1:
            continue     # (no second search at all: aomhip_single_motion_search_batch with try_second_mv = 0, covered by test_gpu_single_caller.py's form)
        bd, w, h = c["bd"], c["w"], c["h"]
        ps, pr, pp = planes[bd]
        b = np.zeros(1, capi.search_block_dtype)
        b["bx"], b["by"] = c["bx"], c["by"]
        b["ref_row"], b["ref_col"] = c["ref_mv"]
        b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
        b["start_row"], b["start_col"] = rawpel(c["ref_mv"][0]), rawpel(c["ref_mv"][1])
        full = capi.SearchParams.make(c["method"], c["step"], 0, c["sadperbit"], c["errorperbit"], mesh_diff_thr=4, mesh=meta["mesh"])
        sub = capi.SubpelParams(TREES[c["tree"]], 0, c["errorperbit"], 2, 1, 0, TAPS[c["taps"]])
        c2 = second_candidate(c)
        yb = np.zeros(1, capi.txfm_yrd_block_dtype)
        yb["bx"], yb["by"] = c["bx"], c["by"]
        yb["tx_size_rate"], yb["no_skip_txfm_rate"], yb["skip_txfm_rate"] = 120, 300, 700
        d_b, d_yb = ctx.to_device(b), ctx.to_device(yb)
        d_s2 = ctx.to_device(np.array(c2, np.int16)) if c2 is not None else None
        outs = [ctx.malloc(64) for _ in range(7)]
        qp = capi.QuantParams.from_tables(oracle.build_quantizer_y(bd, 90))
        rd = capi.SingleRdParams()
        rd.pred, rd.qparams, rd.d_costs, rd.tx_type_rate, rd.rdmult, rd.d_yrd_blocks = C.pointer(pp), C.pointer(qp), d_costs, 200, c["rdmult"], d_yb
        rd.d_stats_first, rd.d_stats_second, rd.d_candidate_mvs = outs[4], outs[5], outs[6]
        ctx.single_motion_search_rd_batch(ps, pr, 0, w, h, full, sub, d_b, 1, rd, outs[0], outs[1], outs[2], *tabs, d_start2=d_s2, use_cost_list=c["costlist"],
                                          d_pred_sse=outs[3])
        best = ctx.from_device(outs[0], (2,), np.int16).tolist()
        rate_mv = int(ctx.from_device(outs[2], (1,), np.int32)[0])
        cand = ctx.from_device(outs[6], (2, 2), np.int16)
        want = measured_mvs(c)
        if len(want) == 2:
            assert [tuple(int(v) for v in cand[0]), tuple(int(v) for v in cand[1])] == want, (c["k"], cand, want)
            sa, sb = (ctx.from_device(o, (1,), capi.txfm_yrd_stats_dtype)[0] for o in outs[4:6])
            r = [oracle.mv_bit_cost(m[0], m[1], c["ref_mv"][0], c["ref_mv"][1], j, c0, c1) for m in want]      # (av1_mv_bit_cost: two table reads)
            second_wins = rdcost(c["rdmult"], int(sb["rate"]) + r[1], int(sb["dist"])) < rdcost(c["rdmult"], r[0] + int(sa["rate"]), int(sa["dist"]))
            assert best == list(want[1] if second_wins else want[0]) and rate_mv == r[1 if second_wins else 0], (c["k"], best, want, second_wins)
            took += second_wins and want[0] != want[1]
            kept += not second_wins
        else:   # the reference measured nothing (second_best_mv invalid / equal / outside the sub-pel limits): the first candidate stands
            assert (cand[1] == -32768).all() and best == c["best_mv"] and rate_mv == c["rate_mv"], (c["k"], cand, best)
            assert int(ctx.from_device(outs[3], (1,), np.uint32)[0]) == c["pred_sse"]
        for d in outs + [d_b, d_yb] + ([d_s2] if d_s2 is not None else []):
            ctx.free(d)
    assert took >= 1 and kept >= 2, (took, kept)     # (with real RD_STATS the first candidate usually stands; test_gpu_single_motion.py has the dense case)
    for d in (d_j, d_c0, d_c1, d_costs):
        ctx.free(d)
    for ps_ in planes.values():
        for p_ in ps_:
            ctx.planes_free(p_)
