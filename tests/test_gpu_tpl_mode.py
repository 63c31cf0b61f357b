"""aomhip_tpl_inter_estimation_batch driven as INTEGRATION.md shows -- one call per wavefront of TPL blocks (block (r, c) in wave c + 2r: its above, left
and above-right neighbours are in earlier waves), the host gathering each block's starting MVs from the neighbours' results (is_alike_mv) -- against
mode_estimation interpreted for the same rows in raster order (tests/golden/ref_eval_tpl_mode.npz).  No oracle search in between: the host side is the
candidate rule and the mode decision (oracle.tpl_gather_candidates / tpl_mode_decision: a few comparisons, pinned by tests/test_golden_tpl_mode.py)."""
import numpy as np
import pytest

from test_golden_joint import TAPS, TREES
from test_golden_tpl_mode import load

pytestmark = pytest.mark.gpu
INT_MAX = 2147483647


def test_wavefront_calls_reproduce_the_interpreted_rows(hip, oracle, ctx):
    capi = hip.capi
    z, meta = load()
    B, W, H, bs, refs = meta["border"], meta["width"], meta["height"], meta["bs"], meta["refs"]
    j, c0, c1 = z["mvjcost"].astype(np.int32), z["mvcost0"].astype(np.int32), z["mvcost1"].astype(np.int32)
    mv_max = c0.size // 2
    d_j, d_c0, d_c1 = ctx.to_device(j), ctx.to_device(c0), ctx.to_device(c1)
    cols = W // bs
    n_waves = n_multi = 0
    for fi, f in enumerate(meta["frames"]):
        cfg = f["config"]
        bd = cfg["bd"]
        ps = ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(ps, 0, np.ascontiguousarray(z["src_%d" % fi][B:B + H, B:B + W]))
        prs = []
        for k in range(len(refs)):
            p = ctx.planes_alloc(W, H, B, bd, 1)
            ctx.planes_upload(p, 0, np.ascontiguousarray(z["ref%d_%d" % (k, fi)][B:B + H, B:B + W]))
            prs.append(p)
        full = capi.SearchParams.make(cfg["search_method"], min(cfg["reduce_first_step_size"], 9), 0, f["sadperbit"], f["errorperbit"], mesh_diff_thr=4, mesh=meta["mesh"])
        sub = capi.SubpelParams(TREES[cfg["subpel_search_method"]], capi.MV_COST_NONE, f["errorperbit"], 2, 1, cfg["subpel_force_stop"], TAPS["USE_2_TAPS"])
        by_pos = {(b["mi_row"] // 4, b["mi_col"] // 4): b for b in f["blocks"]}
        rows = 1 + max(r for r, _ in by_pos)
        done = {}
        for wave in range(cols + 2 * (rows - 1)):
            members = [(r, wave - 2 * r) for r in range(rows) if 0 <= wave - 2 * r < cols]
            if not members:
                continue
            n = len(members)
            blocks = np.zeros(n, capi.search_block_dtype)
            centers, counts = np.zeros((n, len(refs), 4, 2), np.int16), np.zeros((n, len(refs)), np.uint8)
            for i, (r, c) in enumerate(members):
                b = by_pos[(r, c)]
                blocks["bx"][i], blocks["by"][i] = c * bs, r * bs
                blocks["row_min"][i], blocks["row_max"][i], blocks["col_min"][i], blocks["col_max"][i] = b["limits"]
                for k in range(len(refs)):
                    nb = lambda rr, cc: done[(rr, cc)][k] if (rr, cc) in done else None
                    cl = oracle.tpl_gather_candidates(nb(r - 1, c) if r > 0 else None, nb(r, c - 1) if c > 0 else None,
                                                      nb(r - 1, c + 1) if r > 0 and c + 1 < cols else None, cfg["skip_alike_starting_mv"])
                    centers[i, k, :len(cl)] = cl
                    counts[i, k] = len(cl)
            d_b, d_c, d_n = ctx.to_device(blocks), ctx.to_device(centers), ctx.to_device(counts)
            d_mv, d_pe, d_rf, d_bc = ctx.malloc(n * len(refs) * 4), ctx.malloc(n * len(refs) * 4), ctx.malloc(max(n, 4)), ctx.malloc(n * 4)
            ctx.tpl_inter_estimation_batch(ps, prs, 0, bs, full, sub, cfg["use_fullpel_costlist"], cfg["prune_starting_mv"], d_b, d_c, d_n, n, d_mv, d_pe, d_rf, d_bc,
                                           d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
            g_mv, g_pe = ctx.from_device(d_mv, (n, len(refs), 2), np.int16), ctx.from_device(d_pe, (n, len(refs)), np.int32)
            g_rf, g_bc = ctx.from_device(d_rf, (n,), np.int8), ctx.from_device(d_bc, (n,), np.int32)
            for i, (r, c) in enumerate(members):
                s = by_pos[(r, c)]["stats"]
                for k, rf in enumerate(refs):
                    assert g_mv[i, k].tolist() == s["mv"][rf] and int(g_pe[i, k]) == s["pred_error"][rf], (fi, r, c, rf, g_mv[i, k], s["mv"][rf], g_pe[i, k], s["pred_error"][rf])
                dec = oracle.tpl_mode_decision(by_pos[(r, c)]["intra_costs"], g_rf[i], g_bc[i], None)
                rfi = [refs[dec["ref_frame_index"][0]] if dec["ref_frame_index"][0] >= 0 else -1, -1]
                assert (dec["intra_cost"], dec["inter_cost"], rfi) == (s["intra_cost"], s["inter_cost"], s["ref_frame_index"]), (fi, r, c, dec, s)
                done[(r, c)] = [g_mv[i, k].tolist() for k in range(len(refs))]
            n_waves += 1
            n_multi += n > 1
            for d in (d_b, d_c, d_n, d_mv, d_pe, d_rf, d_bc):
                ctx.free(d)
        assert len(done) == len(f["blocks"])
        ctx.planes_free(ps)
        for p in prs:
            ctx.planes_free(p)
    assert n_waves >= 30 and n_multi >= 20
    for d in (d_j, d_c0, d_c1):
        ctx.free(d)
