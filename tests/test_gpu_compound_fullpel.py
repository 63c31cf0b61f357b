"""aomhip_compound_full_pixel_search_batch (csrc/mcomp_compound.hip + the resume form of the general search kernel): av1_full_pixel_search with
ms_buffers.second_pred [/ mask] set (av1/encoder/mcomp.c:1693-1832; the extensive full-pel step of av1_joint_motion_search,
motion_search_facade.c:613-619) -- straight against the values obtained by interpreting the reference
(tests/golden/ref_eval_compound_fullpel.npz) and against the oracle on whole batches, 8 / 10-bit, with and without the mesh follow-up."""
import json
import os

import numpy as np
import pytest

from test_gpu_compound_search import _blocks, _planes, _tables

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_compound_fullpel.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _params(capi, c):
    return capi.SearchParams.make(c["method"], c["step_param"], c["cost_type"], c["sad_per_bit"], c["error_per_bit"], 0, c.get("run_mesh", 0),
                                  c.get("prune_mesh", 0), c.get("mesh_diff_thr", 0), c.get("force_mesh_thresh", 2147483647), 0, c.get("mesh"))


def test_matches_the_interpreted_reference(hip, ctx):
    capi = hip.capi
    z, meta = _load()
    (d_j, d_c0, d_c1), _ = _tables(ctx, z)
    planes = {bd: _planes(ctx, z, meta, bd) for bd in (8, 10)}
    n = 0
    for c in meta["cases"]:
        ps, pr = planes[c["bd"]]
        dt = np.uint8 if c["bd"] == 8 else np.uint16
        k = c["k"]
        d_b = ctx.to_device(_blocks(capi, [c["block"]]))
        d_mv, d_a, d_s = ctx.malloc(16), ctx.malloc(16), ctx.malloc(16)
        d_sp = ctx.to_device(np.ascontiguousarray(z["sp%d" % k].astype(dt)))
        d_m = ctx.to_device(np.ascontiguousarray(z["mask%d" % k])) if c["masked"] else None
        ctx.compound_full_pixel_search_batch(ps, pr, 0, c["w"], c["h"], _params(capi, c), d_b, 1, d_sp, d_m, c["inv"], d_mv, d_a, d_s, d_j, d_c0, d_c1)
        got = (ctx.from_device(d_mv, (2,), np.int16).tolist(), int(ctx.from_device(d_a, (1,), np.int32)[0]), ctx.from_device(d_s, (2,), np.int16).tolist())
        assert got == (c["mv"], c["cost"], c["second_best"]), (c, got)
        n += 1
        for d in (d_b, d_mv, d_a, d_s, d_sp):
            ctx.free(d)
        if d_m:
            ctx.free(d_m)
    assert n >= 30


@pytest.mark.parametrize("bd", [8, 10, 12])   # (12 bits with a mask: the 32-bit blend)
@pytest.mark.parametrize("bw,bh", [(16, 16), (8, 8), (8, 16), (32, 16), (32, 32), (64, 16), (64, 64), (128, 64), (4, 8)])   # 4 / 2 candidates per wavefront, <= 512 px, <= 1024 px, streamed
def test_batches_match_the_oracle(hip, oracle, ctx, bd, bw, bh):
    capi = hip.capi
    W, H, B = 320, 192, 96
    rng = np.random.default_rng(2000 * bd + bw + bh)
    src, ref = hip.synth.shifted_smooth_pair(W, H, 9, bd, shift=(-3, 2), frac8=(2, 6))
    mx = (1 << bd) - 1
    ref = np.clip(ref.astype(np.int32) + rng.integers(-3 << (bd - 8), (3 << (bd - 8)) + 1, ref.shape), 0, mx).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bw, H // bh
    n = min(gc * gr, 240)
    blocks = np.zeros(n, capi.search_block_dtype)
    pick = rng.permutation(gc * gr)[:n]
    blocks["bx"], blocks["by"] = (pick % gc) * bw, (pick // gc) * bh
    blocks["start_row"], blocks["start_col"] = rng.integers(-6, 7, n), rng.integers(-6, 7, n)
    blocks["ref_row"], blocks["ref_col"] = rng.integers(-60, 61, n), rng.integers(-60, 61, n)
    ext = B - 8 - 16
    blocks["col_min"], blocks["col_max"] = np.maximum(-(blocks["bx"] + ext), -40), np.minimum(W - blocks["bx"] - bw + ext, 40)
    blocks["row_min"], blocks["row_max"] = np.maximum(-(blocks["by"] + ext), -40), np.minimum(H - blocks["by"] - bh + ext, 40)
    blocks["row_max"][::7] = 2; blocks["col_min"][::5] = -1          # tight limits: sites out of range, clamped starts
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    d_b = ctx.to_device(blocks)
    d_mv, d_a, d_s = ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 4)
    sp = np.zeros((n, bh, bw), src.dtype)
    for i in range(n):   # the other reference's predictor: the reference block a few pixels off the start MV, perturbed
        y = B + blocks["by"][i] + int(np.clip(blocks["start_row"][i] + rng.integers(-3, 4), -8, 8))
        x = B + blocks["bx"][i] + int(np.clip(blocks["start_col"][i] + rng.integers(-3, 4), -8, 8))
        sp[i] = np.clip(rb[y:y + bh, x:x + bw].astype(np.int32) + rng.integers(-4 << (bd - 8), (4 << (bd - 8)) + 1, (bh, bw)), 0, mx)
    mask = np.clip((np.arange(bw)[None, None, :] * 64 // bw + rng.integers(-6, 7, (n, bh, bw))), 0, 64).astype(np.uint8)
    d_sp, d_m = ctx.to_device(sp), ctx.to_device(mask)
    mesh = [(12, 4), (6, 2), (4, 1), (3, 1)]
    configs = [("NSTEP", 5, capi.MV_COST_ENTROPY, None, 0, {}),                                       # the joint search's call (speed 0)
               ("NSTEP", 5, capi.MV_COST_L1_HDRES, mask, 0, dict(force_mesh_thresh=0, mesh=mesh)),   # every block through the plain mesh passes
               ("DIAMOND", 4, capi.MV_COST_NONE, mask, 1, dict(run_mesh=1, prune_mesh=1, mesh_diff_thr=2, mesh=mesh)),
               ("NSTEP_8PT", 6, capi.MV_COST_ENTROPY, mask, 1, dict(force_mesh_thresh=40000 >> (0 if bw * bh >= 256 else 3), mesh=mesh)),
               ("CLAMPED_DIAMOND", 3, capi.MV_COST_L1_LOWRES, None, 0, {})]
    if bw * bh >= 2048:
        configs = configs[:3]
    for (method, step, ct, m, inv, kw) in configs:
        q = capi.SearchParams.make(method, step, ct, 23, 71, **kw)
        ctx.compound_full_pixel_search_batch(ps, pr, 0, bw, bh, q, d_b, n, d_sp, None if m is None else d_m, inv, d_mv, d_a, d_s, d_j, d_c0 + mv_max * 4,
                                             d_c1 + mv_max * 4)
        got = (ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_a, (n,), np.int32), ctx.from_device(d_s, (n, 2), np.int16))
        oq = oracle.search_params(method, step, ct, 23, 71, no_cost_list=1, **kw)
        want = oracle.compound_full_pixel_search_batch(sb, rb, B, bw, bh, blocks, oq, sp, m, inv, mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8)
        for g, w_, name in zip(got, want, ("mv", "cost", "second_best")):
            assert np.array_equal(g, w_), (method, name, np.flatnonzero((g != w_).reshape(n, -1).any(1))[:8])
        if kw.get("force_mesh_thresh", 1) == 0:   # the mesh did change results (else the resume form is not exercised)
            oq2 = oracle.search_params(method, step, ct, 23, 71, no_cost_list=1)
            plain = oracle.compound_full_pixel_search_batch(sb, rb, B, bw, bh, blocks, oq2, sp, m, inv, mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8)
            assert (plain[0] != want[0]).any() or (plain[1] != want[1]).any()
    for d in (d_j, d_c0, d_c1, d_b, d_mv, d_a, d_s, d_sp, d_m):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_rejects_what_the_reference_does_not_do_on_a_compound(hip, ctx):
    capi = hip.capi
    ps, pr = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 8, 1)
    d = ctx.malloc(4096)
    for q in (capi.SearchParams.make("HEX", 2, capi.MV_COST_NONE), capi.SearchParams.make("NSTEP", 2, capi.MV_COST_NONE, skip_sad=1),
              capi.SearchParams.make("NSTEP", 30, capi.MV_COST_NONE)):
        with pytest.raises(capi.AomHipError):
            ctx.compound_full_pixel_search_batch(ps, pr, 0, 16, 16, q, d, 1, d, None, 0, d, d, d)
    ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
