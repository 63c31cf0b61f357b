"""aomhip_first_pass_motion_search_batch (csrc/tf_search.hip): NSTEP on the first-pass site table + av1_get_mvpred_sse + the new-MV penalty for
a list of blocks, against the interpreted-reference vectors (tests/golden/ref_eval_fp.npz) and against the oracle on whole-frame lists (the
two zero-MV legs of a first-pass frame) with entropy and L1 MV costs, 8 and 10 bit."""
import numpy as np
import pytest

from test_oracle_fp import BLOCK_FIELDS, block_of, fixture

pytestmark = pytest.mark.gpu


def centre_ptr(ctx, table):
    t = np.ascontiguousarray(table, np.int32)
    d = ctx.to_device(t)
    return d, d + (t.size // 2) * 4


def test_device_reproduces_the_interpreted_reference(hip, oracle, ctx):
    z, meta = fixture()
    B, W, H = meta["border"], meta["W"], meta["H"]
    d_j = ctx.to_device(np.ascontiguousarray(z["mvjcost"], np.int32))
    d_c0, c0 = centre_ptr(ctx, z["mvcost0"])
    d_c1, c1 = centre_ptr(ctx, z["mvcost1"])
    planes = {}
    for bd in (8, 10):
        ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
        ctx.planes_upload(ps, 0, z["src%d" % bd][B:B + H, B:B + W]); ctx.planes_upload(pr, 0, z["ref%d" % bd][B:B + H, B:B + W])
        planes[bd] = (ps, pr)
    d_mv, d_err = ctx.malloc(16), ctx.malloc(16)
    for c in meta["cases"]:
        q = hip.capi.SearchParams.make("NSTEP_FPF", c["step_param"], hip.capi.MV_COST_ENTROPY, sad_per_bit=c["sad_per_bit"], error_per_bit=c["error_per_bit"])
        d_b = ctx.to_device(block_of(c, hip.capi.search_block_dtype))
        ps, pr = planes[c["bd"]]
        ctx.first_pass_motion_search_batch(ps, pr, 0, c["w"], c["h"], q, d_b, 1, d_mv, d_err, d_j, c0, c1)
        assert ctx.from_device(d_mv, (2,), np.int16).tolist() == c["mv"], c
        assert int(ctx.from_device(d_err, (1,), np.int32)[0]) == c["err"], c
        ctx.free(d_b)
    for d in (d_j, d_c0, d_c1, d_mv, d_err):
        ctx.free(d)
    for ps, pr in planes.values():
        ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd,cost", [(8, "ENTROPY"), (10, "ENTROPY"), (8, "L1_HDRES"), (10, "NONE")])
def test_whole_frame_zero_mv_leg_matches_the_oracle(hip, oracle, ctx, bd, cost):
    """Every 16x16 block of a 352x288 frame (the first pass's fixed block size), started at MV 0 (the leg that needs no neighbour)."""
    capi = hip.capi
    W, H, B, bs = 352, 288, 64, 16
    rng = np.random.default_rng(bd + len(cost))
    src, ref = hip.synth.shifted_smooth_pair(W, H, 3, bd, shift=(2, -3), frac8=(0, 0))
    ref = np.clip(ref.astype(np.int32) + rng.integers(-4, 5, ref.shape), 0, (1 << bd) - 1).astype(ref.dtype)
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    gc, gr = W // bs, H // bs
    n = gc * gr
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    half = n // 2                                       # second half: a non-zero ref_mv (the chained leg's shape)
    blocks["ref_row"][half:], blocks["ref_col"][half:] = rng.integers(-24, 25, n - half), rng.integers(-24, 25, n - half)
    blocks["start_row"] = (blocks["ref_row"].astype(np.int32) + 3 + (blocks["ref_row"] >= 0)) >> 3    # get_fullmv_from_mv
    blocks["start_col"] = (blocks["ref_col"].astype(np.int32) + 3 + (blocks["ref_col"] >= 0)) >> 3
    ext = B - 8
    blocks["col_min"] = np.maximum(-(blocks["bx"] + ext), -1023); blocks["col_max"] = np.minimum(W - blocks["bx"] - bs + ext, 1023)
    blocks["row_min"] = np.maximum(-(blocks["by"] + ext), -1023); blocks["row_max"] = np.minimum(H - blocks["by"] - bs + ext, 1023)
    ct = {"ENTROPY": 0, "L1_HDRES": 3, "NONE": 4}[cost]
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (150 + bits * 310).astype(np.int32), (170 + bits * 290 + (v & 7) * 3).astype(np.int32)
    tj = np.array([200, 650, 640, 1050], np.int32)
    q = capi.SearchParams.make("NSTEP_FPF", 2, ct, sad_per_bit=24, error_per_bit=70)
    d_b = ctx.to_device(blocks)
    d_mv, d_err = ctx.malloc(n * 4), ctx.malloc(n * 4)
    d_j = ctx.to_device(tj)
    d_c0, d_c1 = ctx.to_device(t0), ctx.to_device(t1)
    ctx.first_pass_motion_search_batch(ps, pr, 0, bs, bs, q, d_b, n, d_mv, d_err, d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    mv, err = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_err, (n,), np.int32)
    sb, rb = oracle.extend_plane(src, B, ps.stride), oracle.extend_plane(ref, B, pr.stride)
    oq = oracle.search_params("NSTEP_FPF", 2, ct, sad_per_bit=24, error_per_bit=70, no_cost_list=1)
    wmv, werr = oracle.first_pass_motion_search_batch(sb, rb, B, bs, bs, blocks, oq, tj, t0, t1, bd=bd, threads=8)
    assert np.array_equal(mv, wmv) and np.array_equal(err, werr)
    assert mv.any() and (err < 2147483647).all()
    for d in (d_b, d_mv, d_err, d_j, d_c0, d_c1):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
