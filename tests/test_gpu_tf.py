"""aomhip_tf_motion_search_frames (csrc/tf_search.hip): tf_motion_search for every 32x32 block of a filter window in device memory,
against the interpreted-reference vectors (tests/golden/ref_eval_tf.npz) and against the oracle's composition on larger windows:
every parameter branch (pruned / unpruned mesh, the three sub-pel trees, cost list, down-sampled SAD, force_integer_mv), 8 and 10
bit, absent frames, a frame size that is not a multiple of 32, and a whole 1920x1080 window."""
import numpy as np
import pytest

from test_oracle_tf import GOOD_MESH, check_against_fixture, fixture_cases, oracle_params

pytestmark = pytest.mark.gpu


def run_device(hip, ctx, frames, filter_frame, border, width, height, bd, tp, frame_present=None):
    F = len(frames)
    planes = ctx.planes_alloc(width, height, border, bd, F)
    for f, fr in enumerate(frames):
        ctx.planes_upload(planes, f, fr)
    blocks = hip.capi.tf_block_list(width, height, border)
    n = len(blocks)
    d_b = ctx.to_device(blocks)
    d_mv, d_mse, d_ref = ctx.malloc(F * n * 16), ctx.malloc(F * n * 16), ctx.malloc(n * 4)
    ctx.tf_motion_search_frames(planes, filter_frame, tp, d_b, n, d_mv, d_mse, d_ref, frame_present)
    out = (ctx.from_device(d_mv, (F, n, 4, 2), np.int16), ctx.from_device(d_mse, (F, n, 4), np.int32), ctx.from_device(d_ref, (n, 2), np.int16))
    for d in (d_b, d_mv, d_mse, d_ref):
        ctx.free(d)
    ctx.planes_free(planes)
    return out


def device_params(hip, width, height, s):
    return hip.capi.TfParams.default(width, height, s["bd"], s["q"], s["prune_level"], GOOD_MESH, subpel_tree=s["tree"], iters_per_step=s["iters"],
                                     allow_hp=s["allow_hp"], use_cost_list=s["use_cost_list"], use_downsampled_sad=s["skip_sad"],
                                     force_integer_mv=s["force_integer_mv"])


def test_device_pass_reproduces_the_interpreted_reference(hip, oracle, ctx):
    for ci, case, frames, meta in fixture_cases():
        s, B, W, H = case["spec"], meta["border"], meta["W"], meta["H"]
        vis = [f[B:B + H, B:B + W] for f in frames]
        mvs, mses, ref_mv = run_device(hip, ctx, vis, s["filter_frame"], B, W, H, s["bd"], device_params(hip, W, H, s))
        assert check_against_fixture(case, meta, mvs, mses, ref_mv) > 0
        # ... and every block (the fixture holds a few) against the oracle
        p = oracle_params(oracle, case, meta)
        want = oracle.tf_motion_search_frames(frames, s["filter_frame"], B, oracle.tf_block_list(W, H, B), p, threads=8)
        for g, w_, name in zip((mvs, mses, ref_mv), want, ("mvs", "mses", "ref_mv")):
            assert np.array_equal(g, w_), (s["name"], name)


def window(hip, rng, W, H, bd, F, noise=0.8):
    """A moving smooth field with two motions (left / right half) + noise; the last frame has a block of pure noise."""
    base, _ = hip.synth.shifted_smooth_pair(W + 64, H + 64, 5, bd)
    out = []
    for f in range(F):
        img = np.empty((H, W), np.float64)
        half = (W // 2) & ~15
        img[:, :half] = base[32 + f:32 + f + H, 32 - 2 * f:32 - 2 * f + half]
        img[:, half:] = base[32 - f:32 - f + H, 32 + half + 3 * f:32 + 3 * f + W]
        img += rng.normal(0, noise * (1 << (bd - 8)), img.shape)
        if f == F - 1:
            img[:32, :32] = rng.integers(0, 1 << bd, (32, 32))
        out.append(np.clip(np.rint(img), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16))
    return out


VARIANTS = [
    # W, H, bd, F, filter, spec overrides
    (352, 288, 8, 5, 2, dict(q=30, prune_level=1, tree=2)),
    (352, 288, 10, 3, 1, dict(q=10, prune_level=1, tree=2)),                       # q <= 20: the mesh search always runs
    (200, 150, 8, 4, 0, dict(q=40, prune_level=2, tree=0, use_cost_list=1)),      # not a multiple of 32: blocks reach into the border
    (200, 150, 10, 3, 2, dict(q=40, prune_level=0, tree=1, use_cost_list=1, allow_hp=0, iters=1)),
    (352, 288, 8, 3, 1, dict(q=30, prune_level=1, tree=2, skip_sad=1)),
    (352, 288, 10, 3, 1, dict(q=30, prune_level=1, tree=2, force_integer_mv=1)),
    (1280, 720, 8, 3, 1, dict(q=30, prune_level=1, tree=2)),                       # MV_COST_L1_HDRES, thresh 12
]


@pytest.mark.parametrize("W,H,bd,F,filt,over", VARIANTS)
def test_device_pass_matches_the_oracle(hip, oracle, ctx, W, H, bd, F, filt, over):
    rng = np.random.default_rng(W + bd + F)
    s = dict(bd=bd, q=30, prune_level=1, tree=2, iters=2, allow_hp=1, use_cost_list=0, skip_sad=0, force_integer_mv=0)
    s.update(over)
    border = 64
    frames = window(hip, rng, W, H, bd, F)
    present = None
    if F >= 4:
        present = np.ones(F, np.uint8); present[F - 2 if F - 2 != filt else F - 1] = 0
    got = run_device(hip, ctx, frames, filt, border, W, H, bd, device_params(hip, W, H, s), present)
    p = oracle.tf_params(W, H, bd, s["q"], s["prune_level"], GOOD_MESH, subpel_tree=s["tree"], iters_per_step=s["iters"], allow_hp=s["allow_hp"],
                         use_cost_list=s["use_cost_list"], use_downsampled_sad=s["skip_sad"], force_integer_mv=s["force_integer_mv"])
    stride = hip.capi.lib.aomhip_calc_stride(W, border)
    fb = [oracle.extend_plane(f, border, stride) for f in frames]
    want = oracle.tf_motion_search_frames(fb, filt, border, oracle.tf_block_list(W, H, border), p, present, threads=8)
    for g, w_, name in zip(got, want, ("mvs", "mses", "ref_mv")):
        assert np.array_equal(g, w_), name
    mvs, mses, _ = got
    assert (mses[filt] == 2147483647).all() and not mvs[filt].any()
    searched = [f for f in range(F) if f != filt and (present is None or present[f])]
    assert any((mvs[f][:, 0] != mvs[f][:, 3]).any() for f in searched) or s["force_integer_mv"]   # some blocks split
    assert any(mvs[f].any() for f in searched)


def test_whole_1080p_10bit_window(hip, oracle, ctx):
    W, H, bd, F, filt, border = 1920, 1080, 10, 3, 1, 160
    rng = np.random.default_rng(77)
    frames = window(hip, rng, W, H, bd, F)
    s = dict(bd=bd, q=30, prune_level=1, tree=2, iters=2, allow_hp=1, use_cost_list=0, skip_sad=0, force_integer_mv=0)
    got = run_device(hip, ctx, frames, filt, border, W, H, bd, device_params(hip, W, H, s))
    p = oracle.tf_params(W, H, bd, 30, 1, GOOD_MESH)
    stride = hip.capi.lib.aomhip_calc_stride(W, border)
    fb = [oracle.extend_plane(f, border, stride) for f in frames]
    want = oracle.tf_motion_search_frames(fb, filt, border, oracle.tf_block_list(W, H, border), p, threads=8)
    assert got[0].shape == (F, 60 * 34, 4, 2)
    for g, w_, name in zip(got, want, ("mvs", "mses", "ref_mv")):
        assert np.array_equal(g, w_), name


def test_rejects_parameters_that_are_not_tf_motion_search(hip, ctx):
    capi = hip.capi
    planes = ctx.planes_alloc(96, 96, 48, 8, 2)
    tp = capi.TfParams.default(96, 96, 8, 30, 1, GOOD_MESH)
    d = ctx.malloc(4096)
    bad = []
    for field, value in (("full.search_method", 0), ("full.run_mesh_search", 0), ("sub.subpel_search_type", 0), ("sub.forced_stop", 1),
                         ("sub.mv_cost_type", 3), ("full.mv_cost_type", 0), ("sub.tree", 3)):
        t = capi.TfParams.default(96, 96, 8, 30, 1, GOOD_MESH)
        obj, name = (t.full, field[5:]) if field.startswith("full.") else (t.sub, field[4:])
        setattr(obj, name, value)
        bad.append(t)
    import ctypes as C
    for t in bad:
        assert capi.lib.aomhip_tf_motion_search_frames(ctx.h, C.byref(planes), 0, None, C.byref(t), d, 9, d, d, None) == capi.ERR_INVALID
    assert capi.lib.aomhip_tf_motion_search_frames(ctx.h, C.byref(planes), 2, None, C.byref(tp), d, 9, d, d, None) == capi.ERR_INVALID
    assert capi.lib.aomhip_tf_motion_search_frames(ctx.h, C.byref(planes), 0, None, C.byref(tp), d, 0, None, None, None) == 0
    ctx.free(d)
    ctx.planes_free(planes)
