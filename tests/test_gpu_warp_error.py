"""aomhip_warp_error_batch / aomhip_segmented_frame_error (csrc/warp.hip) against (a) the interpreted reference's av1_warp_error /
av1_segmented_frame_error (tests/golden/ref_eval_warp_error.npz, directly) and (b) the oracle on larger frames with many candidate models per call."""
import numpy as np
import pytest

from test_golden_warp_error import INT64_MAX, load, oracle_frame_error, oracle_shear, oracle_warp_error, planes_of

pytestmark = pytest.mark.gpu


def _model(capi, mat):
    rec = np.zeros(1, capi.warp_model_dtype)
    rec["mat"][0] = mat
    return rec, capi.get_shear_params(rec)[0]


def test_device_model_error_reproduces_the_interpreted_reference(hip, ctx):
    z, cases = load()
    capi = hip.capi
    n_checked = 0
    for c in cases:
        rec, ok = _model(capi, c["mat"])
        assert ok == c["valid"], c["k"]           # the host's shear decomposition = av1_get_shear_params' verdict ...
        if c["mat"][2] > 0:
            assert [int(rec[f][0]) for f in ("alpha", "beta", "gamma", "delta")] == c["shear"], c["k"]     # ... and values
        if not ok:
            continue                              # av1_warp_error returns INT64_MAX without warping
        ref, cur = planes_of(z, c)
        H, W = ref.shape
        pr, pc = ctx.planes_alloc(W, H, 16, c["bd"], 1), ctx.planes_alloc(W, H, 16, c["bd"], 1)
        ctx.planes_upload(pr, 0, ref)
        ctx.planes_upload(pc, 0, cur)
        d_m, d_s, d_e = ctx.to_device(rec), ctx.to_device(np.asarray(c["seg"], np.uint8)), ctx.malloc(8)
        ctx.warp_error_batch(pr, 0, pc, 0, c["ss"], c["ss"], d_m, 1, c["p_col"], c["p_row"], c["pw"], c["ph"], d_s, c["seg_stride"], d_e)
        got = int(ctx.from_device(d_e, (1,), np.int64)[0])
        assert got == int(c["error"]), c["k"]
        if "best_error" in c:                     # the early exit is a comparison on the total (all terms >= 0)
            assert got > int(c["best_error"]) and int(c["error_bounded"]) == INT64_MAX
        if "frame_error" in c:
            ctx.segmented_frame_error(pr, 0, pc, 0, W, H, d_s, c["seg_stride"], d_e)
            assert int(ctx.from_device(d_e, (1,), np.int64)[0]) == int(c["frame_error"]), c["k"]
        n_checked += 1
        for d in (d_m, d_s, d_e):
            ctx.free(d)
        ctx.planes_free(pr); ctx.planes_free(pc)
    assert n_checked >= 14


@pytest.mark.parametrize("bd,W,H,ss", [(8, 352, 288, 0), (10, 330, 270, 0), (12, 176, 144, 1), (10, 100, 50, 0)])
def test_many_models_per_call_equal_the_oracle(hip, oracle, ctx, bd, W, H, ss):
    capi = hip.capi
    rng = np.random.default_rng(bd * 100 + W + ss)
    mx = (1 << bd) - 1
    yy, xx = np.mgrid[0:H + 8, 0:W + 8]
    base = (np.sin(xx / 11.0) + np.cos(yy / 6.0) + 2) * 0.25 * mx
    dt = np.uint8 if bd == 8 else np.uint16
    ref = np.clip(base[4:H + 4, 4:W + 4] + rng.integers(-mx // 16, mx // 16 + 1, (H, W)), 0, mx).astype(dt)
    cur = np.clip(base[3:H + 3, 6:W + 6] + rng.integers(-mx // 16, mx // 16 + 1, (H, W)), 0, mx).astype(dt)
    cur[-5:, -9:] = mx - ref[-5:, -9:]
    sw, sh = (W + 31) // 32, (H + 31) // 32
    seg = (rng.random((sh, sw + 2)) < 0.7).astype(np.uint8)          # a map wider than the region: the stride is the map's
    seg[0, 0] = seg[-1, sw - 1] = 1
    models = np.zeros(24, capi.warp_model_dtype)
    for i in range(len(models)):
        while True:
            models["mat"][i] = [rng.integers(-4 << 16, 4 << 16), rng.integers(-4 << 16, 4 << 16), (1 << 16) + rng.integers(-(1 << 12), 1 << 12),
                                rng.integers(-(1 << 12), 1 << 12), rng.integers(-(1 << 12), 1 << 12), (1 << 16) + rng.integers(-(1 << 12), 1 << 12)]
            if i == 0:
                models["mat"][i] = [0, 0, 1 << 16, 0, 0, 1 << 16]
            if i == 1:
                models["mat"][i][:2] = [90 << 16, -(70 << 16)]       # the whole prediction from clamped samples
            if capi.get_shear_params(models[i:i + 1])[0]:
                break
    for i in range(len(models)):                                      # the host helper against the restatement
        ok, sh4 = oracle_shear(models["mat"][i])
        assert ok and sh4.tolist() == [int(models[f][i]) for f in ("alpha", "beta", "gamma", "delta")]
    pr, pc = ctx.planes_alloc(W, H, 32, bd, 1), ctx.planes_alloc(W, H, 32, bd, 1)
    ctx.planes_upload(pr, 0, ref)
    ctx.planes_upload(pc, 0, cur)
    d_m, d_s, d_e = ctx.to_device(models), ctx.to_device(seg), ctx.malloc(8 * len(models))
    ctx.warp_error_batch(pr, 0, pc, 0, ss, ss, d_m, len(models), 0, 0, W, H, d_s, sw + 2, d_e)
    got = ctx.from_device(d_e, (len(models),), np.int64)
    c = {"bd": bd, "W": W, "H": H, "ss": ss, "p_col": 0, "p_row": 0, "pw": W, "ph": H, "seg": seg.ravel().tolist(), "seg_stride": sw + 2}
    for i in range(len(models)):
        c["mat"] = models["mat"][i].tolist()
        want = oracle_warp_error(c, ref, cur, [int(models[f][i]) for f in ("alpha", "beta", "gamma", "delta")])
        assert int(got[i]) == want, i
    assert len(set(got.tolist())) > 20
    ctx.segmented_frame_error(pr, 0, pc, 0, W, H, d_s, sw + 2, d_e)
    assert int(ctx.from_device(d_e, (1,), np.int64)[0]) == oracle_frame_error(c, ref, cur)
    # the same call again gives the same totals (integer atomics), and a sub-region starting inside the frame
    ctx.warp_error_batch(pr, 0, pc, 0, ss, ss, d_m, len(models), 0, 0, W, H, d_s, sw + 2, d_e)
    assert np.array_equal(ctx.from_device(d_e, (len(models),), np.int64), got)
    if W > 128:
        c.update(p_col=64, p_row=32, pw=W - 64 - 7, ph=H - 32 - 3)
        ctx.warp_error_batch(pr, 0, pc, 0, ss, ss, d_m, 3, 64, 32, c["pw"], c["ph"], d_s, sw + 2, d_e)
        sub = ctx.from_device(d_e, (3,), np.int64)
        for i in range(3):
            c["mat"] = models["mat"][i].tolist()
            assert int(sub[i]) == oracle_warp_error(c, ref, cur, [int(models[f][i]) for f in ("alpha", "beta", "gamma", "delta")]), i
    for d in (d_m, d_s, d_e):
        ctx.free(d)
    ctx.planes_free(pr); ctx.planes_free(pc)


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    pr, pc = ctx.planes_alloc(64, 64, 16, 8, 1), ctx.planes_alloc(64, 64, 16, 10, 1)
    d = ctx.malloc(256)
    with pytest.raises(capi.AomHipError):
        ctx.warp_error_batch(pr, 0, pc, 0, 0, 0, d, 1, 0, 0, 64, 64, d, 2, d)      # bit depths differ
    with pytest.raises(capi.AomHipError):
        ctx.warp_error_batch(pr, 0, pr, 0, 0, 0, d, 1, 0, 0, 65, 64, d, 3, d)      # region outside the frame
    with pytest.raises(capi.AomHipError):
        ctx.warp_error_batch(pr, 0, pr, 0, 0, 0, d, 1, 0, 0, 64, 64, d, 1, d)      # segment map narrower than the region
    ctx.free(d)
    ctx.planes_free(pr); ctx.planes_free(pc)
