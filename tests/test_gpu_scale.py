"""aomhip_scaled_pred_batch / aomhip_scaled_pred_compound_batch (csrc/scale_pred.hip) against (a) the interpreted reference's
av1_convolve_2d_scale_c / av1_highbd_convolve_2d_scale_c (tests/golden/ref_eval_scale.npz, directly) and (b) the oracle on frames of blocks."""
import numpy as np
import pytest

from test_golden_scale import load, orc_scale

pytestmark = pytest.mark.gpu
B = 160     # border: a 2:1 down-scaled 128-wide block reads 264 source columns


def run_case(ctx, capi, planes, c, W, H):
    pr = [ctx.planes_alloc(W, H, B, c["bd"], 1) for _ in range(2)]
    pp = ctx.planes_alloc(W, H, B, c["bd"], 1)
    for r in range(2):
        ctx.planes_upload(pr[r], 0, planes[r])
    ctx.planes_upload(pp, 0, np.zeros_like(planes[0]))
    d_conv = ctx.to_device(np.zeros(H * W, np.uint16))
    conv = None
    dx, dy = 8, 16
    for r in range(2 if c["compound"] else 1):
        rec = np.zeros(1, capi.scaled_block_dtype)
        rec["src_x"], rec["src_y"] = c["pos"][r]
        rec["subpel_x_qn"], rec["subpel_y_qn"] = c["subs"][r]
        rec["dst_x"], rec["dst_y"] = dx, dy
        d_b = ctx.to_device(rec)
        if c["compound"]:
            ctx.scaled_pred_compound_batch(pr[r], 0, pp if r else None, 0, c["w"], c["h"], c["fx"], c["fy"], c["xs"], c["ys"], d_b, 1, d_conv, W, r, c["weights"])
            if r == 0:
                conv = ctx.from_device(d_conv, (H, W), np.uint16)[dy:dy + c["h"], dx:dx + c["w"]].copy()
        else:
            ctx.scaled_pred_batch(pr[r], 0, pp, 0, c["w"], c["h"], c["fx"], c["fy"], c["xs"], c["ys"], d_b, 1)
        ctx.free(d_b)
    got = ctx.planes_download(pp, 0)[B:B + H, B:B + W]
    out = got[dy:dy + c["h"], dx:dx + c["w"]].copy()
    got[dy:dy + c["h"], dx:dx + c["w"]] = 0
    assert not got.any()        # nothing outside the block is written
    ctx.free(d_conv)
    for p in pr + [pp]:
        ctx.planes_free(p)
    return conv, out


def test_device_scaled_predictor_reproduces_the_interpreted_reference(hip, ctx):
    z, cases = load()
    for c in cases:
        planes = [z["p%d_%d" % (c["bd"], r)] for r in range(2)]
        H, W = planes[0].shape
        conv, out = run_case(ctx, hip.capi, planes, c, W, H)
        if c["compound"]:
            assert np.array_equal(conv.ravel(), z["c%d" % c["k"]]), c
        assert np.array_equal(out.ravel().astype(np.uint16), z["d%d" % c["k"]]), c


@pytest.mark.parametrize("bd,bw,bh,xs,ys,compound,weights", [(8, 16, 16, 2048, 2048, 0, None), (10, 64, 64, 1536, 1229, 0, None), (10, 128, 128, 2048, 2048, 1, None),
                                                            (12, 32, 64, 512, 700, 1, (9, 7)), (8, 4, 16, 1024, 1820, 0, None), (10, 128, 32, 64, 64, 0, None)])
def test_frames_of_blocks_equal_the_oracle(hip, oracle, ctx, bd, bw, bh, xs, ys, compound, weights):
    capi = hip.capi
    rng = np.random.default_rng(bd + bw * 3 + xs)
    W, H = 512, 384
    mx = (1 << bd) - 1
    planes = [rng.integers(0, mx + 1, (H, W)).astype(np.uint16) for _ in range(2)]
    pr = [ctx.planes_alloc(W, H, B, bd, 1) for _ in range(2)]
    pp = ctx.planes_alloc(W, H, B, bd, 1)
    for r in range(2):
        ctx.planes_upload(pr[r], 0, planes[r])
    ctx.planes_upload(pp, 0, np.zeros_like(planes[0]))
    ext = [oracle.extend_plane(p, B, W + 2 * B) for p in planes]
    gc, gr = W // bw, H // bh
    n = gc * gr
    recs = []
    for r in range(2):
        rec = np.zeros(n, capi.scaled_block_dtype)
        rec["dst_x"], rec["dst_y"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bh
        # source positions as a scaled motion vector would give them, some reaching into the border
        rec["src_x"] = np.clip((rec["dst_x"].astype(np.int64) * xs >> 10) + rng.integers(-40, 41, n), -(B - 12), W + B - 12 - ((bw * xs) >> 10) - 8)
        rec["src_y"] = np.clip((rec["dst_y"].astype(np.int64) * ys >> 10) + rng.integers(-40, 41, n), -(B - 12), H + B - 12 - ((bh * ys) >> 10) - 8)
        rec["subpel_x_qn"], rec["subpel_y_qn"] = rng.integers(0, 1024, n), rng.integers(0, 1024, n)
        recs.append(rec)
    fx, fy = (bd // 2) % 4, (bw // 8) % 4
    d_conv = ctx.to_device(np.zeros(H * W, np.uint16))
    for r in range(2 if compound else 1):
        d_b = ctx.to_device(recs[r])
        if compound:
            ctx.scaled_pred_compound_batch(pr[r], 0, pp if r else None, 0, bw, bh, fx, fy, xs, ys, d_b, n, d_conv, W, r, weights)
        else:
            ctx.scaled_pred_batch(pr[r], 0, pp, 0, bw, bh, fx, fy, xs, ys, d_b, n)
        ctx.free(d_b)
    got = ctx.planes_download(pp, 0)[B:B + H, B:B + W]
    for i in range(0, n, max(1, n // 48)):
        c = {"bd": bd, "w": bw, "h": bh, "xs": xs, "ys": ys, "fx": fx, "fy": fy, "compound": compound, "weights": weights,
             "pos": [(int(recs[r]["src_x"][i]) + B, int(recs[r]["src_y"][i]) + B) for r in range(2)],
             "subs": [(int(recs[r]["subpel_x_qn"][i]), int(recs[r]["subpel_y_qn"][i])) for r in range(2)]}
        _, want = orc_scale(oracle, ext, c)
        x0, y0 = int(recs[0]["dst_x"][i]), int(recs[0]["dst_y"][i])
        assert np.array_equal(got[y0:y0 + bh, x0:x0 + bw], want.astype(got.dtype)), (i, c)
    ctx.free(d_conv)
    for p in pr + [pp]:
        ctx.planes_free(p)


def test_bad_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p = ctx.planes_alloc(64, 64, 32, 8, 1)
    d = ctx.malloc(4096)
    with pytest.raises(capi.AomHipError):
        ctx.scaled_pred_batch(p, 0, p, 0, 16, 16, 0, 0, 4096, 1024, d, 1)      # more than 2:1
    with pytest.raises(capi.AomHipError):
        ctx.scaled_pred_batch(p, 0, p, 0, 16, 16, 4, 0, 1024, 1024, d, 1)      # no such filter here (the 12-tap set is the temporal filter's)
    with pytest.raises(capi.AomHipError):
        ctx.scaled_pred_compound_batch(p, 0, None, 0, 16, 16, 0, 0, 1024, 1024, d, 1, None, 64, 0)   # a compound needs its CONV_BUF
    ctx.scaled_pred_batch(p, 0, p, 0, 16, 16, 0, 0, 1024, 1024, None, 0)
    ctx.free(d)
    ctx.planes_free(p)
