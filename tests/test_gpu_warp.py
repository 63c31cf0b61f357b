"""aomhip_warp_affine_batch (csrc/warp.hip) against (a) the interpreted reference's av1_warp_affine_c / av1_highbd_warp_affine_c
(tests/golden/ref_eval_warp.npz, directly) and (b) the oracle on frames of blocks with one model per block."""
import json
import os

import numpy as np
import pytest

from test_golden_warp import orc_warp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def shear_of(mat):   # valid shear parameters for a near-identity model (av1_get_shear_params's form, rounded to WARP_PARAM_REDUCE_BITS)
    def red(v):
        v = int(np.clip(v, -32768, 32767))
        r = (abs(v) + 32) >> 6
        return (r if v >= 0 else -r) * 64
    return (red(mat[2] - (1 << 16)), red(mat[3]), red(int(round(mat[4] * 65536.0 / mat[2]))),
            red(mat[5] - int(round(mat[3] * mat[4] / float(mat[2]))) - (1 << 16)))


def test_device_warp_reproduces_the_interpreted_reference(hip, ctx):
    z = np.load(os.path.join(HERE, "golden", "ref_eval_warp.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    capi = hip.capi
    for c in cases:
        plane = z["ref%d" % c["bd"]]
        H, W = plane.shape
        pr, pp = ctx.planes_alloc(W, H, 16, c["bd"], 1), ctx.planes_alloc(W, H, 16, c["bd"], 1)
        ctx.planes_upload(pr, 0, plane)
        ctx.planes_upload(pp, 0, np.zeros_like(plane))
        rec = np.zeros(1, capi.warp_block_dtype)
        rec["mat"][0] = c["mat"]
        rec["alpha"], rec["beta"], rec["gamma"], rec["delta"] = c["shear"]
        rec["p_col"], rec["p_row"], rec["p_width"], rec["p_height"] = c["p_col"], c["p_row"], c["pw"], c["ph"]
        d_b = ctx.to_device(rec)
        ctx.warp_affine_batch(pr, 0, pp, 0, c["ss"], c["ss"], d_b, 1, c["pw"], c["ph"])
        got = ctx.planes_download(pp, 0)[16:16 + H, 16:16 + W]
        want = z["d%d" % c["k"]].reshape(c["ph"], c["pw"])
        assert np.array_equal(got[c["p_row"]:c["p_row"] + c["ph"], c["p_col"]:c["p_col"] + c["pw"]].astype(np.uint16), want), c
        mask = np.ones((H, W), bool)
        mask[c["p_row"]:c["p_row"] + c["ph"], c["p_col"]:c["p_col"] + c["pw"]] = False
        assert not got[mask].any()      # nothing outside the block is written
        ctx.free(d_b)
        ctx.planes_free(pr); ctx.planes_free(pp)


@pytest.mark.parametrize("bd,bw,bh,ss", [(8, 16, 16, 0), (10, 32, 16, 0), (10, 8, 8, 1), (12, 64, 64, 0), (8, 4, 4, 1), (10, 128, 128, 0), (8, 12, 20, 0)])
def test_frames_of_blocks_equal_the_oracle(hip, oracle, ctx, bd, bw, bh, ss):
    capi = hip.capi
    rng = np.random.default_rng(bd * 1000 + bw + bh + ss)
    W, H = 384, 256
    mx = (1 << bd) - 1
    yy, xx = np.mgrid[0:H, 0:W]
    plane = np.clip(((np.sin(xx / 9.0) + np.cos(yy / 7.0) + 2) * 0.25 * mx + rng.integers(-mx // 8, mx // 8 + 1, (H, W))), 0, mx).astype(np.uint16)
    pr, pp = ctx.planes_alloc(W, H, 32, bd, 1), ctx.planes_alloc(W, H, 32, bd, 1)
    ctx.planes_upload(pr, 0, plane)
    ctx.planes_upload(pp, 0, np.zeros_like(plane))
    gc, gr = W // bw, H // bh
    n = gc * gr
    rec = np.zeros(n, capi.warp_block_dtype)
    for i in range(n):
        mat = [int(rng.integers(-9 << 16, 9 << 16)), int(rng.integers(-9 << 16, 9 << 16)), (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12)),
               int(rng.integers(-(1 << 12), 1 << 12)), int(rng.integers(-(1 << 12), 1 << 12)), (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12))]
        if i % 11 == 3:
            mat[i % 2] += (500 << 16) * (1 if i % 4 else -1)      # far outside: every sample clamped
        rec["mat"][i] = mat
        rec["alpha"][i], rec["beta"][i], rec["gamma"][i], rec["delta"][i] = shear_of(mat)
    rec["p_col"], rec["p_row"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bh
    rec["p_width"], rec["p_height"] = bw, bh
    d_b = ctx.to_device(rec)
    ctx.warp_affine_batch(pr, 0, pp, 0, ss, ss, d_b, n, bw, bh)
    got = ctx.planes_download(pp, 0)[32:32 + H, 32:32 + W]
    step = max(1, n // 160)
    for i in range(0, n, step):
        c = {"mat": [int(v) for v in rec["mat"][i]], "shear": [int(rec[k][i]) for k in ("alpha", "beta", "gamma", "delta")], "p_col": int(rec["p_col"][i]),
             "p_row": int(rec["p_row"][i]), "pw": bw, "ph": bh, "ss": ss, "round_0": 5 if bd == 12 else 3}
        want = orc_warp(oracle, plane, bd, c)
        assert np.array_equal(got[c["p_row"]:c["p_row"] + bh, c["p_col"]:c["p_col"] + bw], want.astype(got.dtype)), (i, c)
    assert not got[gr * bh:, :].any() and not got[:, gc * bw:].any()
    ctx.free(d_b)
    ctx.planes_free(pr); ctx.planes_free(pp)


def test_bad_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p8, p10 = ctx.planes_alloc(64, 64, 16, 8, 1), ctx.planes_alloc(64, 64, 16, 10, 1)
    d = ctx.malloc(4096)
    with pytest.raises(capi.AomHipError):
        ctx.warp_affine_batch(p8, 0, p10, 0, 0, 0, d, 1, 8, 8)      # pixel types differ
    with pytest.raises(capi.AomHipError):
        ctx.warp_affine_batch(p8, 1, p8, 0, 0, 0, d, 1, 8, 8)       # no such frame
    with pytest.raises(capi.AomHipError):
        ctx.warp_affine_batch(p8, 0, p8, 0, 0, 0, d, 1, 256, 8)     # blocks are at most 128 wide
    ctx.warp_affine_batch(p8, 0, p8, 0, 0, 0, None, 0, 8, 8)        # an empty batch is not an error
    ctx.free(d)
    ctx.planes_free(p8); ctx.planes_free(p10)


def test_device_compound_warp_reproduces_the_interpreted_reference(hip, ctx):
    z = np.load(os.path.join(HERE, "golden", "ref_eval_warp_compound.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    capi = hip.capi
    for c in cases:
        planes = [z["ref%d_%d" % (c["bd"], r)] for r in range(2)]
        H, W = planes[0].shape
        pr = [ctx.planes_alloc(W, H, 16, c["bd"], 1) for _ in range(2)]
        pp = ctx.planes_alloc(W, H, 16, c["bd"], 1)
        for r in range(2):
            ctx.planes_upload(pr[r], 0, planes[r])
        ctx.planes_upload(pp, 0, np.zeros_like(planes[0]))
        d_conv = ctx.to_device(np.zeros(H * W, np.uint16))     # a plane-shaped CONV_BUF
        for r in range(2):
            rec = np.zeros(1, capi.warp_block_dtype)
            rec["mat"][0] = c["mat"][r]
            rec["alpha"], rec["beta"], rec["gamma"], rec["delta"] = c["shear"][r]
            rec["p_col"], rec["p_row"], rec["p_width"], rec["p_height"] = c["p_col"], c["p_row"], c["pw"], c["ph"]
            d_b = ctx.to_device(rec)
            ctx.warp_affine_compound_batch(pr[r], 0, pp if r else None, 0, c["ss"], c["ss"], d_b, 1, c["pw"], c["ph"], d_conv, W, r, c["weights"])
            ctx.free(d_b)
            if r == 0:
                conv = ctx.from_device(d_conv, (H, W), np.uint16)
                assert np.array_equal(conv[c["p_row"]:c["p_row"] + c["ph"], c["p_col"]:c["p_col"] + c["pw"]].ravel(), z["c%d" % c["k"]]), c
        got = ctx.planes_download(pp, 0)[16:16 + H, 16:16 + W]
        assert np.array_equal(got[c["p_row"]:c["p_row"] + c["ph"], c["p_col"]:c["p_col"] + c["pw"]].ravel().astype(np.uint16), z["d%d" % c["k"]]), c
        ctx.free(d_conv)
        for p in pr + [pp]:
            ctx.planes_free(p)


@pytest.mark.parametrize("bd,bw,bh,weights", [(8, 16, 16, None), (10, 32, 32, (9, 7)), (12, 64, 32, (4, 12)), (10, 8, 8, None)])
def test_compound_frames_of_blocks_equal_the_oracle(hip, oracle, ctx, bd, bw, bh, weights):
    from test_golden_warp import orc_warp_compound
    capi = hip.capi
    rng = np.random.default_rng(bd * 77 + bw)
    W, H = 256, 192
    mx = (1 << bd) - 1
    planes = [np.clip(rng.integers(0, mx + 1, (H, W)), 0, mx).astype(np.uint16) for _ in range(2)]
    pr = [ctx.planes_alloc(W, H, 32, bd, 1) for _ in range(2)]
    pp = ctx.planes_alloc(W, H, 32, bd, 1)
    for r in range(2):
        ctx.planes_upload(pr[r], 0, planes[r])
    ctx.planes_upload(pp, 0, np.zeros_like(planes[0]))
    gc, gr = W // bw, H // bh
    n = gc * gr
    recs = []
    for r in range(2):
        rec = np.zeros(n, capi.warp_block_dtype)
        for i in range(n):
            mat = [int(rng.integers(-9 << 16, 9 << 16)), int(rng.integers(-9 << 16, 9 << 16)), (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12)),
                   int(rng.integers(-(1 << 12), 1 << 12)), int(rng.integers(-(1 << 12), 1 << 12)), (1 << 16) + int(rng.integers(-(1 << 12), 1 << 12))]
            rec["mat"][i] = mat
            rec["alpha"][i], rec["beta"][i], rec["gamma"][i], rec["delta"][i] = shear_of(mat)
        rec["p_col"], rec["p_row"] = (np.arange(n) % gc) * bw, (np.arange(n) // gc) * bh
        rec["p_width"], rec["p_height"] = bw, bh
        recs.append(rec)
    d_conv = ctx.to_device(np.zeros(H * W, np.uint16))
    for r in range(2):
        d_b = ctx.to_device(recs[r])
        ctx.warp_affine_compound_batch(pr[r], 0, pp if r else None, 0, 0, 0, d_b, n, bw, bh, d_conv, W, r, weights)
        ctx.free(d_b)
    got = ctx.planes_download(pp, 0)[32:32 + H, 32:32 + W]
    for i in range(0, n, max(1, n // 60)):
        c = {"mat": [[int(v) for v in recs[r]["mat"][i]] for r in range(2)], "shear": [[int(recs[r][k][i]) for k in ("alpha", "beta", "gamma", "delta")] for r in range(2)],
             "p_col": int(recs[0]["p_col"][i]), "p_row": int(recs[0]["p_row"][i]), "pw": bw, "ph": bh, "ss": 0, "weights": weights, "round_0": 5 if bd == 12 else 3}
        _, want = orc_warp_compound(oracle, planes, bd, c)
        assert np.array_equal(got[c["p_row"]:c["p_row"] + bh, c["p_col"]:c["p_col"] + bw], want.astype(got.dtype)), (i, c)
    ctx.free(d_conv)
    for p in pr + [pp]:
        ctx.planes_free(p)
