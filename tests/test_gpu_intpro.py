"""aomhip_int_pro_motion_estimation_batch (csrc/int_pro.hip) against (a) av1_int_pro_motion_estimation interpreted as it is written
(tests/golden/ref_eval_intpro.npz, directly) and (b) the oracle on frames of blocks."""
import numpy as np
import pytest

from test_golden_intpro import load, oracle_int_pro, planes

pytestmark = pytest.mark.gpu


def test_device_matches_the_interpreted_function(hip, ctx):
    capi = hip.capi
    z, meta = load()
    B, W, H = meta["border"], meta["width"], meta["height"]
    pl = {}
    for bd in (8, 10):
        pl[bd] = [ctx.planes_alloc(W, H, B, bd, 1) for _ in range(2)]
        for p_, name in zip(pl[bd], ("src%d", "ref%d")):
            ctx.planes_upload(p_, 0, np.ascontiguousarray(z[name % bd][B:B + H, B:B + W]))
    for c in meta["cases"]:
        b = np.zeros(1, capi.search_block_dtype)
        b["bx"], b["by"] = c["bx"], c["by"]
        b["ref_row"], b["ref_col"] = c["ref_mv"]
        b["row_min"], b["row_max"], b["col_min"], b["col_max"] = c["limits"]
        d_b, d_mv, d_sad = ctx.to_device(b), ctx.malloc(4), ctx.malloc(4)
        ps, pr = pl[c["bd"]]
        ctx.int_pro_motion_estimation_batch(ps, 0, pr, 0, c["w"], c["h"], d_b, 1, d_mv, d_sad)
        assert ctx.from_device(d_mv, (2,), np.int16).tolist() == c["mv"], c
        assert int(ctx.from_device(d_sad, (1,), np.uint32)[0]) == c["best_sad"], c
        for d in (d_b, d_mv, d_sad):
            ctx.free(d)
    for bd in pl:
        for p_ in pl[bd]:
            ctx.planes_free(p_)


@pytest.mark.parametrize("bd,bw,bh", [(8, 16, 16), (8, 32, 32), (8, 64, 64), (8, 128, 128), (8, 64, 32), (8, 16, 64), (8, 128, 64), (10, 64, 64), (12, 32, 32)])
def test_frames_of_blocks_equal_the_oracle(hip, oracle, ctx, bd, bw, bh):
    capi = hip.capi
    rng = np.random.default_rng(bd * 1000 + bw * 7 + bh)
    W, H, B = 512, 384, 160
    mx = (1 << bd) - 1
    yy, xx = np.mgrid[0:H + 64, 0:W + 64]
    base = (np.sin(xx / 13.0) * np.cos(yy / 9.0) + np.sin((xx + 2 * yy) / 31.0) + 2) * 0.25 * mx
    dt = np.uint8 if bd == 8 else np.uint16
    src = np.clip(base[32:32 + H, 32:32 + W] + rng.integers(-mx // 32, mx // 32 + 1, (H, W)), 0, mx).astype(dt)
    ref = np.empty_like(src)
    for qy in range(0, H, 128):           # the reference: the source displaced by another vector in every 128 x 128 region
        for qx in range(0, W, 128):
            dy, dx = int(rng.integers(-12, 13)), int(rng.integers(-12, 13))
            ref[qy:qy + 128, qx:qx + 128] = np.clip(base[32 + qy + dy:32 + qy + dy + 128, 32 + qx + dx:32 + qx + dx + 128][:H - qy, :W - qx]
                                                    + rng.integers(-mx // 32, mx // 32 + 1, (min(128, H - qy), min(128, W - qx))), 0, mx)
    ps, pr = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src)
    ctx.planes_upload(pr, 0, ref)
    sb, rb = np.pad(src, B, mode="edge"), np.pad(ref, B, mode="edge")
    pos = [(x, y) for y in range(0, H - bh + 1, bh) for x in range(0, W - bw + 1, bw)]
    pos += [(W - bw, H - bh), (4, 8), (W - bw - 4, 12)]
    blocks = np.zeros(len(pos), capi.search_block_dtype)
    for i, (x, y) in enumerate(pos):
        blocks["bx"][i], blocks["by"][i] = x, y
        blocks["ref_row"][i], blocks["ref_col"][i] = rng.integers(-200, 201, 2)
        lim = (-(y + B - 16), H - y - bh + B - 16, -(x + B - 16), W - x - bw + B - 16)       # av1_set_mv_limits' shape
        if i % 5 == 3:
            lim = (-3, 2, -2, 4)
        blocks["row_min"][i], blocks["row_max"][i], blocks["col_min"][i], blocks["col_max"][i] = lim
    n = len(pos)
    d_b, d_mv, d_sad = ctx.to_device(blocks), ctx.malloc(4 * n), ctx.malloc(4 * n)
    ctx.int_pro_motion_estimation_batch(ps, 0, pr, 0, bw, bh, d_b, n, d_mv, d_sad)
    mv, sad = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_sad, (n,), np.uint32)
    for i, (x, y) in enumerate(pos):
        lim = [int(blocks[k][i]) for k in ("row_min", "row_max", "col_min", "col_max")]
        want_sad, want_mv = oracle_int_pro(sb, rb, B, x, y, bw, bh, bd, lim, [int(blocks["ref_row"][i]), int(blocks["ref_col"][i])])
        assert (int(sad[i]), mv[i].tolist()) == (want_sad, want_mv), (i, x, y)
    if bd == 8:
        assert len({tuple(v) for v in mv.tolist()}) >= 4          # the vectors differ between regions
    else:
        assert not mv.any()
    for d in (d_b, d_mv, d_sad):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_invalid_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p8, p8s, p10 = ctx.planes_alloc(128, 128, 64, 8, 1), ctx.planes_alloc(128, 128, 16, 8, 1), ctx.planes_alloc(128, 128, 64, 10, 1)
    d = ctx.malloc(64)
    for args in ((p8, 0, p8, 0, 8, 8), (p8, 0, p8, 0, 16, 24), (p8, 0, p10, 0, 16, 16), (p8, 0, p8s, 0, 64, 32), (p8, 1, p8, 0, 16, 16)):
        with pytest.raises(capi.AomHipError):
            ctx.int_pro_motion_estimation_batch(*args, d, 1, d, d)
    ctx.free(d)
    for p_ in (p8, p8s, p10):
        ctx.planes_free(p_)
