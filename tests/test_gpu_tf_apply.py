"""aomhip_tf_apply_frames (csrc/tf_apply.hip) == oracle/aomref_tf.c, which tests/test_golden_tf_apply.py pins to the interpreted reference:
the 12-tap predictors, the pixel weights, accum / count and the normalised frame for every 32x32 block of a window -- luma only and with
4:2:0 / 4:4:4 chroma, 8 / 10 / 12 bits, frames whose size is not a multiple of 32, absent frames, and the FRAME_DIFF sums.  The filtered
pixels are integers: compared exactly, with a counted allowance of one unit for pixels whose weight straddles an integer within the 1 ulp
the device's exp() may differ from libm's (none observed)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _window(hip, oracle, ctx, W, H, bd, F, planes, ssx, ssy, border, seed):
    rng = np.random.default_rng(seed)
    rings, host = [], []
    for p in range(planes):
        w, h = ((W + ssx) >> ssx, (H + ssy) >> ssy) if p else (W, H)
        ring = ctx.planes_alloc(w, h, border, bd, F)
        frames = []
        for f in range(F):
            base = hip.synth.shifted_smooth_pair(w, h, 3 * p + 1, bd, shift=(f, 2 * f), frac8=(0, 0))[1].astype(np.int64)
            img = np.clip(base + rng.integers(-(3 << (bd - 8)), (3 << (bd - 8)) + 1, base.shape), 0, (1 << bd) - 1).astype(np.uint8 if bd == 8 else np.uint16)
            ctx.planes_upload(ring, f, img)
            frames.append(oracle.extend_plane(img, border, ring.stride))
        rings.append(ring); host.append(frames)
    return rings, host


@pytest.mark.parametrize("W,H,bd,planes,ssx,ssy,q,strength", [(160, 96, 8, 3, 1, 1, 40, 5), (176, 112, 10, 3, 1, 1, 160, 2), (128, 72, 10, 1, 0, 0, 20, 4),
                                                            (96, 64, 12, 3, 0, 0, 64, 6), (200, 120, 8, 1, 0, 0, 255, 1)])
def test_apply_frames_equals_oracle(hip, oracle, ctx, W, H, bd, planes, ssx, ssy, q, strength):
    F, filt, border = 5, 2, 96
    rng = np.random.default_rng(W + 7 * bd + planes)
    rings, host = _window(hip, oracle, ctx, W, H, bd, F, planes, ssx, ssy, border, seed=W * 3 + bd)
    mb_rows, mb_cols = (H + 31) // 32, (W + 31) // 32
    n = mb_rows * mb_cols
    assert n == hip.capi.lib.aomhip_tf_block_list(W, H, border, None)
    mvs = rng.integers(-120, 121, (F, n, 4, 2)).astype(np.int16)     # 1/8 pel: up to +-15 pixels, every phase
    mvs[:, ::3] = (mvs[:, ::3] // 8) * 8                             # some full-pel vectors (the copy path), some half-aligned ones
    mvs[:, 1::5, :, 0] = (mvs[:, 1::5, :, 0] // 8) * 8              # x-only
    mvs[:, 2::7, :, 1] = (mvs[:, 2::7, :, 1] // 8) * 8              # y-only
    mses = (rng.integers(0, 90, (F, n, 4)) << (bd - 8)).astype(np.int32)
    mses[:, ::4] = rng.integers(0, 6, (F, (n + 3) // 4, 4))
    mvs[filt] = 0; mses[filt] = 2147483647
    present = np.ones(F, np.uint8); present[F - 1] = 0              # one absent frame
    noise = [1.7, 0.8, 1.2]
    outs = [ctx.planes_alloc(r.width, r.height, border, bd, 2) for r in rings]
    d_mvs, d_mses, d_diff = ctx.to_device(mvs), ctx.to_device(mses), ctx.malloc(16)
    params = hip.capi.TfApplyParams.make(noise, q, strength, planes, ssx, ssy)
    ctx.tf_apply_frames(rings, filt, params, n, d_mvs, d_mses, outs, 1, frame_present=present, d_diff=d_diff)
    want = oracle.tf_apply_frames(host, border, W, H, filt, mvs, mses, noise, q, strength, bd=bd, ss_x=ssx, ss_y=ssy, present=present)
    total, off = 0, 0
    for p in range(planes):
        sx, sy = (ssx, ssy) if p else (0, 0)
        w32, h32 = (mb_cols * 32) >> sx, (mb_rows * 32) >> sy
        got = ctx.planes_download(outs[p], 1)   # the whole bordered plane
        b = border
        g = got[b:b + h32, b:b + w32].astype(np.int64)
        wv = want[p][b:b + h32, b:b + w32].astype(np.int64)
        d = np.abs(g - wv)
        assert d.max() <= 1, (p, d.max())
        off += int((d != 0).sum()); total += d.size
        if p == 0: luma_dev, luma_off = g, int((d != 0).sum())
    assert off <= max(1, total // 20000), (off, total)
    # FRAME_DIFF: sse of every luma block (source vs filtered), highbd forms rounded to the 8-bit scale
    diff = ctx.from_device(d_diff, (2,), np.int64)
    src = host[0][filt][border:border + mb_rows * 32, border:border + mb_cols * 32].astype(np.int64)

    def frame_diff(flt):
        sse = ((src - flt) ** 2).reshape(mb_rows, 32, mb_cols, 32).sum(axis=(1, 3))
        if bd == 10: sse = (sse + 8) >> 4
        if bd == 12: sse = (sse + 128) >> 8
        return int(sse.sum()), int((sse * sse).sum())
    # unconditionally: the sums are those of the plane the device wrote (the fp64 weights may move a pixel by one against this host's libm,
    # the sums must follow the device's own pixels exactly) ...
    assert (int(diff[0]), int(diff[1])) == frame_diff(luma_dev)
    # ... and the oracle's sums whenever no luma pixel differs; otherwise within what the counted +-1 pixels can move them
    want_sum, want_sq = frame_diff(want[0][border:border + mb_rows * 32, border:border + mb_cols * 32].astype(np.int64))
    if luma_off == 0:
        assert (int(diff[0]), int(diff[1])) == (want_sum, want_sq)
    else:
        peak = (1 << bd) - 1
        assert abs(int(diff[0]) - want_sum) <= luma_off * (2 * peak + 1) + mb_rows * mb_cols
    for d_ in (d_mvs, d_mses, d_diff):
        ctx.free(d_)
    for r in rings + outs:
        ctx.planes_free(r)


@pytest.mark.parametrize("W,H,bd", [(1280, 720, 10), (704, 400, 8)])
def test_search_then_apply_stays_on_the_device(hip, oracle, ctx, W, H, bd):
    """The whole temporal filter of one frame as av1_tf_do_filtering_row runs it -- tf_motion_search for every block and window frame, then
    predictor / weights / accumulation / normalisation -- in two calls with the MVs and errors never leaving HBM; the filtered luma frame
    equals the oracle's chain (search oracle -> apply oracle) and so does FRAME_DIFF."""
    from test_gpu_tf import window, GOOD_MESH
    from test_oracle_tf import oracle_params
    F, filt, border, q = 5, 2, 160, 30
    rng = np.random.default_rng(W + bd)
    frames = window(hip, rng, W, H, bd, F)
    ring = ctx.planes_alloc(W, H, border, bd, F)
    for f, fr in enumerate(frames):
        ctx.planes_upload(ring, f, fr)
    out = ctx.planes_alloc(W, H, border, bd, 1)
    blocks = hip.capi.tf_block_list(W, H, border)
    n = len(blocks)
    d_b = ctx.to_device(blocks)
    d_mv, d_mse, d_ref, d_diff = ctx.malloc(F * n * 16), ctx.malloc(F * n * 16), ctx.malloc(n * 4), ctx.malloc(16)
    tp = hip.capi.TfParams.default(W, H, bd, q, 1, GOOD_MESH, subpel_tree=2, iters_per_step=2, allow_hp=1, use_cost_list=0, use_downsampled_sad=0,
                                   force_integer_mv=0)
    ctx.tf_motion_search_frames(ring, filt, tp, d_b, n, d_mv, d_mse, d_ref, None)
    ap = hip.capi.TfApplyParams.make([2.1, 0, 0], q, 5, 1, 0, 0)
    ctx.tf_apply_frames([ring], filt, ap, n, d_mv, d_mse, [out], 0, d_diff=d_diff)
    got = ctx.planes_download(out, 0)
    mvs, mses = ctx.from_device(d_mv, (F, n, 4, 2), np.int16), ctx.from_device(d_mse, (F, n, 4), np.int32)
    host = [oracle.extend_plane(fr, border, ring.stride) for fr in frames]
    want = oracle.tf_apply_frames([host], border, W, H, filt, mvs, mses, [2.1, 0, 0], q, 5, bd=bd)[0]
    h32, w32 = (H + 31) // 32 * 32, (W + 31) // 32 * 32
    d = np.abs(got[border:border + h32, border:border + w32].astype(np.int64) - want[border:border + h32, border:border + w32].astype(np.int64))
    assert d.max() <= 1 and int((d != 0).sum()) <= d.size // 20000, (d.max(), int((d != 0).sum()))
    # the filter did something: the result differs from the source and is closer to the window's mean than the noisy source is
    src = frames[filt].astype(np.int64)
    assert (got[border:border + H, border:border + W].astype(np.int64) != src).mean() > 0.2
    for d_ in (d_b, d_mv, d_mse, d_ref, d_diff):
        ctx.free(d_)
    ctx.planes_free(ring); ctx.planes_free(out)
