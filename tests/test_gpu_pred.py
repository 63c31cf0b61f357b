"""Full-pel prediction gather (aom_convolve_copy case of av1_build_inter_predictor) vs numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bd", [8, 10])
def test_pred_copy(hip, oracle, ctx, bd):
    rng = np.random.default_rng(bd)
    W, H, border = 192, 128, 64
    ref = hip.synth.lcg_frame(W, H, 1, 0, bd)
    pr, pp = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pr, 1, ref)
    ctx.planes_upload(pp, 0, np.zeros_like(ref))
    rb = oracle.extend_plane(ref, border, pr.stride)
    for bw, bh in ((16, 16), (8, 32), (64, 64), (4, 4)):
        xs, ys = np.meshgrid(np.arange(0, W - bw + 1, bw), np.arange(0, H - bh + 1, bh))
        n = xs.size
        blocks = np.zeros(n, hip.capi.search_block_dtype)
        blocks["bx"], blocks["by"] = xs.ravel(), ys.ravel()
        mv = rng.integers(-40, 41, (n, 2)).astype(np.int16)
        d_b, d_mv = ctx.to_device(blocks), ctx.to_device(mv)
        ctx.build_pred_fullpel(pr, 1, pp, 0, bw, bh, d_b, d_mv, n)
        got = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
        want = np.zeros_like(ref)
        for i in range(n):
            x, y = int(blocks["bx"][i]), int(blocks["by"][i])
            want[y:y + bh, x:x + bw] = rb[border + y + mv[i, 0]:border + y + mv[i, 0] + bh, border + x + mv[i, 1]:border + x + mv[i, 1] + bw]
        assert np.array_equal(got[:(H // bh) * bh, :(W // bw) * bw], want[:(H // bh) * bh, :(W // bw) * bw])
        ctx.free(d_b); ctx.free(d_mv)
    ctx.planes_free(pr); ctx.planes_free(pp)
