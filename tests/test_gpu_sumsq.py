"""aomhip_sum_sse_2d_i16_batch (csrc/rd_helpers.hip) against aom_sum_squares_2d_i16_c / aom_sum_sse_2d_i16_c interpreted (tests/golden/ref_eval_sumsq.npz,
directly) and the oracle on lists of blocks."""
import numpy as np
import pytest

from test_golden_sumsq import load, oracle_sum_sse

pytestmark = pytest.mark.gpu


def test_device_matches_the_interpreted_functions(hip, ctx):
    capi = hip.capi
    z, cases = load()
    dev = {k: ctx.to_device(np.ascontiguousarray(z[k])) for k in ("r8", "r12", "r16")}
    for c in cases:
        b = np.zeros(1, capi.txb_dtype)
        b["x"], b["y"] = c["x"], c["y"]
        d_b, d_s, d_m = ctx.to_device(b), ctx.malloc(8), ctx.malloc(4)
        ctx.sum_sse_2d_i16_batch(dev[c["plane"]], z[c["plane"]].shape[1], c["w"], c["h"], d_b, 1, d_s, d_m)
        assert int(ctx.from_device(d_s, (1,), np.int64)[0]) == int(c["ss"]), c
        assert int(ctx.from_device(d_m, (1,), np.int32)[0]) == c["sum_out"] - c["sum_in"], c     # (the reference adds to *sum)
        for d in (d_b, d_s, d_m):
            ctx.free(d)
    for d in dev.values():
        ctx.free(d)


@pytest.mark.parametrize("w,h", [(4, 4), (16, 16), (64, 64), (128, 128), (32, 8), (12, 20)])
def test_lists_of_blocks_equal_the_oracle(hip, ctx, w, h):
    capi = hip.capi
    rng = np.random.default_rng(w * 131 + h)
    S, R = 384, 256
    plane = rng.integers(-32768, 32768, (R, S)).astype(np.int16)
    nb = 77
    b = np.zeros(nb, capi.txb_dtype)
    b["x"], b["y"] = rng.integers(0, S - w + 1, nb), rng.integers(0, R - h + 1, nb)
    d_p, d_b, d_s = ctx.to_device(plane), ctx.to_device(b), ctx.malloc(8 * nb)
    ctx.sum_sse_2d_i16_batch(d_p, S, w, h, d_b, nb, d_s)                       # without the sum
    got = ctx.from_device(d_s, (nb,), np.int64)
    for i in range(nb):
        assert int(got[i]) == oracle_sum_sse(plane, int(b["x"][i]), int(b["y"][i]), w, h)[0], i
    for d in (d_p, d_b, d_s):
        ctx.free(d)
