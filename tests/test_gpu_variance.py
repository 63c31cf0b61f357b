"""Parity of the HIP variance / sub-pixel variance kernels with the oracle (bit-exact), mirroring
test/variance_test.cc: Zero, Ref, RefStride, OneQuarter, SubpelVariance Ref / ExtremeRef, for all 22
block sizes, 8/10/12 bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import BLOCK_SIZES

pytestmark = pytest.mark.gpu


def _ptr(a, y=0, x=0):
    return a.ctypes.data + (int(y) * a.shape[1] + int(x)) * a.itemsize


@pytest.mark.parametrize("w,h", BLOCK_SIZES)
def test_rtcd_variance_cases(hip, oracle, ctx, w, h):
    lib = hip.capi.lib
    rng = np.random.default_rng(w * 31 + h)
    sse = C.c_uint()
    # OneQuarter known answer (variance_test.cc:823-839)
    a = np.full((h, w), 255, np.uint8)
    b = np.full(h * w, 255, np.uint8); b[h * w // 2:] = 0; b = b.reshape(h, w)
    assert lib.aomhip_variance(a.ctypes.data, w, b.ctypes.data, w, w, h, C.byref(sse)) == w * h * 255 * 255 // 4
    # Zero
    z = np.zeros((h, w), np.uint8)
    assert lib.aomhip_variance(a.ctypes.data, w, z.ctypes.data, w, w, h, C.byref(sse)) == 0 and sse.value == 255 * 255 * w * h
    for _ in range(3):  # Ref / RefStride
        a = rng.integers(0, 256, (h + 1, w + 9), dtype=np.uint8)
        b = rng.integers(0, 256, (h + 1, w + 5), dtype=np.uint8)
        got = lib.aomhip_variance(_ptr(a, 1, 3), a.shape[1], _ptr(b, 0, 2), b.shape[1], w, h, C.byref(sse))
        wv, wsse, _ = oracle.variance(a, 1, 3, b, 0, 2, w, h)
        assert (got, sse.value) == (wv, wsse)
    # sub-pixel: every (xoff, yoff) on one random pair + extreme pattern
    ref = rng.integers(0, 256, (h + 2, w + 4), dtype=np.uint8)
    src = rng.integers(0, 256, (h, w), dtype=np.uint8)
    for xo in range(8):
        for yo in range(8):
            got = lib.aomhip_sub_pixel_variance(_ptr(ref, 1, 2), ref.shape[1], xo, yo, src.ctypes.data, w, w, h, C.byref(sse))
            assert (got, sse.value) == oracle.sub_pixel_variance(ref, 1, 2, xo, yo, src, 0, 0, w, h), (xo, yo)
    ref = np.zeros((h + 1, w + 1), np.uint8); ref[:h // 2] = 255
    src = np.zeros((h, w), np.uint8); src[h // 2:] = 255
    got = lib.aomhip_sub_pixel_variance(ref.ctypes.data, w + 1, 3, 5, src.ctypes.data, w, w, h, C.byref(sse))
    assert (got, sse.value) == oracle.sub_pixel_variance(ref, 0, 0, 3, 5, src, 0, 0, w, h)


@pytest.mark.parametrize("bd", [10, 12])
@pytest.mark.parametrize("w,h", [(4, 4), (8, 8), (16, 16), (32, 16), (64, 64), (128, 128), (16, 64)])
def test_rtcd_highbd_variance(hip, oracle, ctx, w, h, bd):
    lib = hip.capi.lib
    rng = np.random.default_rng(bd * 100 + w + h)
    mx = (1 << bd) - 1
    sse = C.c_uint()
    cases = [(rng.integers(0, mx + 1, (h + 1, w + 3), dtype=np.uint16), rng.integers(0, mx + 1, (h + 1, w + 2), dtype=np.uint16))]
    ext_a = np.zeros((h + 1, w + 3), np.uint16); ext_a[:h // 2] = mx
    ext_b = np.zeros((h + 1, w + 2), np.uint16); ext_b[h // 2:] = mx
    cases.append((ext_a, ext_b))
    cases.append((np.zeros((h + 1, w + 3), np.uint16), np.full((h + 1, w + 2), mx, np.uint16)))  # negative sum rounding
    for a, b in cases:
        got = lib.aomhip_highbd_variance(_ptr(a, 0, 1) >> 1, a.shape[1], _ptr(b, 0, 0) >> 1, b.shape[1], w, h, bd, C.byref(sse))
        wv, wsse, _ = oracle.variance(a, 0, 1, b, 0, 0, w, h, bd)
        assert (got, sse.value) == (wv, wsse)
        for xo, yo in [(0, 0), (3, 0), (0, 5), (7, 7), (4, 4)]:
            got = lib.aomhip_highbd_sub_pixel_variance(_ptr(a, 0, 1) >> 1, a.shape[1], xo, yo, _ptr(b, 0, 0) >> 1,
                                                       b.shape[1], w, h, bd, C.byref(sse))
            assert (got, sse.value) == oracle.sub_pixel_variance(a, 0, 1, xo, yo, b, 0, 0, w, h, bd)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("w,h", [(16, 16), (4, 4), (8, 16), (32, 32), (64, 32), (128, 128), (4, 16)])
def test_batched_variance_matches_oracle(hip, oracle, ctx, w, h, bd):
    W, H, border, F = 256, 160, 160, 2
    src = [hip.synth.lcg_frame(W, H, 30 + f, 0, bd) for f in range(F)]
    ref = [hip.synth.lcg_frame(W, H, 40 + f, 0, bd) for f in range(F)]
    ps, pr = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    for f in range(F):
        ctx.planes_upload(ps, f, src[f]); ctx.planes_upload(pr, f, ref[f])
    rng = np.random.default_rng(w * h + bd)
    n = 203
    c = np.zeros((F, n), hip.capi.var_cand_dtype)
    lim = border - 4
    c["sx"] = rng.integers(0, W - w + 1, (F, n)); c["sy"] = rng.integers(0, H - h + 1, (F, n))
    c["rx"] = rng.integers(-lim, W + lim - w, (F, n)); c["ry"] = rng.integers(-lim, H + lim - h, (F, n))
    c["xoff"] = rng.integers(0, 8, (F, n)); c["yoff"] = rng.integers(0, 8, (F, n))
    d_c = ctx.to_device(c)
    d_v, d_s = ctx.malloc(F * n * 4), ctx.malloc(F * n * 4)
    for subpel in (False, True):
        ctx.variance_batch(ps, pr, 0, F, w, h, d_c, n, n, d_v, d_s, subpel=subpel)
        gv, gs = ctx.from_device(d_v, (F, n), np.uint32), ctx.from_device(d_s, (F, n), np.uint32)
        for f in range(F):
            sb = oracle.extend_plane(src[f], border, ps.stride); rb = oracle.extend_plane(ref[f], border, pr.stride)
            want = oracle.variance_cands(sb, rb, border, w, h, c[f], subpel, bd)
            assert np.array_equal(gv[f], want[:, 0]) and np.array_equal(gs[f], want[:, 1]), (w, h, bd, subpel, f)
    ctx.planes_free(ps); ctx.planes_free(pr)
    for d in (d_c, d_v, d_s):
        ctx.free(d)
