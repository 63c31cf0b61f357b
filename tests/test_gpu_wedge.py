"""aomhip_wedge_sse_from_residuals_batch / _sign_from_residuals_batch / _compute_delta_squares_batch (csrc/rd_helpers.hip) against (a) the
interpreted reference's av1_wedge_*_c (tests/golden/ref_eval_wedge.npz, directly) and (b) the oracle on batches shaped like pick_wedge's use:
every block of a frame x the 16 wedges of its size."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from test_golden_wedge import bind

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_wedge_helpers_reproduce_the_interpreted_reference(hip, ctx):
    z = np.load(os.path.join(HERE, "golden", "ref_eval_wedge.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    for c in cases:
        k, n = c["k"], c["N"]
        d_r1, d_d, d_m = (ctx.to_device(np.ascontiguousarray(z["%s_%d" % (s, k)])) for s in ("r1", "d", "m"))
        d_sse = ctx.malloc(8)
        ctx.wedge_sse_from_residuals_batch(d_r1, d_d, d_m, n, 1, 1, d_sse)
        assert int(ctx.from_device(d_sse, (1,), np.uint64)[0]) == c["sse"], c
        d_a, d_b, d_ds = ctx.to_device(np.ascontiguousarray(z["a_%d" % k])), ctx.to_device(np.ascontiguousarray(z["b_%d" % k])), ctx.malloc(2 * n)
        ctx.wedge_compute_delta_squares_batch(d_a, d_b, n, 1, d_ds)
        assert np.array_equal(ctx.from_device(d_ds, (n,), np.int16), z["ds_%d" % k]), c
        d_sign = ctx.malloc(1)
        for limit, want in zip(c["limits"], c["signs"]):
            d_lim = ctx.to_device(np.array([limit], np.int64))
            ctx.wedge_sign_from_residuals_batch(d_ds, d_m, n, 1, 1, d_lim, d_sign)
            assert int(ctx.from_device(d_sign, (1,), np.int8)[0]) == want, (c, limit)
            ctx.free(d_lim)
        for d in (d_r1, d_d, d_m, d_sse, d_a, d_b, d_ds, d_sign):
            ctx.free(d)


@pytest.mark.parametrize("n,n_blocks,bits", [(64, 400, 8), (256, 300, 10), (512, 100, 12), (1024, 64, 15), (4096, 9, 12)])
def test_batches_equal_the_oracle(hip, oracle, ctx, n, n_blocks, bits):
    lib = bind(oracle)
    rng = np.random.default_rng(n + bits)
    lim = (1 << bits) - 1
    n_masks = 16   # the wedge codebook of a block size (av1_wedge_params_lookup: 16 wedge types)
    r0 = rng.integers(-lim, lim + 1, (n_blocks, n)).astype(np.int16)
    r1 = rng.integers(-lim, lim + 1, (n_blocks, n)).astype(np.int16)
    d = np.clip(r0.astype(np.int32) - r1, -32768, 32767).astype(np.int16)   # diff10 of pick_wedge
    masks = np.clip(rng.integers(-20, 85, (n_masks, n)), 0, 64).astype(np.uint8)
    d_r0, d_r1, d_d, d_m = ctx.to_device(r0), ctx.to_device(r1), ctx.to_device(d), ctx.to_device(masks)
    d_ds, d_sse, d_sign = ctx.malloc(2 * n * n_blocks), ctx.malloc(8 * n_blocks * n_masks), ctx.malloc(n_blocks * n_masks)
    ctx.wedge_compute_delta_squares_batch(d_r0, d_r1, n, n_blocks, d_ds)
    ds = ctx.from_device(d_ds, (n_blocks, n), np.int16)
    want_ds = np.zeros_like(ds)
    lib.orc_wedge_compute_delta_squares(want_ds.ctypes.data, r0.ctypes.data, r1.ctypes.data, n * n_blocks)
    assert np.array_equal(ds, want_ds)
    # sign_limit of pick_wedge: ((sum r0^2 - sum r1^2) * (1 << WEDGE_WEIGHT_BITS)) / 2; two blocks get limits at the decision point of mask 0
    limits = (((r0.astype(np.int64) ** 2).sum(1) - (r1.astype(np.int64) ** 2).sum(1)) * 64) // 2
    acc0 = (ds.astype(np.int64) * masks[0]).sum(1)
    limits[0], limits[1] = acc0[0], acc0[1] - 1
    d_lim = ctx.to_device(limits.astype(np.int64))
    ctx.wedge_sign_from_residuals_batch(d_ds, d_m, n, n_blocks, n_masks, d_lim, d_sign)
    ctx.wedge_sse_from_residuals_batch(d_r1, d_d, d_m, n, n_blocks, n_masks, d_sse)
    sign, sse = ctx.from_device(d_sign, (n_blocks, n_masks), np.int8), ctx.from_device(d_sse, (n_blocks, n_masks), np.uint64)
    assert sign[0, 0] == 0 and sign[1, 0] == 1
    for i in range(n_blocks):
        for k in range(n_masks):
            assert sign[i, k] == lib.orc_wedge_sign_from_residuals(ds[i].ctypes.data, masks[k].ctypes.data, n, int(limits[i])), (i, k)
            assert int(sse[i, k]) == lib.orc_wedge_sse_from_residuals(r1[i].ctypes.data, d[i].ctypes.data, masks[k].ctypes.data, n), (i, k)
    assert 0 < int(sign.sum()) < sign.size
    for x in (d_r0, d_r1, d_d, d_m, d_ds, d_sse, d_sign, d_lim):
        ctx.free(x)


def test_bad_arguments_are_refused(hip, ctx):
    capi = hip.capi
    d = ctx.malloc(4096)
    with pytest.raises(capi.AomHipError):
        ctx.wedge_sse_from_residuals_batch(d, d, d, 48, 1, 1, d)        # N is a multiple of 64
    with pytest.raises(capi.AomHipError):
        ctx.wedge_sign_from_residuals_batch(d, d, 64, 1, 1, None, d)    # no limits
    with pytest.raises(capi.AomHipError):
        ctx.wedge_compute_delta_squares_batch(d, None, 64, 1, d)
    ctx.wedge_sse_from_residuals_batch(None, None, None, 64, 0, 16, None)   # an empty batch is not an error
    ctx.free(d)
