"""aomhip_calc_proj_params_batch / aomhip_pixel_proj_error_batch (csrc/restoration.hip) against (a) the interpreted reference
(tests/golden/ref_eval_proj.npz: every unit through the device, directly) and (b) the oracle on the restoration units of a frame."""
import numpy as np
import pytest

from test_golden_proj import bind, load, planes_of

pytestmark = pytest.mark.gpu


def test_device_projection_statistics_reproduce_the_interpreted_reference(hip, ctx):
    z, cases = load()
    capi = hip.capi
    for c in cases:
        src, dat, f0, f1 = planes_of(z, c)
        h, S = src.shape
        ps, pd = ctx.planes_alloc(S, h, 8, c["bd"], 1), ctx.planes_alloc(S, h, 8, c["bd"], 1)
        ctx.planes_upload(ps, 0, src); ctx.planes_upload(pd, 0, dat)
        unit = np.zeros(1, capi.rect_dtype)
        unit["h_start"], unit["h_end"], unit["v_start"], unit["v_end"] = 0, c["w"], 0, c["h"]
        d_u, d_f0, d_f1 = ctx.to_device(unit), ctx.to_device(f0), ctx.to_device(f1)
        d_r = ctx.to_device(np.array(c["r"], np.int32))
        d_H, d_C = ctx.malloc(32), ctx.malloc(16)
        ctx.calc_proj_params_batch(ps, 0, pd, 0, d_u, 1, d_f0, d_f1, c["FS"], f0.size, d_r, d_H, d_C)
        assert ctx.from_device(d_H, (4,), np.int64).tolist() == c["H"] and ctx.from_device(d_C, (2,), np.int64).tolist() == c["C"], c
        xq = np.array(c["xq"], np.int32)
        d_xq, d_e = ctx.to_device(xq), ctx.malloc(8 * len(xq))
        ctx.pixel_proj_error_batch(ps, 0, pd, 0, d_u, 1, d_f0, d_f1, c["FS"], f0.size, d_r, d_xq, len(xq), d_e)
        assert ctx.from_device(d_e, (len(xq),), np.int64).tolist() == c["err"], c
        for d in (d_u, d_f0, d_f1, d_r, d_H, d_C, d_xq, d_e):
            ctx.free(d)
        ctx.planes_free(ps); ctx.planes_free(pd)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_units_of_a_frame_equal_the_oracle(hip, oracle, ctx, bd):
    capi = hip.capi
    lib = bind(oracle)
    rng = np.random.default_rng(40 + bd)
    W, H, B = 400, 272, 16
    mx = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    src = rng.integers(0, mx + 1, (H, W)).astype(dt)
    dat = np.clip(src.astype(np.int32) + rng.integers(-mx // 20, mx // 20 + 1, (H, W)), 0, mx).astype(dt)
    ps, pd = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pd, 0, dat)
    # restoration units of 128 x 128 with the frame's remainder columns / rows, each with its own sgr_params radii
    units = [(x, min(x + 128, W), y, min(y + 128, H)) for y in range(0, H, 128) for x in range(0, W, 128)]
    n = len(units)
    rec = np.zeros(n, capi.rect_dtype)
    for i, (x0, x1, y0, y1) in enumerate(units):
        rec["h_start"][i], rec["h_end"][i], rec["v_start"][i], rec["v_end"][i] = x0, x1, y0, y1
    radii = np.array([[2, 1], [2, 0], [0, 1]] * n, np.int32)[:n]
    FS, pitch = 136, 136 * 128
    f0 = np.zeros((n, 128, FS), np.int32); f1 = np.zeros((n, 128, FS), np.int32)
    for i, (x0, x1, y0, y1) in enumerate(units):
        blk = dat[y0:y1, x0:x1].astype(np.int32) << 4
        f0[i, :y1 - y0, :x1 - x0] = blk + rng.integers(-mx, mx + 1, blk.shape)
        f1[i, :y1 - y0, :x1 - x0] = blk + rng.integers(-mx, mx + 1, blk.shape)
    n_xq = 7
    xq = np.stack([rng.integers(-96, 32, (n, n_xq)), rng.integers(-32, 96, (n, n_xq))], 2).astype(np.int32)
    d_u, d_f0, d_f1, d_r, d_xq = ctx.to_device(rec), ctx.to_device(f0), ctx.to_device(f1), ctx.to_device(radii), ctx.to_device(xq)
    d_H, d_C, d_e = ctx.malloc(32 * n), ctx.malloc(16 * n), ctx.malloc(8 * n * n_xq)
    ctx.calc_proj_params_batch(ps, 0, pd, 0, d_u, n, d_f0, d_f1, FS, pitch, d_r, d_H, d_C)
    ctx.pixel_proj_error_batch(ps, 0, pd, 0, d_u, n, d_f0, d_f1, FS, pitch, d_r, d_xq, n_xq, d_e)
    Hg, Cg, eg = ctx.from_device(d_H, (n, 4), np.int64), ctx.from_device(d_C, (n, 2), np.int64), ctx.from_device(d_e, (n, n_xq), np.int64)
    for i, (x0, x1, y0, y1) in enumerate(units):
        s_, d_ = np.ascontiguousarray(src[y0:y1, x0:x1]), np.ascontiguousarray(dat[y0:y1, x0:x1])
        w, h = x1 - x0, y1 - y0
        Hw, Cw = np.zeros(4, np.int64), np.zeros(2, np.int64)
        a, b = np.ascontiguousarray(f0[i]), np.ascontiguousarray(f1[i])
        lib.orc_calc_proj_params(s_.ctypes.data, w, h, w, d_.ctypes.data, w, a.ctypes.data, FS, b.ctypes.data, FS, int(bd > 8), int(radii[i, 0]), int(radii[i, 1]),
                                 Hw.ctypes.data, Cw.ctypes.data)
        assert np.array_equal(Hg[i], Hw) and np.array_equal(Cg[i], Cw), i
        for k in range(n_xq):
            want = lib.orc_pixel_proj_error(s_.ctypes.data, w, h, w, d_.ctypes.data, w, a.ctypes.data, FS, b.ctypes.data, FS, int(bd > 8), int(radii[i, 0]),
                                            int(radii[i, 1]), int(xq[i, k, 0]), int(xq[i, k, 1]))
            assert int(eg[i, k]) == want, (i, k)
    for d in (d_u, d_f0, d_f1, d_r, d_xq, d_H, d_C, d_e):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pd)


def test_bad_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p8, p10 = ctx.planes_alloc(64, 64, 8, 8, 1), ctx.planes_alloc(64, 64, 8, 10, 1)
    d = ctx.malloc(65536)
    with pytest.raises(capi.AomHipError):
        ctx.calc_proj_params_batch(p8, 0, p10, 0, d, 1, d, d, 64, 4096, d, d, d)     # bit depths differ
    with pytest.raises(capi.AomHipError):
        ctx.pixel_proj_error_batch(p8, 0, p8, 0, d, 1, d, d, 64, 4096, d, None, 3, d)   # no xq
    ctx.calc_proj_params_batch(p8, 0, p8, 0, None, 0, None, None, 64, 4096, None, None, None)   # an empty batch is not an error
    ctx.free(d)
    ctx.planes_free(p8); ctx.planes_free(p10)
