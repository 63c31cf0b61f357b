"""aomhip_selfguided_restoration_batch (csrc/restoration.hip) against (a) the interpreted reference's av1_selfguided_restoration_c
(tests/golden/ref_eval_sgr.npz, directly) and (b) the oracle on the restoration units of a frame -- then the whole device chain of
search_sgrproj's inner evaluation: filter -> projection statistics -> projection error, against the oracle's chain."""
import numpy as np
import pytest

from test_golden_proj import bind as bind_proj
from test_golden_sgr import load, orc_sgr

pytestmark = pytest.mark.gpu


def test_device_self_guided_filter_reproduces_the_interpreted_reference(hip, ctx):
    z, cases = load()
    capi = hip.capi
    # (planes are uploaded without their border, so the fixtures go through a larger plane: unit at (3, 3))
    for c in cases:
        k, w, h = c["k"], c["w"], c["h"]
        img = z["img%d" % k]
        Hh, S = img.shape
        p = ctx.planes_alloc(S, Hh, 8, c["bd"], 1)
        ctx.planes_upload(p, 0, img)
        unit = np.zeros(1, capi.rect_dtype)
        unit["h_start"], unit["h_end"], unit["v_start"], unit["v_end"] = 3, 3 + w, 3, 3 + h
        d_u, d_i = ctx.to_device(unit), ctx.to_device(np.array([c["idx"]], np.int32))
        init = np.full(w * h, -7, np.int32)
        d_f0, d_f1 = ctx.to_device(init), ctx.to_device(init)
        ctx.selfguided_restoration_batch(p, 0, d_u, unit, 1, d_i, w, h, d_f0, d_f1, w, w * h)
        assert np.array_equal(ctx.from_device(d_f0, (w * h,), np.int32), z["f0_%d" % k]), c
        assert np.array_equal(ctx.from_device(d_f1, (w * h,), np.int32), z["f1_%d" % k]), c
        for d in (d_u, d_i, d_f0, d_f1):
            ctx.free(d)
        ctx.planes_free(p)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_units_of_a_frame_and_the_search_chain_equal_the_oracle(hip, oracle, ctx, bd):
    capi = hip.capi
    lib = bind_proj(oracle)
    rng = np.random.default_rng(60 + bd)
    W, H, B = 328, 200, 16
    mx = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    yy, xx = np.mgrid[0:H, 0:W]
    src = np.clip((np.sin(xx / 11.0) + np.cos(yy / 8.0) + 2) * 0.25 * mx + rng.integers(-mx // 30, mx // 30 + 1, (H, W)), 0, mx).astype(dt)
    dat = np.clip(src.astype(np.int32) + rng.integers(-mx // 12, mx // 12 + 1, (H, W)), 0, mx).astype(dt)
    ps, pd = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pd, 0, dat)
    ext = oracle.extend_plane(dat, B, W + 2 * B)     # what the device plane holds around the frame
    units = [(x, min(x + 128, W), y, min(y + 96, H)) for y in range(0, H, 96) for x in range(0, W, 128)]   # remainders 72 wide / 8 high
    n = len(units)
    rec = np.zeros(n, capi.rect_dtype)
    for i, (x0, x1, y0, y1) in enumerate(units):
        rec["h_start"][i], rec["h_end"][i], rec["v_start"][i], rec["v_end"][i] = x0, x1, y0, y1
    idx = np.array([(3 * i + 1) % 16 for i in range(n)], np.int32)
    FS, pitch = 128, 128 * 96
    d_u, d_i = ctx.to_device(rec), ctx.to_device(idx)
    init = np.full(n * pitch, -7, np.int32)
    d_f0, d_f1 = ctx.to_device(init), ctx.to_device(init)
    ctx.selfguided_restoration_batch(pd, 0, d_u, rec, n, d_i, 128, 96, d_f0, d_f1, FS, pitch)
    f0 = ctx.from_device(d_f0, (n, 96, FS), np.int32); f1 = ctx.from_device(d_f1, (n, 96, FS), np.int32)
    sgr_r = np.array([[2, 1]] * 10 + [[0, 1]] * 4 + [[2, 0]] * 2, np.int32)
    radii = sgr_r[idx]
    for i, (x0, x1, y0, y1) in enumerate(units):
        w, h = x1 - x0, y1 - y0
        want0, want1 = orc_sgr(oracle, ext, bd, B + x0, B + y0, w, h, int(idx[i]))
        assert np.array_equal(f0[i, :h, :w], want0) and np.array_equal(f1[i, :h, :w], want1), (i, units[i], int(idx[i]))
        assert (f0[i, h:] == -7).all() and (f0[i, :, w:] == -7).all()
    # the chain: the filter's outputs stay on the device and feed the statistics and the error of a few xq
    n_xq = 3
    xq = np.stack([rng.integers(-96, 32, (n, n_xq)), rng.integers(-32, 96, (n, n_xq))], 2).astype(np.int32)
    d_r, d_xq = ctx.to_device(radii), ctx.to_device(xq)
    d_H, d_C, d_e = ctx.malloc(32 * n), ctx.malloc(16 * n), ctx.malloc(8 * n * n_xq)
    ctx.calc_proj_params_batch(ps, 0, pd, 0, d_u, n, d_f0, d_f1, FS, pitch, d_r, d_H, d_C)
    ctx.pixel_proj_error_batch(ps, 0, pd, 0, d_u, n, d_f0, d_f1, FS, pitch, d_r, d_xq, n_xq, d_e)
    Hg, Cg, eg = ctx.from_device(d_H, (n, 4), np.int64), ctx.from_device(d_C, (n, 2), np.int64), ctx.from_device(d_e, (n, n_xq), np.int64)
    for i, (x0, x1, y0, y1) in enumerate(units):
        w, h = x1 - x0, y1 - y0
        s_, d_ = np.ascontiguousarray(src[y0:y1, x0:x1]), np.ascontiguousarray(dat[y0:y1, x0:x1])
        a, b = np.ascontiguousarray(f0[i]), np.ascontiguousarray(f1[i])
        Hw, Cw = np.zeros(4, np.int64), np.zeros(2, np.int64)
        lib.orc_calc_proj_params(s_.ctypes.data, w, h, w, d_.ctypes.data, w, a.ctypes.data, FS, b.ctypes.data, FS, int(bd > 8), int(radii[i, 0]), int(radii[i, 1]),
                                 Hw.ctypes.data, Cw.ctypes.data)
        assert np.array_equal(Hg[i], Hw) and np.array_equal(Cg[i], Cw), i
        for k in range(n_xq):
            assert int(eg[i, k]) == lib.orc_pixel_proj_error(s_.ctypes.data, w, h, w, d_.ctypes.data, w, a.ctypes.data, FS, b.ctypes.data, FS, int(bd > 8),
                                                            int(radii[i, 0]), int(radii[i, 1]), int(xq[i, k, 0]), int(xq[i, k, 1])), (i, k)
    for d in (d_u, d_i, d_f0, d_f1, d_r, d_xq, d_H, d_C, d_e):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pd)


def test_bad_arguments_are_refused(hip, ctx):
    capi = hip.capi
    p = ctx.planes_alloc(64, 64, 2, 8, 1)     # border too small for the filter's 3-pixel reach
    q = ctx.planes_alloc(64, 64, 8, 8, 1)
    d = ctx.malloc(65536)
    bad = np.zeros(1, capi.rect_dtype); bad["h_end"], bad["v_end"] = 80, 32
    with pytest.raises(capi.AomHipError):
        ctx.selfguided_restoration_batch(p, 0, d, None, 1, d, 32, 32, d, d, 32, 1024)
    with pytest.raises(capi.AomHipError):
        ctx.selfguided_restoration_batch(q, 0, d, bad, 1, d, 128, 32, d, d, 128, 4096)   # the unit leaves the plane
    with pytest.raises(capi.AomHipError):
        ctx.selfguided_restoration_batch(q, 0, d, None, 1, d, 64, 64, d, d, 32, 4096)    # rows of 32 cannot hold 64-wide units
    ctx.free(d)
    ctx.planes_free(p); ctx.planes_free(q)
