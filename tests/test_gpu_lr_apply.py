"""aomhip_apply_selfguided_restoration_batch / aomhip_wiener_convolve_add_src_batch (csrc/restoration.hip) against (a) the interpreted reference
(tests/golden/ref_eval_lr_apply.npz, directly) and (b) the oracle on the restoration units of a frame, each unit with its own parameters."""
import numpy as np
import pytest

from test_golden_lr_apply import load, orc_lr

pytestmark = pytest.mark.gpu


def test_device_restoration_filters_reproduce_the_interpreted_reference(hip, ctx):
    z, cases = load()
    capi = hip.capi
    for c in cases:
        img = z["img%d" % c["k"]]
        Hh, S = img.shape
        w, h = c["w"], c["h"]
        p, q = ctx.planes_alloc(S, Hh, 8, c["bd"], 1), ctx.planes_alloc(S, Hh, 8, c["bd"], 1)
        ctx.planes_upload(p, 0, img)
        ctx.planes_upload(q, 0, np.zeros_like(img))
        unit = np.zeros(1, capi.rect_dtype)
        unit["h_start"], unit["h_end"], unit["v_start"], unit["v_end"] = 3, 3 + w, 3, 3 + h
        d_u = ctx.to_device(unit)
        if c["kind"] == "sgr":
            d_i, d_x = ctx.to_device(np.array([c["idx"]], np.int32)), ctx.to_device(np.array(c["xqd"], np.int32))
            d_f0, d_f1 = ctx.malloc(4 * w * h), ctx.malloc(4 * w * h)
            ctx.apply_selfguided_restoration_batch(p, 0, q, 0, d_u, unit, 1, d_i, d_x, w, h, d_f0, d_f1, w, w * h)
            extra = (d_i, d_x, d_f0, d_f1)
        else:
            d_f = ctx.to_device(np.array(c["fx"] + c["fy"], np.int16))
            ctx.wiener_convolve_add_src_batch(p, 0, q, 0, d_u, unit, 1, d_f, w, h)
            extra = (d_f,)
        got = ctx.planes_download(q, 0)[8:8 + Hh, 8:8 + S]
        assert np.array_equal(got[3:3 + h, 3:3 + w].ravel().astype(np.uint16), z["out%d" % c["k"]]), c
        got[3:3 + h, 3:3 + w] = 0
        assert not got.any()
        for d in (d_u,) + extra:
            ctx.free(d)
        ctx.planes_free(p); ctx.planes_free(q)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_units_of_a_frame_equal_the_oracle(hip, oracle, ctx, bd):
    capi = hip.capi
    rng = np.random.default_rng(90 + bd)
    W, H, B = 328, 200, 16
    mx = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    dat = rng.integers(0, mx + 1, (H, W)).astype(dt)
    p, q = ctx.planes_alloc(W, H, B, bd, 1), ctx.planes_alloc(W, H, B, bd, 1)
    ctx.planes_upload(p, 0, dat)
    ext = oracle.extend_plane(dat, B, W + 2 * B)
    units = [(x, min(x + 128, W), y, min(y + 96, H)) for y in range(0, H, 96) for x in range(0, W, 128)]
    n = len(units)
    rec = np.zeros(n, capi.rect_dtype)
    for i, (x0, x1, y0, y1) in enumerate(units):
        rec["h_start"][i], rec["h_end"][i], rec["v_start"][i], rec["v_end"][i] = x0, x1, y0, y1
    d_u = ctx.to_device(rec)
    # Wiener: a filter pair per unit
    filt = np.zeros((n, 16), np.int16)
    for i in range(n):
        for o in (0, 8):
            t0, t1, t2 = int(rng.integers(-5, 11)), int(rng.integers(-23, 9)), int(rng.integers(-17, 47))
            filt[i, o:o + 8] = [t0, t1, t2, -2 * (t0 + t1 + t2), t2, t1, t0, 0]
    d_f = ctx.to_device(filt)
    ctx.planes_upload(q, 0, np.zeros_like(dat))
    ctx.wiener_convolve_add_src_batch(p, 0, q, 0, d_u, rec, n, d_f, 128, 96)
    got = ctx.planes_download(q, 0)[B:B + H, B:B + W]
    for i, (x0, x1, y0, y1) in enumerate(units):
        c = {"kind": "wiener", "bd": bd, "w": x1 - x0, "h": y1 - y0, "fx": filt[i, :8].tolist(), "fy": filt[i, 8:].tolist()}
        assert np.array_equal(got[y0:y1, x0:x1], orc_lr(oracle, ext, c, B + x0, B + y0)), (i, units[i])
    # self-guided: a parameter set and an xqd pair per unit
    idx = np.array([(5 * i + 2) % 16 for i in range(n)], np.int32)
    xqd = np.stack([rng.integers(-96, 32, n), rng.integers(-32, 96, n)], 1).astype(np.int32)
    d_i, d_x = ctx.to_device(idx), ctx.to_device(xqd)
    d_f0, d_f1 = ctx.malloc(4 * n * 128 * 96), ctx.malloc(4 * n * 128 * 96)
    ctx.planes_upload(q, 0, np.zeros_like(dat))
    ctx.apply_selfguided_restoration_batch(p, 0, q, 0, d_u, rec, n, d_i, d_x, 128, 96, d_f0, d_f1, 128, 128 * 96)
    got = ctx.planes_download(q, 0)[B:B + H, B:B + W]
    for i, (x0, x1, y0, y1) in enumerate(units):
        c = {"kind": "sgr", "bd": bd, "w": x1 - x0, "h": y1 - y0, "idx": int(idx[i]), "xqd": xqd[i].tolist()}
        assert np.array_equal(got[y0:y1, x0:x1], orc_lr(oracle, ext, c, B + x0, B + y0)), (i, units[i], int(idx[i]))
    for d in (d_u, d_f, d_i, d_x, d_f0, d_f1):
        ctx.free(d)
    ctx.planes_free(p); ctx.planes_free(q)
