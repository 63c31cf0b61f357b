"""The RD helpers on the device through the C ABI: aom_[highbd_]sse, the Hadamard family + SATD
(aom_hadamard_{4x4..32x32}, _lp_{8x8,16x16}, aom_highbd_hadamard_*), av1_txb_init_levels -- against the interpreted
reference's vectors (tests/golden/ref_eval_rdhelp.npz) and against the oracle on seeded lists (mirrors
test/hadamard_test.cc's random / extreme-value inputs, test/sum_squares_test.cc's SSE size sweep and
test/encodetxb_test.cc's EncodeTxbInitLevelTest)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _txb(hip, xs, ys, offs):
    b = np.zeros(len(xs), hip.capi.txb_dtype)
    b["x"], b["y"], b["out_offset"] = xs, ys, offs
    return b


def test_rd_helper_goldens(hip, ctx):
    z = np.load(os.path.join(GOLD, "ref_eval_rdhelp.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    K = hip.capi
    planes = {}
    for bd in (8, 12):
        dt = np.uint8 if bd == 8 else np.uint16
        a, b = np.ascontiguousarray(z["sa%d" % bd], dt), np.ascontiguousarray(z["sb%d" % bd], dt)
        pa, pb = ctx.planes_alloc(a.shape[1], a.shape[0], 16, bd, 1), ctx.planes_alloc(a.shape[1], a.shape[0], 16, bd, 1)
        ctx.planes_upload(pa, 0, a); ctx.planes_upload(pb, 0, b)
        planes[bd] = (pa, pb)
    n_checked = 0
    for c in cases:
        if c["kind"] == "sse":
            pa, pb = planes[c["bd"]]
            cand = np.zeros(1, K.sad_cand_dtype)
            cand["sx"], cand["sy"], cand["rx"], cand["ry"] = c["ox"], c["oy"], 2, 3
            d_c, d_o = ctx.to_device(cand), ctx.malloc(8)
            ctx.sse_batch(pa, pb, 0, c["w"], c["h"], d_c, 1, d_o)
            assert int(ctx.from_device(d_o, (1,), np.int64)[0]) == c["value"], c
            ctx.free(d_c); ctx.free(d_o)
        elif c["kind"] == "hadamard":
            r = np.ascontiguousarray(z["r%d" % c["k"]])
            n = c["n"]
            d_r, d_b = ctx.to_device(r), ctx.to_device(_txb(hip, [c["x"]], [c["y"]], [0]))
            d_c, d_s = ctx.malloc(4 * n * n), ctx.malloc(4)
            ctx.hadamard_batch(d_r, r.shape[1], n, c["flavour"], d_b, 1, d_c, d_s)
            got = ctx.from_device(d_c, (n * n,), np.int16 if c["flavour"] == 1 else np.int32)
            assert np.array_equal(got.astype(np.int32), z["c%d" % c["k"]]), c
            assert int(ctx.from_device(d_s, (1,), np.int32)[0]) == c["satd"], c
            for d in (d_r, d_b, d_c, d_s):
                ctx.free(d)
        else:
            coeff, want = np.ascontiguousarray(z["tc%d" % c["k"]]), z["tl%d" % c["k"]]
            d_cf, d_l = ctx.to_device(coeff), ctx.malloc(want.size + 64)
            ctx.txb_init_levels_batch(d_cf, c["w"], c["h"], None, 1, d_l, want.size)
            assert np.array_equal(ctx.from_device(d_l, (want.size,), np.uint8), want), c
            ctx.free(d_cf); ctx.free(d_l)
        n_checked += 1
    assert n_checked == len(cases) >= 50
    for pa, pb in planes.values():
        ctx.planes_free(pa); ctx.planes_free(pb)


@pytest.mark.parametrize("flavour,sizes", [(0, (4, 8, 16, 32)), (1, (8, 16)), (2, (8, 16, 32))])
def test_hadamard_batch_vs_oracle(hip, oracle, ctx, flavour, sizes):
    rng = np.random.default_rng(flavour)
    lib = oracle.lib
    lib.orc_hadamard.restype = C.c_int
    W, H = 160, 128
    for n in sizes:
        for lim in (255, 4095, 32767):
            res = rng.integers(-lim, lim + 1, (H, W)).astype(np.int16)
            res[:32, :32] = lim
            res[32:64, :32] = np.where(rng.integers(0, 2, (32, 32)) > 0, lim, -lim)
            nb = 77
            xs, ys = rng.integers(0, W - n + 1, nb), rng.integers(0, H - n + 1, nb)
            xs[:2], ys[:2] = (0, 0), (0, 32)
            blocks = _txb(hip, xs, ys, rng.permutation(nb) * (n * n))
            d_r, d_b = ctx.to_device(res), ctx.to_device(blocks)
            esz = 2 if flavour == 1 else 4
            d_c, d_s = ctx.malloc(esz * nb * n * n), ctx.malloc(4 * nb)
            ctx.hadamard_batch(d_r, W, n, flavour, d_b, nb, d_c, d_s)
            got = ctx.from_device(d_c, (nb * n * n,), np.int16 if flavour == 1 else np.int32).astype(np.int32)
            satd = ctx.from_device(d_s, (nb,), np.int32)
            out = np.zeros(n * n, np.int32)
            for i in range(nb):
                ws = lib.orc_hadamard(C.c_void_p(res.ctypes.data + (int(ys[i]) * W + int(xs[i])) * 2), C.c_ssize_t(W), n, flavour, C.c_void_p(out.ctypes.data))
                off = int(blocks["out_offset"][i])
                assert np.array_equal(got[off:off + n * n], out), (n, flavour, lim, i)
                assert int(satd[i]) == ws, (n, flavour, lim, i)
            # SATD alone (no coefficient buffer) gives the same numbers
            ctx.hadamard_batch(d_r, W, n, flavour, d_b, nb, None, d_s)
            assert np.array_equal(ctx.from_device(d_s, (nb,), np.int32), satd)
            for d in (d_r, d_b, d_c, d_s):
                ctx.free(d)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_sse_batch_vs_oracle(hip, oracle, ctx, bd):
    rng = np.random.default_rng(bd)
    lib = oracle.lib
    lib.orc_sse.restype = C.c_int64
    W, H, border = 192, 160, 32
    a, b = hip.synth.lcg_frame(W, H, 1, 0, bd), hip.synth.lcg_frame(W, H, 2, 1, bd)
    b[:64, :64] = 0
    a[:64, :64] = (1 << bd) - 1        # the largest possible sum: 128 x 128 would need the 64-bit accumulator at 12 bits
    pa, pb = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pa, 0, a); ctx.planes_upload(pb, 0, b)
    ab, bb = oracle.extend_plane(a, border, pa.stride), oracle.extend_plane(b, border, pb.stride)
    e16 = int(bd > 8)
    for (w, h) in ((4, 4), (8, 8), (16, 16), (64, 64), (128, 128), (5, 7), (33, 17), (64, 1), (1, 64), (128, 3)):
        n = 40
        cand = np.zeros(n, hip.capi.sad_cand_dtype)
        cand["sx"], cand["sy"] = rng.integers(0, W - w + 1, n), rng.integers(0, H - h + 1, n)
        cand["rx"], cand["ry"] = rng.integers(-border, W + border - w + 1, n), rng.integers(-border, H + border - h + 1, n)
        cand[0] = (0, 0, 0, 0)
        d_c, d_o = ctx.to_device(cand), ctx.malloc(8 * n)
        ctx.sse_batch(pa, pb, 0, w, h, d_c, n, d_o)
        got = ctx.from_device(d_o, (n,), np.int64)
        for i in range(n):
            want = lib.orc_sse(C.c_void_p(oracle._addr(ab, border + int(cand["sy"][i]), border + int(cand["sx"][i]))), ab.shape[1],
                               C.c_void_p(oracle._addr(bb, border + int(cand["ry"][i]), border + int(cand["rx"][i]))), bb.shape[1], w, h, e16)
            assert int(got[i]) == want, (w, h, bd, i)
        ctx.free(d_c); ctx.free(d_o)
    ctx.planes_free(pa); ctx.planes_free(pb)


def test_txb_init_levels_batch_vs_oracle(hip, oracle, ctx):
    rng = np.random.default_rng(5)
    lib = oracle.lib
    lib.orc_txb_init_levels.restype = None
    for (w, h) in ((4, 4), (8, 8), (16, 16), (32, 32), (4, 8), (8, 4), (4, 16), (16, 4), (8, 32), (32, 8), (16, 32), (32, 16), (8, 16), (16, 8)):
        nb = 33
        coeff = rng.integers(-200, 201, (nb, w * h)).astype(np.int32)
        coeff[0] = -(1 << 20)
        coeff[1] = 0
        coeff[2, ::3] = 127
        offs = (rng.permutation(nb) * (w * h)).astype(np.uint32)
        size = (h + 4) * (w + 4) + 16
        pitch = size + 13
        d_cf, d_off, d_l = ctx.to_device(coeff), ctx.to_device(offs), ctx.malloc(pitch * nb)
        K = hip.capi
        K.check(K.lib.aomhip_memset(ctx.h, d_l, 0xAA, pitch * nb))
        ctx.txb_init_levels_batch(d_cf, w, h, d_off, nb, d_l, pitch)
        got = ctx.from_device(d_l, (nb, pitch), np.uint8)
        want = np.zeros(size, np.uint8)
        flat = coeff.reshape(-1)
        for i in range(nb):
            blk = np.ascontiguousarray(flat[int(offs[i]):int(offs[i]) + w * h])
            lib.orc_txb_init_levels(C.c_void_p(blk.ctypes.data), w, h, C.c_void_p(want.ctypes.data))
            assert np.array_equal(got[i, :size], want), (w, h, i)
            assert (got[i, size:] == 0xAA).all()      # nothing written past the block's levels
        for d in (d_cf, d_off, d_l):
            ctx.free(d)


def test_rd_helpers_reject_bad_arguments(hip, ctx):
    K = hip.capi
    d = ctx.malloc(4096)
    for n, fl in ((4, 1), (32, 1), (4, 2), (64, 0), (12, 0), (8, 3)):
        with pytest.raises(K.AomHipError):
            ctx.hadamard_batch(d, 64, n, fl, d, 1, d, d)
    with pytest.raises(K.AomHipError):
        ctx.hadamard_batch(d, 64, 8, 0, d, 1, None, None)
    with pytest.raises(K.AomHipError):
        ctx.txb_init_levels_batch(d, 8, 8, None, 1, d, 100)      # pitch too small
    with pytest.raises(K.AomHipError):
        ctx.txb_init_levels_batch(d, 64, 8, None, 1, d, 4096)    # levels exist up to 32 x 32 (64-point transforms code 32)
    ps = ctx.planes_alloc(64, 64, 16, 8, 1)
    with pytest.raises(K.AomHipError):
        ctx.sse_batch(ps, ps, 0, 0, 8, d, 1, d)
    ctx.hadamard_batch(d, 64, 8, 0, None, 0, d, d)
    ctx.planes_free(ps); ctx.free(d)
