"""The compound / masked / OBMC members of the encoder's kernel table (aom_variance_fn_ptr_t: sdaf, svaf, jsdaf,
jsvaf, msdf, msvf, osdf, ovf, osvf) on the device, through the C ABI: against the golden vectors of the interpreted
reference (tests/golden/ref_eval_compound.npz), against the oracle on seeded candidate lists for all 22 block
sizes x 8/10/12-bit, and through the rtcd-signature vtable entries (mirrors test/variance_test.cc's
AvxSubpelAvgVarianceTest / AvxDistWtdSubpelAvgVarianceTest / AvxObmcSubpelVarianceTest, test/masked_variance_test.cc,
test/masked_sad_test.cc and test/obmc_sad_test.cc: random blocks, extreme blocks, every sub-pel offset)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOCK_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32),
               (64, 64), (64, 128), (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def _params(hip, kind, subpel, fwd=0, bck=0, mask_stride=0, invert=0):
    return hip.capi.CompoundParams(kind, subpel, fwd, bck, mask_stride, invert)


def _one(hip, ctx, planes_a, planes_b, c, p, w, h, sp=None, mask=None, ws=None, om=None, xo=0, yo=0):
    """One candidate through aomhip_compound_batch: the `a` operand (interpolated / blended) is the fixture's plane a at
    (ax, ay) in the ref slot, the compared block is plane b at (bx, by) in the src slot.  Returns (var, sse, sad)."""
    cand = np.zeros(1, hip.capi.var_cand_dtype)
    cand["sx"], cand["sy"], cand["rx"], cand["ry"], cand["xoff"], cand["yoff"] = c["bx"], c["by"], c["ax"], c["ay"], xo, yo
    bufs = [ctx.to_device(cand), ctx.malloc(16)]
    d_sp = d_m = d_ws = d_om = None
    if sp is not None:
        d_sp = ctx.to_device(sp); bufs.append(d_sp)
    if mask is not None:
        d_m = ctx.to_device(mask); bufs.append(d_m)
    if ws is not None:
        d_ws, d_om = ctx.to_device(ws), ctx.to_device(om)
        bufs += [d_ws, d_om]
    d_o = bufs[1]
    ctx.compound_batch(planes_b, planes_a, 0, 1, w, h, bufs[0], 1, 0, p, d_sp, d_m, d_ws, d_om, None, None, d_o, d_o + 4, d_o + 8)
    out = ctx.from_device(d_o, (3,), np.uint32)
    for d in bufs:
        ctx.free(d)
    return int(out[0]), int(out[1]), int(out[2])


def test_compound_goldens(hip, ctx):
    z = np.load(os.path.join(GOLD, "ref_eval_compound.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    K = hip.capi
    planes = {}
    for bd in (8, 10, 12):
        a, b = z["a%d" % bd], z["b%d" % bd]
        dt = np.uint8 if bd == 8 else np.uint16
        H, W = a.shape
        pa, pb = ctx.planes_alloc(W, H, 32, bd, 1), ctx.planes_alloc(W, H, 32, bd, 1)
        ctx.planes_upload(pa, 0, np.ascontiguousarray(a, dt)); ctx.planes_upload(pb, 0, np.ascontiguousarray(b, dt))
        planes[bd] = (pa, pb, dt)
    checked = 0
    for c in cases:
        if c.get("hbd8"):
            continue  # 8-bit content in 16-bit containers: covered through the vtable test below
        bd, w, h, k = c["bd"], c["w"], c["h"], c["k"]
        pa, pb, dt = planes[bd]
        sp, mask, ms = np.ascontiguousarray(z["sp%d" % k], dt), np.ascontiguousarray(z["mask%d" % k]), c["mask_stride"]
        ws, om = np.ascontiguousarray(z["ws%d" % k]), np.ascontiguousarray(z["om%d" % k])
        sh = {8: 0, 10: 2, 12: 4}[bd]
        for xo, yo, v, sse in c["svaf"]:
            assert _one(hip, ctx, pa, pb, c, _params(hip, K.COMP_AVG, 1), w, h, sp=sp, xo=xo, yo=yo)[:2] == (v, sse), ("svaf", c)
        for xo, yo, fwd, bck, v, sse in c["jsvaf"]:
            assert _one(hip, ctx, pa, pb, c, _params(hip, K.COMP_DIST_WTD, 1, fwd, bck), w, h, sp=sp, xo=xo, yo=yo)[:2] == (v, sse), ("jsvaf", c)
        for xo, yo, inv, v, sse in c["msvf"]:
            got = _one(hip, ctx, pa, pb, c, _params(hip, K.COMP_MASK, 1, 0, 0, ms, inv), w, h, sp=sp, mask=mask, xo=xo, yo=yo)
            assert got[:2] == (v, sse), ("msvf", c)
        for inv, v in c["msdf"]:
            got = _one(hip, ctx, pa, pb, c, _params(hip, K.COMP_MASK, 0, 0, 0, ms, inv), w, h, sp=sp, mask=mask)
            assert got[2] == v >> sh, ("msdf", c)
        got = _one(hip, ctx, pa, pb, c, _params(hip, K.COMP_OBMC, 0), w, h, ws=ws, om=om)
        assert got[2] == c["osdf"] >> sh and list(got[:2]) == c["ovf"], ("osdf/ovf", c, got)
        for xo, yo, v, sse in c["osvf"]:
            assert _one(hip, ctx, pa, pb, c, _params(hip, K.COMP_OBMC, 1), w, h, ws=ws, om=om, xo=xo, yo=yo)[:2] == (v, sse), ("osvf", c)
        checked += 1
    assert checked >= 25
    for pa, pb, _ in planes.values():
        ctx.planes_free(pa); ctx.planes_free(pb)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_compound_batch_vs_oracle(hip, oracle, ctx, bd):
    """All 22 sizes, candidate lists with their own second-predictor / mask / OBMC block each, every kind, sub-pel and
    full-pel, positions reaching into the border, saturated and zero blocks."""
    rng = np.random.default_rng(100 + bd)
    K = hip.capi
    W, H, border = 256, 192, 64
    src, ref = hip.synth.lcg_frame(W, H, 3, 0, bd), hip.synth.lcg_frame(W, H, 4, 1, bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 2)
    ctx.planes_upload(ps, 1, src); ctx.planes_upload(pr, 1, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    dt = np.uint8 if bd == 8 else np.uint16
    mx = (1 << bd) - 1
    for (w, h) in BLOCK_SIZES:
        n, nblk = (60, 5) if w * h <= 1024 else (24, 3)
        cands = np.zeros(n, K.var_cand_dtype)
        cands["sx"], cands["sy"] = rng.integers(0, W - w + 1, n), rng.integers(0, H - h + 1, n)
        cands["rx"], cands["ry"] = rng.integers(-border + 1, W + border - w - 1, n), rng.integers(-border + 1, H + border - h - 1, n)
        cands["xoff"], cands["yoff"] = rng.integers(0, 8, n), rng.integers(0, 8, n)
        cands["xoff"][:3], cands["yoff"][:3] = (0, 7, 0), (0, 0, 7)
        preds = rng.integers(0, mx + 1, (nblk, h, w)).astype(dt)
        preds[0], preds[1] = mx, 0
        pidx = rng.integers(0, nblk, n).astype(np.uint32)
        ms = w + 5
        masks = rng.integers(0, 65, (nblk, h, ms)).astype(np.uint8)
        masks[0], masks[1] = 64, 0
        moff = (rng.integers(0, nblk, n) * (h * ms)).astype(np.uint32)
        om = rng.integers(0, 4097, (nblk, h, w)).astype(np.int32)
        ws = (rng.integers(0, mx + 1, (nblk, h, w)) * 4096 - rng.integers(0, mx + 1, (nblk, h, w)) * (4096 - om)).astype(np.int32)
        om[0], ws[0] = 4096, mx * 4096
        d_c, d_p, d_i, d_m, d_mo = ctx.to_device(cands), ctx.to_device(preds), ctx.to_device(pidx), ctx.to_device(masks), ctx.to_device(moff)
        d_ws, d_om, d_o = ctx.to_device(ws), ctx.to_device(om), ctx.malloc(12 * n)
        runs = [(K.COMP_AVG, 0, 0, 0), (K.COMP_DIST_WTD, 9, 7, 0), (K.COMP_DIST_WTD, 4, 12, 0), (K.COMP_MASK, 0, 0, 0), (K.COMP_MASK, 0, 0, 1),
                (K.COMP_OBMC, 0, 0, 0)]
        for kind, fwd, bck, inv in runs:
            for subpel in (0, 1):
                p = _params(hip, kind, subpel, fwd, bck, ms, inv)
                ctx.compound_batch(ps, pr, 1, 1, w, h, d_c, n, 0, p, d_p, d_m, d_ws, d_om, d_i, d_mo, d_o, d_o + 4 * n, d_o + 8 * n)
                got = ctx.from_device(d_o, (3, n), np.uint32)
                wv, wq, wsad = oracle.compound_batch(sb, rb, border, w, h, cands, kind, subpel, bd, preds, pidx, fwd, bck, masks.reshape(-1), ms,
                                                     moff, inv, ws.reshape(nblk, -1), om.reshape(nblk, -1))
                assert np.array_equal(got[0], wv) and np.array_equal(got[1], wq), (w, h, bd, kind, subpel, fwd, inv)
                if not subpel:
                    assert np.array_equal(got[2], wsad), (w, h, bd, kind, fwd, inv)
        # sdaf through this entry == the dedicated compound-average SAD kernel
        ctx.compound_batch(ps, pr, 1, 1, w, h, d_c, n, 0, _params(hip, K.COMP_AVG, 0), d_p, None, None, None, d_i, None, None, None, d_o)
        sc = np.zeros(n, K.sad_cand_dtype)
        for f in ("sx", "sy", "rx", "ry"):
            sc[f] = cands[f]
        d_sc, d_o2 = ctx.to_device(sc), ctx.malloc(4 * n)
        ctx.sad_avg_batch(ps, pr, 1, 1, w, h, d_sc, n, 0, d_p, d_i, 0, 0, d_o2)
        assert np.array_equal(ctx.from_device(d_o, (n,), np.uint32), ctx.from_device(d_o2, (n,), np.uint32))
        for d in (d_c, d_p, d_i, d_m, d_mo, d_ws, d_om, d_o, d_sc, d_o2):
            ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_compound_rejects_bad_arguments(hip, ctx):
    K = hip.capi
    ps = ctx.planes_alloc(64, 64, 32, 8, 1)
    d = ctx.malloc(64)
    with pytest.raises(K.AomHipError):   # weights must sum to 16
        ctx.compound_batch(ps, ps, 0, 1, 16, 16, d, 1, 0, _params(hip, K.COMP_DIST_WTD, 1, 9, 9), d, None, None, None, None, None, d, d, d)
    with pytest.raises(K.AomHipError):   # masked without a mask
        ctx.compound_batch(ps, ps, 0, 1, 16, 16, d, 1, 0, _params(hip, K.COMP_MASK, 1, 0, 0, 16, 0), d, None, None, None, None, None, d, d, d)
    with pytest.raises(K.AomHipError):   # OBMC without its buffers
        ctx.compound_batch(ps, ps, 0, 1, 16, 16, d, 1, 0, _params(hip, K.COMP_OBMC, 0), None, None, None, None, None, None, d, d, d)
    with pytest.raises(K.AomHipError):   # no output at all
        ctx.compound_batch(ps, ps, 0, 1, 16, 16, d, 1, 0, _params(hip, K.COMP_AVG, 0), d, None, None, None, None, None, None, None, None)
    with pytest.raises(K.AomHipError):   # not a block size
        ctx.compound_batch(ps, ps, 0, 1, 16, 12, d, 1, 0, _params(hip, K.COMP_AVG, 0), d, None, None, None, None, None, d, d, d)
    ctx.compound_batch(ps, ps, 0, 1, 16, 16, None, 0, 0, _params(hip, K.COMP_AVG, 0), d, None, None, None, None, None, d, d, d)  # empty list
    ctx.free(d); ctx.planes_free(ps)


class _Jcp(C.Structure):
    _fields_ = [("use_dist_wtd_comp_avg", C.c_int), ("fwd_offset", C.c_int), ("bck_offset", C.c_int)]


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_vtable_compound_members(hip, oracle, bd):
    """aomhip_bind_variance_vtable fills sdaf, svaf, msdf, msvf, osdf, ovf, osvf, jsdaf, jsvaf with functions of the
    reference's signatures (aom_dsp/variance.h:29-82); call them on host buffers the way mcomp.c / reconinter_enc.c do."""
    K = hip.capi
    lib = K.lib
    names = ["sdf", "sdsf", "sdaf", "vf", "svf", "svaf", "sdx4df", "sdx3df", "sdsx4df", "msdf", "msvf", "osdf", "ovf", "osvf", "jsdaf", "jsvaf"]
    table = (C.c_void_p * (16 * 22))()
    lib.aomhip_bind_variance_vtable.argtypes = [C.c_void_p, C.c_int]
    assert lib.aomhip_bind_variance_vtable(table, bd) == 0
    rng = np.random.default_rng(7 + bd)
    dt = np.uint8 if bd == 8 else np.uint16
    e16 = int(bd > 8)
    mx = (1 << bd) - 1
    S = 96
    a, b = rng.integers(0, mx + 1, (80, S)).astype(dt), rng.integers(0, mx + 1, (80, S)).astype(dt)
    enc = (lambda addr: addr >> 1) if e16 else (lambda addr: addr)      # CONVERT_TO_BYTEPTR (aom_ports/mem.h:80)
    u, vp, i32 = C.c_uint, C.c_void_p, C.c_int
    protos = {
        "sdaf": C.CFUNCTYPE(u, vp, i32, vp, i32, vp), "jsdaf": C.CFUNCTYPE(u, vp, i32, vp, i32, vp, vp),
        "svaf": C.CFUNCTYPE(u, vp, i32, i32, i32, vp, i32, vp, vp), "jsvaf": C.CFUNCTYPE(u, vp, i32, i32, i32, vp, i32, vp, vp, vp),
        "msdf": C.CFUNCTYPE(u, vp, i32, vp, i32, vp, vp, i32, i32), "msvf": C.CFUNCTYPE(u, vp, i32, i32, i32, vp, i32, vp, vp, i32, i32, vp),
        "osdf": C.CFUNCTYPE(u, vp, i32, vp, vp), "ovf": C.CFUNCTYPE(u, vp, i32, vp, vp, vp), "osvf": C.CFUNCTYPE(u, vp, i32, i32, i32, vp, vp, vp),
    }
    orc = oracle.lib
    orc.orc_compound_sub_pixel_variance.restype = C.c_uint32
    orc.orc_masked_sad.restype = orc.orc_obmc_sad.restype = orc.orc_sad_avg_any.restype = C.c_uint
    orc.orc_obmc_variance.restype = C.c_uint32
    for bi, (w, h) in enumerate(BLOCK_SIZES):
        if bi % 3 != bd % 3 and (w, h) != (16, 16):
            continue        # one launch per call: a third of the sizes per bit depth keeps this quick
        fn = {nm: protos[nm](table[bi * 16 + names.index(nm)]) for nm in protos}
        ax, ay, bx, by = (int(v) for v in rng.integers(0, 80 - 65, 4)) if max(w, h) <= 64 else (0, 0, 0, 0)
        if max(w, h) > 64:
            continue        # the 96 x 80 host planes hold blocks up to 64
        A, B = a.ctypes.data + (ay * S + ax) * a.itemsize, b.ctypes.data + (by * S + bx) * b.itemsize
        sp = rng.integers(0, mx + 1, w * h).astype(dt)
        ms = w + 2
        mask = rng.integers(0, 65, (h, ms)).astype(np.uint8)
        om = rng.integers(0, 4097, w * h).astype(np.int32)
        ws = (rng.integers(0, mx + 1, w * h) * 4096 - rng.integers(0, mx + 1, w * h) * (4096 - om)).astype(np.int32)
        SP, M, WS, OM = sp.ctypes.data, mask.ctypes.data, ws.ctypes.data, om.ctypes.data
        jcp = _Jcp(1, 11, 5)
        sse, q = C.c_uint(), C.c_uint32()
        xo, yo = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        # sdaf(src = a, ref = b, second_pred)
        assert fn["sdaf"](enc(A), S, enc(B), S, enc(SP)) == orc.orc_sad_avg_any(vp(A), S, vp(B), S, vp(SP), w, h, e16, bd, 0, 0)
        assert fn["jsdaf"](enc(A), S, enc(B), S, enc(SP), C.addressof(jcp)) == orc.orc_sad_avg_any(vp(A), S, vp(B), S, vp(SP), w, h, e16, bd, 11, 5)
        got = fn["svaf"](enc(A), S, xo, yo, enc(B), S, C.addressof(sse), enc(SP))
        assert (got, sse.value) == (orc.orc_compound_sub_pixel_variance(vp(A), S, xo, yo, vp(B), S, w, h, e16, bd, 0, vp(SP), 0, 0, None, 0, 0, C.byref(q)), q.value)
        got = fn["jsvaf"](enc(A), S, xo, yo, enc(B), S, C.addressof(sse), enc(SP), C.addressof(jcp))
        assert (got, sse.value) == (orc.orc_compound_sub_pixel_variance(vp(A), S, xo, yo, vp(B), S, w, h, e16, bd, 1, vp(SP), 11, 5, None, 0, 0, C.byref(q)), q.value)
        for inv in (0, 1):
            assert fn["msdf"](enc(A), S, enc(B), S, enc(SP), M, ms, inv) == orc.orc_masked_sad(vp(A), S, vp(B), S, vp(SP), vp(M), ms, inv, w, h, e16, bd)
            got = fn["msvf"](enc(A), S, xo, yo, enc(B), S, enc(SP), M, ms, inv, C.addressof(sse))
            assert (got, sse.value) == (orc.orc_compound_sub_pixel_variance(vp(A), S, xo, yo, vp(B), S, w, h, e16, bd, 2, vp(SP), 0, 0, vp(M), ms, inv, C.byref(q)), q.value)
        assert fn["osdf"](enc(A), S, WS, OM) == orc.orc_obmc_sad(vp(A), S, vp(WS), vp(OM), w, h, e16, bd)
        got = fn["ovf"](enc(A), S, WS, OM, C.addressof(sse))
        assert (got, sse.value) == (orc.orc_obmc_variance(vp(A), S, 0, 0, 0, vp(WS), vp(OM), w, h, e16, bd, C.byref(q)), q.value)
        got = fn["osvf"](enc(A), S, xo, yo, WS, OM, C.addressof(sse))
        assert (got, sse.value) == (orc.orc_obmc_variance(vp(A), S, 1, xo, yo, vp(WS), vp(OM), w, h, e16, bd, C.byref(q)), q.value)


@pytest.mark.parametrize("bd", [10, 12])
def test_vtable_highbd_skip_members(hip, oracle, bd):
    """sdsf / sdsx4df of the 10 / 12-bit tables: aom_highbd_sad_skip_{W}x{H}[x4d] with the _bits wrappers
    (av1/encoder/encoder_utils.h:413-470), called through the table on CONVERT_TO_BYTEPTR pointers."""
    lib = hip.capi.lib
    table = (C.c_void_p * (16 * 22))()
    assert lib.aomhip_bind_variance_vtable(table, bd) == 0
    rng = np.random.default_rng(bd)
    S = 160
    a, b = rng.integers(0, 1 << bd, (140, S)).astype(np.uint16), rng.integers(0, 1 << bd, (140, S)).astype(np.uint16)
    SAD = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_int)
    X4D = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p)
    for bi, (w, h) in enumerate(BLOCK_SIZES):
        if h < 8:
            continue                # the reference installs the down-sampled pair for heights >= 8 only (encoder.c:1193-1221)
        A = a.ctypes.data + (3 * S + 5) * 2
        offs = [(int(rng.integers(0, 8)), int(rng.integers(0, 8))) for _ in range(4)]
        Bs = [b.ctypes.data + (y * S + x) * 2 for (x, y) in offs]
        want = [oracle.sad(a, 3, 5, b, y, x, w, h, skip=True, bd=bd) for (x, y) in offs]
        assert SAD(table[bi * 16 + 1])(A >> 1, S, Bs[0] >> 1, S) == want[0], (w, h, bd)
        ptrs = (C.c_void_p * 4)(*[p >> 1 for p in Bs])
        out = (C.c_uint * 4)()
        X4D(table[bi * 16 + 8])(A >> 1, S, ptrs, S, out)
        assert list(out) == want, (w, h, bd)
