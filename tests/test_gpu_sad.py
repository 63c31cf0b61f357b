"""Parity of the HIP SAD kernels with the oracle, through the C ABI (ctypes), bit-exact.
Mirrors test/sad_test.cc of the reference: MaxRef / MaxSrc / ShortRef / UnalignedRef /
ShortSrc / SrcAlignedByWidth for all 22 block sizes, the x4d forms, skip forms, 10/12-bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import BLOCK_SIZES

pytestmark = pytest.mark.gpu


def _ptr(a, y=0, x=0):
    return a.ctypes.data + (int(y) * a.shape[1] + int(x)) * a.itemsize


@pytest.mark.parametrize("w,h", BLOCK_SIZES)
def test_rtcd_signature_sad_cases(hip, oracle, ctx, w, h):
    lib = hip.capi.lib
    rng = np.random.default_rng(w * 100 + h)
    ss, rs = (w + 31) & ~31, 2 * w  # sad_test.cc:168-169
    cases = []
    z = np.zeros((h, ss), np.uint8)
    m = np.full((h + 1, rs + 1), 255, np.uint8)
    cases.append((z, 0, ss, m, 0, rs))                      # MaxRef
    cases.append((np.full((h, ss), 255, np.uint8), 0, ss, np.zeros((h + 1, rs + 1), np.uint8), 0, rs))  # MaxSrc
    s = rng.integers(0, 256, (h, ss), dtype=np.uint8)
    r = rng.integers(0, 256, (h + 1, rs + 1), dtype=np.uint8)
    cases.append((s, 0, ss, r, 0, rs))                      # random
    cases.append((s, 0, ss, r, 0, w))                       # ShortRef: ref stride = w (rows overlap the array rows)
    cases.append((s, 0, ss, r, 0, rs - 1))                  # UnalignedRef: stride - 1
    cases.append((s, 0, w, r, 1, rs))                       # ShortSrc-like + odd ref offset
    for (sa, so, sst, ra, ro, rst) in cases:
        sp, rp = sa.ctypes.data + so, ra.ctypes.data + ro
        got = lib.aomhip_sad(sp, sst, rp, rst, w, h)
        want = oracle.lib.orc_sad(sp, sst, rp, rst, w, h)
        assert got == want
        assert lib.aomhip_sad_skip(sp, sst, rp, rst, w, h) == oracle.lib.orc_sad_skip(sp, sst, rp, rst, w, h)


@pytest.mark.parametrize("w,h", BLOCK_SIZES)
def test_rtcd_signature_x4d(hip, oracle, ctx, w, h):
    lib = hip.capi.lib
    rng = np.random.default_rng(w * 7 + h * 3)
    s = rng.integers(0, 256, (h, w), dtype=np.uint8)
    r = rng.integers(0, 256, (h + 8, 2 * w + 8), dtype=np.uint8)
    offs = [(0, 0), (1, 3), (5, 1), (7, 7)]
    ptrs = (C.c_void_p * 4)(*[_ptr(r, y, x) for (y, x) in offs])
    got = np.zeros(4, np.uint32)
    lib.aomhip_sad_x4d(s.ctypes.data, w, ptrs, r.shape[1], got.ctypes.data, w, h)
    want = [oracle.sad(s, 0, 0, r, y, x, w, h) for (y, x) in offs]
    assert got.tolist() == want
    lib.aomhip_sad_skip_x4d(s.ctypes.data, w, ptrs, r.shape[1], got.ctypes.data, w, h)
    assert got.tolist() == [oracle.sad(s, 0, 0, r, y, x, w, h, skip=True) for (y, x) in offs]


def test_fixed_size_symbols_16x16(hip, oracle, ctx):
    lib = hip.capi.lib
    rng = np.random.default_rng(5)
    s = rng.integers(0, 256, (16, 32), dtype=np.uint8)
    r = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    assert lib.aomhip_sad16x16(s.ctypes.data, 32, _ptr(r, 3, 5), 40) == oracle.sad(s, 0, 0, r, 3, 5, 16, 16)
    ptrs = (C.c_void_p * 4)(*[_ptr(r, y, x) for (y, x) in [(0, 0), (1, 1), (2, 7), (8, 24)]])
    got = np.zeros(4, np.uint32)
    lib.aomhip_sad16x16x4d(s.ctypes.data, 32, ptrs, 40, got.ctypes.data)
    assert got.tolist() == [oracle.sad(s, 0, 0, r, y, x, 16, 16) for (y, x) in [(0, 0), (1, 1), (2, 7), (8, 24)]]


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_rtcd_signature_highbd(hip, oracle, ctx, bd):
    lib = hip.capi.lib
    rng = np.random.default_rng(bd)
    mx = (1 << bd) - 1
    for (w, h) in [(4, 4), (8, 16), (16, 16), (64, 64), (128, 128), (16, 64)]:
        s = rng.integers(0, mx + 1, (h, w + 4), dtype=np.uint16)
        r = rng.integers(0, mx + 1, (h + 2, w + 6), dtype=np.uint16)
        for (sa, ra) in [(s, r), (np.zeros_like(s), np.full_like(r, mx))]:
            sp, rp = _ptr(sa, 0, 1), _ptr(ra, 1, 3)
            # CONVERT_TO_BYTEPTR (aom_ports/mem.h:80): the reference passes uint16 pointers >> 1
            got = lib.aomhip_highbd_sad(sp >> 1, sa.shape[1], rp >> 1, ra.shape[1], w, h, bd)
            assert got == oracle.lib.orc_highbd_sad(sp, sa.shape[1], rp, ra.shape[1], w, h, bd)


def _upload_pair(hip, ctx, src, ref, border, bd, n_frames=1):
    h, w = src[0].shape
    ps = ctx.planes_alloc(w, h, border, bd, n_frames)
    pr = ctx.planes_alloc(w, h, border, bd, n_frames)
    for f in range(n_frames):
        ctx.planes_upload(ps, f, src[f])
        ctx.planes_upload(pr, f, ref[f])
    return ps, pr


def test_plane_upload_replicates_borders(hip, oracle, ctx):
    pix = hip.synth.lcg_frame(100, 37, 3)
    p = ctx.planes_alloc(100, 37, 32, 8, 2)
    ctx.planes_upload(p, 1, pix)
    got = ctx.planes_download(p, 1)
    assert p.stride == hip.capi.lib.aomhip_calc_stride(100, 32)
    want = oracle.extend_plane(pix, 32, p.stride)
    # columns past aligned_width + 2*border replicate the last pixel in both models
    assert np.array_equal(got, want)
    ctx.planes_free(p)


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("w,h", [(16, 16), (4, 4), (8, 8), (32, 32), (64, 64), (128, 128), (4, 16), (16, 4), (64, 16),
                                 (8, 32), (32, 64)])
def test_batched_sad_matches_oracle(hip, oracle, ctx, w, h, bd):
    """Seeded frames, candidates anywhere the MV limits allow (incl. the replicated border)."""
    W, H, border, F = 320, 192, 160, 3
    src = [hip.synth.lcg_frame(W, H, 2 * f, 0, bd) for f in range(F)]
    ref = [hip.synth.lcg_frame(W, H, 2 * f + 1, 0, bd) for f in range(F)]
    ps, pr = _upload_pair(hip, ctx, src, ref, border, bd, F)
    rng = np.random.default_rng(w * h + bd)
    n = 1501  # ragged: not a multiple of the candidates-per-workgroup
    cands = np.zeros((F, n), hip.capi.sad_cand_dtype)
    lim = border - 4  # av1_set_mv_limits keeps blocks inside border - interp extend
    cands["sx"] = rng.integers(0, W - w + 1, (F, n))
    cands["sy"] = rng.integers(0, H - h + 1, (F, n))
    cands["rx"] = rng.integers(-lim, W + lim - w + 1, (F, n))
    cands["ry"] = rng.integers(-lim, H + lim - h + 1, (F, n))
    d_c = ctx.to_device(cands)
    d_o = ctx.malloc(F * n * 4)
    for flags in (0, hip.capi.SAD_SKIP_ROWS):
        ctx.sad_batch(ps, pr, 0, F, w, h, flags, d_c, n, n, d_o)
        got = ctx.from_device(d_o, (F, n), np.uint32)
        for f in range(F):
            sb = oracle.extend_plane(src[f], border, ps.stride)
            rb = oracle.extend_plane(ref[f], border, pr.stride)
            want = oracle.sad_batch(sb, rb, border, w, h, cands[f], skip=bool(flags), bd=bd, threads=4)
            assert np.array_equal(got[f], want), (w, h, bd, flags, f)
    # shared list (cand_frame_stride = 0), sub-range of frames
    ctx.sad_batch(ps, pr, 1, 2, w, h, 0, d_c, n, 0, d_o)
    got = ctx.from_device(d_o, (2, n), np.uint32)
    for k, f in enumerate((1, 2)):
        sb = oracle.extend_plane(src[f], border, ps.stride)
        rb = oracle.extend_plane(ref[f], border, pr.stride)
        assert np.array_equal(got[k], oracle.sad_batch(sb, rb, border, w, h, cands[0], bd=bd, threads=4))
    for p in (ps, pr):
        ctx.planes_free(p)
    ctx.free(d_c); ctx.free(d_o)


@pytest.mark.parametrize("bd", [8, 10])
@pytest.mark.parametrize("w,h", [(16, 16), (8, 8), (32, 32), (64, 64), (128, 64), (4, 8)])
def test_batched_x4d_matches_oracle(hip, oracle, ctx, w, h, bd):
    W, H, border, F = 256, 128, 160, 2
    src = [hip.synth.lcg_frame(W, H, 10 + f, 0, bd) for f in range(F)]
    ref = [hip.synth.lcg_frame(W, H, 20 + f, 0, bd) for f in range(F)]
    ps, pr = _upload_pair(hip, ctx, src, ref, border, bd, F)
    rng = np.random.default_rng(w + h + bd)
    n = 777
    g = np.zeros((F, n), hip.capi.sad_x4d_dtype)
    lim = border - 4
    g["sx"] = rng.integers(0, W - w + 1, (F, n))
    g["sy"] = rng.integers(0, H - h + 1, (F, n))
    g["rx"] = rng.integers(-lim, W + lim - w + 1, (F, n, 4))
    g["ry"] = rng.integers(-lim, H + lim - h + 1, (F, n, 4))
    d_g = ctx.to_device(g)
    d_o = ctx.malloc(F * n * 16)
    for flags in (0, hip.capi.SAD_SKIP_ROWS):
        ctx.sad_x4d_batch(ps, pr, 0, F, w, h, flags, d_g, n, n, d_o)
        got = ctx.from_device(d_o, (F, n, 4), np.uint32)
        for f in range(F):
            sb = oracle.extend_plane(src[f], border, ps.stride)
            rb = oracle.extend_plane(ref[f], border, pr.stride)
            want = oracle.sad_x4d_batch(sb, rb, border, w, h, g[f], skip=bool(flags), bd=bd, threads=4)
            assert np.array_equal(got[f], want)
    for p in (ps, pr):
        ctx.planes_free(p)
    ctx.free(d_g); ctx.free(d_o)


def test_empty_and_invalid_batches(hip, ctx):
    ps = ctx.planes_alloc(64, 64, 32, 8, 1)
    d_o = ctx.malloc(64)
    ctx.sad_batch(ps, ps, 0, 1, 16, 16, 0, None, 0, 0, d_o)  # empty list: no-op, no error
    with pytest.raises(hip.capi.AomHipError):
        ctx.sad_batch(ps, ps, 0, 1, 16, 12, 0, d_o, 1, 0, d_o)  # 16x12 is not an AV1 block size
    with pytest.raises(hip.capi.AomHipError):
        ctx.sad_batch(ps, ps, 0, 2, 16, 16, 0, d_o, 1, 0, d_o)  # frame range out of bounds
    ctx.planes_free(ps); ctx.free(d_o)


def test_full_size_1080p_mode_a_properties(hip, oracle, ctx):
    """BASELINE configs[1] at full size: exact check against the oracle on one frame pair plus
    size-independent properties on the rest (SAD(x,x) == 0; x4d == 4 single SADs)."""
    W, H, border = 1920, 1080, 160
    src = hip.synth.lcg_frame(W, H, 0)
    ref = hip.synth.lcg_frame(W, H, 1)
    ps, pr = _upload_pair(hip, ctx, [src], [ref], border, 8, 1)
    cands, groups = hip.synth.mode_a_worklist(W, H, 16, seed=3)
    assert len(cands) == 8040
    d_c, d_g = ctx.to_device(cands), ctx.to_device(groups)
    d_o1, d_o4 = ctx.malloc(len(cands) * 4), ctx.malloc(len(groups) * 16)
    ctx.sad_batch(ps, pr, 0, 1, 16, 16, 0, d_c, len(cands), 0, d_o1)
    ctx.sad_x4d_batch(ps, pr, 0, 1, 16, 16, 0, d_g, len(groups), 0, d_o4)
    got1 = ctx.from_device(d_o1, (len(cands),), np.uint32)
    got4 = ctx.from_device(d_o4, (len(groups), 4), np.uint32)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    assert np.array_equal(got1, oracle.sad_batch(sb, rb, border, 16, 16, cands, threads=8))
    assert np.array_equal(got4, oracle.sad_x4d_batch(sb, rb, border, 16, 16, groups, threads=8))
    # property: x4d group == the same four positions as single candidates
    flat = np.zeros(len(groups) * 4, hip.capi.sad_cand_dtype)
    flat["sx"], flat["sy"] = np.repeat(groups["sx"], 4), np.repeat(groups["sy"], 4)
    flat["rx"], flat["ry"] = groups["rx"].ravel(), groups["ry"].ravel()
    d_f, d_of = ctx.to_device(flat), ctx.malloc(len(flat) * 4)
    ctx.sad_batch(ps, pr, 0, 1, 16, 16, 0, d_f, len(flat), 0, d_of)
    assert np.array_equal(ctx.from_device(d_of, (len(groups), 4), np.uint32), got4)
    # property: a plane against itself at mv (0,0) is all zeros
    ctx.sad_batch(ps, ps, 0, 1, 16, 16, 0, d_c, len(cands), 0, d_o1)
    assert not ctx.from_device(d_o1, (len(cands),), np.uint32).any()
    for p in (ps, pr):
        ctx.planes_free(p)
    for d in (d_c, d_g, d_o1, d_o4, d_f, d_of):
        ctx.free(d)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_compound_average_sad(hip, oracle, ctx, bd):
    """sdaf / jsdaf: aom_sadWxH_avg and aom_dist_wtd_sadWxH_avg (+ highbd, + bits10/12 wrappers), all 22 sizes, the
    four quant_dist_lookup_table weight pairs (av1/common/reconinter.c) and the plain average."""
    rng = np.random.default_rng(bd)
    W, H, border = 256, 192, 64
    src, ref = hip.synth.lcg_frame(W, H, 1, 0, bd), hip.synth.lcg_frame(W, H, 2, 1, bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 2)
    ctx.planes_upload(ps, 1, src); ctx.planes_upload(pr, 1, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    dt = np.uint8 if bd == 8 else np.uint16
    for (w, h) in BLOCK_SIZES:
        n, npred = 150, 7
        cands = np.zeros(n, hip.capi.sad_cand_dtype)
        cands["sx"], cands["sy"] = rng.integers(0, W - w + 1, n), rng.integers(0, H - h + 1, n)
        cands["rx"], cands["ry"] = rng.integers(-border, W + border - w + 1, n), rng.integers(-border, H + border - h + 1, n)
        preds = rng.integers(0, 1 << bd, (npred, h, w)).astype(dt)
        preds[0] = (1 << bd) - 1  # saturated block: the weighted blend must not wrap
        pidx = rng.integers(0, npred, n).astype(np.uint32)
        d_c, d_p, d_i, d_o = ctx.to_device(cands), ctx.to_device(preds), ctx.to_device(pidx), ctx.malloc(n * 4)
        for fwd, bck in ((0, 0), (9, 7), (11, 5), (12, 4), (13, 3), (4, 12)):
            ctx.sad_avg_batch(ps, pr, 1, 1, w, h, d_c, n, 0, d_p, d_i, fwd, bck, d_o)
            want = oracle.sad_avg_batch(sb, rb, border, w, h, cands, preds, pidx, fwd, bck, bd)
            assert np.array_equal(ctx.from_device(d_o, (n,), np.uint32), want), (w, h, bd, fwd, bck)
        ctx.sad_avg_batch(ps, pr, 1, 1, w, h, d_c, n, 0, d_p, None, 0, 0, d_o)  # NULL index: block 0 for everyone
        assert np.array_equal(ctx.from_device(d_o, (n,), np.uint32),
                              oracle.sad_avg_batch(sb, rb, border, w, h, cands, preds, np.zeros(n, np.uint32), 0, 0, bd))
        for d in (d_c, d_p, d_i, d_o):
            ctx.free(d)
    with pytest.raises(hip.capi.AomHipError):
        ctx.sad_avg_batch(ps, pr, 1, 1, 16, 16, 0, 0, 0, None, None, 0, 0, 1)
    ctx.planes_free(ps); ctx.planes_free(pr)
