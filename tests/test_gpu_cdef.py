"""Parity of the HIP CDEF luma kernel with the oracle (bit-exact): random strengths per 64x64 filter block,
random skip maps, all dampings, 8/10/12-bit, partial filter blocks at the right / bottom frame edge,
direction / variance side outputs; plus the 4K 10-bit plane of BASELINE configs[4] (pri 4 / sec 2, damping 6)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _content(rng, W, H, bd):
    mx = (1 << bd) - 1
    base = rng.integers(mx // 8, mx - mx // 8, (H // 8 + 1, W // 8 + 1))
    i, j = np.indices((H, W))
    pix = np.kron(base, np.ones((8, 8), np.int64))[:H, :W] + ((i * 3 + j * 5) % 17) * (mx // 255 + 1) + rng.integers(-8, 9, (H, W))
    return np.clip(pix, 0, mx).astype(np.uint8 if bd == 8 else np.uint16)


def _run(hip, ctx, pix, pri, sec, skip, damping, bd):
    H, W = pix.shape
    ps, pd = ctx.planes_alloc(W, H, 32, bd, 1), ctx.planes_alloc(W, H, 32, bd, 2)
    ctx.planes_upload(ps, 0, pix)
    d_pri, d_sec, d_skip = ctx.to_device(pri), ctx.to_device(sec), ctx.to_device(skip)
    nb = (H // 8) * (W // 8)
    d_dir, d_var = ctx.malloc(max(nb, 16)), ctx.malloc(nb * 4)
    ctx.cdef_luma_plane(ps, 0, pd, 1, d_pri, d_sec, pri.shape[1], d_skip, damping, d_dir, d_var)
    out = ctx.planes_download(pd, 1)[32:32 + H, 32:32 + W]
    gdir = ctx.from_device(d_dir, (H // 8, W // 8), np.uint8)
    gvar = ctx.from_device(d_var, (H // 8, W // 8), np.int32)
    ctx.planes_free(ps); ctx.planes_free(pd)
    for d in (d_pri, d_sec, d_skip, d_dir, d_var):
        ctx.free(d)
    return out, gdir, gvar


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_random_strength_maps(hip, oracle, ctx, bd):
    rng = np.random.default_rng(bd)
    for trial in range(5):
        W, H = int(rng.choice([64, 136, 200, 320])), int(rng.choice([64, 72, 200]))
        pix = _content(rng, W, H, bd)
        fbh, fbw = (H + 63) // 64, (W + 63) // 64
        pri = rng.integers(0, 16, (fbh, fbw)).astype(np.uint8)
        sec = rng.choice([0, 1, 2, 4], (fbh, fbw)).astype(np.uint8)
        if trial == 0:
            pri[0, 0], sec[0, 0] = 0, 0
        skip = (rng.random((H // 8, W // 8)) < 0.25).astype(np.uint8)
        damping = int(rng.integers(3, 7))
        got, gdir, gvar = _run(hip, ctx, pix, pri, sec, skip, damping, bd)
        want, wdir, wvar = oracle.cdef_plane_luma(pix, pri, sec, skip, damping, bd)
        assert np.array_equal(got, want), (bd, trial)
        assert np.array_equal(gdir, wdir) and np.array_equal(gvar, wvar)
        assert not np.array_equal(got, pix)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_extreme_direction_costs(hip, oracle, ctx, bd):
    """Blocks whose line sums reach +-8 * 128 (all-zero / all-maximum pixels), and the sharpest lines of each direction between them: the
    direction cost's products sit at the top of the 24-bit multiplier's range (cdef_find_dir, cdef_block.c:103-197)."""
    rng = np.random.default_rng(40 + bd)
    mx = (1 << bd) - 1
    W, H = 192, 128
    pix = np.zeros((H, W), np.int64)
    pix[:, 64:128] = mx
    i, j = np.indices((H, 64))
    stripes = [i, j, i + j, i - j, i + j // 2, i - j // 2, i // 2 + j, j - i // 2]       # lines of the eight directions, 0 / max alternating
    for k, line in enumerate(stripes):
        blk = np.where(line[k * 16:k * 16 + 16] % 2 == 0, 0, mx)
        pix[k * 16:k * 16 + 16, 128:192] = blk
    pix = pix.astype(np.uint8 if bd == 8 else np.uint16)
    pri = np.full((2, 3), 7, np.uint8); sec = np.full((2, 3), 2, np.uint8)
    skip = np.zeros((H // 8, W // 8), np.uint8)
    got, gdir, gvar = _run(hip, ctx, pix, pri, sec, skip, 6, bd)
    want, wdir, wvar = oracle.cdef_plane_luma(pix, pri, sec, skip, 6, bd)
    assert np.array_equal(gdir, wdir) and np.array_equal(gvar, wvar) and np.array_equal(got, want)
    assert wvar.max() > (1 << 17) and len(np.unique(wdir[:, 16:])) >= 4     # huge variances, several directions present


def test_each_enable_combination(hip, oracle, ctx):
    """strength_index 0..3 of av1_cdef_filter_fb: {pri, sec} x {on, off}, odd / even primary strengths
    (the two cdef_pri_taps rows) and every damping."""
    rng = np.random.default_rng(3)
    pix = _content(rng, 128, 64, 8)
    noskip = np.zeros((8, 16), np.uint8)
    for pri_v in (0, 1, 2, 7, 15):
        for sec_v in (0, 1, 2, 4):
            for damping in (3, 6):
                pri = np.full((1, 2), pri_v, np.uint8); sec = np.full((1, 2), sec_v, np.uint8)
                got, _, _ = _run(hip, ctx, pix, pri, sec, noskip, damping, 8)
                want, _, _ = oracle.cdef_plane_luma(pix, pri, sec, noskip, damping, 8)
                assert np.array_equal(got, want), (pri_v, sec_v, damping)


def test_full_size_4k_10bit(hip, oracle, ctx):
    W, H, bd = 3840, 2160, 10
    src, _ = hip.synth.shifted_smooth_pair(W, H, 2, bd)
    rng = np.random.default_rng(6)
    pix = np.clip(src.astype(np.int32) + rng.integers(-20, 21, (H, W)), 0, 1023).astype(np.uint16)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    pri = np.full((fbh, fbw), 4, np.uint8); sec = np.full((fbh, fbw), 2, np.uint8)
    skip = np.zeros((H // 8, W // 8), np.uint8)
    got, gdir, gvar = _run(hip, ctx, pix, pri, sec, skip, 6, bd)
    want, wdir, wvar = oracle.cdef_plane_luma(pix, pri, sec, skip, 6, bd)
    assert np.array_equal(got, want) and np.array_equal(gdir, wdir) and np.array_equal(gvar, wvar)
    assert got.max() <= pix.max() and got.min() >= pix.min()


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("xdec,ydec", [(1, 1), (0, 0), (1, 0), (0, 1)])
def test_chroma_planes(hip, oracle, ctx, bd, xdec, ydec):
    """pli > 0: directions from the luma launch (also in filter blocks whose LUMA strengths are zero), uv strengths,
    no variance adjustment, damping - 1; 4:2:0 / 4:4:4 / 4:2:2 / 4:4:0."""
    rng = np.random.default_rng(bd * 4 + xdec * 2 + ydec)
    for trial in range(3):
        W, H = int(rng.choice([64, 136, 320])), int(rng.choice([64, 72, 200]))
        cw, ch = W >> xdec, H >> ydec
        luma, chroma = _content(rng, W, H, bd), _content(rng, cw, ch, bd)
        fbh, fbw = (H + 63) // 64, (W + 63) // 64
        pri_y = rng.integers(0, 16, (fbh, fbw)).astype(np.uint8); sec_y = rng.choice([0, 1, 2, 4], (fbh, fbw)).astype(np.uint8)
        pri_uv = rng.integers(0, 16, (fbh, fbw)).astype(np.uint8); sec_uv = rng.choice([0, 1, 2, 4], (fbh, fbw)).astype(np.uint8)
        pri_y[0, 0] = sec_y[0, 0] = 0  # luma off, chroma on: directions must still be there
        pri_uv[0, 0], sec_uv[0, 0] = 5, 2
        if fbw > 1:
            pri_uv[0, 1] = sec_uv[0, 1] = 0  # chroma off
        skip = (rng.random((H // 8, W // 8)) < 0.2).astype(np.uint8)
        damping = int(rng.integers(3, 7))
        _, gdir, _ = _run(hip, ctx, luma, pri_y, sec_y, skip, damping, bd)
        _, wdir, _ = oracle.cdef_plane_luma(luma, pri_y, sec_y, skip, damping, bd)
        assert np.array_equal(gdir, wdir)
        assert wdir[:8, :8][skip[:8, :8] == 0].any()  # directions exist in the zero-luma-strength filter block
        ps, pd = ctx.planes_alloc(cw, ch, 32, bd, 1), ctx.planes_alloc(cw, ch, 32, bd, 2)
        ctx.planes_upload(ps, 0, chroma)
        d_dir, d_pri, d_sec, d_skip = ctx.to_device(gdir), ctx.to_device(pri_uv), ctx.to_device(sec_uv), ctx.to_device(skip)
        ctx.cdef_chroma_plane(ps, 0, pd, 1, xdec, ydec, d_dir, d_pri, d_sec, fbw, d_skip, damping)
        got = ctx.planes_download(pd, 1)[32:32 + ch, 32:32 + cw]
        want = oracle.cdef_plane_chroma(chroma, xdec, ydec, wdir, pri_uv, sec_uv, skip, damping, bd)
        assert np.array_equal(got, want), (bd, xdec, ydec, trial)
        assert not np.array_equal(got, chroma)
        ctx.planes_free(ps); ctx.planes_free(pd)
        for d in (d_dir, d_pri, d_sec, d_skip):
            ctx.free(d)


def test_chroma_full_size_4k_10bit_420(hip, oracle, ctx):
    W, H, bd = 3840, 2160, 10
    rng = np.random.default_rng(8)
    src, _ = hip.synth.shifted_smooth_pair(W, H, 2, bd)
    luma = np.clip(src.astype(np.int32) + rng.integers(-20, 21, (H, W)), 0, 1023).astype(np.uint16)
    chroma = np.clip(src[::2, ::2].astype(np.int32) + rng.integers(-20, 21, (H // 2, W // 2)), 0, 1023).astype(np.uint16)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    pri = np.full((fbh, fbw), 4, np.uint8); sec = np.full((fbh, fbw), 2, np.uint8)
    skip = np.zeros((H // 8, W // 8), np.uint8)
    _, gdir, _ = _run(hip, ctx, luma, pri, sec, skip, 6, bd)
    ps, pd = ctx.planes_alloc(W // 2, H // 2, 32, bd, 1), ctx.planes_alloc(W // 2, H // 2, 32, bd, 1)
    ctx.planes_upload(ps, 0, chroma)
    d_dir, d_pri, d_sec, d_skip = ctx.to_device(gdir), ctx.to_device(pri), ctx.to_device(sec), ctx.to_device(skip)
    ctx.cdef_chroma_plane(ps, 0, pd, 0, 1, 1, d_dir, d_pri, d_sec, fbw, d_skip, 6)
    got = ctx.planes_download(pd, 0)[32:32 + H // 2, 32:32 + W // 2]
    assert np.array_equal(got, oracle.cdef_plane_chroma(chroma, 1, 1, gdir, pri, sec, skip, 6, bd))
    ctx.planes_free(ps); ctx.planes_free(pd)
    for d in (d_dir, d_pri, d_sec, d_skip):
        ctx.free(d)
