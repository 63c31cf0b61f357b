"""Parity of the HIP deblocking passes with the oracle (bit-exact): random transform partitions, every
filter length, luma and chroma length rules, 8/10/12-bit, all sharpness values, each pass alone and both;
plus the full-size 4K 10-bit plane of BASELINE configs[4] with the synthetic 'all 8x8 edges, level 32' map."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _content(rng, W, H, bd):
    mx = (1 << bd) - 1
    base = rng.integers(0, mx // 4, (H // 8 + 1, W // 8 + 1))
    pix = np.kron(base, np.ones((8, 8), np.int64))[:H, :W] + rng.integers(-2, 3, (H, W)) + mx // 3
    return np.clip(pix, 0, mx).astype(np.uint8 if bd == 8 else np.uint16)


def _run(hip, ctx, pix, params, bd, sharp, passes):
    H, W = pix.shape
    p = ctx.planes_alloc(W, H, 32, bd, 2)
    ctx.planes_upload(p, 1, pix)
    d = ctx.to_device(params)
    ctx.deblock_plane(p, 1, d, params.shape[1], sharp, passes)
    out = ctx.planes_download(p, 1)[32:32 + H, 32:32 + W]
    ctx.planes_free(p); ctx.free(d)
    return out


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("chroma", [False, True])
def test_random_partitions(hip, oracle, ctx, bd, chroma):
    rng = np.random.default_rng(bd * 2 + chroma)
    any_changed = False
    for trial in range(4):
        W, H = int(rng.choice([64, 136, 200, 384])), int(rng.choice([64, 72, 192]))
        pix = _content(rng, W, H, bd)
        params = oracle.random_edge_params(rng, W, H, chroma=chroma)
        sharp = int(rng.integers(0, 8))
        for passes in (1, 2, 3):
            want_params = params.copy()
            if passes == 1: want_params[..., 2:] = 0
            if passes == 2: want_params[..., :2] = 0
            want = oracle.deblock_plane(pix, want_params, sharp, bd, order=0)
            got = _run(hip, ctx, pix, params, bd, sharp, passes)
            assert np.array_equal(got, want), (bd, chroma, trial, passes)
        any_changed |= not np.array_equal(want, pix)
    assert any_changed


def test_every_length_and_level(hip, oracle, ctx):
    rng = np.random.default_rng(9)
    W, H = 256, 64
    for bd in (8, 10):
        pix = _content(rng, W, H, bd)
        for length, spacing in ((4, 1), (6, 2), (8, 2), (14, 4)):
            for level in (1, 8, 32, 63):
                params = np.zeros((H // 4, W // 4, 4), np.uint8)
                params[:, spacing::spacing, 0] = length; params[:, spacing::spacing, 1] = level
                params[spacing::spacing, :, 2] = length; params[spacing::spacing, :, 3] = level
                want = oracle.deblock_plane(pix, params, 0, bd, order=0)
                assert np.array_equal(_run(hip, ctx, pix, params, bd, 0, 3), want), (bd, length, level)
    # level 0 or length 0: untouched
    params = np.zeros((H // 4, W // 4, 4), np.uint8); params[..., 0] = 8; params[..., 2] = 8
    assert np.array_equal(_run(hip, ctx, pix, params, 10, 0, 3), pix)


def test_full_size_4k_10bit_all_8x8_edges(hip, oracle, ctx):
    """configs[4] synthetic edge map: every 8x8 edge, level 32; exact vs the oracle and
    idempotence-style sanity: frame borders and the first row/column of edges are never touched."""
    W, H, bd = 3840, 2160, 10
    src, _ = hip.synth.shifted_smooth_pair(W, H, 1, bd)
    rng = np.random.default_rng(4)
    pix = np.clip(src.astype(np.int32) + (rng.integers(0, 2, (H // 8, W // 8)) * 6).repeat(8, 0).repeat(8, 1), 0, 1023).astype(np.uint16)
    params = np.zeros((H // 4, W // 4, 4), np.uint8)
    params[:, 2::2, 0] = 8; params[:, 2::2, 1] = 32
    params[2::2, :, 2] = 8; params[2::2, :, 3] = 32
    got = _run(hip, ctx, pix, params, bd, 0, 3)
    want = oracle.deblock_plane(pix, params, 0, bd, order=0)
    assert np.array_equal(got, want)
    changed = got != pix
    assert changed.any()
    # filter8 writes p2..q2 only: pixels in columns 3/4 AND rows 3/4 of every 8x8 cell are out of reach of both passes
    cx, cy = np.arange(W) % 8, np.arange(H) % 8
    assert not changed[np.ix_((cy == 3) | (cy == 4), (cx == 3) | (cx == 4))].any()
    assert not changed[:5, :5].any()  # the frame edges themselves are never filtered


@pytest.mark.parametrize("bd", [8, 10])
def test_mode_info_grid_to_filtered_planes(hip, oracle, ctx, bd):
    """The whole chain of row E3: a random mode-info grid -> the host producers (aomhip_lf_build_edge_params,
    aomhip_cdef_build_skip8x8 / _strengths) -> the deblocking and CDEF kernels, luma and 4:2:0 chroma deblocking, against the
    oracle fed the planes its own mode-info walk derives (which tests/test_filter_maps.py pins to the interpreted reference)."""
    import ctypes as C
    from test_filter_maps import _random_grid, _product_edges
    lib = hip.capi.lib
    rng = np.random.default_rng(60 + bd)
    grid = _random_grid(oracle, rng, 32, 48, consistent_tx=True)   # 192 x 128 luma
    f = oracle.LfFrame()
    f.filter_level[0], f.filter_level[1], f.filter_level_u, f.filter_level_v = 30, 26, 22, 18
    f.mode_ref_delta_enabled = 1
    for i, v in enumerate([1, 0, 0, 0, -1, 0, -1, -1]):
        f.ref_deltas[i] = v
    lvl = oracle.lf_frame_init(f)
    for plane, (ssx, ssy) in ((0, (0, 0)), (1, (1, 1))):
        W, H = (grid.mi_cols * 4) >> ssx, (grid.mi_rows * 4) >> ssy
        pix = _content(rng, W, H, bd)
        units = oracle.lf_units(grid, f, lvl, plane, ssx, ssy)
        params = np.ascontiguousarray(_product_edges(lib, units, W, H, int(plane > 0)))
        want_params = np.ascontiguousarray(oracle.lf_edge_plane(grid, f, lvl, plane, ssx, ssy)[..., :4].astype(np.uint8))
        got = _run(hip, ctx, pix, params, bd, 0, 3)
        want = oracle.deblock_plane(pix, want_params, 0, bd)
        assert np.array_equal(got, want), (plane, bd)
        assert not np.array_equal(got, pix)
    # CDEF luma with producer-made skip / strength planes
    W, H = grid.mi_cols * 4, grid.mi_rows * 4
    pix = _content(rng, W, H, bd)
    mi_skip = np.ascontiguousarray(grid.blocks["skip_txfm"][grid.owner]).astype(np.uint8)
    skip = np.zeros((H // 8, W // 8), np.uint8)
    fsk = lib.aomhip_cdef_build_skip8x8
    fsk.restype, fsk.argtypes = C.c_int, None
    assert fsk(C.c_void_p(mi_skip.ctypes.data), C.c_int(grid.mi_cols), C.c_int(grid.mi_rows), C.c_int(grid.mi_cols), C.c_void_p(skip.ctypes.data),
               C.c_int(W // 8)) == 0
    assert np.array_equal(skip, oracle.cdef_skip_map(grid))
    fbw, fbh = (W + 63) // 64, (H + 63) // 64
    idx = rng.integers(-1, 4, fbw * fbh).astype(np.int8)
    strengths = np.array([0, 9, 22, 63], np.int32)
    pri, sec = np.zeros(fbw * fbh, np.uint8), np.zeros(fbw * fbh, np.uint8)
    fst = lib.aomhip_cdef_build_strengths
    fst.restype, fst.argtypes = C.c_int, None
    assert fst(C.c_void_p(idx.ctypes.data), C.c_int(fbw * fbh), C.c_void_p(strengths.ctypes.data), None, C.c_void_p(pri.ctypes.data),
               C.c_void_p(sec.ctypes.data), None, None) == 0
    ps, pd = ctx.planes_alloc(W, H, 32, bd, 1), ctx.planes_alloc(W, H, 32, bd, 1)
    ctx.planes_upload(ps, 0, pix)
    d_pri, d_sec, d_skip = ctx.to_device(pri), ctx.to_device(sec), ctx.to_device(skip)
    ctx.cdef_luma_plane(ps, 0, pd, 0, d_pri, d_sec, fbw, d_skip, 5)
    got = ctx.planes_download(pd, 0)[32:32 + H, 32:32 + W]
    want = oracle.cdef_plane_luma(pix, pri.reshape(fbh, fbw), sec.reshape(fbh, fbw), skip, 5, bd)
    want = want[0] if isinstance(want, tuple) else want
    assert np.array_equal(got, want)
    for d in (d_pri, d_sec, d_skip):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pd)


def _run_fused(hip, ctx, pix, params, bd, sharp, border=32):
    H, W = pix.shape
    p, q = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(p, 1, pix)
    d = ctx.to_device(params)
    ctx.deblock_plane_fused(p, 1, q, 0, d, params.shape[1], sharp)
    out = ctx.planes_download(q, 0)[border:border + H, border:border + W]
    src_after = ctx.planes_download(p, 1)[border:border + H, border:border + W]
    ctx.planes_free(p); ctx.planes_free(q); ctx.free(d)
    assert np.array_equal(src_after, pix)  # out of place: the source is untouched
    return out


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("chroma", [False, True])
def test_fused_single_launch_equals_the_two_passes(hip, oracle, ctx, bd, chroma):
    """aomhip_deblock_plane_fused (one launch, LDS tiles with halos, out of place) == the oracle's vertical-then-horizontal result on random
    transform partitions: sizes that are not multiples of the 128 x 64 tile, every length, all sharpness values."""
    rng = np.random.default_rng(100 + bd * 2 + chroma)
    changed = False
    for trial in range(5):
        W, H = int(rng.choice([64, 136, 200, 384, 520])), int(rng.choice([64, 72, 136, 192]))
        pix = _content(rng, W, H, bd)
        params = oracle.random_edge_params(rng, W, H, chroma=chroma)
        sharp = int(rng.integers(0, 8))
        want = oracle.deblock_plane(pix, params, sharp, bd, order=0)
        got = _run_fused(hip, ctx, pix, params, bd, sharp, border=int(rng.choice([8, 32, 160])))
        assert np.array_equal(got, want), (bd, chroma, trial, W, H)
        changed |= not np.array_equal(want, pix)
    assert changed


def test_fused_full_size_4k_10bit(hip, oracle, ctx):
    rng = np.random.default_rng(4)
    W, H, bd = 3840, 2160, 10
    pix = _content(rng, W, H, bd)
    params = np.zeros((H // 4, W // 4, 4), np.uint8)
    # left half: 8x8 transforms (8-tap edges every 8 pixels); right half: 16x16 transforms (14-tap edges every 16) -- edge zones never overlap
    hw = (W // 8) // 4 * 4   # unit column where the 16x16 region starts (a multiple of 16 pixels)
    params[:, 2:hw:2, 0] = 8; params[:, 2:hw:2, 1] = 32; params[2::2, :hw, 2] = 8; params[2::2, :hw, 3] = 32
    params[:, hw::4, 0] = 14; params[:, hw::4, 1] = 40; params[4::4, hw:, 2] = 14; params[4::4, hw:, 3] = 40
    params[:, hw, 0] = 8   # (the seam: an 8x8 transform on its left side limits the edge to 8 taps, as get_filter_length does)
    want = oracle.deblock_plane(pix, params, 0, bd, order=0)
    assert np.array_equal(_run_fused(hip, ctx, pix, params, bd, 0, border=160), want)
