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
