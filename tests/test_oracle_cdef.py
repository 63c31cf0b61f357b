"""CDEF oracle: definitional properties (the reference's own CDEF tests are SIMD-vs-C only).
Direction numbering per cdef_block.c:50-56: 0 = 45 degrees up-right, 2 = horizontal, 6 = vertical."""
import numpy as np


def test_find_dir_on_synthetic_edges(oracle):
    i, j = np.indices((8, 8))
    horizontal = (np.where(i < 4, 40, 200)).astype(np.uint16)           # constant along rows -> direction 2
    vertical = (np.where(j < 4, 40, 200)).astype(np.uint16)             # constant along columns -> 6
    diag_up_right = (np.where(i + j < 7, 40, 200)).astype(np.uint16)    # constant along i + j -> 0
    diag_down_right = (np.where(i - j < 0, 40, 200)).astype(np.uint16)  # constant along i - j -> 4
    assert oracle.cdef_find_dir(horizontal)[0] == 2
    assert oracle.cdef_find_dir(vertical)[0] == 6
    assert oracle.cdef_find_dir(diag_up_right)[0] == 0
    assert oracle.cdef_find_dir(diag_down_right)[0] == 4
    d, var = oracle.cdef_find_dir(np.full((8, 8), 99, np.uint16))
    assert (d, var) == (0, 0)  # flat block: every cost equal -> first direction, zero variance
    # 10-bit input with coeff_shift 2 sees the same structure
    assert oracle.cdef_find_dir(horizontal * 4, 2) == oracle.cdef_find_dir(horizontal, 0)


def test_plane_identity_cases(oracle):
    rng = np.random.default_rng(0)
    pix = rng.integers(0, 256, (128, 192), dtype=np.uint8)
    z = np.zeros((2, 3), np.uint8)
    noskip = np.zeros((16, 24), np.uint8)
    out, _, _ = oracle.cdef_plane_luma(pix, z, z, noskip, 3)
    assert np.array_equal(out, pix)                                   # zero strength
    out, _, _ = oracle.cdef_plane_luma(pix, z + 7, z + 2, noskip + 1, 4)
    assert np.array_equal(out, pix)                                   # every 8x8 skipped
    flat = np.full((64, 64), 123, np.uint8)
    out, _, _ = oracle.cdef_plane_luma(flat, np.full((1, 1), 15, np.uint8), np.full((1, 1), 4, np.uint8),
                                       np.zeros((8, 8), np.uint8), 6)
    assert np.array_equal(out, flat)                                  # flat area: all differences are zero


def test_plane_filtering_is_bounded_and_local(oracle):
    rng = np.random.default_rng(1)
    pix = (np.kron(rng.integers(40, 200, (16, 24)), np.ones((8, 8), np.int64)) + rng.integers(-6, 7, (128, 192)))
    pix = np.clip(pix, 0, 255).astype(np.uint8)
    pri = np.array([[4, 0, 9], [15, 2, 0]], np.uint8); sec = np.array([[2, 0, 1], [4, 0, 4]], np.uint8)
    skip = (rng.random((16, 24)) < 0.2).astype(np.uint8)
    out, d, v = oracle.cdef_plane_luma(pix, pri, sec, skip, 5)
    assert not np.array_equal(out, pix)
    # the filter moves a pixel by at most the sum of tap weights times the strengths / 16
    assert np.abs(out.astype(int) - pix.astype(int)).max() <= 15 + 4 + 1
    # skipped blocks and zero-strength filter blocks are untouched
    for by in range(16):
        for bx in range(24):
            blk = (slice(by * 8, by * 8 + 8), slice(bx * 8, bx * 8 + 8))
            if skip[by, bx] or (pri[by // 8, bx // 8] == 0 and sec[by // 8, bx // 8] == 0):
                assert np.array_equal(out[blk], pix[blk])
    # with primary and secondary both on, the output is clamped to the range of the taps it read (cdef_block.c:186-188)
    assert out.max() <= pix.max() and out.min() >= pix.min()


def test_chroma_identity_and_direction_reuse(oracle):
    """Chroma driver: zero uv strength or skip = copy; directions exist for blocks of a zero-LUMA-strength filter block
    when requested (the chroma planes need them, cdef.c:334-345); 4:2:2 / 4:4:0 direction conversion tables."""
    rng = np.random.default_rng(2)
    W, H = 128, 64
    luma = rng.integers(0, 256, (H, W)).astype(np.uint8)
    zero = np.zeros((1, 2), np.uint8)
    skip = np.zeros((H // 8, W // 8), np.uint8)
    out, d, v = oracle.cdef_plane_luma(luma, zero, zero, skip, 6, 8)
    assert np.array_equal(out, luma) and d.max() > 0
    u = rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8)
    assert np.array_equal(oracle.cdef_plane_chroma(u, 1, 1, d, zero, zero, skip, 6, 8), u)
    allskip = np.ones_like(skip)
    four, two = np.full((1, 2), 4, np.uint8), np.full((1, 2), 2, np.uint8)
    assert np.array_equal(oracle.cdef_plane_chroma(u, 1, 1, d, four, two, allskip, 6, 8), u)
    f = oracle.cdef_plane_chroma(u, 1, 1, d, four, two, skip, 6, 8)
    assert not np.array_equal(f, u) and np.abs(f.astype(int) - u).max() <= 4 + 2 * 2  # bounded by the strengths
    # 4:2:2 with every luma direction forced to 0 equals 4:2:2 filtering along converted direction 7 (conv422[0])
    u422 = rng.integers(0, 256, (H, W // 2)).astype(np.uint8)
    a = oracle.cdef_plane_chroma(u422, 1, 0, np.zeros_like(d), four, zero, skip, 6, 8)
    taps = a.astype(int) - u422
    assert np.abs(taps).max() <= 4 and np.abs(taps).max() > 0
