"""aomhip_cdef_search_sse_luma (the distortion table of av1_cdef_search: av1_cdef_mse_calc_block / get_filt_error,
av1/encoder/pickcdef.c:401-615) through the C ABI: against the interpreted reference's get_filt_error values, and
against the oracle on whole frames with random skip maps for all 64 full-search strength pairs."""
import numpy as np
import pytest

from cdef_search_fixture import load_cases, mapped_strengths, planes_of

pytestmark = pytest.mark.gpu


def _table(hip, ctx, recon, source, skip, strengths, damping, bd):
    H, W = recon.shape
    pr, ps = ctx.planes_alloc(W, H, 16, bd, 1), ctx.planes_alloc(W, H, 16, bd, 1)
    ctx.planes_upload(pr, 0, recon); ctx.planes_upload(ps, 0, source)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    st = np.asarray(strengths, np.uint8).reshape(-1, 2)
    d_st, d_skip = ctx.to_device(st), ctx.to_device(np.ascontiguousarray(skip, np.uint8))
    d_sse, d_dir = ctx.malloc(8 * len(st) * fbh * fbw), ctx.malloc((H // 8) * (W // 8))
    ctx.cdef_search_sse_luma(pr, 0, ps, 0, d_st, len(st), d_skip, damping, fbw, d_sse, d_dir, None)
    got = ctx.from_device(d_sse, (len(st), fbh, fbw), np.uint64)
    dirs = ctx.from_device(d_dir, (H // 8, W // 8), np.uint8)
    for d in (d_st, d_skip, d_sse, d_dir):
        ctx.free(d)
    ctx.planes_free(pr); ctx.planes_free(ps)
    return got, dirs


def _table_chroma(hip, ctx, recon, source, skip, ldir, strengths, damping, bd, xdec, ydec):
    H, W = recon.shape
    pr, ps = ctx.planes_alloc(W, H, 16, bd, 1), ctx.planes_alloc(W, H, 16, bd, 1)
    ctx.planes_upload(pr, 0, recon); ctx.planes_upload(ps, 0, source)
    fh, fw = 64 >> ydec, 64 >> xdec
    fbh, fbw = (H + fh - 1) // fh, (W + fw - 1) // fw
    st = np.asarray(strengths, np.uint8).reshape(-1, 2)
    d_st, d_skip, d_dir = ctx.to_device(st), ctx.to_device(np.ascontiguousarray(skip, np.uint8)), ctx.to_device(np.ascontiguousarray(ldir, np.uint8))
    d_sse = ctx.malloc(8 * len(st) * fbh * fbw)
    ctx.cdef_search_sse_chroma(pr, 0, ps, 0, xdec, ydec, d_dir, d_st, len(st), d_skip, damping, fbw, d_sse)
    got = ctx.from_device(d_sse, (len(st), fbh, fbw), np.uint64)
    for d in (d_st, d_skip, d_dir, d_sse):
        ctx.free(d)
    ctx.planes_free(pr); ctx.planes_free(ps)
    return got


def test_cdef_search_goldens(hip, ctx):
    z, cases = load_cases()
    assert len(cases) == 6
    for c in cases:
        recon, source, skip, fb, ldir = planes_of(z, c)
        if c["pli"]:
            got = _table_chroma(hip, ctx, recon, source, skip, ldir, mapped_strengths(c), c["damping"], c["bd"], 1, 1)
        else:
            got, _ = _table(hip, ctx, recon, source, skip, mapped_strengths(c), c["damping"], c["bd"])
        shift = 2 * (c["bd"] - 8)
        assert [int(v) >> shift for v in got[:, fb[0], fb[1]]] == c["errors"], (c["variant"], c["pli"])
        got[:, fb[0], fb[1]] = 0
        assert not got.any()          # all-skip filter blocks contribute nothing


@pytest.mark.parametrize("bd,xdec,ydec", [(8, 1, 1), (10, 1, 1), (10, 0, 0), (8, 1, 0), (12, 0, 1)])
def test_cdef_search_chroma_vs_oracle(hip, oracle, ctx, bd, xdec, ydec):
    rng = np.random.default_rng(10 * bd + 2 * xdec + ydec)
    LW, LH = 256, 192                           # luma size; the chroma plane is subsampled from it
    W, H = LW >> xdec, LH >> ydec
    recon = hip.synth.lcg_frame(W, H, 3, 0, bd)
    source = np.clip(recon.astype(np.int64) + rng.integers(-(6 << (bd - 8)), (6 << (bd - 8)) + 1, (H, W)), 0, (1 << bd) - 1).astype(recon.dtype)
    skip = (rng.integers(0, 4, (LH // 8, LW // 8)) == 0).astype(np.uint8)
    ldir = rng.integers(0, 8, (LH // 8, LW // 8)).astype(np.uint8)
    full = [(gi // 4, (gi % 4) + (gi % 4 == 3)) for gi in range(64)]
    strengths = full if (bd, xdec, ydec) == (10, 1, 1) else [full[i] for i in (0, 2, 5, 11, 23, 36, 47, 63)]
    got = _table_chroma(hip, ctx, recon, source, skip, ldir, strengths, 4, bd, xdec, ydec)
    want = oracle.cdef_search_sse_chroma(recon, source, xdec, ydec, ldir, strengths, skip, 4, bd)
    assert np.array_equal(got, want), (bd, xdec, ydec)


@pytest.mark.parametrize("bd,W,H,n_strengths", [(8, 256, 192, 64), (10, 320, 200, 64), (12, 136, 72, 12)])
def test_cdef_search_vs_oracle(hip, oracle, ctx, bd, W, H, n_strengths):
    rng = np.random.default_rng(bd)
    recon = hip.synth.lcg_frame(W, H, 2, 0, bd)
    noise = rng.integers(-(5 << (bd - 8)), (5 << (bd - 8)) + 1, (H, W))
    source = np.clip(recon.astype(np.int64) + noise, 0, (1 << bd) - 1).astype(recon.dtype)
    recon[8:40, 8:72] = np.where(rng.integers(0, 2, (32, 64)) > 0, (1 << bd) - 1, 0).astype(recon.dtype)   # ringing-prone texture
    skip = (rng.integers(0, 4, (H // 8, W // 8)) == 0).astype(np.uint8)
    skip[:, -3:] = 1
    # CDEF_FULL_SEARCH: pri = gi / 4, sec = gi % 4 mapped 3 -> 4 (get_cdef_filter_strengths, pickcdef.c:29-84)
    full = [(gi // 4, (gi % 4) + (gi % 4 == 3)) for gi in range(64)]
    strengths = full if n_strengths == 64 else [full[i] for i in (0, 1, 5, 7, 12, 19, 27, 33, 42, 51, 60, 63)]
    for damping in ((3, 6) if bd == 8 else (5,)):
        got, dirs = _table(hip, ctx, recon, source, skip, strengths, damping, bd)
        want = oracle.cdef_search_sse_luma(recon, source, strengths, skip, damping, bd)
        assert np.array_equal(got, want), (bd, damping)
        _, wdir, _ = oracle.cdef_plane_luma(recon, np.ones(((H + 63) // 64, (W + 63) // 64), np.uint8), np.zeros(((H + 63) // 64, (W + 63) // 64), np.uint8), skip, damping, bd)
        assert np.array_equal(dirs, wdir)
    # the zero-strength column is the plain distortion of the unfiltered reconstruction over the non-skip units
    keep = np.kron((skip == 0).astype(np.int64), np.ones((8, 8), np.int64))
    e = ((recon.astype(np.int64) - source.astype(np.int64)) ** 2 * keep)
    assert int(got[0].sum()) == int(e.sum())


def test_cdef_search_rejects_bad_arguments(hip, ctx):
    K = hip.capi
    p8, p10 = ctx.planes_alloc(64, 64, 16, 8, 1), ctx.planes_alloc(64, 64, 16, 10, 1)
    odd = ctx.planes_alloc(60, 64, 16, 8, 1)
    d = ctx.malloc(4096)
    for args in ((p8, 0, p10, 0, d, 4, d, 5, 1, d), (p8, 0, p8, 0, d, 0, d, 5, 1, d), (p8, 0, p8, 0, d, 65, d, 5, 1, d),
                 (p8, 0, p8, 0, d, 4, d, 7, 1, d), (p8, 0, p8, 0, d, 4, d, 5, 0, d), (odd, 0, odd, 0, d, 4, d, 5, 1, d),
                 (p8, 1, p8, 0, d, 4, d, 5, 1, d)):
        with pytest.raises(K.AomHipError):
            ctx.cdef_search_sse_luma(*args)
    ctx.free(d)
    for p in (p8, p10, odd):
        ctx.planes_free(p)
