"""Superblock-bucketed SAD batch (aomhip_sad_sb_batch) == oracle aom_sadWxH / x4d, bit-exact: all block sizes,
8/10/12-bit, skip forms, cell / range geometries incl. ragged right/bottom cells and windows clipped by the
frame border, entries outside the declared range (served from global memory), empty buckets, either list
alone, per-frame lists, and the full-size 4K Mode-A work list."""
import numpy as np
import pytest

from conftest import BLOCK_SIZES

pytestmark = pytest.mark.gpu


def _lists(hip, rng, W, H, bw, bh, search, n_extra_far=0, border=160):
    cands, groups = hip.synth.mode_a_worklist(W, H, bw, seed=int(rng.integers(1 << 30)), search=search)
    if bw != bh:  # mode_a_worklist is square: rebuild for rectangular blocks
        bx, by = np.meshgrid(np.arange(0, W - bw + 1, bw, dtype=np.int16), np.arange(0, H - bh + 1, bh, dtype=np.int16))
        n = bx.size
        cands, groups = np.zeros(n, hip.capi.sad_cand_dtype), np.zeros(n, hip.capi.sad_x4d_dtype)
        cands["sx"] = cands["rx"] = groups["sx"] = bx.ravel(); cands["sy"] = cands["ry"] = groups["sy"] = by.ravel()
        groups["rx"] = groups["sx"][:, None] + rng.integers(-search, search + 1, (n, 4))
        groups["ry"] = groups["sy"][:, None] + rng.integers(-search, search + 1, (n, 4))
    # keep every reference block inside the bordered plane
    groups["rx"] = np.clip(groups["rx"], -border, W + border - bw); groups["ry"] = np.clip(groups["ry"], -border, H + border - bh)
    if n_extra_far:  # entries that violate the declared range
        idx = rng.choice(len(groups), n_extra_far, replace=False)
        groups["rx"][idx, 1] = rng.integers(-border, W + border - bw + 1, n_extra_far)
        groups["ry"][idx, 2] = rng.integers(-border, H + border - bh + 1, n_extra_far)
        cands["rx"][idx] = rng.integers(-border, W + border - bw + 1, n_extra_far)
    return cands, groups


def _run(hip, ctx, ps, pr, frame, nf, bw, bh, flags, sbw, sbh, search, cands, groups, W, H, cfs=0, gfs=0):
    res = []
    pg, og = hip.synth.bucket_order(groups["sx"][:len(groups) // max(nf if gfs else 1, 1)], groups["sy"][:len(groups) // max(nf if gfs else 1, 1)], W, H, sbw, sbh)
    n = len(pg)
    if gfs:
        gs = np.concatenate([groups[f * n:(f + 1) * n][pg] for f in range(nf)]); cs = np.concatenate([cands[f * n:(f + 1) * n][pg] for f in range(nf)])
    else:
        gs, cs = groups[pg], cands[pg]
    d_g, d_c, d_o = ctx.to_device(gs), ctx.to_device(cs), ctx.to_device(og)
    d_out4, d_out1 = ctx.malloc(nf * n * 16), ctx.malloc(nf * n * 4)
    ctx.memset(d_out4, 0xff, nf * n * 16); ctx.memset(d_out1, 0xff, nf * n * 4)
    ctx.sad_sb_batch(ps, pr, frame, nf, bw, bh, flags, sbw, sbh, search, len(og) - 1, d_g, d_o, n, n if gfs else 0, d_out4,
                     d_c, d_o, n, n if cfs else 0, d_out1)
    out4, out1 = ctx.from_device(d_out4, (nf, n, 4), np.uint32), ctx.from_device(d_out1, (nf, n), np.uint32)
    for d in (d_g, d_c, d_o, d_out4, d_out1):
        ctx.free(d)
    return gs, cs, out4, out1


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("w,h", BLOCK_SIZES)
def test_all_block_sizes(hip, oracle, ctx, w, h, bd):
    rng = np.random.default_rng(w * 131 + h * 7 + bd)
    W, H, border = 400, 272, 160  # not multiples of the 128-cell: ragged last column / row of cells
    src = hip.synth.lcg_frame(W, H, 1, 0, bd); ref = hip.synth.lcg_frame(W, H, 2, 1, bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 2)
    ctx.planes_upload(ps, 1, src); ctx.planes_upload(pr, 1, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    cands, groups = _lists(hip, rng, W, H, w, h, 24, n_extra_far=5)
    if len(groups) > 600:
        keep = np.sort(rng.choice(len(groups), 600, replace=False)); cands, groups = cands[keep], groups[keep]
    sbw, sbh = (128, 32) if bd > 8 else (128, 64)
    for flags in (0, 1):
        gs, cs, out4, out1 = _run(hip, ctx, ps, pr, 1, 1, w, h, flags, sbw, sbh, 32, cands, groups, W, H)
        assert np.array_equal(out4[0], oracle.sad_x4d_batch(sb, rb, border, w, h, gs, skip=bool(flags), bd=bd)), (w, h, bd, flags)
        assert np.array_equal(out1[0], oracle.sad_batch(sb, rb, border, w, h, cs, skip=bool(flags), bd=bd))
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("sbw,sbh,search", [(96, 48, 64), (240, 48, 64), (240, 64, 64), (320, 48, 64), (384, 32, 64), (480, 32, 64), (64, 64, 16), (128, 64, 64), (256, 32, 8), (48, 80, 20), (128, 128, 0), (640, 16, 64)])
def test_geometries_and_partial_lists(hip, oracle, ctx, sbw, sbh, search):
    rng = np.random.default_rng(sbw + sbh + search)
    W, H, border, bd = 704, 416, 160, 8
    src = hip.synth.lcg_frame(W, H, 3, 0, bd); ref = hip.synth.lcg_frame(W, H, 4, 1, bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    cands, groups = _lists(hip, rng, W, H, 16, 16, max(search, 1), n_extra_far=40)
    # leave some buckets empty
    keep = (groups["sx"] // sbw + groups["sy"] // sbh) % 3 != 1
    cands, groups = cands[keep], groups[keep]
    gs, cs, out4, out1 = _run(hip, ctx, ps, pr, 0, 1, 16, 16, 0, sbw, sbh, search, cands, groups, W, H)
    assert np.array_equal(out4[0], oracle.sad_x4d_batch(sb, rb, border, 16, 16, gs))
    assert np.array_equal(out1[0], oracle.sad_batch(sb, rb, border, 16, 16, cs))
    # groups only / cands only
    pg, og = hip.synth.bucket_order(groups["sx"], groups["sy"], W, H, sbw, sbh)
    d_g, d_o, d_out = ctx.to_device(groups[pg]), ctx.to_device(og), ctx.malloc(len(pg) * 16)
    ctx.sad_sb_batch(ps, pr, 0, 1, 16, 16, 0, sbw, sbh, search, len(og) - 1, d_g, d_o, len(pg), 0, d_out)
    assert np.array_equal(ctx.from_device(d_out, (len(pg), 4), np.uint32), out4[0])
    d_c = ctx.to_device(cands[pg])
    ctx.sad_sb_batch(ps, pr, 0, 1, 16, 16, 0, sbw, sbh, search, len(og) - 1, d_cands=d_c, d_cand_off=d_o, n_cands=len(pg), d_out_cands=d_out)
    assert np.array_equal(ctx.from_device(d_out, (len(pg),), np.uint32), out1[0])
    # argument errors: wrong bucket count, window too large for LDS
    with pytest.raises(hip.capi.AomHipError):
        ctx.sad_sb_batch(ps, pr, 0, 1, 16, 16, 0, sbw, sbh, search, len(og), d_g, d_o, len(pg), 0, d_out)
    with pytest.raises(hip.capi.AomHipError):
        ctx.sad_sb_batch(ps, pr, 0, 1, 16, 16, 0, 512, 512, 64, ((W + 511) // 512) * ((H + 511) // 512), d_g, d_o, len(pg), 0, d_out)
    for d in (d_g, d_o, d_out, d_c):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_per_frame_lists_and_frame_ring(hip, oracle, ctx):
    rng = np.random.default_rng(12)
    W, H, border, bd, F = 256, 256, 160, 8, 3
    ps, pr = ctx.planes_alloc(W, H, border, bd, F + 1), ctx.planes_alloc(W, H, border, bd, F + 1)
    frames = []
    for f in range(F):
        s, r = hip.synth.lcg_frame(W, H, 10 + f, 0, bd), hip.synth.lcg_frame(W, H, 20 + f, 1, bd)
        ctx.planes_upload(ps, 1 + f, s); ctx.planes_upload(pr, 1 + f, r)
        frames.append((oracle.extend_plane(s, border, ps.stride), oracle.extend_plane(r, border, pr.stride)))
    per = [_lists(hip, rng, W, H, 16, 16, 64) for _ in range(F)]
    cands = np.concatenate([p[0] for p in per]); groups = np.concatenate([p[1] for p in per])
    gs, cs, out4, out1 = _run(hip, ctx, ps, pr, 1, F, 16, 16, 0, 128, 64, 64, cands, groups, W, H, cfs=1, gfs=1)
    n = len(per[0][0])
    for f in range(F):
        sb, rb = frames[f]
        assert np.array_equal(out4[f], oracle.sad_x4d_batch(sb, rb, border, 16, 16, gs[f * n:(f + 1) * n]))
        assert np.array_equal(out1[f], oracle.sad_batch(sb, rb, border, 16, 16, cs[f * n:(f + 1) * n]))
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10])
def test_full_size_4k_mode_a(hip, oracle, ctx, bd):
    """BASELINE configs[1] 4K variant, whole frame: bucketed kernel == direct kernels == oracle."""
    W, H, border = 3840, 2160, 160
    src, ref = hip.synth.lcg_frame(W, H, 1, 0, bd), hip.synth.lcg_frame(W, H, 1, 1, bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    cands, groups = hip.synth.mode_a_worklist(W, H, 16, seed=5, search=64)
    sbw, sbh = (320, 48) if bd == 8 else (160, 32)  # the bench's cells at 4K
    gs, cs, out4, out1 = _run(hip, ctx, ps, pr, 0, 1, 16, 16, 0, sbw, sbh, 64, cands, groups, W, H)
    n = len(gs)
    d_g, d_c, d_o4, d_o1 = ctx.to_device(gs), ctx.to_device(cs), ctx.malloc(n * 16), ctx.malloc(n * 4)
    ctx.sad_x4d_batch(ps, pr, 0, 1, 16, 16, 0, d_g, n, 0, d_o4)
    ctx.sad_batch(ps, pr, 0, 1, 16, 16, 0, d_c, n, 0, d_o1)
    assert np.array_equal(ctx.from_device(d_o4, (n, 4), np.uint32), out4[0])
    assert np.array_equal(ctx.from_device(d_o1, (n,), np.uint32), out1[0])
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    assert np.array_equal(out4[0], oracle.sad_x4d_batch(sb, rb, border, 16, 16, gs, bd=bd, threads=8))
    assert np.array_equal(out1[0], oracle.sad_batch(sb, rb, border, 16, 16, cs, bd=bd, threads=8))
    for d in (d_g, d_c, d_o4, d_o1):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_crowded_buckets_overflow_the_descriptor_buffers(hip, oracle, ctx, monkeypatch):
    """More entries in a cell than one LDS descriptor buffer holds: the cell is evaluated in several slices."""
    monkeypatch.setenv("AOMHIP_SB_DESC_CAP", "24")
    rng = np.random.default_rng(77)
    W, H, border, bd = 320, 192, 160, 8
    src = hip.synth.lcg_frame(W, H, 5, 0, bd); ref = hip.synth.lcg_frame(W, H, 6, 1, bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    cands, groups = _lists(hip, rng, W, H, 8, 8, 16)   # 8x8 blocks: 128 entries per 128x64 cell
    cands = np.concatenate([cands, cands[::3]])        # unequal list lengths: 171 candidates vs 128 groups per cell
    cands["rx"] += rng.integers(-8, 9, len(cands)).astype(np.int16)
    pg, og = hip.synth.bucket_order(groups["sx"], groups["sy"], W, H, 128, 64)
    pc, oc = hip.synth.bucket_order(cands["sx"], cands["sy"], W, H, 128, 64)
    gs, cs = groups[pg], cands[pc]
    d_g, d_c, d_og, d_oc = ctx.to_device(gs), ctx.to_device(cs), ctx.to_device(og), ctx.to_device(oc)
    d_o4, d_o1 = ctx.malloc(len(gs) * 16), ctx.malloc(len(cs) * 4)
    ctx.sad_sb_batch(ps, pr, 0, 1, 8, 8, 0, 128, 64, 16, len(og) - 1, d_g, d_og, len(gs), 0, d_o4, d_c, d_oc, len(cs), 0, d_o1)
    assert np.array_equal(ctx.from_device(d_o4, (len(gs), 4), np.uint32), oracle.sad_x4d_batch(sb, rb, border, 8, 8, gs))
    assert np.array_equal(ctx.from_device(d_o1, (len(cs),), np.uint32), oracle.sad_batch(sb, rb, border, 8, 8, cs))
    for d in (d_g, d_c, d_og, d_oc, d_o4, d_o1):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10])
def test_bench_geometry_1080p_ring_of_frames_last_slot(hip, oracle, ctx, bd):
    """bench.py's own launch at 1080p: 240 x 64 cells (160 x 32 for 16-bit planes), a ring of 9 frame pairs in ONE launch, per-frame group
    lists (group_frame_stride = n) with a shared single-candidate list -- and the slots checked against the oracle are the first, a middle
    one and the LAST one (the far end of every per-frame stride the kernel uses: plane, list, output)."""
    W, H, border, F = 1920, 1080, 160, 9
    rng = np.random.default_rng(1080 + bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    frames = []
    for f in range(F):
        s, r = hip.synth.lcg_frame(W, H, 2 * f, 0, bd), hip.synth.lcg_frame(W, H, 2 * f + 1, 0, bd)
        ctx.planes_upload(ps, f, s); ctx.planes_upload(pr, f, r)
        frames.append((s, r))
    cands, groups = hip.synth.mode_a_worklist(W, H, 16, seed=3, search=64)
    n = len(groups)
    allg = np.tile(groups, (F, 1))
    allg["rx"] = allg["sx"][..., None] + rng.integers(-64, 65, (F, n, 4), dtype=np.int16)   # distinct positions per ring slot
    allg["ry"] = allg["sy"][..., None] + rng.integers(-64, 65, (F, n, 4), dtype=np.int16)
    sbw, sbh = (240, 64) if bd == 8 else (160, 32)
    perm, off = hip.synth.bucket_order(groups["sx"], groups["sy"], W, H, sbw, sbh)
    gs, cs = np.ascontiguousarray(allg[:, perm]), cands[perm]
    d_g, d_c, d_o = ctx.to_device(gs), ctx.to_device(cs), ctx.to_device(off)
    d_out4, d_out1 = ctx.malloc(F * n * 16), ctx.malloc(F * n * 4)
    ctx.memset(d_out4, 0xff, F * n * 16); ctx.memset(d_out1, 0xff, F * n * 4)
    ctx.sad_sb_batch(ps, pr, 0, F, 16, 16, 0, sbw, sbh, 64, len(off) - 1, d_g, d_o, n, n, d_out4, d_c, d_o, n, 0, d_out1)
    out4, out1 = ctx.from_device(d_out4, (F, n, 4), np.uint32), ctx.from_device(d_out1, (F, n), np.uint32)
    for f in (0, F // 2, F - 1):
        sb, rb = oracle.extend_plane(frames[f][0], border, ps.stride), oracle.extend_plane(frames[f][1], border, pr.stride)
        assert np.array_equal(out4[f], oracle.sad_x4d_batch(sb, rb, border, 16, 16, gs[f], bd=bd, threads=8)), f
        assert np.array_equal(out1[f], oracle.sad_batch(sb, rb, border, 16, 16, cs, bd=bd, threads=8)), f
    # every slot was written (none left at the 0xff fill) and the slots differ from each other
    assert not (out4 == 0xFFFFFFFF).any() and not (out1 == 0xFFFFFFFF).any()
    assert not np.array_equal(out4[0], out4[F - 1])
    for d in (d_g, d_c, d_o, d_out4, d_out1):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
