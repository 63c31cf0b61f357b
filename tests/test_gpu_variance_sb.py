"""aomhip_variance_sb_batch (the variance form of the strip walk, csrc/sad_sb.hip) == oracle aom_varianceWxH / aom_highbd_{10,12}_varianceWxH
(aom_dsp/variance.c:56-163,383-420), bit-exact: every block size of at most 256 pixels, 8/10/12-bit, ragged cells, entries outside the
declared range (served from global memory), either list alone, per-frame lists, the extreme planes of every bit depth (largest sse a 32-bit
sum must hold), and the whole 1080p Mode-A list against the direct kernel."""
import numpy as np
import pytest

from test_gpu_sad_sb import _lists

pytestmark = pytest.mark.gpu
VAR_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (4, 16), (16, 4), (8, 32), (32, 8)]


def _var_cands(hip, gs, cs):
    """the same evaluations as a flat aomhip_var_cand list: 4 per group, then the single candidates"""
    vc = np.zeros(4 * len(gs) + len(cs), hip.capi.var_cand_dtype)
    vc["sx"][:4 * len(gs)] = np.repeat(gs["sx"], 4); vc["sy"][:4 * len(gs)] = np.repeat(gs["sy"], 4)
    vc["rx"][:4 * len(gs)] = gs["rx"].ravel(); vc["ry"][:4 * len(gs)] = gs["ry"].ravel()
    for k in ("sx", "sy", "rx", "ry"):
        vc[k][4 * len(gs):] = cs[k]
    return vc


def _run(hip, ctx, ps, pr, frame, nf, bw, bh, sbw, sbh, search, cands, groups, W, H, use_g=True, use_c=True):
    pg, og = hip.synth.bucket_order(groups["sx"], groups["sy"], W, H, sbw, sbh)
    n = len(pg)
    gs, cs = groups[pg], cands[pg]
    d_g, d_c, d_o = ctx.to_device(gs), ctx.to_device(cs), ctx.to_device(og)
    outs = [ctx.malloc(nf * n * 16), ctx.malloc(nf * n * 16), ctx.malloc(nf * n * 4), ctx.malloc(nf * n * 4)]
    for d, b in zip(outs, (16, 16, 4, 4)):
        ctx.memset(d, 0xff, nf * n * b)
    ctx.variance_sb_batch(ps, pr, frame, nf, bw, bh, sbw, sbh, search, len(og) - 1, d_g if use_g else None, d_o if use_g else None, n if use_g else 0, 0,
                          outs[0] if use_g else None, outs[1] if use_g else None, d_c if use_c else None, d_o if use_c else None, n if use_c else 0, 0,
                          outs[2] if use_c else None, outs[3] if use_c else None)
    v4, s4 = ctx.from_device(outs[0], (nf, n, 4), np.uint32), ctx.from_device(outs[1], (nf, n, 4), np.uint32)
    v1, s1 = ctx.from_device(outs[2], (nf, n), np.uint32), ctx.from_device(outs[3], (nf, n), np.uint32)
    for d in [d_g, d_c, d_o] + outs:
        ctx.free(d)
    return gs, cs, v4, s4, v1, s1


@pytest.mark.parametrize("bd", [8, 10, 12])
@pytest.mark.parametrize("w,h", VAR_SIZES)
def test_block_sizes_and_bit_depths(hip, oracle, ctx, w, h, bd):
    rng = np.random.default_rng(w * 17 + h * 3 + bd)
    W, H, border = 400, 272, 160
    src = hip.synth.lcg_frame(W, H, 1, 0, bd); ref = hip.synth.lcg_frame(W, H, 2, 1, bd)
    ps, pr = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 2)
    ctx.planes_upload(ps, 1, src); ctx.planes_upload(pr, 1, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    cands, groups = _lists(hip, rng, W, H, w, h, 24, n_extra_far=5)
    if len(groups) > 300:
        keep = np.sort(rng.choice(len(groups), 300, replace=False)); cands, groups = cands[keep], groups[keep]
    sbw, sbh = (128, 32) if bd > 8 else (128, 64)
    gs, cs, v4, s4, v1, s1 = _run(hip, ctx, ps, pr, 1, 1, w, h, sbw, sbh, 32, cands, groups, W, H)
    want = oracle.variance_cands(sb, rb, border, w, h, _var_cands(hip, gs, cs), bd=bd)
    n = len(gs)
    assert np.array_equal(v4[0].ravel(), want[:4 * n, 0]) and np.array_equal(s4[0].ravel(), want[:4 * n, 1]), (w, h, bd)
    assert np.array_equal(v1[0], want[4 * n:, 0]) and np.array_equal(s1[0], want[4 * n:, 1])
    ctx.planes_free(ps); ctx.planes_free(pr)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_extreme_planes_and_either_list_alone(hip, oracle, ctx, bd):
    """source all zero against reference all (2^bd - 1): sse = 256 (2^bd - 1)^2 -- 4 292 870 400 at 12 bits, the largest value the 32-bit sums hold."""
    rng = np.random.default_rng(bd)
    W, H, border = 256, 128, 64
    mx = (1 << bd) - 1
    dt = np.uint8 if bd == 8 else np.uint16
    src = np.zeros((H, W), dt); ref = np.full((H, W), mx, dt)
    src[64:, :] = rng.integers(0, mx + 1, (64, W)); ref[64:, 128:] = 0
    ps, pr = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    sb, rb = oracle.extend_plane(src, border, ps.stride), oracle.extend_plane(ref, border, pr.stride)
    cands, groups = _lists(hip, rng, W, H, 16, 16, 16, border=border)
    for use_g, use_c in ((True, True), (True, False), (False, True)):
        gs, cs, v4, s4, v1, s1 = _run(hip, ctx, ps, pr, 0, 1, 16, 16, 128, 32, 16, cands, groups, W, H, use_g, use_c)
        want = oracle.variance_cands(sb, rb, border, 16, 16, _var_cands(hip, gs, cs), bd=bd)
        n = len(gs)
        if use_g:
            assert np.array_equal(v4[0].ravel(), want[:4 * n, 0]) and np.array_equal(s4[0].ravel(), want[:4 * n, 1])
        if use_c:
            assert np.array_equal(v1[0], want[4 * n:, 0]) and np.array_equal(s1[0], want[4 * n:, 1])
    assert int(want[:, 1].max()) == (256 * mx * mx if bd == 8 else (256 * mx * mx + (8 if bd == 10 else 128)) >> (4 if bd == 10 else 8))
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_whole_1080p_mode_a_list_against_the_direct_kernel(hip, ctx):
    W, H, border, bd, F = 1920, 1080, 160, 8, 2
    ps, pr = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    for f in range(F):
        ctx.planes_upload(ps, f, hip.synth.lcg_frame(W, H, 2 * f, 0, bd)); ctx.planes_upload(pr, f, hip.synth.lcg_frame(W, H, 2 * f + 1, 0, bd))
    cands, groups = hip.synth.mode_a_worklist(W, H, 16, seed=3, search=64)
    gs, cs, v4, s4, v1, s1 = _run(hip, ctx, ps, pr, 0, F, 16, 16, 240, 64, 64, cands, groups, W, H)
    vc = _var_cands(hip, gs, cs)
    d_vc = ctx.to_device(vc)
    d_v, d_s = ctx.malloc(F * len(vc) * 4), ctx.malloc(F * len(vc) * 4)
    ctx.variance_batch(ps, pr, 0, F, 16, 16, d_vc, len(vc), 0, d_v, d_s)
    dv, ds = ctx.from_device(d_v, (F, len(vc)), np.uint32), ctx.from_device(d_s, (F, len(vc)), np.uint32)
    n = len(gs)
    for f in range(F):
        assert np.array_equal(v4[f].ravel(), dv[f, :4 * n]) and np.array_equal(s4[f].ravel(), ds[f, :4 * n])
        assert np.array_equal(v1[f], dv[f, 4 * n:]) and np.array_equal(s1[f], ds[f, 4 * n:])
    for d in (d_vc, d_v, d_s):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)


def test_large_blocks_and_flags_are_refused(hip, ctx):
    capi = hip.capi
    ps, pr = ctx.planes_alloc(128, 128, 64, 8, 1), ctx.planes_alloc(128, 128, 64, 8, 1)
    d = ctx.malloc(4096)
    off = ctx.to_device(np.zeros(2, np.int32))
    with pytest.raises(capi.AomHipError):
        ctx.variance_sb_batch(ps, pr, 0, 1, 32, 32, 128, 128, 16, 1, d, off, 0, 0, d, d)
    with pytest.raises(capi.AomHipError):
        ctx.variance_sb_batch(ps, pr, 0, 1, 16, 16, 128, 128, 16, 1, d, off, 1, 0, d, None)   # no sse array
    ctx.free(d); ctx.free(off)
    ctx.planes_free(ps); ctx.planes_free(pr)
