"""Sub-pel motion-compensated prediction (aomhip_build_inter_pred_batch: av1_enc_build_inter_predictor for a single,
unscaled reference -> av1_[highbd_]convolve_2d_facade) through the C ABI: against the interpreted reference's vectors
(tests/golden/ref_eval_convolve.npz), against the oracle for all 22 block sizes x 8/10/12-bit x the four interpolation
filters with MVs covering every 1/8-pel phase pair (copy / x-only / y-only / 2-D cases), extreme content, MVs that
reach into the border, and the integer-MV case against the full-pel gather."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOCK_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32),
               (64, 64), (64, 128), (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64), (64, 16)]


def test_inter_pred_goldens(hip, ctx):
    """The fixture's phases are in 1/16 pel (subpel_x_qn).  A luma MV (1/8 pel) reaches the even ones; with chroma subsampling
    the MV itself is in sixteenths, so the 4:2:0 form of the call replays EVERY fixture case, and the luma form the even ones."""
    z = np.load(os.path.join(GOLD, "ref_eval_convolve.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    planes = {}
    border = 16
    for bd in (8, 10, 12):
        p = z["p%d" % bd]
        dt = np.uint8 if bd == 8 else np.uint16
        H, W = p.shape
        pr, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(pr, 0, np.ascontiguousarray(p, dt))
        planes[bd] = (pr, pp, W, H)
    luma = 0
    for c in cases:
        pr, pp, W, H = planes[c["bd"]]
        w, h = c["w"], c["h"]
        blk = np.zeros(1, hip.capi.search_block_dtype)
        blk["bx"], blk["by"] = c["x0"], c["y0"]
        d_b = ctx.to_device(blk)
        forms = [(1, np.array([[c["sy"], c["sx"]]], np.int16))]
        if c["sx"] % 2 == 0 and c["sy"] % 2 == 0:
            forms.append((0, np.array([[c["sy"] // 2, c["sx"] // 2]], np.int16)))
            luma += 1
        for ss, mv in forms:
            d_mv = ctx.to_device(mv)
            ctx.build_inter_pred_batch(pr, 0, pp, 0, w, h, d_b, d_mv, 1, c["fx"], c["fy"], ss, ss)
            got = ctx.planes_download(pp, 0)[border + c["y0"]:border + c["y0"] + h, border + c["x0"]:border + c["x0"] + w]
            assert np.array_equal(got.ravel().astype(np.uint16), z["d%d" % c["k"]]), (ss, c)
            ctx.free(d_mv)
        ctx.free(d_b)
    assert len(cases) >= 200 and luma >= 40
    for pr, pp, _, _ in planes.values():
        ctx.planes_free(pr); ctx.planes_free(pp)


@pytest.mark.parametrize("bd,ss_x,ss_y", [(8, 1, 1), (10, 1, 1), (10, 1, 0), (12, 0, 1)])
def test_chroma_inter_pred_vs_oracle(hip, oracle, ctx, bd, ss_x, ss_y):
    """Chroma planes: luma MV x (1 << (1 - subsampling)) sixteenths -- all 16 phases, 4-tap sets for the small chroma blocks."""
    rng = np.random.default_rng(60 + bd + ss_x)
    W, H, border = 192, 128, 32
    ref = hip.synth.lcg_frame(W, H, 7, 0, bd)
    pr, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pr, 0, ref)
    rb = oracle.extend_plane(ref, border, pr.stride)
    for (bw, bh) in ((4, 4), (8, 8), (4, 8), (16, 16), (8, 4), (32, 32), (16, 8)):
        xs, ys = np.meshgrid(np.arange(0, W - bw + 1, bw), np.arange(0, H - bh + 1, bh))
        n = xs.size
        blocks = np.zeros(n, hip.capi.search_block_dtype)
        blocks["bx"], blocks["by"] = xs.ravel(), ys.ravel()
        lim_x, lim_y = (border - 8) * (16 >> (1 - ss_x)), (border - 8) * (16 >> (1 - ss_y))
        mv = np.stack([rng.integers(-lim_y, lim_y + 1, n), rng.integers(-lim_x, lim_x + 1, n)], axis=1).astype(np.int16)
        d_b, d_mv = ctx.to_device(blocks), ctx.to_device(mv)
        for fx, fy in ((0, 0), (2, 1), (1, 3)):
            ctx.planes_upload(pp, 0, np.zeros_like(ref))
            ctx.build_inter_pred_batch(pr, 0, pp, 0, bw, bh, d_b, d_mv, n, fx, fy, ss_x, ss_y)
            got = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
            want = oracle.build_inter_pred(rb, border, W, H, bw, bh, blocks, mv, fx, fy, bd, ss_x, ss_y)
            assert np.array_equal(got, want), (bw, bh, bd, fx, fy)
        ctx.free(d_b); ctx.free(d_mv)
    ctx.planes_free(pr); ctx.planes_free(pp)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_inter_pred_vs_oracle(hip, oracle, ctx, bd):
    rng = np.random.default_rng(40 + bd)
    W, H, border = 256, 256, 64
    ref = hip.synth.lcg_frame(W, H, 5, 0, bd)
    mx = (1 << bd) - 1
    ref[40:72, 100:180] = np.where(rng.integers(0, 2, (32, 80)) > 0, mx, 0).astype(ref.dtype)   # saturating checkerboard noise
    pr, pp = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pr, 1, ref)
    rb = oracle.extend_plane(ref, border, pr.stride)
    for si, (bw, bh) in enumerate(BLOCK_SIZES):
        xs, ys = np.meshgrid(np.arange(0, W - bw + 1, bw), np.arange(0, H - bh + 1, bh))
        n = xs.size
        blocks = np.zeros(n, hip.capi.search_block_dtype)
        blocks["bx"], blocks["by"] = xs.ravel(), ys.ravel()
        # MVs in 1/8 pel: up to +-(border - 8) pixels, all 64 phase pairs, with the integer / x-only / y-only cases forced in
        lim = (border - 8) * 8
        mv = rng.integers(-lim, lim + 1, (n, 2)).astype(np.int16)
        mv[0::7] &= ~7
        mv[1::7, 0] &= ~7
        mv[2::7, 1] &= ~7
        d_b, d_mv = ctx.to_device(blocks), ctx.to_device(mv)
        for fx, fy in (((0, 0), (1, 2), (2, 1), (3, 3), (2, 0), (0, 1)) if (bw, bh) in ((16, 16), (4, 8), (64, 64)) else ((si % 4, (si + 1) % 3),)):
            ctx.planes_upload(pp, 0, np.zeros_like(ref))
            ctx.build_inter_pred_batch(pr, 1, pp, 0, bw, bh, d_b, d_mv, n, fx, fy)
            got = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
            want = oracle.build_inter_pred(rb, border, W, H, bw, bh, blocks, mv, fx, fy, bd)
            assert np.array_equal(got, want), (bw, bh, bd, fx, fy)
        if (bw, bh) == (16, 16):
            # integer MVs: the interpolating path equals the full-pel gather
            mvi = (mv >> 3).astype(np.int16)
            d_i, d_i8 = ctx.to_device(mvi), ctx.to_device((mvi * 8).astype(np.int16))
            ctx.build_pred_fullpel(pr, 1, pp, 0, bw, bh, d_b, d_i, n)
            a = ctx.planes_download(pp, 0).copy()
            ctx.build_inter_pred_batch(pr, 1, pp, 0, bw, bh, d_b, d_i8, n, 2, 1)
            assert np.array_equal(a, ctx.planes_download(pp, 0))
            ctx.free(d_i); ctx.free(d_i8)
        ctx.free(d_b); ctx.free(d_mv)
    ctx.planes_free(pr); ctx.planes_free(pp)


def test_inter_pred_rejects_bad_arguments(hip, ctx):
    K = hip.capi
    pr, pp = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 8, 1)
    small = ctx.planes_alloc(64, 64, 4, 8, 1)
    p10 = ctx.planes_alloc(64, 64, 32, 10, 1)
    d = ctx.malloc(64)
    with pytest.raises(K.AomHipError):
        ctx.build_inter_pred_batch(pr, 0, pp, 0, 16, 16, d, d, 1, 4, 0)      # MULTITAP_SHARP2 is not served
    with pytest.raises(K.AomHipError):
        ctx.build_inter_pred_batch(pr, 0, pp, 0, 16, 12, d, d, 1, 0, 0)      # not a block size
    with pytest.raises(K.AomHipError):
        ctx.build_inter_pred_batch(small, 0, pp, 0, 16, 16, d, d, 1, 0, 0)   # border too small for 8 taps
    with pytest.raises(K.AomHipError):
        ctx.build_inter_pred_batch(pr, 0, p10, 0, 16, 16, d, d, 1, 0, 0)     # bit depths differ
    with pytest.raises(K.AomHipError):
        ctx.build_inter_pred_batch(pr, 1, pp, 0, 16, 16, d, d, 1, 0, 0)      # no such frame
    ctx.build_inter_pred_batch(pr, 0, pp, 0, 16, 16, None, None, 0, 0, 0)    # empty list
    ctx.free(d)
    for p in (pr, pp, small, p10):
        ctx.planes_free(p)


def test_compound_pred_goldens(hip, ctx):
    """Every case of ref_eval_convolve_compound.npz through aomhip_build_compound_pred_batch (4:2:0 form: MVs in sixteenths)."""
    z = np.load(os.path.join(GOLD, "ref_eval_convolve_compound.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    border = 16
    planes = {}
    for bd in (8, 10, 12):
        dt = np.uint8 if bd == 8 else np.uint16
        p0, p1 = np.ascontiguousarray(z["p%d_0" % bd], dt), np.ascontiguousarray(z["p%d_1" % bd], dt)
        H, W = p0.shape
        r0, r1, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(r0, 0, p0); ctx.planes_upload(r1, 0, p1)
        planes[bd] = (r0, r1, pp)
    for c in cases:
        r0, r1, pp = planes[c["bd"]]
        w, h = c["w"], c["h"]
        (x0, y0), (x1, y1) = c["pos"]
        (sx0, sy0), (sx1, sy1) = c["subs"]
        # the block is placed at reference 0's position; reference 1's displacement goes into its MV (sixteenths)
        blk = np.zeros(1, hip.capi.search_block_dtype)
        blk["bx"], blk["by"] = x0, y0
        mv0 = np.array([[sy0, sx0]], np.int16)
        mv1 = np.array([[(y1 - y0) * 16 + sy1, (x1 - x0) * 16 + sx1]], np.int16)
        wts = c["weights"] or (0, 0)
        d_b, d_0, d_1 = ctx.to_device(blk), ctx.to_device(mv0), ctx.to_device(mv1)
        ctx.build_compound_pred_batch(r0, 0, r1, 0, pp, 0, w, h, d_b, d_0, d_1, 1, c["fx"], c["fy"], wts[0], wts[1], 1, 1)
        got = ctx.planes_download(pp, 0)[border + y0:border + y0 + h, border + x0:border + x0 + w]
        assert np.array_equal(got.ravel().astype(np.uint16), z["d%d" % c["k"]]), c
        for d in (d_b, d_0, d_1):
            ctx.free(d)
    assert len(cases) >= 40
    for t in planes.values():
        for p in t:
            ctx.planes_free(p)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_compound_pred_vs_oracle(hip, oracle, ctx, bd):
    rng = np.random.default_rng(80 + bd)
    W, H, border = 256, 128, 64
    ref0, ref1 = hip.synth.lcg_frame(W, H, 8, 0, bd), hip.synth.lcg_frame(W, H, 9, 1, bd)
    ref0[:24, :48] = (1 << bd) - 1
    ref1[:24, :48] = np.where(rng.integers(0, 2, (24, 48)) > 0, (1 << bd) - 1, 0).astype(ref1.dtype)
    r0, r1, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(r0, 0, ref0); ctx.planes_upload(r1, 1, ref1)
    b0, b1 = oracle.extend_plane(ref0, border, r0.stride), oracle.extend_plane(ref1, border, r1.stride)
    for si, (bw, bh) in enumerate(BLOCK_SIZES):
        xs, ys = np.meshgrid(np.arange(0, W - bw + 1, bw), np.arange(0, H - bh + 1, bh))
        n = xs.size
        blocks = np.zeros(n, hip.capi.search_block_dtype)
        blocks["bx"], blocks["by"] = xs.ravel(), ys.ravel()
        lim = (border - 8) * 8
        mv0, mv1 = rng.integers(-lim, lim + 1, (n, 2)).astype(np.int16), rng.integers(-lim, lim + 1, (n, 2)).astype(np.int16)
        mv0[0::5] &= ~7
        mv1[1::5, 0] &= ~7
        mv0[2::5, 1] &= ~7
        d_b, d_0, d_1 = ctx.to_device(blocks), ctx.to_device(mv0), ctx.to_device(mv1)
        for fx, fy, fwd, bck in ((si % 4, (si + 2) % 4, 0, 0), ((si + 1) % 3, si % 3, (9, 11, 12, 13)[si % 4], (7, 5, 4, 3)[si % 4])):
            ctx.build_compound_pred_batch(r0, 0, r1, 1, pp, 0, bw, bh, d_b, d_0, d_1, n, fx, fy, fwd, bck)
            got = ctx.planes_download(pp, 0)[border:border + H, border:border + W][:(H // bh) * bh, :(W // bw) * bw]
            want = oracle.build_compound_pred(b0, b1, border, W, H, bw, bh, blocks, mv0, mv1, fx, fy, fwd, bck, bd)[:(H // bh) * bh, :(W // bw) * bw]
            assert np.array_equal(got, want), (bw, bh, bd, fx, fy, fwd)
        for d in (d_b, d_0, d_1):
            ctx.free(d)
    with pytest.raises(hip.capi.AomHipError):
        ctx.build_compound_pred_batch(r0, 0, r1, 1, pp, 0, 16, 16, 1, 1, 1, 1, 0, 0, 9, 9)      # weights must sum to 16
    for p in (r0, r1, pp):
        ctx.planes_free(p)


def test_masked_compound_pred_goldens(hip, ctx):
    """Every case of ref_eval_convolve_masked.npz through aomhip_build_masked_compound_pred_batch."""
    z = np.load(os.path.join(GOLD, "ref_eval_convolve_masked.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    border = 16
    planes = {}
    for bd in (8, 10, 12):
        dt = np.uint8 if bd == 8 else np.uint16
        p0, p1 = np.ascontiguousarray(z["p%d_0" % bd], dt), np.ascontiguousarray(z["p%d_1" % bd], dt)
        H, W = p0.shape
        r0, r1, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(r0, 0, p0); ctx.planes_upload(r1, 0, p1)
        planes[bd] = (r0, r1, pp)
    n_diffwtd = 0
    for c in cases:
        r0, r1, pp = planes[c["bd"]]
        w, h = c["w"], c["h"]
        (x0, y0), (x1, y1) = c["pos"]
        (sx0, sy0), (sx1, sy1) = c["subs"]
        blk = np.zeros(1, hip.capi.search_block_dtype)
        blk["bx"], blk["by"] = x0, y0
        mv0 = np.array([[sy0, sx0]], np.int16)
        mv1 = np.array([[(y1 - y0) * 16 + sy1, (x1 - x0) * 16 + sx1]], np.int16)
        pad = np.zeros(7, np.uint8)
        mask = np.concatenate([pad, np.ascontiguousarray(z["m%d" % c["k"]]).ravel()])          # the block's mask at a byte offset
        d_b, d_0, d_1, d_m, d_o = ctx.to_device(blk), ctx.to_device(mv0), ctx.to_device(mv1), ctx.to_device(mask), ctx.to_device(np.array([7], np.uint32))
        if c.get("diffwtd"):
            # the luma form takes MVs in 1/8 pel; the fixture's phases are sixteenths, so only even-phase cases replay exactly
            if any(v % 2 for pair in c["subs"] for v in pair):
                for d in (d_b, d_0, d_1, d_m, d_o):
                    ctx.free(d)
                continue
            mv0l = np.array([[sy0 // 2, sx0 // 2]], np.int16)
            mv1l = np.array([[(y1 - y0) * 8 + sy1 // 2, (x1 - x0) * 8 + sx1 // 2]], np.int16)
            d_0l, d_1l, d_mo = ctx.to_device(mv0l), ctx.to_device(mv1l), ctx.malloc(w * h + 16)
            ctx.build_diffwtd_compound_pred_batch(r0, 0, r1, 0, pp, 0, w, h, d_b, d_0l, d_1l, 1, c["fx"], c["fy"], c["diffwtd"] - 1, d_mo)
            assert np.array_equal(ctx.from_device(d_mo, (h, w), np.uint8), z["m%d" % c["k"]]), c
            n_diffwtd += 1
            for d in (d_0l, d_1l, d_mo):
                ctx.free(d)
        else:
            ctx.build_masked_compound_pred_batch(r0, 0, r1, 0, pp, 0, w, h, d_b, d_0, d_1, 1, c["fx"], c["fy"], d_m, d_o, c["mask_stride"], c["subw"],
                                                 c["subh"], 1, 1)
        got = ctx.planes_download(pp, 0)[border + y0:border + y0 + h, border + x0:border + x0 + w]
        assert np.array_equal(got.ravel().astype(np.uint16), z["d%d" % c["k"]]), c
        for d in (d_b, d_0, d_1, d_m, d_o):
            ctx.free(d)
    assert len(cases) >= 30 and n_diffwtd >= 2
    for t in planes.values():
        for p in t:
            ctx.planes_free(p)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_diffwtd_compound_pred_vs_oracle(hip, oracle, ctx, bd):
    import ctypes as C
    rng = np.random.default_rng(95 + bd)
    W, H, border = 192, 128, 48
    ref0 = hip.synth.lcg_frame(W, H, 12, 0, bd)
    ref1 = np.clip(ref0.astype(np.int64) + rng.integers(-(40 << (bd - 8)), (40 << (bd - 8)) + 1, (H, W)), 0, (1 << bd) - 1).astype(ref0.dtype)
    ref1[:, 96:] = hip.synth.lcg_frame(W, H, 13, 1, bd)[:, 96:]          # half the frame: unrelated content -> the mask saturates at 64
    r0, r1, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(r0, 0, ref0); ctx.planes_upload(r1, 0, ref1)
    b0, b1 = oracle.extend_plane(ref0, border, r0.stride), oracle.extend_plane(ref1, border, r1.stride)
    g = oracle.lib.orc_convolve_compound_diffwtd
    g.restype = None
    for (bw, bh), mtype in (((8, 8), 0), ((16, 16), 1), ((32, 16), 0), ((8, 32), 1), ((64, 64), 0)):
        xs, ys = np.meshgrid(np.arange(0, W - bw + 1, bw), np.arange(0, H - bh + 1, bh))
        n = xs.size
        blocks = np.zeros(n, hip.capi.search_block_dtype)
        blocks["bx"], blocks["by"] = xs.ravel(), ys.ravel()
        mv0, mv1 = rng.integers(-64, 65, (n, 2)).astype(np.int16), rng.integers(-64, 65, (n, 2)).astype(np.int16)
        d_b, d_0, d_1, d_mo = ctx.to_device(blocks), ctx.to_device(mv0), ctx.to_device(mv1), ctx.malloc(n * bw * bh)
        fx, fy = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        ctx.build_diffwtd_compound_pred_batch(r0, 0, r1, 0, pp, 0, bw, bh, d_b, d_0, d_1, n, fx, fy, mtype, d_mo)
        got = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
        gmask = ctx.from_device(d_mo, (n, bh, bw), np.uint8)
        want, wm = np.zeros((H, W), ref0.dtype), np.zeros((bh, bw), np.uint8)
        for i in range(n):
            x, y = int(blocks["bx"][i]), int(blocks["by"][i])
            p = []
            for ref, mv in ((b0, mv0[i]), (b1, mv1[i])):
                px, py = (x << 4) + int(mv[1]) * 2, (y << 4) + int(mv[0]) * 2
                p.append((C.c_void_p(oracle._addr(ref, border + (py >> 4), border + (px >> 4))), ref.shape[1], px & 15, py & 15))
            g(p[0][0], p[0][1], p[0][2], p[0][3], p[1][0], p[1][1], p[1][2], p[1][3], C.c_void_p(oracle._addr(want, y, x)), W, bw, bh, fx, fy, int(bd > 8), bd,
              mtype, C.c_void_p(wm.ctypes.data))
            assert np.array_equal(gmask[i], wm), (bw, bh, bd, i)
        hh, ww = (H // bh) * bh, (W // bw) * bw
        assert np.array_equal(got[:hh, :ww], want[:hh, :ww]), (bw, bh, bd)
        assert gmask.min() >= (0 if mtype else 38) and gmask.max() <= (26 if mtype else 64)
        for d in (d_b, d_0, d_1, d_mo):
            ctx.free(d)
    for p in (r0, r1, pp):
        ctx.planes_free(p)


@pytest.mark.parametrize("bd,subw,subh", [(8, 0, 0), (10, 0, 0), (10, 1, 1), (8, 1, 0), (12, 0, 1)])
def test_masked_compound_pred_vs_oracle(hip, oracle, ctx, bd, subw, subh):
    import ctypes as C
    rng = np.random.default_rng(90 + bd + subw)
    W, H, border = 192, 128, 48
    ref0, ref1 = hip.synth.lcg_frame(W, H, 10, 0, bd), hip.synth.lcg_frame(W, H, 11, 1, bd)
    r0, r1, pp = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(r0, 0, ref0); ctx.planes_upload(r1, 0, ref1)
    b0, b1 = oracle.extend_plane(ref0, border, r0.stride), oracle.extend_plane(ref1, border, r1.stride)
    f = oracle.lib.orc_convolve_compound_mask
    f.restype = None
    for (bw, bh) in ((4, 4), (8, 8), (16, 16), (8, 16), (32, 32), (64, 64), (16, 4)):
        xs, ys = np.meshgrid(np.arange(0, W - bw + 1, bw), np.arange(0, H - bh + 1, bh))
        n = xs.size
        blocks = np.zeros(n, hip.capi.search_block_dtype)
        blocks["bx"], blocks["by"] = xs.ravel(), ys.ravel()
        lim = (border - 8) * 8
        mv0, mv1 = rng.integers(-lim, lim + 1, (n, 2)).astype(np.int16), rng.integers(-lim, lim + 1, (n, 2)).astype(np.int16)
        mw, mh = bw << subw, bh << subh
        nmask, ms = 5, mw + 3
        masks = rng.integers(0, 65, (nmask, mh, ms)).astype(np.uint8)
        masks[0], masks[1] = 64, 0
        moff = (rng.integers(0, nmask, n) * (mh * ms)).astype(np.uint32)
        d_b, d_0, d_1, d_m, d_o = ctx.to_device(blocks), ctx.to_device(mv0), ctx.to_device(mv1), ctx.to_device(masks), ctx.to_device(moff)
        fx, fy = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        ctx.build_masked_compound_pred_batch(r0, 0, r1, 0, pp, 0, bw, bh, d_b, d_0, d_1, n, fx, fy, d_m, d_o, ms, subw, subh)
        got = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
        e16 = int(bd > 8)
        want = np.zeros((H, W), ref0.dtype)
        flat = masks.reshape(-1)
        for i in range(n):
            x, y = int(blocks["bx"][i]), int(blocks["by"][i])
            p = []
            for ref, mv in ((b0, mv0[i]), (b1, mv1[i])):
                px, py = (x << 4) + int(mv[1]) * 2, (y << 4) + int(mv[0]) * 2
                p.append((C.c_void_p(oracle._addr(ref, border + (py >> 4), border + (px >> 4))), ref.shape[1], px & 15, py & 15))
            f(p[0][0], p[0][1], p[0][2], p[0][3], p[1][0], p[1][1], p[1][2], p[1][3], C.c_void_p(oracle._addr(want, y, x)), W, bw, bh, fx, fy, 0, 0, e16, bd,
              C.c_void_p(flat.ctypes.data + int(moff[i])), ms, subw, subh)
        hh, ww = (H // bh) * bh, (W // bw) * bw
        assert np.array_equal(got[:hh, :ww], want[:hh, :ww]), (bw, bh, bd, subw, subh)
        for d in (d_b, d_0, d_1, d_m, d_o):
            ctx.free(d)
    for p in (r0, r1, pp):
        ctx.planes_free(p)


def test_obmc_blend_goldens_and_batch(hip, ctx):
    """aomhip_blend_a64_1d_batch: every interpreted-reference case (one item per call, then all non-overlapping ones of a
    bit depth together), masks = av1_get_obmc_mask's tables from the fixture."""
    z = np.load(os.path.join(GOLD, "ref_eval_obmc_blend.npz"))
    cases = json.loads(bytes(z["cases"]).decode())
    K = hip.capi
    sizes = (1, 2, 4, 8, 16, 32, 64)
    table = np.concatenate([z["obmc_mask_%d" % n] for n in sizes])
    offs = {n: int(sum(s for s in sizes if s < n)) for n in sizes}
    d_masks = ctx.to_device(table)
    border = 8
    for bd in (8, 10, 12):
        dt = np.uint8 if bd == 8 else np.uint16
        pred, adj = np.ascontiguousarray(z["pred%d" % bd], dt), np.ascontiguousarray(z["adj%d" % bd], dt)
        H, W = pred.shape
        pp, pa = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(pa, 0, adj)
        mine = [c for c in cases if c["bd"] == bd]
        for c in mine:
            ctx.planes_upload(pp, 0, pred)
            it = np.zeros(1, K.blend_item_dtype)
            it[0] = (c["x"], c["y"], c["w"], c["h"], offs[c["h"] if c["vertical"] else c["w"]], c["vertical"], 0)
            d_it = ctx.to_device(it)
            ctx.blend_a64_1d_batch(pp, 0, pa, 0, d_it, 1, d_masks)
            got = ctx.planes_download(pp, 0)[border:border + H, border:border + W]
            assert np.array_equal(got[c["y"]:c["y"] + c["h"], c["x"]:c["x"] + c["w"]].astype(np.uint16), z["o%d" % c["k"]]), c
            keep = got.copy(); keep[c["y"]:c["y"] + c["h"], c["x"]:c["x"] + c["w"]] = pred[c["y"]:c["y"] + c["h"], c["x"]:c["x"] + c["w"]]
            assert np.array_equal(keep, pred)          # nothing outside the rectangle moved
            ctx.free(d_it)
        ctx.planes_free(pp); ctx.planes_free(pa)
    with pytest.raises(K.AomHipError):
        ctx.blend_a64_1d_batch(pp, 0, pa, 0, None, 1, d_masks)
    ctx.free(d_masks)
