"""The encoder-side searches of SURVEY 8(f) (informational workloads): CDEF strength search, Wiener statistics, global-motion model error,
projection-based motion estimation, the temporal filter, the compound / OBMC searches."""
import json
import os
import sys
import time

import numpy as np

from . import common
from .common import HBM_PEAK_GBS, ROOT, kernel_avg_ms, ramp


def run_cdef_search(pkg, ctx, orc, steps, warmup):
    """The distortion table of av1_cdef_search (pickcdef.c:401-615) for a 4K 10-bit luma plane, CDEF_FULL_SEARCH (64 strength
    pairs per 64x64 filter block), one launch; also the 16-pair list of CDEF_FAST_SEARCH_LVL1-sized searches.  Informational."""
    W, H, bd, border = 3840, 2160, common.BD_OVERRIDE or 10, 64
    recon = pkg.synth.lcg_frame(W, H, 2, 0, bd)
    rng = np.random.default_rng(9)
    source = np.clip(recon.astype(np.int64) + rng.integers(-20, 21, recon.shape), 0, (1 << bd) - 1).astype(recon.dtype)
    pr, ps = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
    ctx.planes_upload(pr, 0, recon); ctx.planes_upload(ps, 0, source)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    skip = np.zeros((H // 8, W // 8), np.uint8)
    d_skip = ctx.to_device(skip)
    full = np.array([(gi // 4, (gi % 4) + (gi % 4 == 3)) for gi in range(64)], np.uint8)
    d_st, d_sse = ctx.to_device(full), ctx.malloc(8 * 64 * fbh * fbw)
    out = {"workload": "cdef_search_luma_4k_10bit", "filter_blocks": fbh * fbw}
    for name, n in (("full_search_64", 64), ("fast_search_16", 16)):
        ms = kernel_avg_ms(ctx, lambda n=n: ctx.cdef_search_sse_luma(pr, 0, ps, 0, d_st, n, d_skip, 5, fbw, d_sse), max(steps, 4))
        out[name] = {"ms_per_frame": ms, "strength_evaluations_per_s": fbh * fbw * n / ms * 1e3,
                     "filtered_pixels_per_s": float(W) * H * n / ms * 1e3}
    # exact check of one filter-block row against the oracle (4 strengths)
    sub = slice(0, 64)
    want = orc.cdef_search_sse_luma(recon[sub, :256], source[sub, :256], [tuple(int(v) for v in full[i]) for i in (0, 5, 30, 63)], skip[:8, :32], 5, bd)
    p2, s2 = ctx.planes_alloc(256, 64, border, bd, 1), ctx.planes_alloc(256, 64, border, bd, 1)
    ctx.planes_upload(p2, 0, np.ascontiguousarray(recon[sub, :256])); ctx.planes_upload(s2, 0, np.ascontiguousarray(source[sub, :256]))
    d_s4, d_o4, d_k4 = ctx.to_device(np.ascontiguousarray(full[[0, 5, 30, 63]])), ctx.malloc(8 * 4 * 4), ctx.to_device(np.zeros((8, 32), np.uint8))
    ctx.cdef_search_sse_luma(p2, 0, s2, 0, d_s4, 4, d_k4, 5, 4, d_o4)
    out["parity_sample"] = bool(np.array_equal(ctx.from_device(d_o4, (4, 1, 4), np.uint64), want))
    out["value"], out["unit"] = out["full_search_64"]["strength_evaluations_per_s"], "filter-block strength evaluations/s"
    for d in (d_skip, d_st, d_sse, d_s4, d_o4, d_k4):
        ctx.free(d)
    for p in (pr, ps, p2, s2):
        ctx.planes_free(p)
    return out


def run_wiener_stats(pkg, ctx, orc, steps, warmup):
    """av1_compute_stats for every restoration unit of a 4K luma plane (7x7 window): 8-bit with 64x64 and 256x256 units, and
    10-bit 64x64.  Informational; 1274 multiply-adds per pixel (1225 H entries + 49 M entries)."""
    import ctypes as C
    W, H, border = 3840, 2160, 16
    out = {"workload": "wiener_stats_luma_4k"}
    for name, bd, unit in (("8bit_units64", 8, 64), ("8bit_units256", 8, 256), ("10bit_units64", 10, 64)):
        dgd = pkg.synth.lcg_frame(W, H, 3, 0, bd)
        src = pkg.synth.lcg_frame(W, H, 3, 1, bd)
        pd, ps = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(pd, 0, dgd); ctx.planes_upload(ps, 0, src)
        rects = [(x, min(x + unit, W), y, min(y + unit, H)) for y in range(0, H, unit) for x in range(0, W, unit)]
        units = np.zeros(len(rects), pkg.capi.rect_dtype)
        for i, r in enumerate(rects):
            units[i] = r
        d_u, d_M, d_H = ctx.to_device(units), ctx.malloc(8 * 49 * len(rects)), ctx.malloc(8 * 2401 * len(rects))
        ms = kernel_avg_ms(ctx, lambda: ctx.compute_stats_batch(pd, 0, ps, 0, 7, d_u, None, len(rects), 0, d_M, d_H), max(steps, 3))
        out[name] = {"ms_per_frame": ms, "units": len(rects), "mac_per_s": float(W) * H * 1274 / ms * 1e3}
        if name == "8bit_units64":          # exact check of two units against the oracle
            Hm = ctx.from_device(d_H, (len(rects), 2401), np.int64)
            db, sb = orc.extend_plane(dgd, border), orc.extend_plane(src, border)
            ok = True
            f = orc.lib.orc_compute_stats
            f.restype = None
            for i in (0, len(rects) - 1):
                wm, wh = np.zeros(49, np.int64), np.zeros(2401, np.int64)
                hs, he, vs, ve = rects[i]
                f(7, C.c_void_p(orc._addr(db, border, border)), C.c_void_p(orc._addr(sb, border, border)), hs, he, vs, ve, db.shape[1], sb.shape[1], 0, 8, 0,
                  C.c_void_p(wm.ctypes.data), C.c_void_p(wh.ctypes.data))
                ok = ok and bool(np.array_equal(Hm[i], wh))
            out["parity_sample"] = ok
        for d in (d_u, d_M, d_H):
            ctx.free(d)
        ctx.planes_free(pd); ctx.planes_free(ps)
    out["value"], out["unit"] = out["8bit_units64"]["mac_per_s"], "window multiply-adds/s"
    return out


def run_warp_error(pkg, ctx, orc, steps, warmup):
    """The global-motion search's inner loop (av1_warp_error, av1/encoder/global_motion.c:128-224) on a 4K luma plane: 14 candidate models per call
    (the +step / -step pair of one parameter for 7 references' worth of candidates), every 32 x 32 tile active, 10 and 8 bits; and the baseline
    av1_segmented_frame_error.  Informational.  Algorithmic bytes per model: the reference and the current frame once each."""
    import ctypes as C
    capi = pkg.capi
    W, H, border, n_models = 3840, 2160, 32, 14
    out = {"workload": "global_motion_warp_error_luma_4k", "models_per_call": n_models}
    rng = np.random.default_rng(5)
    for name, bd in (("10bit", 10), ("8bit", 8)):
        ref = pkg.synth.lcg_frame(W, H, 3, 0, bd)
        cur = pkg.synth.lcg_frame(W, H, 3, 1, bd)
        pr, pc = ctx.planes_alloc(W, H, border, bd, 1), ctx.planes_alloc(W, H, border, bd, 1)
        ctx.planes_upload(pr, 0, ref); ctx.planes_upload(pc, 0, cur)
        models = np.zeros(n_models, capi.warp_model_dtype)
        for i in range(n_models):
            while True:
                models["mat"][i] = [rng.integers(-8 << 16, 8 << 16), rng.integers(-8 << 16, 8 << 16), (1 << 16) + rng.integers(-(1 << 10), 1 << 10),
                                    rng.integers(-(1 << 10), 1 << 10), rng.integers(-(1 << 10), 1 << 10), (1 << 16) + rng.integers(-(1 << 10), 1 << 10)]
                if capi.get_shear_params(models[i:i + 1])[0]:
                    break
        sw, sh = (W + 31) // 32, (H + 31) // 32
        seg = np.ones((sh, sw), np.uint8)
        d_m, d_s, d_e = ctx.to_device(models), ctx.to_device(seg), ctx.malloc(8 * n_models)
        once = lambda: ctx.warp_error_batch(pr, 0, pc, 0, 0, 0, d_m, n_models, 0, 0, W, H, d_s, sw, d_e)
        for _ in range(warmup):
            once()
        ms = kernel_avg_ms(ctx, once, max(steps, 3))
        es = 2 if bd > 8 else 1
        out[name] = {"ms_per_call": ms, "ms_per_model": ms / n_models, "model_pixels_per_s": float(W) * H * n_models / ms * 1e3,
                     "algorithmic_GBps": 2.0 * W * H * es * n_models / ms / 1e6}
        ms_f = kernel_avg_ms(ctx, lambda: ctx.segmented_frame_error(pr, 0, pc, 0, W, H, d_s, sw, d_e), max(steps, 3))
        out[name]["segmented_frame_error_ms"] = ms_f
        if name == "10bit":   # exact check of one model over the whole frame against the oracle
            once()
            got = ctx.from_device(d_e, (n_models,), np.int64)
            f = orc.lib.orc_warp_error
            f.restype = C.c_int64
            m = np.ascontiguousarray(models["mat"][3], np.int32)
            sh4 = np.array([models[k][3] for k in ("alpha", "beta", "gamma", "delta")], np.int16)
            rc, cc = np.ascontiguousarray(ref), np.ascontiguousarray(cur)
            t0 = time.perf_counter()
            want = f(C.c_void_p(m.ctypes.data), C.c_void_p(sh4.ctypes.data), C.c_void_p(rc.ctypes.data), 1, W, H, W, C.c_void_p(cc.ctypes.data), 0, 0, W, H, W, 0, 0, bd,
                     C.c_int64((1 << 63) - 1), C.c_void_p(seg.ctypes.data), sw)
            out["cpu_port_ms_per_model"] = (time.perf_counter() - t0) * 1e3     # the C restatement, one host core, the same frame
            out["parity_sample"] = bool(int(got[3]) == int(want))
        for d in (d_m, d_s, d_e):
            ctx.free(d)
        ctx.planes_free(pr); ctx.planes_free(pc)
    out["value"], out["unit"] = out["10bit"]["model_pixels_per_s"], "model pixels/s"
    return out


def run_int_pro(pkg, ctx, orc, steps, warmup):
    """av1_int_pro_motion_estimation (av1/encoder/mcomp.c:1897-2105) for every block of a 4K 8-bit luma plane: the 64 x 64 superblocks (the vector
    variance partitioning starts from) and all 16 x 16 / 32 x 32 blocks.  Informational."""
    import ctypes as C
    capi = pkg.capi
    W, H, border = 3840, 2160, 160
    base, _ = pkg.synth.shifted_smooth_pair(W + 64, H + 64, 0, 8)
    rng = np.random.default_rng(3)
    src = np.clip(base[32:32 + H, 32:32 + W].astype(np.int32) + rng.integers(-3, 4, (H, W)), 0, 255).astype(np.uint8)
    ref = np.clip(base[29:29 + H, 37:37 + W].astype(np.int32) + rng.integers(-3, 4, (H, W)), 0, 255).astype(np.uint8)
    ps, pr = ctx.planes_alloc(W, H, border, 8, 1), ctx.planes_alloc(W, H, border, 8, 1)
    ctx.planes_upload(ps, 0, src); ctx.planes_upload(pr, 0, ref)
    out = {"workload": "int_pro_motion_estimation_luma_4k_8bit"}
    for bs in (64, 32, 16):
        pos = [(x, y) for y in range(0, H - bs + 1, bs) for x in range(0, W - bs + 1, bs)]
        blocks = np.zeros(len(pos), capi.search_block_dtype)
        blocks["bx"], blocks["by"] = [p[0] for p in pos], [p[1] for p in pos]
        blocks["row_min"], blocks["row_max"], blocks["col_min"], blocks["col_max"] = -1023, 1023, -1023, 1023
        n = len(pos)
        d_b, d_mv, d_sad = ctx.to_device(blocks), ctx.malloc(4 * n), ctx.malloc(4 * n)
        once = lambda: ctx.int_pro_motion_estimation_batch(ps, 0, pr, 0, bs, bs, d_b, n, d_mv, d_sad)
        for _ in range(warmup):
            once()
        ms = kernel_avg_ms(ctx, once, max(steps, 3))
        out["%dx%d" % (bs, bs)] = {"ms_per_frame": ms, "blocks": n, "blocks_per_s": n / ms * 1e3}
        if bs == 64:   # a sample of blocks against the oracle, and the CPU restatement's rate on them
            mv, sad = ctx.from_device(d_mv, (n, 2), np.int16), ctx.from_device(d_sad, (n,), np.uint32)
            sb, rb = np.pad(src, border, mode="edge"), np.pad(ref, border, mode="edge")
            f = orc.lib.orc_int_pro_motion_estimation
            f.restype = C.c_uint
            lim, rm, o = np.array([-1023, 1023, -1023, 1023], np.int32), np.zeros(2, np.int16), np.zeros(2, np.int16)
            ok, t0, sample = True, time.perf_counter(), range(0, n, 17)
            for i in sample:
                off = (border + pos[i][1]) * sb.shape[1] + border + pos[i][0]
                w = f(C.c_void_p(sb.ctypes.data + off), sb.shape[1], C.c_void_p(rb.ctypes.data + off), rb.shape[1], bs, bs, 8, C.c_void_p(lim.ctypes.data),
                      C.c_void_p(rm.ctypes.data), C.c_void_p(o.ctypes.data))
                ok = ok and int(w) == int(sad[i]) and o.tolist() == mv[i].tolist()
            out["cpu_port_blocks_per_s_64x64"] = len(sample) / (time.perf_counter() - t0)
            out["parity_sample"] = bool(ok)
            out["vectors_found"] = int(len({tuple(v) for v in mv.tolist()}))
        for d in (d_b, d_mv, d_sad):
            ctx.free(d)
    # the variance tree's leaves on the same pair of planes (what follows the vector in av1_choose_var_based_partitioning)
    n8x, n8y = W // 8, (H + 7) // 8
    d_s8, d_mm, d_s4 = ctx.malloc(2 * n8x * n8y), ctx.malloc(4 * (W // 16) * ((H + 15) // 16)), ctx.malloc(2 * (W // 4) * (H // 4))
    ms8 = kernel_avg_ms(ctx, lambda: ctx.vbp_8x8_stats_plane(ps, 0, pr, 0, W, H, d_s8, n8x, d_mm, W // 16), max(steps, 3))
    ms4 = kernel_avg_ms(ctx, lambda: ctx.vbp_4x4_avg_plane(ps, 0, W, H, 0, d_s4, W // 4), max(steps, 3))
    out["vbp_leaves"] = {"ms_8x8_stats": ms8, "GBps_8x8_stats": 2.0 * W * H / ms8 / 1e6, "ms_4x4_avg": ms4, "GBps_4x4_avg": 1.0 * W * H / ms4 / 1e6}
    for d in (d_s8, d_mm, d_s4):
        ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pr)
    out["value"], out["unit"] = out["64x64"]["blocks_per_s"], "64x64 blocks/s"
    return out


def run_tf(pkg, ctx, orc, steps, warmup, width=3840, height=2160, bd=10, n_frames=5):
    """the temporal filter's motion search (tf_motion_search, temporal_filter.c:87-253) for every 32x32 block of a
    4K 10-bit frame against the 4 other frames of a 5-frame window, one aomhip_tf_motion_search_frames call per filtered frame: per
    reference frame the 32x32 NSTEP + mesh full-pel search, the 8-tap sub-pel tree, the same pair for the four 16x16 sub-blocks, the
    partition decision and the ref_mv hand-over, all in device memory."""
    capi, synth = pkg.capi, pkg.synth
    border, filt = 160, n_frames // 2
    mesh = [(64, 8), (28, 4), (15, 1), (7, 1)]   # good_quality_mesh_patterns[0] (speed_features.c:25-33)
    planes = ctx.planes_alloc(width, height, border, bd, n_frames)
    base, _ = synth.shifted_smooth_pair(width + 64, height + 64, 0, bd)
    rng = np.random.default_rng(11)
    host = []
    for f in range(n_frames):
        d = f - filt
        img = base[32 + d:32 + d + height, 32 - 2 * d:32 - 2 * d + width].astype(np.int32) + rng.integers(-3, 4, (height, width))
        host.append(np.clip(img, 0, (1 << bd) - 1).astype(np.uint16 if bd > 8 else np.uint8))
        ctx.planes_upload(planes, f, host[-1])
    blocks = capi.tf_block_list(width, height, border)
    n = len(blocks)
    d_b = ctx.to_device(blocks)
    d_mv, d_mse, d_ref = ctx.malloc(n_frames * n * 16), ctx.malloc(n_frames * n * 16), ctx.malloc(n * 4)
    out = {}
    for name, q in (("q30_mesh_pruned_when_close", 30), ("q12_mesh_always", 12)):
        tp = capi.TfParams.default(width, height, bd, q, 1, mesh)
        once = lambda: ctx.tf_motion_search_frames(planes, filt, tp, d_b, n, d_mv, d_mse, d_ref)
        for _ in range(warmup):
            once()
        out[name] = {"ms_per_filtered_frame": kernel_avg_ms(ctx, once, max(3, steps // 4))}
    # what follows the search in av1_tf_do_filtering_row, on the MVs / errors still in HBM: predictors (12-tap), pixel weights, accumulation and
    # normalisation of the whole frame in one launch (aomhip_tf_apply_frames) -- luma + 4:2:0 chroma planes of the same window
    cw, ch = (width + 1) >> 1, (height + 1) >> 1
    chroma = [ctx.planes_alloc(cw, ch, border, bd, n_frames) for _ in range(2)]
    for f in range(n_frames):
        for c in chroma:
            ctx.planes_upload(c, f, host[f][::2, ::2][:ch, :cw])
    outs = [ctx.planes_alloc(width, height, border, bd, 1)] + [ctx.planes_alloc(cw, ch, border, bd, 1) for _ in range(2)]
    ap3 = capi.TfApplyParams.make([2.0, 1.5, 1.5], 30, 5, 3, 1, 1)
    ap1 = capi.TfApplyParams.make([2.0, 0, 0], 30, 5, 1, 0, 0)
    d_diff = ctx.malloc(16)
    apply3 = lambda: ctx.tf_apply_frames([planes] + chroma, filt, ap3, n, d_mv, d_mse, outs, 0, d_diff=d_diff)
    apply1 = lambda: ctx.tf_apply_frames([planes], filt, ap1, n, d_mv, d_mse, outs[:1], 0)
    vis = width * height * (2 if bd > 8 else 1)
    for nm, fn, planes_n in (("apply_yuv420", apply3, 1.5), ("apply_luma", apply1, 1.0)):
        ms_a = kernel_avg_ms(ctx, fn, max(3, steps // 4))
        moved = vis * planes_n * (n_frames + 1)   # every window frame read once + the filtered frame written
        out[nm] = {"ms_per_filtered_frame": ms_a, "GBs_window_plus_output": moved / (ms_a * 1e-3) / 1e9, "frac_of_8TBs": moved / (ms_a * 1e-3) / 1e9 / HBM_PEAK_GBS}
    apply_ok = None
    if orc is not None:   # the luma launch against the oracle on every 61st block (blocks are independent)
        apply1(); ctx.sync()
        mvs_a = ctx.from_device(d_mv, (n_frames, n, 4, 2), np.int16)
        mses_a = ctx.from_device(d_mse, (n_frames, n, 4), np.int32)
        mb_cols = (width + 31) // 32
        fb = [orc.extend_plane(h, border, planes.stride) for h in host]
        want = orc.tf_apply_frames([fb], border, width, height, filt, mvs_a, mses_a, [2.0, 0, 0], 30, 5, bd=bd, block_first=0, block_step=61)[0]
        got = ctx.planes_download(outs[0], 0)
        okb = []
        for bi in range(0, n, 61):
            r0, c0 = border + 32 * (bi // mb_cols), border + 32 * (bi % mb_cols)
            okb.append(np.array_equal(got[r0:r0 + 32, c0:c0 + 32], want[r0:r0 + 32, c0:c0 + 32]))
        apply_ok = {"blocks_checked": len(okb), "identical": bool(all(okb))}
    out["apply_parity_sample"] = apply_ok
    for pl in chroma + outs:
        ctx.planes_free(pl)
    ctx.free(d_diff)
    ok = None
    if orc is not None:   # the last call (q 12) against the oracle on every 97th block (blocks are independent)
        mvs = ctx.from_device(d_mv, (n_frames, n, 4, 2), np.int16)
        mses = ctx.from_device(d_mse, (n_frames, n, 4), np.int32)
        idx = np.arange(0, n, 97)
        fb = [orc.extend_plane(h, border, planes.stride) for h in host]
        wmv, wmse, _ = orc.tf_motion_search_frames(fb, filt, border, orc.tf_block_list(width, height, border)[idx], orc.tf_params(width, height, bd, 12, 1, mesh),
                                                   threads=8)
        ok = bool(np.array_equal(mvs[:, idx], wmv) and np.array_equal(mses[:, idx], wmse))
    for d in (d_b, d_mv, d_mse, d_ref):
        ctx.free(d)
    ctx.planes_free(planes)
    ms = out["q30_mesh_pruned_when_close"]["ms_per_filtered_frame"]
    return dict(out, workload="tf_motion_search_4k_10bit", value=n * (n_frames - 1) / (ms * 1e-3), unit="block searches/s (32x32 block x reference frame)",
                blocks_per_frame=n, reference_frames=n_frames - 1, parity_sample=ok,
                config={"frame": "%dx%d %d-bit" % (width, height, bd), "window": n_frames, "search": "NSTEP + mesh (run_mesh_search 1, prune LVL_1), L1_HDRES; "
                        "av1_find_best_sub_pixel_tree USE_8_TAPS; 32x32 + four 16x16 per block and frame"})


def run_compound_search(pkg, ctx, orc, steps, warmup, width=3840, height=2160, bd=10, bs=16):
    """SURVEY 8(f) row 1, the RD path's compound searches of handle_newmv on every 16x16 block of a 4K 10-bit frame against two references:
    av1_joint_motion_search on both branches (8-neighbour refinement: speed >= 1; av1_full_pixel_search on the compound prediction with the second
    sub-pel start: speed 0), av1_compound_single_motion_search_interinter (masked), and the OBMC pair (av1_obmc_full_pixel_search +
    av1_find_best_obmc_sub_pixel_tree_up).  One call per frame each; ms per frame.  A sample of blocks is checked against the oracle."""
    capi, synth = pkg.capi, pkg.synth
    border = 160
    src, ref0 = synth.shifted_smooth_pair(width, height, 61, bd, shift=(2, -3), frac8=(3, 0))
    _, ref1 = synth.shifted_smooth_pair(width, height, 61, bd, shift=(-3, 2), frac8=(0, 5))
    ps, p0, p1 = (ctx.planes_alloc(width, height, border, bd, 1) for _ in range(3))
    for p_, a in ((ps, src), (p0, ref0), (p1, ref1)):
        ctx.planes_upload(p_, 0, a)
    gc, gr = width // bs, height // bs
    n = gc * gr
    rng = np.random.default_rng(5)
    blocks = np.zeros(n, capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    ext = border - 8 - 16
    blocks["col_min"], blocks["col_max"] = np.maximum(-(blocks["bx"] + ext), -1000), np.minimum(width - blocks["bx"] - bs + ext, 1000)
    blocks["row_min"], blocks["row_max"] = np.maximum(-(blocks["by"] + ext), -1000), np.minimum(height - blocks["by"] - bs + ext, 1000)
    ref_mv = rng.integers(-24, 25, (n, 2, 2)).astype(np.int16)
    cur = np.zeros((n, 2, 2), np.int16)
    cur[:, 0] = np.array([-3 * 8, 2 * 8]) + rng.integers(-20, 21, (n, 2))      # the single-reference results: a few pixels off the true motion
    cur[:, 1] = np.array([2 * 8, -3 * 8]) + rng.integers(-20, 21, (n, 2))
    mask = np.clip((np.arange(bs)[None, None, :] * 64 // bs + rng.integers(-6, 7, (n, bs, bs))), 0, 64).astype(np.uint8)
    mv_max = (1 << 14) - 1
    v = np.abs(np.arange(-mv_max, mv_max + 1))
    bits = np.where(v == 0, 0, np.floor(np.log2(np.maximum(v, 1))) + 1).astype(np.int64)
    t0, t1 = (140 + bits * 305).astype(np.int32), (165 + bits * 285 + (v & 7) * 5).astype(np.int32)
    tj = np.array([190, 660, 655, 1040], np.int32)
    d_j, d_c0, d_c1 = ctx.to_device(tj), ctx.to_device(t0), ctx.to_device(t1)
    tabs = (d_j, d_c0 + mv_max * 4, d_c1 + mv_max * 4)
    d_b, d_r, d_m = ctx.to_device(blocks), ctx.to_device(ref_mv), ctx.to_device(mask)
    d_cur = ctx.malloc(n * 8)
    d_rate, d_err = ctx.malloc(n * 4), ctx.malloc(n * 4)
    sub8 = capi.SubpelParams(2, 0, 61, 2, 1, 0, 3)      # SUBPEL_TREE, USE_8_TAPS (speed 0)
    sub4 = capi.SubpelParams(2, 0, 61, 2, 1, 0, 2)      # SUBPEL_TREE, USE_4_TAPS (speed 1 - 2)
    full = capi.SearchParams.make("NSTEP", 5, 0, 22, 61, mesh_diff_thr=4, mesh=[(64, 8), (28, 4), (15, 1), (7, 1)])
    # every call starts from the single-reference results again: a 261 KB host copy on the stream, inside the timed region (~1 % of the shortest call)
    reset = lambda: ctx.memcpy_h2d(d_cur, cur)
    out = {}

    def timed(name, fn, note):
        def once():
            reset()
            fn()
        for _ in range(max(1, warmup)):
            once()
        ms = kernel_avg_ms(ctx, once, max(3, steps // 2))
        out[name] = {"ms_per_frame": ms, "blocks_per_s": n / (ms * 1e-3), "what": note}
    timed("joint_refining_4tap", lambda: ctx.joint_motion_search_batch(ps, p0, p1, 0, bs, bs, 0, 22, sub4, 0, d_b, d_r, d_cur, None, n, d_rate, d_err, *tabs),
          "av1_joint_motion_search, disable_extensive_joint_motion_search (speed >= 1): 4 iterations of {predictor, av1_refining_search_8p_c, compound sub-pel tree USE_4_TAPS}")
    timed("joint_extensive_8tap", lambda: ctx.joint_motion_search_extensive_batch(ps, p0, p1, 0, bs, bs, full, sub8, 1, 0, d_b, d_r, d_cur, None, n, d_rate, d_err, *tabs),
          "av1_joint_motion_search, speed 0: 4 iterations of {predictor, av1_full_pixel_search(.., 5, ..) on the compound, compound sub-pel tree USE_8_TAPS twice (second MV)}")
    want = None
    if orc is not None:   # the extensive call against the oracle's composition on every 211th block
        got_mv = ctx.from_device(d_cur, (n, 2, 2), np.int16)
        got_rate, got_err = ctx.from_device(d_rate, (n,), np.int32), ctx.from_device(d_err, (n,), np.int32)
        idx = np.arange(0, n, 211)
        sb, r0b, r1b = (orc.extend_plane(a, border, ps.stride) for a in (src, ref0, ref1))
        oq = orc.search_params("NSTEP", 5, 0, 22, 61, 0, 0, 0, 4, 2147483647, 0, [(64, 8), (28, 4), (15, 1), (7, 1)], no_cost_list=1)
        w_mv, w_rate, w_err, _ = orc.joint_motion_search_batch(sb, r0b, r1b, border, width, height, bs, bs, blocks[idx], ref_mv[idx], cur[idx], None, cost_type=0,
                                                               sad_per_bit=22, sub=dict(tree=2, subpel_search_type=3, error_per_bit=61, iters_per_step=2, allow_hp=1),
                                                               mvjcost=tj, mvcost0=t0, mvcost1=t1, bd=bd, threads=8, full=oq, allow_second_mv=1)
        want = bool(np.array_equal(got_mv[idx], w_mv) and np.array_equal(got_rate[idx], w_rate) and np.array_equal(got_err[idx], w_err))
    d_this, d_other = ctx.to_device(np.ascontiguousarray(cur[:, 0])), ctx.to_device(np.ascontiguousarray(cur[:, 1]))
    d_this_w, d_ref0 = ctx.malloc(n * 4), ctx.to_device(np.ascontiguousarray(ref_mv[:, 0]))
    this0 = np.ascontiguousarray(cur[:, 0])
    reset = lambda: ctx.memcpy_h2d(d_this_w, this0)
    timed("compound_single_masked_4tap", lambda: ctx.compound_single_motion_search_batch(ps, p0, p1, 0, bs, bs, full, sub4, 0, d_b, d_ref0, d_this_w, d_other, 0, 0, None, d_m, 0,
                                                                                         n, d_rate, d_err, *tabs),
          "av1_compound_single_motion_search_interinter with a mask: predictor of the other side, av1_full_pixel_search(.., 5, ..) on the masked compound, sub-pel tree USE_4_TAPS")
    # OBMC: weighted source / mask of calc_target_weighted_pred (synthetic: top / left neighbours overlap half a block)
    om = np.full((bs, bs), 4096, np.int64)
    om[:bs // 2, :] = (np.linspace(36, 64, bs // 2).astype(np.int64)[:, None]) * 64
    om[:, :bs // 2] = np.minimum(om[:, :bs // 2], (np.linspace(34, 64, bs // 2).astype(np.int64)[None, :]) * 64)
    sblk = src[:gr * bs, :gc * bs].reshape(gr, bs, gc, bs).transpose(0, 2, 1, 3).reshape(n, bs, bs).astype(np.int64)
    nb = np.clip(sblk + rng.integers(-(10 << (bd - 8)), (10 << (bd - 8)) + 1, sblk.shape), 0, (1 << bd) - 1)
    ws = (sblk * 4096 - nb * (4096 - om[None])).astype(np.int32)
    d_ws, d_om = ctx.to_device(ws), ctx.to_device(np.broadcast_to(om.astype(np.int32), (n, bs, bs)).copy())
    ob = blocks.copy()
    ob["ref_row"], ob["ref_col"] = ref_mv[:, 0, 0], ref_mv[:, 0, 1]
    ob["start_row"], ob["start_col"] = cur[:, 0, 0] >> 3, cur[:, 0, 1] >> 3
    ob["row_min"], ob["row_max"] = np.maximum(ob["row_min"], -64), np.minimum(ob["row_max"], 64)
    ob["col_min"], ob["col_max"] = np.maximum(ob["col_min"], -64), np.minimum(ob["col_max"], 64)
    sbl = ob.copy()
    for k_ in ("start_row", "start_col", "row_min", "row_max", "col_min", "col_max"):
        sbl[k_] = ob[k_] * 8
    d_ob, d_sbl = ctx.to_device(ob), ctx.to_device(sbl)
    d_mv, d_dist, d_sse = ctx.malloc(n * 4), ctx.malloc(n * 4), ctx.malloc(n * 4)
    reset = lambda: None
    timed("obmc_full_pixel_nstep", lambda: ctx.obmc_full_pixel_search_batch(p0, 0, bs, bs, "NSTEP", 4, 0, 0, 22, 61, d_ob, n, d_ws, d_om, d_mv, d_err, *tabs),
          "av1_obmc_full_pixel_search: obmc_full_pixel_diamond, NSTEP from step_param 4")
    timed("obmc_subpel_tree_4tap", lambda: ctx.obmc_subpel_tree_batch(p0, 0, bs, bs, sub4, d_sbl, n, d_ws, d_om, d_mv, d_err, d_dist, d_sse, *tabs),
          "av1_find_best_obmc_sub_pixel_tree_up, USE_4_TAPS, from the full-pel start")
    for d in (d_j, d_c0, d_c1, d_b, d_r, d_m, d_cur, d_rate, d_err, d_this, d_other, d_this_w, d_ref0, d_ws, d_om, d_ob, d_sbl, d_mv, d_dist, d_sse):
        ctx.free(d)
    for p_ in (ps, p0, p1):
        ctx.planes_free(p_)
    ms = out["joint_refining_4tap"]["ms_per_frame"]
    return dict(out, workload="compound_search_4k_10bit", value=n / (ms * 1e-3), unit="compound blocks/s (av1_joint_motion_search, refining branch)", ms_per_frame=ms,
                blocks_per_frame=n, parity_sample_extensive=want)
