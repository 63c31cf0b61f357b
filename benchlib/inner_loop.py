"""BASELINE.json configs[4] on one GPU: the 4K 10-bit encode inner loop, its per-stage timings and VALU floors, the 4:2:0 leg."""
import json
import os
import sys
import time

import numpy as np

from . import common
from .common import HBM_PEAK_GBS, ROOT, kernel_avg_ms, ramp
from .search import SearchPipeline


def latest_profile_json(suffix):
    """The newest profiles/r0N*<suffix> (rounds sort by name); {} when there is none."""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*" + suffix)))
    try:
        return json.load(open(fs[-1])) if fs else {}
    except Exception:  # noqa: BLE001
        return {}


VALU_CLASS_OPS = {"fast": ("v_add_u32", "v_mov_b32", "v_and_b32", "v_ashrrev_i32"),
                  "slow": ("v_mad_i32_i24", "v_add3_u32", "v_sad_u16", "v_dot2_i32_i16", "v_perm_b32", "v_lshl_add_u64", "v_lshlrev_b32"),
                  "trans": ("v_exp_f32",), "trans64": ("v_rcp_f64",)}


_VALU_RATES = {}


def valu_class_rates(ctx):
    """Wave-instructions per second per SIMD of each issue class, measured on this box in this run (8 wavefronts per SIMD, 8 independent
    chains each, ~4 ms per opcode after a ramp launch of the same kernel): the harmonic mean over the class's probe opcodes."""
    if _VALU_RATES:
        return _VALU_RATES
    import aom_av1_psy_amd as pkg
    names = pkg.capi.valu_issue_probe_names()
    per_op, cus, hz = {}, 256, []
    for cls, ops in VALU_CLASS_OPS.items():
        inv = []
        for op in ops:
            r = ctx.valu_issue_probe(names.index(op), 8, 300)
            iters = max(200, int(4e-3 * r["wave_insts_per_s_per_simd"] / 8 / 128))
            r = ctx.valu_issue_probe(names.index(op), 8, iters)
            per_op[op] = r["wave_insts_per_s_per_simd"]
            inv.append(1.0 / r["wave_insts_per_s_per_simd"])
            cus = r["compute_units"]
            hz.append(r["memtime_hz"])
        _VALU_RATES[cls] = len(inv) / sum(inv)
    _VALU_RATES["per_op"] = per_op
    _VALU_RATES["compute_units"] = cus
    _VALU_RATES["clock_hz_median"] = sorted(hz)[len(hz) // 2]
    _VALU_RATES["clocks_per_wave_inst"] = {c: _VALU_RATES["clock_hz_median"] / _VALU_RATES[c] for c in VALU_CLASS_OPS}
    return _VALU_RATES


def valu_seconds_per_inst(mix_entry, rates):
    """Seconds of one SIMD per wave-instruction of a kernel with this static class mix (no mix known: everything at the 4-clock rate)."""
    share = (mix_entry or {}).get("share") or {"slow": 1.0}
    return sum(v / rates[c] for c, v in share.items())


def run_inner_loop(pkg, ctx, orc, steps, warmup):
    """BASELINE.json configs[4] on one GPU: the whole 4K 10-bit encode inner loop per frame, every stage on the
    device and chained through HBM: full-pel diamond search + bilinear sub-pel refinement (16x16 blocks) ->
    motion-compensated prediction at the sub-pel MV (8-tap interpolation) -> subtract + fwd_txfm2d 16x16 + quantize_b (qindex 100) ->
    inverse transform + reconstruction -> deblocking (every 8x8 edge, level 32) -> CDEF (pri 4, sec 2, damping 6)."""
    sp = SearchPipeline(pkg, ctx, None, 0, 1, frames=2)
    W, H, bd, border, F = sp.W, sp.H, sp.BD, sp.BORDER, sp.F
    capi = pkg.capi
    pred = ctx.planes_alloc(W, H, border, bd, F)  # slot f: prediction, then reconstruction, of ring frame f
    out = ctx.planes_alloc(W, H, border, bd, 1)
    dbk = ctx.planes_alloc(W, H, border, bd, 1)
    fused_middle = os.environ.get("AOMHIP_BENCH_MIDDLE", "fused") != "three_calls"   # (three_calls: the separate predictor / transform / inverse launches)
    fused_deblock = os.environ.get("AOMHIP_BENCH_DEBLOCK", "two_pass") == "fused"   # (round 4: the two in-place passes with four lines per lane are the faster form)
    n = sp.n
    nc = 256
    d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    qp = capi.QuantParams.from_tables(orc.build_quantizer_y(bd, 100))
    params = np.zeros((H // 4, W // 4, 4), np.uint8)
    params[:, 2::2, 0] = 8; params[:, 2::2, 1] = 32; params[2::2, :, 2] = 8; params[2::2, :, 3] = 32
    d_params = ctx.to_device(params)
    fbh, fbw = (H + 63) // 64, (W + 63) // 64
    d_pri, d_sec = ctx.to_device(np.full((fbh, fbw), 4, np.uint8)), ctx.to_device(np.full((fbh, fbw), 2, np.uint8))
    d_skip = ctx.to_device(np.zeros((H // 8, W // 8), np.uint8))
    for f in range(F):
        sp.d_sub_blocks(f)
    state = {"f": 0}

    def frame(f=None):
        if f is None:
            f = state["f"] % F
            state["f"] += 1
        ctx.fullpel_diamond_batch(sp.src, sp.ref, f, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, sp.d_blocks, n, sp.d_mv, sp.d_cost)
        ctx.subpel_bilinear_batch(sp.src, sp.ref, f, 16, 16, capi.MV_COST_L1_HDRES, 2, 1, 0, sp.d_sub_blocks(f), n,
                                  sp.d_smv, sp.d_err, sp.d_dist, sp.d_sse)
        if fused_middle:
            # prediction -> residual -> transform + quantise -> inverse + add in one kernel (csrc/encode_block.hip); EIGHTTAP_REGULAR both ways
            ctx.encode_inter_blocks_batch(sp.src, f, sp.ref, f, pred, f, 16, sp.d_blocks, sp.d_smv, n, qp, d_q, d_dq, d_e, 0, 0, 0)
        else:
            ctx.build_inter_pred_batch(sp.ref, f, pred, f, 16, 16, sp.d_blocks, sp.d_smv, n, 0, 0)
            # grid mode: block i of the plane == block i of the raster list used above
            ctx.subtract_xform_quant_batch(sp.src, pred, f, 2, None, n, W // 16, 0, qp, None, d_q, d_dq, d_e)
            ctx.inv_txfm_add_batch(d_dq, 2, None, n, W // 16, 0, d_e, pred, f)
        # both deblocking passes in one launch, out of place into `dbk` (CDEF reads a second buffer anyway); AOMHIP_BENCH_DEBLOCK=two_pass
        # keeps the in-place vertical + horizontal launches
        if fused_deblock:
            ctx.deblock_plane_fused(pred, f, dbk, 0, d_params, W // 4, 0)
            ctx.cdef_luma_plane(dbk, 0, out, 0, d_pri, d_sec, fbw, d_skip, 6)
        else:
            ctx.deblock_plane(pred, f, d_params, W // 4, 0, 3)
            ctx.cdef_luma_plane(pred, f, out, 0, d_pri, d_sec, fbw, d_skip, 6)

    for _ in range(max(warmup, F)):
        frame()
    ctx.sync()
    # The frame's chain replayed as one hipGraph per ring slot (aomhip_graph_*): the eight launches then follow each other without the
    # queue's per-launch dispatch latency (AOMHIP_BENCH_GRAPH=0: enqueue them one by one; both figures are reported)
    use_graph = os.environ.get("AOMHIP_BENCH_GRAPH", "1") != "0"
    def timed(step_fn):
        step_fn(); ctx.sync()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(steps):
            step_fn()
        ev = ctx.timer_end()
        return time.perf_counter() - t0, ev
    wall_plain, ev_plain = timed(frame)
    wall, ev_ms, graph_note = wall_plain, ev_plain, None
    if use_graph:
        graphs = [ctx.capture(lambda f=f: frame(f)) for f in range(F)]
        def replay():
            f = state["f"] % F
            state["f"] += 1
            ctx.graph_launch(graphs[f])
        wall, ev_ms = timed(replay)
        graph_note = {"frames_per_s_launches_one_by_one": steps / wall_plain, "ms_per_frame_launches_one_by_one": wall_plain / steps * 1e3}
        # Between two graph launches the queue idles ~9 us (profiles/r05_inner_loop_timeline.md): the ring's F frames as ONE graph halve that
        # share per frame (an encoder submits its frames back to back; AOMHIP_BENCH_GRAPH=frame keeps one graph per frame).  Exactly `steps`
        # frames are run: steps // F ring graphs, then the remainder frame by frame.
        if os.environ.get("AOMHIP_BENCH_GRAPH", "ring") != "frame" and F > 1 and steps >= F:
            state["f"] = 0
            ring = ctx.capture(lambda: [frame(f) for f in range(F)])
            def timed_ring():
                ctx.graph_launch(ring); ctx.sync()
                t0 = time.perf_counter()
                ctx.timer_begin()
                for _ in range(steps // F):
                    ctx.graph_launch(ring)
                for f in range(steps % F):
                    ctx.graph_launch(graphs[f])
                ev = ctx.timer_end()
                return time.perf_counter() - t0, ev
            wall_frame, ev_frame = wall, ev_ms
            wall, ev_ms = timed_ring()
            state["f"] = steps % F if steps % F else F
            graph_note.update({"frames_per_s_one_graph_per_frame": steps / wall_frame, "ms_per_frame_one_graph_per_frame": wall_frame / steps * 1e3,
                               "frames_per_graph": F})
            ctx.sync()
            ctx.graph_destroy(ring)
        ctx.sync()
        for g in graphs:
            ctx.graph_destroy(g)
    # ---- the same frame with its two 4:2:0 chroma planes (8x8 chroma blocks at the luma block's MV, TX_8X8, deblock at level 32 on the 8x8
    # chroma grid's 4-sample units, CDEF chroma with the luma directions): luma chain + two chroma chains as ONE graph per ring slot
    # (tests/test_gpu_full_size.py::test_config4... checks this chain bit for bit against the oracle)
    yuv = None
    if use_graph and os.environ.get("AOMHIP_BENCH_420", "1") != "0":
        CW, CH, cbd = W // 2, H // 2, border // 2
        cs = [ctx.planes_alloc(CW, CH, cbd, bd, F) for _ in range(2)]
        cr = [ctx.planes_alloc(CW, CH, cbd, bd, F) for _ in range(2)]
        cp = [ctx.planes_alloc(CW, CH, cbd, bd, F) for _ in range(2)]
        co = ctx.planes_alloc(CW, CH, cbd, bd, 1)
        for f in range(F):
            ys, yr = ctx.planes_download(sp.src, f)[border:border + H, border:border + W], ctx.planes_download(sp.ref, f)[border:border + H, border:border + W]
            for pl, off in enumerate((200, 330)):
                ctx.planes_upload(cs[pl], f, np.clip(ys[::2, ::2].astype(np.int32) // 2 + off, 0, 1023).astype(np.uint16))
                ctx.planes_upload(cr[pl], f, np.clip(yr[::2, ::2].astype(np.int32) // 2 + off, 0, 1023).astype(np.uint16))
        cblocks = sp.h_blocks.copy()
        cblocks["bx"] //= 2; cblocks["by"] //= 2
        d_cb = ctx.to_device(cblocks)
        cparams = np.zeros((CH // 4, CW // 4, 4), np.uint8)
        cparams[:, 2::2, 0] = 6; cparams[:, 2::2, 1] = 32; cparams[2::2, :, 2] = 6; cparams[2::2, :, 3] = 32
        d_cparams = ctx.to_device(cparams)
        d_cq, d_cdq, d_ce = ctx.malloc(n * 64 * 4), ctx.malloc(n * 64 * 4), ctx.malloc(2 * n)
        d_dir, d_var = ctx.malloc((H // 8) * (W // 8)), ctx.malloc((H // 8) * (W // 8) * 4)

        def frame_420(f):
            ctx.fullpel_diamond_batch(sp.src, sp.ref, f, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, sp.d_blocks, n, sp.d_mv, sp.d_cost)
            ctx.subpel_bilinear_batch(sp.src, sp.ref, f, 16, 16, capi.MV_COST_L1_HDRES, 2, 1, 0, sp.d_sub_blocks(f), n, sp.d_smv, sp.d_err, sp.d_dist, sp.d_sse)
            ctx.encode_inter_blocks_batch(sp.src, f, sp.ref, f, pred, f, 16, sp.d_blocks, sp.d_smv, n, qp, d_q, d_dq, d_e, 0, 0, 0)
            ctx.deblock_plane(pred, f, d_params, W // 4, 0, 3)
            ctx.cdef_luma_plane(pred, f, out, 0, d_pri, d_sec, fbw, d_skip, 6, d_dir, d_var)
            for pl in range(2):
                ctx.build_inter_pred_batch(cr[pl], f, cp[pl], f, 8, 8, d_cb, sp.d_smv, n, 0, 0, 1, 1)
                ctx.subtract_xform_quant_batch(cs[pl], cp[pl], f, 1, None, n, CW // 8, 0, qp, None, d_cq, d_cdq, d_ce)
                ctx.inv_txfm_add_batch(d_cdq, 1, None, n, CW // 8, 0, d_ce, cp[pl], f)
                ctx.deblock_plane(cp[pl], f, d_cparams, CW // 4, 0, 3)
                ctx.cdef_chroma_plane(cp[pl], f, co, 0, 1, 1, d_dir, d_pri, d_sec, fbw, d_skip, 6)
        for f in range(F):
            frame_420(f)
        ctx.sync()
        ring420 = ctx.capture(lambda: [frame_420(f) for f in range(F)])
        reps = max(2, steps // F)
        ctx.graph_launch(ring420); ctx.sync()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(reps):
            ctx.graph_launch(ring420)
        ev420 = ctx.timer_end()
        wall420 = time.perf_counter() - t0
        ctx.graph_destroy(ring420)
        yuv = {"ms_per_frame": wall420 / (reps * F) * 1e3, "event_ms_per_frame": ev420 / (reps * F), "frames_per_s": reps * F / wall420,
               "chain": "the luma chain + per chroma plane: 8x8 prediction at the luma MV (ss 1, 1), subtract + fwd_txfm2d_8x8 + quantize_b, inverse + add, deblock, CDEF chroma"}
        for pl in range(2):
            for x in (cs[pl], cr[pl], cp[pl]):
                ctx.planes_free(x)
        ctx.planes_free(co)
        for d in (d_cb, d_cparams, d_cq, d_cdq, d_ce, d_dir, d_var):
            ctx.free(d)
    # sanity: the reconstruction of the last frame is close to its source (fine quantiser, converged search)
    f_last = (state["f"] - 1) % F
    rec = ctx.planes_download(out, 0)[border:border + H, border:border + W].astype(np.int32)
    srcf = ctx.planes_download(sp.src, f_last)[border:border + H, border:border + W].astype(np.int32)
    psnr = 10 * np.log10(1023.0 ** 2 / max(np.mean((rec - srcf) ** 2), 1e-9))
    # per-stage launch times (each stage alone, same inputs) and the algorithmic rate of the memory-bound ones
    # (deblock / CDEF read + write each pixel once per pass; transform stages as the txq workload)
    px_bytes = W * H * 2

    def stage_fns_of(f0):
        return {
            "fullpel_diamond": lambda: ctx.fullpel_diamond_batch(sp.src, sp.ref, f0, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, sp.d_blocks, n, sp.d_mv, sp.d_cost),
            "subpel_bilinear": lambda: ctx.subpel_bilinear_batch(sp.src, sp.ref, f0, 16, 16, capi.MV_COST_L1_HDRES, 2, 1, 0, sp.d_sub_blocks(f0), n, sp.d_smv, sp.d_err, sp.d_dist, sp.d_sse),
            "inter_pred_8tap": lambda: ctx.build_inter_pred_batch(sp.ref, f0, pred, f0, 16, 16, sp.d_blocks, sp.d_smv, n, 0, 0),
            "subtract_xform_quant_16x16": lambda: ctx.subtract_xform_quant_batch(sp.src, pred, f0, 2, None, n, W // 16, 0, qp, None, d_q, d_dq, d_e),
            "inv_txfm_add_16x16": lambda: ctx.inv_txfm_add_batch(d_dq, 2, None, n, W // 16, 0, d_e, pred, f0),
            "encode_inter_blocks_16x16": lambda: ctx.encode_inter_blocks_batch(sp.src, f0, sp.ref, f0, pred, f0, 16, sp.d_blocks, sp.d_smv, n, qp, d_q, d_dq, d_e, 0, 0, 0),
            "deblock_vert+horz": lambda: ctx.deblock_plane(pred, f0, d_params, W // 4, 0, 3),
            "deblock_fused": lambda: ctx.deblock_plane_fused(pred, f0, dbk, 0, d_params, W // 4, 0),
            "cdef_luma": lambda: ctx.cdef_luma_plane(pred, f0, out, 0, d_pri, d_sec, fbw, d_skip, 6),
        }
    stage_bytes = {"encode_inter_blocks_16x16": 3 * px_bytes + n * (256 * 8 + 2), "inter_pred_8tap": 2 * px_bytes, "subtract_xform_quant_16x16": 2 * px_bytes + n * (256 * 8 + 2),
                   "inv_txfm_add_16x16": n * 256 * 4 + 2 * px_bytes, "deblock_vert+horz": 2 * 2 * px_bytes, "deblock_fused": 2 * px_bytes, "cdef_luma": 2 * px_bytes}
    # The stages are data dependent (the search by the motion, the inverse transform by the share of blocks with coefficients) and the ring's
    # frames differ (profiles/r05_inner_loop_timeline.md: 408 vs 323 us per frame): every stage is timed on every ring slot, each slot prepared
    # by running the chain up to that stage on it, and the mean over the slots is reported (`ms_by_slot` has them all).
    stages = {}
    order = ["fullpel_diamond", "subpel_bilinear", "inter_pred_8tap", "subtract_xform_quant_16x16", "inv_txfm_add_16x16", "encode_inter_blocks_16x16",
             "deblock_vert+horz", "deblock_fused", "cdef_luma"]
    for f0 in range(F):
        fns = stage_fns_of(f0)
        for name in order:
            if name in ("deblock_fused", "encode_inter_blocks_16x16"):   # out of place / idempotent: re-running them leaves the chain's state as it is
                ms = kernel_avg_ms(ctx, fns[name], max(steps, 8))
            else:
                fns[name](); ctx.sync()     # (the chain's state for the next stage; deblock is in place: its re-runs filter an already filtered plane, same work)
                ms = kernel_avg_ms(ctx, fns[name], max(steps, 8))
                if name in ("inv_txfm_add_16x16", "deblock_vert+horz"):   # in-place stages: restore the chain before the next stage is timed
                    for nm in order[2:order.index(name) + 1]:
                        fns[nm]()
                    ctx.sync()
            stages.setdefault(name, {"ms_by_slot": []})["ms_by_slot"].append(ms)
    for name in order:
        stages[name]["ms"] = sum(stages[name]["ms_by_slot"]) / F
    eob_share = []
    for f0 in range(F):
        fns = stage_fns_of(f0)
        for nm in order[:4]:
            fns[nm]()
        ctx.sync()
        eob_share.append(float((ctx.from_device(d_e, (n,), np.uint16) > 0).mean()))
    for name in order:
        ms = stages[name]["ms"]
        if name in stage_bytes:
            # NOT an HBM figure: the whole luma chain of a 4K frame (~100 MB) lives in the 256 MiB Infinity Cache between the
            # dependent stages, so this is the rate at which the stage moves its algorithmic bytes through the cache hierarchy
            stages[name]["cache_resident_GBs"] = stage_bytes[name] / (ms * 1e-3) / 1e9
            stages[name]["cache_resident_rate_over_8TBs"] = stages[name]["cache_resident_GBs"] / HBM_PEAK_GBS
    # Each stage's own roofline: these kernels are bound by instruction issue, not by bytes.  VALU wave-instructions per launch come from the
    # committed PMC passes (profiles/r0N_inner_loop_pmc.json, tools/gpu_pmc_stages.sh: SQ_INSTS_VALU / SQ_WAVES of the same kernel x the
    # launch's wavefronts).  The issue rate is MEASURED in this run (aomhip_valu_issue_probe, csrc/probe.hip; profiles/r05_valu_issue.md):
    # a SIMD of gfx950 retires one wave64 instruction per ~2 clocks for a small "fast" class (v_add/sub_u32, v_mov, v_and/or/xor,
    # v_lshrrev, v_ashrrev, fp32 add / mul / fma) and one per ~4 clocks for every other integer / packed / dot / SAD / DPP / 64-bit opcode
    # the kernels issue; the kernel's class shares are its static opcode mix (profiles/r05_isa_mix.json, tools/isa_mix.py).
    # floor = insts x sum(share_c / rate_c) / (CUs x 4 SIMDs); valu_frac = floor / the launch time measured HERE.
    pmc_map = {"fullpel_diamond": "fullpel_diamond_kernel", "subpel_bilinear": "subpel_bilinear_kernel", "inter_pred_8tap": "inter_pred_kernel",
               "subtract_xform_quant_16x16": "xform_quant_staged_kernel", "inv_txfm_add_16x16": "inv_txfm_add_kernel", "encode_inter_blocks_16x16": "encode_inter_block_kernel",
               "deblock_vert+horz": ("deblock_vert", "deblock_horz"), "deblock_fused": "deblock_fused_kernel", "cdef_luma": "cdef_luma_kernel"}
    pmc = latest_profile_json("_inner_loop_pmc.json")
    rates = valu_class_rates(ctx)
    mix = (latest_profile_json("_isa_mix.json") or {}).get("kernels", {})
    for name, kn in pmc_map.items():
        kns = kn if isinstance(kn, tuple) else (kn,)
        ents = [next(((k_, e) for k_, e in pmc.items() if k_.startswith(x)), None) for x in kns]
        if name not in stages or any(e is None for e in ents):
            continue
        insts = sum(e["SQ_INSTS_VALU_per_wavefront"] * e["wavefronts_per_launch"] for _, e in ents)
        floor_s = sum(e["SQ_INSTS_VALU_per_wavefront"] * e["wavefronts_per_launch"] * valu_seconds_per_inst(mix.get(k_), rates) for k_, e in ents)
        st = stages[name]
        st["valu_wave_insts_per_launch"] = insts
        st["valu_floor_ms"] = floor_s / (rates["compute_units"] * 4) * 1e3
        st["valu_frac"] = st["valu_floor_ms"] / st["ms"] if st["ms"] > 0 else None
        st["valu_fast_share_static"] = [round((mix.get(k_) or {}).get("share", {}).get("fast", 0.0), 3) for k_, _ in ents]
        st["insts_per_wavefront"] = {k_.replace("SQ_INSTS_", "").replace("_per_wavefront", "").lower(): round(sum(e.get(k_, 0.0) for _, e in ents), 1)
                                     for k_ in ("SQ_INSTS_VALU_per_wavefront", "SQ_INSTS_SALU_per_wavefront", "SQ_INSTS_LDS_per_wavefront",
                                                "SQ_INSTS_VMEM_RD_per_wavefront", "SQ_INSTS_VMEM_WR_per_wavefront")}
    # blocks whose quantised coefficients are all zero skip the inverse transform (and cost the forward stage its coefficient writes only)
    for nm in ("inv_txfm_add_16x16", "subtract_xform_quant_16x16"):
        stages[nm]["eob_nonzero_share_by_slot"] = eob_share
    # BASELINE.json configs[4] asks "fps + HBM-roofline fraction": the frame's algorithmic bytes by SURVEY 8(d)'s units -- 16x16 transform blocks at
    # 10 N + 2 B, one deblocked and one CDEF-filtered pixel at 4 B each (the search has no byte unit there) -- over the frame time.  The chain is
    # bound by the search kernels' instruction issue, not by bytes: the fraction says how far from an HBM limit the frame is, nothing more.
    algo_luma = n * (10 * 256 + 2) + 2 * (4 * W * H)
    algo_420 = algo_luma + 2 * (n * (10 * 64 + 2) + 2 * (4 * (W // 2) * (H // 2)))
    roof = {"bound": "issue/latency (search kernels 2/3 of the frame)", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_frame": algo_luma,
            "achieved": algo_luma / (wall / steps) / 1e9, "frac": algo_luma / (wall / steps) / 1e9 / HBM_PEAK_GBS}
    if yuv:
        yuv["algorithmic_bytes_per_frame"] = algo_420
        yuv["roofline_frac"] = algo_420 / (yuv["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {"workload": "encode_inner_loop_4k_10bit", "value": steps / wall, "unit": "frames/s", "roofline": roof, "roofline_frac": roof["frac"], "yuv420": yuv,
            "ms_per_frame": wall / steps * 1e3, "event_ms_per_frame": ev_ms / steps, "blocks_per_frame": n,
            "recon_psnr_db_last_frame": float(psnr), "stages": stages, "valu_issue_rates": rates, "deblock_in_frame": "fused" if fused_deblock else "two_pass",
            "launch": ("one hipGraph per ring of %d frames (aomhip_graph_launch)" % graph_note["frames_per_graph"] if graph_note and "frames_per_graph" in graph_note
                       else "one hipGraph per frame (aomhip_graph_launch)" if graph_note else "eight launches per frame"), "without_graph": graph_note,
            "middle_of_frame": "one kernel (aomhip_encode_inter_blocks_batch)" if fused_middle else "three launches",
            "config": {"frame": "3840x2160 10-bit luma", "stages": "fullpel diamond + subpel bilinear (16x16) -> inter prediction at the "
                       "sub-pel MV (8-tap regular, av1_highbd_convolve_2d_sr) -> subtract+fwd_txfm2d_16x16+quantize_b q100 -> inv_txfm_add -> deblock (8x8 edges, level 32) -> "
                       "CDEF (pri 4, sec 2, damping 6)", "gpus": 1}}
