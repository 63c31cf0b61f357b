"""Phase timing of sad_strip_kernel (wave 0 of every workgroup, s_memtime), from a library built with -DAOMHIP_SB_PROF:
    AOMHIP_LIB=build/prof/libaomhip_prof.so python tools/gpu_sb_prof.py <4k|1080p> <bd> <frames> <sbw,sbh[,threads]> ..."""
import os, sys, json, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
import aom_av1_psy_amd as pkg
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sb_override
sb_override.apply(pkg)

def main():
    W, H = (1920, 1080) if sys.argv[1] == "1080p" else (3840, 2160)
    bd, F = int(sys.argv[2]), int(sys.argv[3])
    ctx = pkg.capi.Context(0)
    lib = pkg.capi.lib
    border = 160
    ps, pr = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    for f in range(F):
        ctx.planes_upload(ps, f, pkg.synth.lcg_frame(W, H, f, 0, bd)); ctx.planes_upload(pr, f, pkg.synth.lcg_frame(W, H, f, 1, bd))
    cands, groups = pkg.synth.mode_a_worklist(W, H, 16, seed=1, search=64)
    n = len(groups)
    d_p4, d_p1 = ctx.malloc(F * n * 16), ctx.malloc(F * n * 4)
    for spec in sys.argv[4:]:
        v = [int(x) for x in spec.split(",")]
        sbw, sbh = v[0], v[1]
        if len(v) > 2: os.environ["AOMHIP_SB_THREADS"] = str(v[2])
        else: os.environ.pop("AOMHIP_SB_THREADS", None)
        perm, off = pkg.synth.bucket_order(groups["sx"], groups["sy"], W, H, sbw, sbh)
        d_gs, d_cs, d_off = ctx.to_device(groups[perm]), ctx.to_device(cands[perm]), ctx.to_device(off)
        def go():
            ctx.sad_sb_batch(ps, pr, 0, F, 16, 16, 0, sbw, sbh, 64, len(off) - 1, d_gs, d_off, n, 0, d_p4, d_cs, d_off, n, 0, d_p1)
        go(); ctx.sync()
        buf = (C.c_ulonglong * 32)()
        lib.aomhip_debug_sb_prof(buf, 1)
        ctx.timer_begin(); go(); ms = ctx.timer_end()
        lib.aomhip_debug_sb_prof(buf, 1)
        p = list(buf)
        steps, items = max(p[4], 1), max(p[6], 1)
        lsteps = max(p[8 + 5], 1)
        print(json.dumps({"cell": spec, "ms": ms, "steps": steps, "items": items,
                          "evaluator_per_step_cycles": {"bookkeeping": p[0] / steps, "bookkeeping+evaluate": p[1] / steps, "barrier": p[3] / steps},
                          "loader_per_active_step_cycles": {"batch_of(cy+1)": p[8] / lsteps, "commit(incl. wait)": p[9] / lsteps, "wait_for_loads": p[15] / lsteps, "overflow+batch_of(cy+3)": p[10] / lsteps,
                                                            "request(+batch_of)": p[11] / lsteps, "barrier": p[12] / lsteps, "passive_overflow+barrier": p[14] / lsteps},
                          "raw_first_wavefront_per_step": [round(p[i] / steps) for i in range(4)],
                          "prologue_cycles_per_item": p[5] / items,
                          "barrier_wait_per_step_by_wavefront": [round(x / steps) for x in p[16:32]]}), flush=True)
        for d in (d_gs, d_cs, d_off): ctx.free(d)
main()
