"""Same-box A/B of experiment builds of the full-pel search kernels (tools/fps_build_exp.sh): for every library name given, a CHILD process rebinds
aomhip_full_pixel_search_batch of the ctypes binding to explib/libfps_<name>.so, runs the default search (NSTEP step_param 3, 4K 10-bit),
prints a hash of its outputs (equal hashes = bit-identical results) and, for a _prof build, prints the per-phase clocks.
  python tools/fps_ab.py [--reps 3] [--workloads nstep,tf,fp] name [name ...]        ("product" = no override)"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(name, workloads, reps):
    sys.path.insert(0, ROOT)
    import numpy as np
    import aom_av1_psy_amd as pkg
    from benchlib import common, search
    capi = pkg.capi
    exp = None
    if name != "product":
        if os.path.exists(os.path.join(ROOT, "explib", "libmcomp_%s.so" % name)):
            exp = C.CDLL(os.path.join(ROOT, "explib", "libmcomp_%s.so" % name), mode=C.RTLD_GLOBAL)
            entry = "aomhip_fullpel_diamond_batch"
        else:
            exp = C.CDLL(os.path.join(ROOT, "explib", "libfps_%s.so" % name), mode=C.RTLD_GLOBAL)
            entry = "aomhip_full_pixel_search_batch"
        f = getattr(exp, entry)
        f.restype, f.argtypes = getattr(capi.lib, entry).restype, getattr(capi.lib, entry).argtypes
        setattr(capi.lib, entry, f)
    ctx = capi.Context(0)
    out = {"lib": name}
    if "nstep" in workloads:
        wl = search.SearchPipeline(pkg, ctx, None, 0, 1)
        n = wl.n
        d_cl, d_sec = ctx.malloc(n * 20), ctx.malloc(n * 4)
        q = capi.SearchParams.make("NSTEP", 3, capi.MV_COST_L1_HDRES)
        k = [0]
        def once():
            ctx.full_pixel_search_batch(wl.src, wl.ref, k[0] % wl.F, 16, 16, q, wl.d_blocks, n, wl.d_mv, wl.d_cost, d_cl, d_sec); k[0] += 1
        common.ramp(ctx, once)
        out["nstep_ms"] = [round(common.kernel_avg_ms(ctx, once, 40), 4) for _ in range(reps)]
        k[0] = 0; once(); ctx.sync()
        res = np.concatenate([ctx.from_device(wl.d_mv, (n, 2), np.int16).ravel(), ctx.from_device(wl.d_cost, (n,), np.int32), ctx.from_device(d_cl, (n * 5,), np.int32),
                              ctx.from_device(d_sec, (n * 2,), np.int16)])
        import hashlib
        out["nstep_sha"] = hashlib.sha1(res.tobytes()).hexdigest()[:12]
        if name.endswith("_prof"):
            buf = np.zeros((n, 16), np.uint32)
            assert exp.aomhip_debug_fps_prof(C.c_void_p(buf.ctypes.data), n) == 0
            m = buf.astype(np.float64).mean(0)
            out["prof_per_block"] = {"total": m[0], "prologue": m[1], "lds_steps": m[2], "n_lds": m[3], "glob_steps": m[4], "n_glob": m[5], "var": m[6], "n_var": m[7],
                                     "cost_list": m[8], "two_batch_steps": m[9], "runs": m[10], "glob_steps_rad_le_8": m[11], "glob_steps_rad_le_18": m[12], "moves": m[13],
                                     "per_lds_step": m[2] / max(m[3], 1e-9), "per_glob_step": m[4] / max(m[5], 1e-9), "per_var": m[6] / max(m[7], 1e-9)}
        ctx.free(d_cl); ctx.free(d_sec); wl.free()
    if "diamond" in workloads:   # the inner loop's search: fullpel_diamond_kernel, DIAMOND step_param 4 (explib/libmcomp_<name>.so rebinds its entry point)
        wl = search.SearchPipeline(pkg, ctx, None, 0, 1)
        k = [0]
        def onced():
            ctx.fullpel_diamond_batch(wl.src, wl.ref, k[0] % wl.F, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, wl.d_blocks, wl.n, wl.d_mv, wl.d_cost); k[0] += 1
        common.ramp(ctx, onced)
        out["diamond_ms"] = [round(common.kernel_avg_ms(ctx, onced, 60), 4) for _ in range(reps)]
        k[0] = 0; onced(); ctx.sync()
        import hashlib
        out["diamond_sha"] = hashlib.sha1(np.concatenate([ctx.from_device(wl.d_mv, (wl.n, 2), np.int16).ravel(), ctx.from_device(wl.d_cost, (wl.n,), np.int32)]).tobytes()).hexdigest()[:12]
        wl.free()
    if "nstep32" in workloads:   # 32x32 blocks (the temporal filter's first search): NSTEP step_param 3 over the same 4K 10-bit pair
        wl = search.SearchPipeline(pkg, ctx, None, 0, 1)
        W, H = wl.W, wl.H
        gc, gr = W // 32, H // 32
        n = gc * gr
        b = np.zeros(n, capi.search_block_dtype)
        b["bx"], b["by"] = (np.arange(n) % gc) * 32, (np.arange(n) // gc) * 32
        ext = wl.BORDER - 8
        b["col_min"] = np.maximum(-(b["bx"] + ext), -1023); b["col_max"] = np.minimum(W - b["bx"] - 32 + ext, 1023)
        b["row_min"] = np.maximum(-(b["by"] + ext), -1023); b["row_max"] = np.minimum(H - b["by"] - 32 + ext, 1023)
        d_b, d_mv, d_cost = ctx.to_device(b), ctx.malloc(n * 4), ctx.malloc(n * 4)
        q = capi.SearchParams.make("NSTEP", 3, capi.MV_COST_L1_HDRES)
        k = [0]
        def once32():
            ctx.full_pixel_search_batch(wl.src, wl.ref, k[0] % wl.F, 32, 32, q, d_b, n, d_mv, d_cost); k[0] += 1
        common.ramp(ctx, once32)
        out["nstep32_ms"] = [round(common.kernel_avg_ms(ctx, once32, 40), 4) for _ in range(reps)]
        k[0] = 0; once32(); ctx.sync()
        import hashlib
        if name.endswith("_prof"):
            buf = np.zeros((n, 16), np.uint32)
            assert exp.aomhip_debug_fps_prof(C.c_void_p(buf.ctypes.data), n) == 0
            m = buf.astype(np.float64).mean(0)
            out["prof32_per_block"] = {"total": m[0], "prologue": m[1], "lds_steps": m[2], "n_lds": m[3], "glob_steps": m[4], "n_glob": m[5], "var": m[6], "n_var": m[7],
                                       "runs": m[10], "glob_rad_le_8": m[11], "glob_rad_le_18": m[12], "moves": m[13],
                                       "per_lds_step": m[2] / max(m[3], 1e-9), "per_glob_step": m[4] / max(m[5], 1e-9)}
        out["nstep32_sha"] = hashlib.sha1(np.concatenate([ctx.from_device(d_mv, (n, 2), np.int16).ravel(), ctx.from_device(d_cost, (n,), np.int32)]).tobytes()).hexdigest()[:12]
        for d in (d_b, d_mv, d_cost):
            ctx.free(d)
        wl.free()
    if "nstepmesh" in workloads:   # NSTEP + the four mesh passes on every block (run_mesh_search 1, no pruning): 16x16 and 32x32 over the same pair
        import hashlib
        for bs in (16, 32):
            wl = search.SearchPipeline(pkg, ctx, None, 0, 1)
            W, H = wl.W, wl.H
            gc, gr = W // bs, H // bs
            n = gc * gr
            b = np.zeros(n, capi.search_block_dtype)
            b["bx"], b["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
            ext = wl.BORDER - 8
            b["col_min"] = np.maximum(-(b["bx"] + ext), -1023); b["col_max"] = np.minimum(W - b["bx"] - bs + ext, 1023)
            b["row_min"] = np.maximum(-(b["by"] + ext), -1023); b["row_max"] = np.minimum(H - b["by"] - bs + ext, 1023)
            d_b, d_mv, d_cost = ctx.to_device(b), ctx.malloc(n * 4), ctx.malloc(n * 4)
            q = capi.SearchParams.make("NSTEP", 3, capi.MV_COST_L1_HDRES, run_mesh=1, mesh=[(64, 8), (28, 4), (15, 1), (7, 1)])
            k = [0]
            def oncem():
                ctx.full_pixel_search_batch(wl.src, wl.ref, k[0] % wl.F, bs, bs, q, d_b, n, d_mv, d_cost); k[0] += 1
            common.ramp(ctx, oncem)
            out["nstepmesh%d_ms" % bs] = [round(common.kernel_avg_ms(ctx, oncem, 20), 4) for _ in range(reps)]
            k[0] = 0; oncem(); ctx.sync()
            out["nstepmesh%d_sha" % bs] = hashlib.sha1(np.concatenate([ctx.from_device(d_mv, (n, 2), np.int16).ravel(), ctx.from_device(d_cost, (n,), np.int32)]).tobytes()).hexdigest()[:12]
            for d in (d_b, d_mv, d_cost):
                ctx.free(d)
            wl.free()
    if "tf" in workloads:
        from benchlib import encoder
        r = encoder.run_tf(pkg, ctx, None, 6, 2)
        out["tf"] = r and {k_: (v.get("ms_per_filtered_frame") if isinstance(v, dict) else v) for k_, v in r.items() if isinstance(v, dict) and "ms_per_filtered_frame" in v}
    if "fp" in workloads:
        r = search.run_first_pass(pkg, ctx, None, 6, 2)
        out["fp"] = r and {k_: v for k_, v in r.items() if "ms" in k_}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--workloads", default="nstep")
    ap.add_argument("--child", default=None)
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    if a.child:
        return child(a.child, a.workloads.split(","), a.reps)
    for name in a.names:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", name, "--reps", str(a.reps), "--workloads", a.workloads], capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else "FAILED %s: %s" % (name, p.stderr[-1500:]), flush=True)


if __name__ == "__main__":
    main()
