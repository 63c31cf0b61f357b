"""A/B on one MI355X: direct sad_x4d + sad_cand launches vs the strip-walking bucketed launch, Mode A lists.
    python tools/gpu_ab_sadsb.py <4k|1080p> <bit depth> <frames> <sbw,sbh[,threads]> [<sbw,sbh[,threads]> ...]
AB_RANGE=<px> (default 64): the search range of the lists AND the kernel's range contract (the LDS window's halo)."""
import os, sys, json
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
    border = 160
    ps, pr = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    data = os.environ.get("AB_DATA", "lcg")  # lcg: the bench's frames; zero / smooth: the same launch on low-toggle data (is the chip power limited?)
    def frame(f, plane):
        if data == "zero": return np.zeros((H, W), np.uint16 if bd > 8 else np.uint8)
        if data == "smooth":
            g = (np.add.outer(np.arange(H), np.arange(W)) // 8 + 3 * f + plane) % (1 << bd)
            return g.astype(np.uint16 if bd > 8 else np.uint8)
        return pkg.synth.lcg_frame(W, H, f, plane, bd)
    for f in range(F):
        ctx.planes_upload(ps, f, frame(f, 0)); ctx.planes_upload(pr, f, frame(f, 1))
    RANGE = int(os.environ.get("AB_RANGE", "64"))
    cands, groups = pkg.synth.mode_a_worklist(W, H, 16, seed=1, search=RANGE)
    n = len(groups)
    d_g, d_c = ctx.to_device(groups), ctx.to_device(cands)
    d_o4, d_o1 = ctx.malloc(F * n * 16), ctx.malloc(F * n * 4)
    d_p4, d_p1 = ctx.malloc(F * n * 16), ctx.malloc(F * n * 4)
    def timed(fn, reps=int(os.environ.get("AB_REPS", "10"))):
        for _ in range(3): fn()
        ctx.sync(); ctx.timer_begin()
        for _ in range(reps): fn()
        return ctx.timer_end() / reps
    def direct():
        ctx.sad_batch(ps, pr, 0, F, 16, 16, 0, d_c, n, 0, d_o1)
        ctx.sad_x4d_batch(ps, pr, 0, F, 16, 16, 0, d_g, n, 0, d_o4)
    ms = timed(direct)
    compulsory = F * (2 * W * H * (2 if bd > 8 else 1) + n * 20)
    print(json.dumps({"frame": [W, H, bd, F], "direct_ms": ms, "cand_per_s": 5 * n * F / ms * 1e3}), flush=True)
    a4 = ctx.from_device(d_o4, (F, n, 4), np.uint32); a1 = ctx.from_device(d_o1, (F, n), np.uint32)
    for spec in sys.argv[4:]:
        v = [int(x) for x in spec.split(",")]
        sbw, sbh = v[0], v[1]
        if len(v) > 2: os.environ["AOMHIP_SB_THREADS"] = str(v[2])
        else: os.environ.pop("AOMHIP_SB_THREADS", None)
        perm, off = pkg.synth.bucket_order(groups["sx"], groups["sy"], W, H, sbw, sbh)
        d_gs, d_cs, d_off = ctx.to_device(groups[perm]), ctx.to_device(cands[perm]), ctx.to_device(off)
        ctx.memset(d_p4, 0xff, F * n * 16); ctx.memset(d_p1, 0xff, F * n * 4)
        def sb_both():
            ctx.sad_sb_batch(ps, pr, 0, F, 16, 16, 0, sbw, sbh, RANGE, len(off) - 1, d_gs, d_off, n, 0, d_p4, d_cs, d_off, n, 0, d_p1)
        try:
            ms = timed(sb_both)
        except Exception as e:
            print(json.dumps({"cell": spec, "error": str(e)}), flush=True)
            continue
        b4 = ctx.from_device(d_p4, (F, n, 4), np.uint32); b1 = ctx.from_device(d_p1, (F, n), np.uint32)
        ok = bool(np.array_equal(a4[:, perm], b4) and np.array_equal(a1[:, perm], b1))
        print(json.dumps({"cell": spec, "range": RANGE, "ms": ms, "cand_per_s": 5 * n * F / ms * 1e3, "compulsory_GBs": compulsory / ms / 1e6,
                          "frac_of_8TBs": compulsory / ms / 1e6 / 8000, "identical": ok}), flush=True)
        for d in (d_gs, d_cs, d_off): ctx.free(d)

main()
