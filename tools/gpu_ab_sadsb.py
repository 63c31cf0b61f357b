"""A/B on one MI355X: direct sad_x4d + sad_cand launches vs the superblock-bucketed launch, Mode A lists."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.getcwd())
import aom_av1_psy_amd as pkg

def main():
    W, H, bd, F = (3840, 2160, 8, 64)
    if len(sys.argv) > 1 and sys.argv[1] == "1080p": W, H = 1920, 1080
    if len(sys.argv) > 2: bd = int(sys.argv[2])
    sbw, sbh = (128, 128) if bd == 8 else (128, 64)
    if len(sys.argv) > 4: sbw, sbh = int(sys.argv[3]), int(sys.argv[4])
    ctx = pkg.capi.Context(0)
    border = 160
    ps, pr = ctx.planes_alloc(W, H, border, bd, F), ctx.planes_alloc(W, H, border, bd, F)
    for f in range(F):
        ctx.planes_upload(ps, f, pkg.synth.lcg_frame(W, H, f, 0, bd)); ctx.planes_upload(pr, f, pkg.synth.lcg_frame(W, H, f, 1, bd))
    cands, groups = pkg.synth.mode_a_worklist(W, H, 16, seed=1, search=64)
    n = len(groups)
    perm, off = pkg.synth.bucket_order(groups["sx"], groups["sy"], W, H, sbw, sbh)
    d_g, d_c = ctx.to_device(groups), ctx.to_device(cands)
    d_gs, d_cs, d_off = ctx.to_device(groups[perm]), ctx.to_device(cands[perm]), ctx.to_device(off)
    d_o4, d_o1 = ctx.malloc(F * n * 16), ctx.malloc(F * n * 4)
    d_p4, d_p1 = ctx.malloc(F * n * 16), ctx.malloc(F * n * 4)
    def direct():
        ctx.sad_batch(ps, pr, 0, F, 16, 16, 0, d_c, n, 0, d_o1)
        ctx.sad_x4d_batch(ps, pr, 0, F, 16, 16, 0, d_g, n, 0, d_o4)
    def direct4():
        ctx.sad_x4d_batch(ps, pr, 0, F, 16, 16, 0, d_g, n, 0, d_o4)
    def sb_both():
        ctx.sad_sb_batch(ps, pr, 0, F, 16, 16, 0, sbw, sbh, 64, len(off) - 1, d_gs, d_off, n, 0, d_p4, d_cs, d_off, n, 0, d_p1)
    def sb_groups():
        ctx.sad_sb_batch(ps, pr, 0, F, 16, 16, 0, sbw, sbh, 64, len(off) - 1, d_gs, d_off, n, 0, d_p4)
    res = {}
    for name, fn, cand in (("direct cand+x4d", direct, 5), ("direct x4d", direct4, 4), ("sb groups+cands", sb_both, 5), ("sb groups", sb_groups, 4)):
        for _ in range(3): fn()
        ctx.sync(); ctx.timer_begin()
        for _ in range(10): fn()
        ms = ctx.timer_end() / 10
        res[name] = {"ms": ms, "cand_per_s": cand * n * F / ms * 1e3}
    a4 = ctx.from_device(d_o4, (F, n, 4), np.uint32)[:, perm]; b4 = ctx.from_device(d_p4, (F, n, 4), np.uint32)
    a1 = ctx.from_device(d_o1, (F, n), np.uint32)[:, perm]; b1 = ctx.from_device(d_p1, (F, n), np.uint32)
    res["identical"] = bool(np.array_equal(a4, b4) and np.array_equal(a1, b1))
    print(json.dumps({"frame": [W, H, bd], "cell": [sbw, sbh], **res}))

main()
