"""How long does the chip take to reach its sustained clock?  ms per launch of the 4K 8-bit Mode-A strip launch in windows of 200 launches,
from an idle start.    python tools/r03_clock_ramp.py [seconds]"""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.getcwd())
import aom_av1_psy_amd as pkg

def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    W, H, bd, F = 3840, 2160, 8, 64
    ctx = pkg.capi.Context(0)
    ps, pr = ctx.planes_alloc(W, H, 160, bd, F), ctx.planes_alloc(W, H, 160, bd, F)
    for f in range(F):
        ctx.planes_upload(ps, f, pkg.synth.lcg_frame(W, H, f, 0, bd)); ctx.planes_upload(pr, f, pkg.synth.lcg_frame(W, H, f, 1, bd))
    cands, groups = pkg.synth.mode_a_worklist(W, H, 16, seed=1, search=64)
    n = len(groups)
    perm, off = pkg.synth.bucket_order(groups["sx"], groups["sy"], W, H, 320, 48)
    d_gs, d_cs, d_off = ctx.to_device(groups[perm]), ctx.to_device(cands[perm]), ctx.to_device(off)
    d_p4, d_p1 = ctx.malloc(F * n * 16), ctx.malloc(F * n * 4)
    def go():
        ctx.sad_sb_batch(ps, pr, 0, F, 16, 16, 0, 320, 48, 64, len(off) - 1, d_gs, d_off, n, 0, d_p4, d_cs, d_off, n, 0, d_p1)
    go(); ctx.sync()
    time.sleep(2.0)  # idle
    t0 = time.perf_counter()
    rows = []
    while time.perf_counter() - t0 < secs:
        ctx.timer_begin()
        for _ in range(200): go()
        ms = ctx.timer_end() / 200
        rows.append((round(time.perf_counter() - t0, 3), round(ms, 4)))
    print(json.dumps({"windows_of_200_launches": rows[:12] + rows[12::10]}))
main()
