"""The general full-pel kernel (aomhip_full_pixel_search_batch) per search method / step_param / cost type on the bench's 4K 10-bit pair, next
to the lean diamond kernel: where the general kernel's time goes.  python tools/r04_fps_methods.py"""
import os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("aom-av1-psy_amd")
import bench
capi = pkg.capi
ctx = capi.Context(0)
wl = bench.SearchPipeline(pkg, ctx, None, 0, 1, frames=2)
n = wl.n
d_cl, d_sec = ctx.malloc(n * 20), ctx.malloc(n * 4)
def t(fn):
    return bench.kernel_avg_ms(ctx, fn, 10)
print("lean diamond sp4 L1_HDRES: %.4f ms" % t(lambda: ctx.fullpel_diamond_batch(wl.src, wl.ref, 0, 16, 16, 0, 4, capi.MV_COST_L1_HDRES, wl.d_blocks, n, wl.d_mv, wl.d_cost)))
for method, sp, ct, outs in (("DIAMOND", 4, capi.MV_COST_L1_HDRES, 0), ("DIAMOND", 4, capi.MV_COST_L1_HDRES, 1), ("DIAMOND", 3, capi.MV_COST_L1_HDRES, 0),
                             ("NSTEP", 3, capi.MV_COST_L1_HDRES, 0), ("NSTEP", 3, capi.MV_COST_L1_HDRES, 1), ("NSTEP", 4, capi.MV_COST_L1_HDRES, 0),
                             ("NSTEP", 3, capi.MV_COST_NONE, 0), ("NSTEP_8PT", 3, capi.MV_COST_L1_HDRES, 0), ("BIGDIA", 3, capi.MV_COST_L1_HDRES, 0),
                             ("HEX", 3, capi.MV_COST_L1_HDRES, 0)):
    q = capi.SearchParams.make(method, sp, ct)
    fn = (lambda: ctx.full_pixel_search_batch(wl.src, wl.ref, 0, 16, 16, q, wl.d_blocks, n, wl.d_mv, wl.d_cost, d_cl, d_sec)) if outs else \
         (lambda: ctx.full_pixel_search_batch(wl.src, wl.ref, 0, 16, 16, q, wl.d_blocks, n, wl.d_mv, wl.d_cost))
    print("general %-10s sp%d cost %d %s: %.4f ms" % (method, sp, ct, "cost list + second" if outs else "", t(fn)))
