"""Launch aomhip_full_pixel_search_batch (the general kernel) with a given method / step on the 4K 10-bit search workload a few times
(for PMC comparisons against fullpel_diamond_kernel): python3 tools/gpu_fps_methods.py <METHOD> <step_param> [run_mesh]"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import aom_av1_psy_amd as pkg
from benchlib import search as bench
ctx = pkg.capi.Context(0)
wl = bench.SearchPipeline(pkg, ctx, None, 0, 1, frames=2)
q = pkg.capi.SearchParams.make(sys.argv[1], int(sys.argv[2]), pkg.capi.MV_COST_L1_HDRES, run_mesh=int(sys.argv[3]) if len(sys.argv) > 3 else 0,
                               mesh=[(64, 8), (28, 4), (15, 1), (7, 1)])
for it in range(4):
    ctx.full_pixel_search_batch(wl.src, wl.ref, it % 2, 16, 16, q, wl.d_blocks, wl.n, wl.d_mv, wl.d_cost)
    ctx.fullpel_diamond_batch(wl.src, wl.ref, it % 2, 16, 16, 0, 4, pkg.capi.MV_COST_L1_HDRES, wl.d_blocks, wl.n, wl.d_mv, wl.d_cost)
ctx.sync()
print("ok", wl.n)
