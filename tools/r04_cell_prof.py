"""Phase timing of the cell-window diamond kernel (build/exp/libaomhip_prof*.so = mcomp.hip with -DAOMHIP_CELL_PROF: every wavefront
stores its own s_memtime differences, plain stores).  AOMHIP_LIB=build/exp/libaomhip_prof4.so python tools/r04_cell_prof.py"""
import ctypes as C, os, sys, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("aom-av1-psy_amd")
import bench
ctx = pkg.capi.Context(0)
wl = bench.SearchPipeline(pkg, ctx, None, 0, 1, frames=2)
lib = pkg.capi.lib
def once():
    ctx.fullpel_diamond_batch(wl.src, wl.ref, 0, 16, 16, 0, 4, pkg.capi.MV_COST_L1_HDRES, wl.d_blocks, wl.n, wl.d_mv, wl.d_cost)
ms = bench.kernel_avg_ms(ctx, once, 10)
ctx.sync()
out = np.zeros((wl.n, 16), np.uint32)
lib.aomhip_debug_cell_prof(out.ctypes.data_as(C.c_void_p), wl.n)
v = out.astype(np.float64)
m = v.mean(0)
print("kernel %.4f ms (with the timing stores)" % ms)
print("per wavefront (cycles): total %.0f (max %.0f) | prologue+staging %.0f | LDS rounds %.1f x %.0f = %.0f | global rounds %.1f x %.0f = %.0f | variances %.1f x %.0f = %.0f" % (
    m[8], v[:, 8].max(), m[1], m[3], m[2] / max(m[3], 1e-9), m[2], m[5], m[4] / max(m[5], 1e-9), m[4], m[7], m[6] / max(m[7], 1e-9), m[6]))
print("staging per wavefront: record+clamp %.0f | bbox+fit %.0f | loads+writes %.0f | barrier wait %.0f" % (m[9], m[10], m[11], m[12]))
print("rounds per block: LDS mean %.1f max %d, global mean %.2f max %d" % (m[3], v[:, 3].max(), m[5], v[:, 5].max()))
