"""Time the diamond + bilinear sub-pel pipeline (bench.SearchPipeline) at another block size / bit depth, under rocprofv3 or plainly:
    python3 tools/gpu_ab_search_bs.py <block size> <bit depth>      (AOMHIP_LIB selects the library)"""
import sys, os
sys.path.insert(0, os.getcwd())
import aom_av1_psy_amd as pkg
from benchlib import search as bench
class P(bench.SearchPipeline):
    BS, BD = int(sys.argv[1]), int(sys.argv[2])
ctx = pkg.capi.Context(0)
wl = P(pkg, ctx, None, 0, 1, frames=2)
for _ in range(3): wl.step()
ctx.sync(); ctx.timer_begin()
for _ in range(10): wl.step()
print("bs %d bd %d: %.4f ms per frame (full-pel diamond + sub-pel), %d blocks" % (P.BS, P.BD, ctx.timer_end() / 10, wl.n))
