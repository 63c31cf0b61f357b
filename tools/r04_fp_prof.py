"""Phase timing of the first pass's chain kernel (build/exp/libaomhip_fpprof.so = fp_row.hip with -DAOMHIP_FP_PROF: wavefront 0 of block
row 60 sums s_memtime differences over its 240 blocks).  AOMHIP_LIB=build/exp/libaomhip_fpprof.so [AOMHIP_FP_ROW_WAVES=1] python tools/r04_fp_prof.py"""
import ctypes as C, os, sys, importlib, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module("aom-av1-psy_amd")
import bench
ctx = pkg.capi.Context(0)
r = bench.run_first_pass(pkg, ctx, None, 3, 1)
print("ms_per_frame %.3f" % r["ms_per_frame"])
out = (C.c_ulonglong * 16)()
pkg.capi.lib.aomhip_debug_fp_prof(out)
v = [float(x) for x in out]
ns, nb = max(v[12], 1), max(v[13], 1)
print("row 60, wavefront 0: %d blocks, %d searched; shader cycles per searched block (100 MHz s_memtime ticks x 24 if the counter is the constant-rate one):" % (nb, ns))
names = ["fps: source rows + setup", "fps: own run(s)", "fps: wait for the other runs", "fps: replay", "fps: second barrier", "fps: cost list", "fps: whole", "-",
         "row: list entry", "row: fps_block", "row: sse + cost", "row: decision + stores"]
for k, nme in enumerate(names):
    if nme != "-": print("  %-32s %9.0f" % (nme, v[k] / ns))
