"""Phase clocks of the up-sampled sub-pel error in the temporal filter's search (16-bit planes): run with the profiling library,
    AOMHIP_LIB=explib/libaomhip_upsprof.so python tools/ups_prof.py
prints, per sub-pel launch size, the shader clocks per wavefront spent in the horizontal pass, the vertical pass, whole error calls and the whole kernel."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib
pkg = importlib.import_module("aom-av1-psy_amd")
from benchlib import search, common
capi = pkg.capi
ctx = capi.Context(0)
lib = C.CDLL(capi.LIB_PATH)
import numpy as np
buf = np.zeros((40960, 8), np.uint64)
bufp = C.c_void_p(buf.ctypes.data)
for bs in (16, 32):
    class P(search.SearchPipeline):
        BS, BD = bs, 10
    wl = P(pkg, ctx, None, 0, 1, frames=2)
    sp8 = capi.SubpelParams(2, capi.MV_COST_NONE, 0, 2, 1, 0, 3)
    for f in range(wl.F):
        wl.d_sub_blocks(f)
    k = [0]
    def once():
        ctx.subpel_tree_batch(wl.src, wl.ref, k[0] % wl.F, bs, bs, sp8, wl.d_sub_blocks(k[0] % wl.F), wl.n, wl.d_smv, wl.d_err, wl.d_dist, wl.d_sse); k[0] += 1
    common.ramp(ctx, once, 0.1)
    ms = common.kernel_avg_ms(ctx, once, 10)
    lib.aomhip_debug_ups_prof(bufp)
    for _ in range(4):
        once()
    ctx.sync()
    assert lib.aomhip_debug_ups_prof(bufp) == 0
    v = [float(x) for x in buf.astype(np.float64).sum(0)]
    w = max(v[6], 1)
    print(json.dumps({"bs": bs, "blocks": wl.n, "ms_per_launch_with_clocks": round(ms, 4), "clocks_per_wave": {"kernel": v[5] / w, "error_calls": v[3] / w, "horizontal": v[0] / w,
                      "vertical": v[1] / w, "prologue": v[7] / w}, "calls_per_wave": v[4] / w, "strips_per_wave": v[2] / w,
                      "clocks_per_strip": {"horizontal": v[0] / max(v[2], 1), "vertical": v[1] / max(v[2], 1)}}))
    wl.free()
