"""Kernel A/B support: AOMHIP_SB_LIB=<explib/libsadsb_*.so> rebinds aomhip_sad_sb_batch (and the phase-clock read-out) of the loaded
binding to an experiment build of csrc/sad_sb.hip (tools/sb_build_exp.sh).  Tools only -- the package never looks at this variable."""
import ctypes as C, os


def apply(pkg):
    p = os.environ.get("AOMHIP_SB_LIB")
    if not p:
        return None
    o = C.CDLL(os.path.abspath(p), mode=C.RTLD_GLOBAL)
    lib = pkg.capi.lib
    f = o.aomhip_sad_sb_batch
    f.restype, f.argtypes = lib.aomhip_sad_sb_batch.restype, lib.aomhip_sad_sb_batch.argtypes
    lib.aomhip_sad_sb_batch = f
    if hasattr(o, "aomhip_debug_sb_prof"):
        lib.aomhip_debug_sb_prof = o.aomhip_debug_sb_prof
    return p
