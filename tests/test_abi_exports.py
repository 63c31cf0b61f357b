"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares
(no compute calls here).  Also: the product refuses to run without a device instead of
falling back to anything on the CPU."""
import ctypes
import glob
import os
import re

from conftest import ROOT


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(aomhip_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol(hip):
    lib = ctypes.CDLL(hip.capi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the Python binding declares a prototype for each of them too
    assert sorted(set(names) - set(hip.capi.EXPORTED)) == []
    # the macro-stamped rtcd names of the header (AOMHIP_DECL_QUANTIZE_B / _TX / _LPF / _CDEF): 12 + 38 + 40 + 8
    stamped = hip.capi.RTCD_STAMPED
    assert len(stamped) == 98 and not [n for n in stamped if not hasattr(lib, n)]
    src = open(os.path.join(ROOT, "include", "aomhip.h")).read()
    for frag in ("AOMHIP_DECL_QUANTIZE_B(aomhip_highbd_quantize_b_64x64_adaptive)", "AOMHIP_RTCD_TX_SIZES(AOMHIP_DECL_TX)",
                 "AOMHIP_DECL_LPF(vertical, 14)", "AOMHIP_DECL_CDEF(16, 3)"):
        assert frag in src


def test_abi_version_and_stride_rule(hip):
    lib = hip.capi.lib
    assert lib.aomhip_abi_version() == 1
    # aom_calc_y_stride (aom_scale/yv12config.h:204-206): SURVEY 8(d) config table values
    assert lib.aomhip_calc_stride(640, 160) == 960
    assert lib.aomhip_calc_stride(1920, 160) == 2240
    assert lib.aomhip_calc_stride(3840, 160) == 4160


def test_no_cpu_fallback_without_device(hip):
    lib = hip.capi.lib
    if lib.aomhip_device_count() > 0:
        return  # on the GPU box this is covered by the -m gpu tests
    h = ctypes.c_void_p()
    rc = lib.aomhip_ctx_create(0, None, ctypes.byref(h))
    assert rc == 1 and not h.value  # AOMHIP_ERR_NO_DEVICE
    assert b"no HIP device" in lib.aomhip_last_error()


def test_product_does_not_reference_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "aom-av1-psy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "aomref" not in txt, f
