"""The defined results of a FAILED rtcd-signature call (include/aomhip.h, error model): a losing score, in the arithmetic of the
reference's consumers.  Runs where no GPU is visible -- every rtcd-signature call then fails by construction (there is no CPU fallback)."""
import ctypes as C
import importlib

import numpy as np
import pytest

pkg = importlib.import_module("aom-av1-psy_amd")
lib = pkg.capi.lib


def _no_gpu():
    return lib.aomhip_device_count() == 0


def _i32(v):   # what `int x = fn(..)` stores in the reference's C
    return int(np.int32(np.uint32(v & 0xFFFFFFFF)))


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without a GPU: the calls must fail")
def test_failed_variance_and_sse_lose_in_int_arithmetic():
    a = np.zeros((16, 16), np.uint8)
    sse = C.c_uint(123)
    lib.aomhip_variance.restype = C.c_uint
    v = lib.aomhip_variance(a.ctypes.data_as(C.c_void_p), 16, a.ctypes.data_as(C.c_void_p), 16, 16, 16, C.byref(sse))
    lib.aomhip_status_clear()
    assert v == 0x3FFFFFFF and sse.value == 0x3FFFFFFF
    # check_better_fast (av1/encoder/mcomp.c:2441-2448): `int thismse = svf(..); cost += thismse; if (cost < *besterr) take`
    for mv_cost in (0, 1, 5000, 1 << 20, (1 << 30) - 1):
        thismse = _i32(v)
        cost = _i32(mv_cost + thismse)            # 32-bit int addition as in the reference
        assert thismse > 0 and cost > 0           # stays positive: no wrap with any realistic MV cost
        for besterr in (0, 12345, 1 << 28):       # errors of real candidates
            assert not cost < besterr             # the failed candidate never wins
    # av1_get_mvpred_sse (:3661-3677) returns `sse + mv_err_cost` as int; get_mvpred_var_cost (:645-664) likewise
    assert _i32(sse.value + 4096) > (1 << 28)
    # the value the same call would have produced with UINT32_MAX: -1, which wins every comparison -- the bug this guards against
    assert _i32(0xFFFFFFFF) + 100 < 12345


@pytest.mark.skipif(not _no_gpu(), reason="needs a box without a GPU: the calls must fail")
def test_failed_sad_loses_in_unsigned_arithmetic():
    a = np.zeros((16, 16), np.uint8)
    lib.aomhip_sad.restype = C.c_uint
    v = lib.aomhip_sad(a.ctypes.data_as(C.c_void_p), 16, a.ctypes.data_as(C.c_void_p), 16, 16, 16)
    lib.aomhip_status_clear()
    assert v == 0xFFFFFFFF   # diamond_search_sad compares `unsigned int thissad < bestsad` (mcomp.c:1350-1395)
