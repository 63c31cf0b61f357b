"""aomhip_bind_variance_vtable: the table layout mirrors aom_variance_fn_ptr_t (16 pointers, aom_dsp/variance.h
:84-103), the binder fills every member of the 8 / 10 / 12-bit tables with one function
per block size (no GPU needed); on the GPU box, calls THROUGH the bound pointers match the oracle (the compound /
masked / OBMC members: tests/test_gpu_compound.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import BLOCK_SIZES

FIELDS = ["sdf", "sdsf", "sdaf", "vf", "svf", "svaf", "sdx4df", "sdx3df", "sdsx4df", "msdf", "msvf", "osdf", "ovf",
          "osvf", "jsdaf", "jsvaf"]


class VTable(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in FIELDS]


def _bound(hip, bd):
    tbl = (VTable * 22)()
    for t in tbl:
        for n in FIELDS:
            setattr(t, n, 0xDEAD0000)  # sentinel = "the reference's own function"
    assert hip.capi.lib.aomhip_bind_variance_vtable(C.byref(tbl), bd) == 0
    return tbl


def test_layout_and_which_entries_are_bound(hip):
    assert C.sizeof(VTable) == 16 * C.sizeof(C.c_void_p)
    for bd, filled in ((8, set(FIELDS)), (10, set(FIELDS)), (12, set(FIELDS))):
        tbl = _bound(hip, bd)
        for t in tbl:
            for n in FIELDS:
                v = getattr(t, n)
                assert (v != 0xDEAD0000) == (n in filled), (bd, n)
        for n in filled - {"sdx3df"}:
            assert len({getattr(tbl[i], n) for i in range(22)}) == 22, n  # one function per block size
        assert all(tbl[i].sdx3df == tbl[i].sdx4df for i in range(22))  # x3d forwards to x4d (sad.c:124-129)
    assert hip.capi.lib.aomhip_bind_variance_vtable(None, 8) != 0
    assert hip.capi.lib.aomhip_bind_variance_vtable(C.byref((VTable * 22)()), 9) != 0


@pytest.mark.gpu
def test_calls_through_the_table(hip, oracle, ctx):
    rng = np.random.default_rng(2)
    tbl = _bound(hip, 8)
    SAD = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_int)
    VAR = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint))
    SVF = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_uint))
    X4D = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_void_p)
    for i, (w, h) in enumerate(BLOCK_SIZES):
        s = rng.integers(0, 256, (h, w + 3), dtype=np.uint8)
        r = rng.integers(0, 256, (h + 4, w + 9), dtype=np.uint8)
        sp, rp = s.ctypes.data + 1, r.ctypes.data + 2 * r.shape[1] + 3
        assert SAD(tbl[i].sdf)(sp, s.shape[1], rp, r.shape[1]) == oracle.sad(s, 0, 1, r, 2, 3, w, h)
        assert SAD(tbl[i].sdsf)(sp, s.shape[1], rp, r.shape[1]) == oracle.sad(s, 0, 1, r, 2, 3, w, h, skip=True)
        sse = C.c_uint()
        got = VAR(tbl[i].vf)(sp, s.shape[1], rp, r.shape[1], C.byref(sse))
        wv, wsse, _ = oracle.variance(s, 0, 1, r, 2, 3, w, h)
        assert (got, sse.value) == (wv, wsse)
        got = SVF(tbl[i].svf)(rp, r.shape[1], 3, 6, sp, s.shape[1], C.byref(sse))
        assert (got, sse.value) == oracle.sub_pixel_variance(r, 2, 3, 3, 6, s, 0, 1, w, h)
        ptrs = (C.c_void_p * 4)(rp, rp + 1, rp + r.shape[1], rp + 2 * r.shape[1] + 4)
        out = np.zeros(4, np.uint32)
        X4D(tbl[i].sdx4df)(sp, s.shape[1], ptrs, r.shape[1], out.ctypes.data)
        assert out.tolist() == [oracle.sad(s, 0, 1, r, 2, 3, w, h), oracle.sad(s, 0, 1, r, 2, 4, w, h),
                                oracle.sad(s, 0, 1, r, 3, 3, w, h), oracle.sad(s, 0, 1, r, 4, 7, w, h)]
    # 10-bit table: CONVERT_TO_BYTEPTR pointers, >> 2 wrapper folded into sdf
    tbl10 = _bound(hip, 10)
    s = rng.integers(0, 1024, (16, 16), dtype=np.uint16); r = rng.integers(0, 1024, (16, 16), dtype=np.uint16)
    assert SAD(tbl10[6].sdf)(s.ctypes.data >> 1, 16, r.ctypes.data >> 1, 16) == oracle.sad(s, 0, 0, r, 0, 0, 16, 16, bd=10)
