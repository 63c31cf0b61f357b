"""The fast-butterfly bounds of the forward transform kernels (csrc/txfm_safe_max.inc): the committed table equals what the oracle's interval
analysis derives (oracle/gen_txfm_bounds.py, oracle/aomref_txfm.c bound mode), the analysis really bounds what the transform computes (the
recorded maxima dominate the largest operand / sum a sampled run of the exact network reaches), and the bounds cover video residuals."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_committed_table_is_the_analysis(oracle):
    import gen_txfm_bounds as gb
    want = gb.render(gb.table())
    have = open(os.path.join(ROOT, "aom-av1-psy_amd", "csrc", "txfm_safe_max.inc")).read()
    assert have == want


def test_bounds_cover_video_residuals(oracle):
    import gen_txfm_bounds as gb
    t = gb.table()
    # DCT_DCT of every size: 8-bit (|x| <= 255) and 10-bit (|x| <= 1023) residuals take the fast butterfly
    for ts in range(19):
        assert t[ts][0] >= 1023, (ts, t[ts][0])
    # and every existing pair admits at least 8-bit video
    for ts in range(19):
        for tt in range(16):
            if oracle.lib.orc_txfm_valid(ts, tt):
                assert t[ts][tt] >= 255, (ts, tt, t[ts][tt])


def test_analysis_is_monotone_and_scales(oracle):
    """Bound mode is interval arithmetic: maxima grow with the input bound, roughly linearly (rounding slack aside)."""
    lib = oracle.lib
    lib.orc_fwd_txfm2d_bounds.restype = None
    a, b = C.c_int64(), C.c_int64()
    for ts, tt in ((2, 0), (3, 0), (1, 3), (9, 2), (4, 0)):
        prev = (0, 0)
        for m in (1, 10, 100, 1000, 10000):
            lib.orc_fwd_txfm2d_bounds(ts, tt, m, C.byref(a), C.byref(b))
            assert a.value >= prev[0] and b.value >= prev[1]
            prev = (a.value, b.value)
        lib.orc_fwd_txfm2d_bounds(ts, tt, 1000, C.byref(a), C.byref(b))
        s1000 = b.value
        lib.orc_fwd_txfm2d_bounds(ts, tt, 2000, C.byref(a), C.byref(b))
        assert 1.9 < b.value / s1000 < 2.1
