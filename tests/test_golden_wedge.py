"""The oracle's wedge-mask helpers (orc_wedge_sse_from_residuals / _sign_from_residuals / _compute_delta_squares, oracle/aomref_rdhelp.c) against the
reference's own av1_wedge_*_c (av1/encoder/wedge_utils.c:52-125) interpreted where they lie: tests/golden/ref_eval_wedge.npz
(tests/golden/gen_ref_eval_wedge.py), 30 cases x (1 SSE + 1 delta-squares array + 5 sign decisions), bit for bit."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_wedge.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def bind(oracle):
    lib = oracle.lib
    lib.orc_wedge_sse_from_residuals.restype = C.c_uint64
    lib.orc_wedge_sse_from_residuals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_wedge_sign_from_residuals.restype = C.c_int
    lib.orc_wedge_sign_from_residuals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64]
    lib.orc_wedge_compute_delta_squares.restype = None
    lib.orc_wedge_compute_delta_squares.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return lib


def test_oracle_wedge_helpers_reproduce_the_interpreted_reference(oracle):
    z, cases = load()
    lib = bind(oracle)
    assert len(cases) == 30
    clamped = 0
    for c in cases:
        k, n = c["k"], c["N"]
        r1, d, m = (np.ascontiguousarray(z["%s_%d" % (s, k)]) for s in ("r1", "d", "m"))
        assert lib.orc_wedge_sse_from_residuals(r1.ctypes.data, d.ctypes.data, m.ctypes.data, n) == c["sse"], c
        a, b = np.ascontiguousarray(z["a_%d" % k]), np.ascontiguousarray(z["b_%d" % k])
        ds = np.zeros(n, np.int16)
        lib.orc_wedge_compute_delta_squares(ds.ctypes.data, a.ctypes.data, b.ctypes.data, n)
        assert np.array_equal(ds, z["ds_%d" % k]), c
        clamped += int((np.abs(ds.astype(np.int32)) >= 32767).sum())
        for limit, want in zip(c["limits"], c["signs"]):
            assert lib.orc_wedge_sign_from_residuals(ds.ctypes.data, m.ctypes.data, n, limit) == want, (c, limit)
        assert c["signs"][:3] == [1, 0, 0]      # acc > acc - 1, not > acc, not > acc + 1: the comparison is strict
    assert clamped > 1000                       # the int16 saturation of the delta squares is exercised
