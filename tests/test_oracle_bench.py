"""The CPU-baseline drivers of bench.py (oracle/aomref_bench.c): their AVX2 kernels equal the scalar oracle bit for bit,
and the timed drivers visit every candidate / block of every thread's slice (checksum independent of the thread count)."""
import ctypes as C

import numpy as np
import pytest

P = lambda a: C.c_void_p(a.ctypes.data)


def test_avx2_quantize_b_equals_scalar(oracle):
    lib = oracle.lib
    rng = np.random.default_rng(1)
    lib.orc_quantize_b_avx2.argtypes = None
    for qi in (0, 1, 20, 100, 200, 255):
        qt = oracle.build_quantizer_y(8, qi)
        q = np.array([qt[k] for k in ("zbin", "round", "quant", "quant_shift", "dequant")], np.int16)
        for ts, n, ls in ((0, 16, 0), (1, 64, 0), (2, 256, 0), (3, 1024, 1)):
            sc, isc = oracle.get_scan(ts, 0)
            sc, isc = np.ascontiguousarray(sc, np.int16), np.ascontiguousarray(isc, np.int16)
            for amp in (4, 60, 900, 20000, 70000):
                co = rng.integers(-amp, amp + 1, n).astype(np.int32)
                co[rng.integers(0, n, n // 3)] = 0
                want = oracle.quantize_b(co, q, sc, isc, ls) if hasattr(oracle, "quantize_b_raw") else None
                q1, d1, e1 = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(1, np.uint16)
                q2, d2, e2 = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(1, np.uint16)
                f = lib.orc_quantize_b
                saved = f.argtypes
                f.argtypes = None
                f(P(co), C.c_long(n), P(q[0]), P(q[1]), P(q[2]), P(q[3]), P(q1), P(d1), P(q[4]), P(e1), P(sc), P(isc), C.c_int(ls))
                f.argtypes = saved
                lib.orc_quantize_b_avx2(P(co), C.c_long(n), P(q[0]), P(q[1]), P(q[2]), P(q[3]), P(q2), P(d2), P(q[4]), P(e2), P(isc), C.c_int(ls))
                assert np.array_equal(q1, q2) and np.array_equal(d1, d2) and e1[0] == e2[0], (qi, ts, amp)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_avx2_sad16x16_equals_scalar_and_drivers_agree(oracle, bd):
    lib = oracle.lib
    rng = np.random.default_rng(bd)
    W, H, border, F = 192, 96, 80, 3
    dt = np.uint8 if bd == 8 else np.uint16
    sp = [oracle.extend_plane(rng.integers(0, 1 << bd, (H, W)).astype(dt), border) for _ in range(F)]
    rp = [oracle.extend_plane(rng.integers(0, 1 << bd, (H, W)).astype(dt), border) for _ in range(F)]
    cd = np.dtype([("sx", "<i2"), ("sy", "<i2"), ("rx", "<i2"), ("ry", "<i2")])
    gd = np.dtype([("sx", "<i2"), ("sy", "<i2"), ("rx", "<i2", (4,)), ("ry", "<i2", (4,))])
    xs, ys = np.meshgrid(np.arange(0, W - 15, 16), np.arange(0, H - 15, 16))
    n = xs.size
    c = np.zeros(n, cd); c["sx"] = c["rx"] = xs.ravel(); c["sy"] = c["ry"] = ys.ravel()
    g = np.zeros((F, n), gd); g["sx"] = c["sx"]; g["sy"] = c["sy"]
    g["rx"] = c["sx"][:, None] + rng.integers(-64, 65, (F, n, 4)); g["ry"] = c["sy"][:, None] + rng.integers(-64, 65, (F, n, 4))
    # the AVX2 kernel against the oracle's own SAD on every candidate of frame 0
    lib.orc_sad16x16_avx2.restype = C.c_uint32
    lib.orc_sad16x16_avx2.argtypes = None
    want = oracle.sad_x4d_batch(sp[0], rp[0], border, 16, 16, g[0], bd=bd)
    shift = {8: 0, 10: 2, 12: 4}[bd]
    for i in range(0, n, 7):
        for k in range(4):
            got = lib.orc_sad16x16_avx2(C.c_void_p(oracle._addr(sp[0], border + g[0]["sy"][i], border + g[0]["sx"][i])), C.c_int(sp[0].shape[1]),
                                        C.c_void_p(oracle._addr(rp[0], border + g[0]["ry"][i, k], border + g[0]["rx"][i, k])), C.c_int(rp[0].shape[1]),
                                        C.c_int(int(bd > 8)))
            assert got >> shift == want[i, k]
    # drivers: one pass, scalar vs AVX2, 1 vs 3 threads -> the same checksum = sum of all 5 * n * F SADs
    f = lib.orc_bench_sad_mode_a
    f.restype = C.c_longlong
    f.argtypes = None
    so = (C.c_void_p * F)(*[oracle._addr(p, border, border) for p in sp]); ro = (C.c_void_p * F)(*[oracle._addr(p, border, border) for p in rp])
    total = sum(int(oracle.sad_batch(sp[k], rp[k], border, 16, 16, c, bd=bd).sum()) + int(oracle.sad_x4d_batch(sp[k], rp[k], border, 16, 16, g[k], bd=bd).sum())
                for k in range(F))
    gg = np.ascontiguousarray(g.reshape(-1))
    for threads, avx2 in ((1, 0), (1, 1), (3, 0), (3, 1)):
        el, ck = C.c_double(), C.c_ulonglong()
        done = f(so, ro, C.c_int(F), C.c_int(sp[0].shape[1]), C.c_int(rp[0].shape[1]), C.c_int(int(bd > 8)), C.c_int(bd), P(c), P(gg), C.c_int(n),
                 C.c_int(threads), C.c_int(avx2), C.c_double(0.0), C.byref(el), C.byref(ck))
        assert done == 5 * n * F and ck.value == total, (threads, avx2)
