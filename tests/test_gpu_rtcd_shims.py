"""The rtcd-signature conformance surface beyond SAD / variance (include/aomhip.h "the rest of the rtcd surface"): each
entry point is driven through the C ABI on host buffers, with the input classes of the reference's own unit tests, and must
equal the oracle (which tests/test_golden_ref_eval.py pins to the interpreted reference) bit for bit.
  quantize_b*          test/quantize_func_test.cc:202-259 (zero, DC only, extreme DC, constant, random spans; qindex sweep)
  lpf_*                test/lpf_test.cc:177-278 (random thresholds, pixel lines that make every mask / flat path fire)
  cdef_*               test/cdef_test.cc:408-436 (random depth-limited input, strengths, dampings, very-large borders)
  fwd / inv txfm2d     test/av1_fwd_txfm2d_test.cc, av1_inv_txfm2d_test.cc (every size x valid type)
also: the installer table and the sticky status."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = lambda a: C.c_void_p(a.ctypes.data)
TX = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (4, 8), (8, 4), (8, 16), (16, 8), (16, 32), (32, 16), (32, 64), (64, 32), (4, 16), (16, 4),
      (8, 32), (32, 8), (16, 64), (64, 16)]


def _fn(lib, name, restype=None):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = None
    return f


@pytest.mark.parametrize("hbd", [False, True])
@pytest.mark.parametrize("adaptive", [False, True])
def test_quantize_b_family(hip, oracle, hbd, adaptive):
    lib = hip.capi.lib
    rng = np.random.default_rng(7 + hbd + 2 * adaptive)
    for tx_size, suffix, ls in ((0, "", 0), (1, "", 0), (2, "", 0), (7, "", 0), (3, "_32x32", 1), (9, "_32x32", 1), (4, "_64x64", 2)):
        name = "aomhip_%squantize_b%s%s" % ("highbd_" if hbd else "", suffix, "_adaptive" if adaptive else "")
        f = _fn(lib, name)
        sc, isc = oracle.get_scan(tx_size, 0)
        n = len(sc)
        for qindex in (0, 1, 20, 100, 200, 255):
            q = oracle.build_quantizer_y(10 if hbd else 8, qindex)
            tabs = {k: np.ascontiguousarray(v, np.int16) for k, v in q.items()}
            classes = [np.zeros(n, np.int32)]
            dc = np.zeros(n, np.int32); dc[0] = 300; classes.append(dc)
            ex = np.zeros(n, np.int32); ex[0] = -8191; classes.append(ex)
            classes.append(np.full(n, 16, np.int32))
            for span in (32, 1024, 8191 if not hbd else 200000):
                classes.append(rng.integers(-span, span + 1, n).astype(np.int32))
            lone = np.zeros(n, np.int32); lone[int(sc[min(5, n - 1)])] = int(q["dequant"][1]) // (1 << ls) + 1; classes.append(lone)
            for co in classes:
                qc, dq, eob = np.full(n, 77, np.int32), np.full(n, 77, np.int32), C.c_uint16(9)
                f(P(co), C.c_ssize_t(n), P(tabs["zbin"]), P(tabs["round"]), P(tabs["quant"]), P(tabs["quant_shift"]), P(qc), P(dq),
                  P(tabs["dequant"]), C.byref(eob), P(sc), P(isc))
                if adaptive:
                    wq, wd, we = oracle.quantize_b_adaptive(co, q, sc, ls, highbd=hbd)
                else:
                    wq, wd, we = oracle.quantize_b(co, q, sc, isc, ls, highbd=hbd)
                assert np.array_equal(qc, wq) and np.array_equal(dq, wd) and eob.value == we, (name, tx_size, qindex)


def test_fwd_and_inv_txfm2d_every_size_and_type(hip, oracle):
    lib = hip.capi.lib
    rng = np.random.default_rng(3)
    for tx_size, (w, h) in enumerate(TX):
        ffwd, finv = _fn(lib, "aomhip_fwd_txfm2d_%dx%d" % (w, h)), _fn(lib, "aomhip_inv_txfm2d_add_%dx%d" % (w, h))
        nc = min(w, 32) * min(h, 32)
        for tx_type in range(16):
            if not oracle.av1_tx_valid(tx_size, tx_type):
                continue
            for bd in (8, 10, 12):
                stride = w + 5
                res = rng.integers(-(1 << bd) + 1, 1 << bd, (h, stride)).astype(np.int16)
                out = np.full(w * h, 0x5a5a5a5a, np.int32)
                ffwd(P(res), P(out), C.c_int(stride), C.c_int(tx_type), C.c_int(bd))
                want = oracle.fwd_txfm2d(np.ascontiguousarray(res[:, :w]), tx_size, tx_type, bd)
                assert np.array_equal(out[:nc], want[:nc]), ("fwd", w, h, tx_type, bd)
                # inverse: coefficients of a coarse quantisation of those, added to random pixels
                co = np.zeros(w * h, np.int32); co[:nc] = (want[:nc] // 8) * 8
                dst = rng.integers(0, 1 << bd, (h, stride)).astype(np.uint16)
                got = dst.copy()
                finv(P(co), P(got), C.c_int(stride), C.c_int(tx_type), C.c_int(bd))
                wantd = dst.copy()
                wantd[:, :w] = oracle.inv_txfm2d_add(co, np.ascontiguousarray(dst[:, :w]), tx_size, tx_type, bd)
                assert np.array_equal(got, wantd), ("inv", w, h, tx_type, bd)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_subtract_block(hip, bd):
    lib = hip.capi.lib
    rng = np.random.default_rng(bd)
    f = _fn(lib, "aomhip_subtract_block" if bd == 8 else "aomhip_highbd_subtract_block")
    dt = np.uint8 if bd == 8 else np.uint16
    for rows, cols in ((4, 4), (8, 16), (32, 8), (64, 64), (128, 128), (16, 64)):
        src = rng.integers(0, 1 << bd, (rows + 2, cols + 7)).astype(dt); pred = rng.integers(0, 1 << bd, (rows + 1, cols + 3)).astype(dt)
        diff = np.full((rows, cols + 2), 12345, np.int16)
        sp, pp = src.ctypes.data, pred.ctypes.data
        if bd > 8:  # CONVERT_TO_BYTEPTR (aom_ports/mem.h:79-80)
            sp, pp = sp >> 1, pp >> 1
        f(C.c_int(rows), C.c_int(cols), P(diff), C.c_ssize_t(diff.shape[1]), C.c_void_p(sp), C.c_ssize_t(src.shape[1]), C.c_void_p(pp),
          C.c_ssize_t(pred.shape[1]))
        assert np.array_equal(diff[:, :cols], src[:rows, :cols].astype(np.int32) - pred[:rows, :cols].astype(np.int32))
        assert np.all(diff[:, cols:] == 12345)


def _lpf_patch(rng, bd, flat):
    """16 x 16 pixels around an edge at row / column 8: random, or nearly flat so that the flat / flat2 masks fire."""
    mx = (1 << bd) - 1
    if flat:
        base = int(rng.integers(8 << (bd - 8), mx - (8 << (bd - 8))))
        p = base + rng.integers(-(1 << (bd - 8)), (1 << (bd - 8)) + 1, (32, 32))
        p[16:] += int(rng.integers(-3, 4)) << (bd - 8)
    else:
        p = rng.integers(0, mx + 1, (32, 32))
    return np.clip(p, 0, mx).astype(np.uint8 if bd == 8 else np.uint16)


@pytest.mark.parametrize("bd", [8, 10, 12])
def test_lpf_family(hip, oracle, bd):
    lib = hip.capi.lib
    rng = np.random.default_rng(40 + bd)
    hb = "highbd_" if bd > 8 else ""
    for direction, vertical in (("horizontal", 0), ("vertical", 1)):
        for length in (4, 6, 8, 14):
            for kind, count in (("", 4), ("_dual", 8)) + ((("_quad", 16),) if bd == 8 else ()):
                f = _fn(lib, "aomhip_%slpf_%s_%d%s" % (hb, direction, length, kind))
                for trial in range(24):
                    thr = [np.full(16, int(v), np.uint8) for v in (rng.integers(0, 3 * 63 + 5), rng.integers(0, 64), rng.integers(0, 16),
                                                                   rng.integers(0, 3 * 63 + 5), rng.integers(0, 64), rng.integers(0, 16))]
                    px = _lpf_patch(rng, bd, flat=trial % 3 != 0)
                    got, want = px.copy(), px.copy()
                    y, x = 16, 16
                    # the transposed patch makes the same pixel lines a vertical edge
                    s = got.ctypes.data + (y * 32 + x) * got.itemsize
                    args = [C.c_void_p(s), C.c_int(32), P(thr[0]), P(thr[1]), P(thr[2])]
                    if kind == "_dual":
                        args += [P(thr[3]), P(thr[4]), P(thr[5])]
                    if bd > 8:
                        args.append(C.c_int(bd))
                    f(*args)
                    for unit in range(count // 4):
                        t = thr[3:] if (kind == "_dual" and unit == 1) else thr[:3]
                        oy, ox = (y + 4 * unit, x) if vertical else (y, x + 4 * unit)
                        oracle.lpf_edge(want, oy, ox, vertical, length, int(t[0][0]), int(t[1][0]), int(t[2][0]), bd=bd)
                    assert np.array_equal(got, want), (direction, length, kind, bd, trial)


def test_cdef_find_dir_and_filters(hip, oracle):
    lib = hip.capi.lib
    rng = np.random.default_rng(11)
    fdir = _fn(lib, "aomhip_cdef_find_dir", C.c_int)
    fdual = _fn(lib, "aomhip_cdef_find_dir_dual")
    for bd in (8, 10, 12):
        for trial in range(12):
            img = rng.integers(0, 1 << bd, (8, 24)).astype(np.uint16)
            if trial % 3 == 0:  # a directional ramp
                yy, xx = np.mgrid[0:8, 0:24]
                img = np.clip(((yy * (trial % 5) + xx * 3) % 64) << (bd - 6), 0, (1 << bd) - 1).astype(np.uint16)
            var = C.c_int32(-1)
            d = fdir(P(img), C.c_int(24), C.byref(var), C.c_int(bd - 8))
            assert (d, var.value) == oracle.cdef_find_dir(np.ascontiguousarray(img[:, :8]), bd - 8)
            v1, v2, o1, o2 = C.c_int32(), C.c_int32(), C.c_int(), C.c_int()
            fdual(P(img), C.c_void_p(img.ctypes.data + 16), C.c_int(24), C.byref(v1), C.byref(v2), C.c_int(bd - 8), C.byref(o1), C.byref(o2))
            assert (o1.value, v1.value) == oracle.cdef_find_dir(np.ascontiguousarray(img[:, :8]), bd - 8)
            assert (o2.value, v2.value) == oracle.cdef_find_dir(np.ascontiguousarray(img[:, 8:16]), bd - 8)
    olib = oracle.lib
    olib.orc_cdef_filter_block.restype = None
    olib.orc_cdef_filter_block.argtypes = None
    for bits, is16 in ((8, 0), (16, 1)):
        for variant in range(4):
            f = _fn(lib, "aomhip_cdef_filter_%d_%d" % (bits, variant))
            for trial in range(20):
                bd = 8 if not is16 else int(rng.choice([10, 12]))
                bw, bh = [(8, 8), (4, 4), (8, 4), (4, 8)][trial % 4]
                buf = rng.integers(0, 1 << bd, (16, 144)).astype(np.uint16)
                if trial % 2:
                    buf[:, :3] = 0x4000; buf[:2, :] = 0x4000  # CDEF_VERY_LARGE outside the frame
                pri = int(rng.integers(0, 16)) << (bd - 8); sec = int(rng.choice([0, 1, 2, 4])) << (bd - 8)
                dirn, pd, sd = int(rng.integers(0, 8)), int(rng.integers(3, 7)) + (bd - 8), int(rng.integers(3, 7)) + (bd - 8)
                origin = buf.ctypes.data + (3 * 144 + 4) * 2
                got = np.full((bh, 11), 99, np.uint16 if is16 else np.uint8); want = got.copy()
                f(P(got), C.c_int(11), C.c_void_p(origin), C.c_int(pri), C.c_int(sec), C.c_int(dirn), C.c_int(pd), C.c_int(sd), C.c_int(bd - 8),
                  C.c_int(bw), C.c_int(bh))
                olib.orc_cdef_filter_block(P(want) if not is16 else None, P(want) if is16 else None, C.c_int(11), C.c_void_p(origin), C.c_int(pri),
                                           C.c_int(sec), C.c_int(dirn), C.c_int(pd), C.c_int(sd), C.c_int(bd - 8), C.c_int(bw), C.c_int(bh),
                                           C.c_int(int(variant in (0, 1))), C.c_int(int(variant in (0, 2))))
                assert np.array_equal(got, want), (bits, variant, trial)


def test_installer_table_and_sticky_status(hip):
    lib = hip.capi.lib
    lib.aomhip_status_clear.restype = None
    lib.aomhip_status_clear()
    assert lib.aomhip_status() == 0
    n_fixed = 3 + 2 + 12 + 19 + 19 + 8 * 5 + 2 + 8
    n_ptrs = n_fixed + 3 * 22 * 16  # + block_fns[3 depths][22 BLOCK_SIZEs] x the 16 members of aom_variance_fn_ptr_t
    table = (C.c_void_p * n_ptrs)()
    assert lib.aomhip_rtcd(table) == 0
    assert all(table[i] for i in range(n_ptrs)), "every pointer of aomhip_rtcd_table is filled"
    # pointers are the exported symbols
    assert table[5] == C.cast(lib.aomhip_quantize_b, C.c_void_p).value
    # block_fns = what the vtable binder installs: the sdf of BLOCK_64X32 (index 11) at 8 bits computes aom_sad64x32
    vt = (C.c_void_p * (22 * 16))()
    assert lib.aomhip_bind_variance_vtable(vt, 8) == 0
    assert [table[n_fixed + i] for i in range(22 * 16)] == list(vt)
    sdf = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_int)(table[n_fixed + 11 * 16 + 0])
    rng = np.random.default_rng(9)
    a, b = rng.integers(0, 256, (32, 80), dtype=np.uint8), rng.integers(0, 256, (32, 72), dtype=np.uint8)
    assert sdf(a.ctypes.data, 80, b.ctypes.data, 72) == int(np.abs(a[:, :64].astype(np.int32) - b[:, :64].astype(np.int32)).sum())
    hsdf = C.CFUNCTYPE(C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_int)(table[n_fixed + (1 * 22 + 6) * 16 + 0])  # highbd 10-bit, BLOCK_16X16
    a16, b16 = rng.integers(0, 1024, (16, 24), dtype=np.uint16), rng.integers(0, 1024, (16, 16), dtype=np.uint16)
    # (the encoder's _bits10 wrapper: aom_highbd_sad16x16 >> 2, av1/encoder/encoder_utils.h:130-139)
    assert hsdf(a16.ctypes.data >> 1, 24, b16.ctypes.data >> 1, 16) == int(np.abs(a16[:, :16].astype(np.int32) - b16.astype(np.int32)).sum()) >> 2
    # an unsupported call records the sticky status, returns its defined result and does not take the process down
    f = _fn(lib, "aomhip_fwd_txfm2d")
    out = np.full(16, 5, np.int32); res = np.zeros((4, 4), np.int16)
    f(P(res), P(out), C.c_int(4), C.c_int(0), C.c_int(8), C.c_int(4), C.c_int(12))  # no 4x12 transform
    assert lib.aomhip_status() == 2 and lib.aomhip_failure_count() >= 1
    assert b"4x12" in lib.aomhip_last_error()
    lib.aomhip_status_clear()
    assert lib.aomhip_status() == 0
