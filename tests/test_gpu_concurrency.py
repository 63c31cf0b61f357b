"""Concurrent callers (the reference calls its DSP entry points lock-free from tile / row-MT workers,
av1/encoder/ethread.c:488-593).  Eight threads at once, each
  (1) through the rtcd-signature entry points (host pointers; the library gives every thread its own default context), and
  (2) through its OWN aomhip_ctx on batched calls over its own HBM planes,
must get exactly what the same calls return when issued one after the other -- and what the oracle says.  ctypes drops the GIL for the
duration of every foreign call, so the eight Python threads really are inside libaomhip together."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = lambda a: C.c_void_p(a.ctypes.data)
N_THREADS, ROUNDS = 8, 6


def _fn(lib, name, restype=None):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = None
    return f


def _rtcd_work(lib, oracle, t, rounds):
    """One thread's share of rtcd-signature calls; returns a list of results (python ints / arrays) in call order."""
    rng = np.random.default_rng(1000 + t)
    sad = _fn(lib, "aomhip_sad16x16", C.c_uint)
    sad4 = _fn(lib, "aomhip_sad16x16x4d")
    var = _fn(lib, "aomhip_variance16x16", C.c_uint)
    sub = _fn(lib, "aomhip_subtract_block")
    qb = _fn(lib, "aomhip_quantize_b")
    fwd = _fn(lib, "aomhip_fwd_txfm2d_16x16")
    lpf = _fn(lib, "aomhip_lpf_horizontal_8")
    sc, isc = oracle.get_scan(2, 0)
    q = oracle.build_quantizer_y(8, 60 + 10 * t)
    tabs = {k: np.ascontiguousarray(v, np.int16) for k, v in q.items()}
    out = []
    for _ in range(rounds):
        a = rng.integers(0, 256, (16, 40), dtype=np.uint8)
        refs = [rng.integers(0, 256, (16, 48), dtype=np.uint8) for _ in range(4)]
        out.append(int(sad(P(a), C.c_int(40), P(refs[0]), C.c_int(48))))
        arr4 = (C.c_void_p * 4)(*[r.ctypes.data for r in refs])
        res4 = np.zeros(4, np.uint32)
        sad4(P(a), C.c_int(40), arr4, C.c_int(48), P(res4))
        out.append(res4.copy())
        sse = C.c_uint(0)
        v = int(var(P(a), C.c_int(40), P(refs[1]), C.c_int(48), C.byref(sse)))
        out.append((v, int(sse.value)))
        diff = np.zeros((16, 16), np.int16)
        sub(C.c_int(16), C.c_int(16), P(diff), C.c_ssize_t(16), P(a), C.c_ssize_t(40), P(refs[2]), C.c_ssize_t(48))
        out.append(diff.copy())
        coeff = np.zeros(256, np.int32)
        fwd(P(diff), P(coeff), C.c_int(16), C.c_int(0), C.c_int(8))
        out.append(coeff.copy())
        qc, dq, eob = np.zeros(256, np.int32), np.zeros(256, np.int32), C.c_uint16(0)
        qb(P(coeff), C.c_ssize_t(256), P(tabs["zbin"]), P(tabs["round"]), P(tabs["quant"]), P(tabs["quant_shift"]), P(qc), P(dq),
           P(tabs["dequant"]), C.byref(eob), P(sc), P(isc))
        out.append((qc.copy(), dq.copy(), int(eob.value)))
        pix = rng.integers(96, 160, (16, 16), dtype=np.uint8)  # a smooth-ish patch: the filter's masks fire
        th = [np.full(16, v, np.uint8) for v in (20, 12, 4)]
        lpf(C.c_void_p(pix.ctypes.data + 8 * 16), C.c_int(16), P(th[0]), P(th[1]), P(th[2]))
        out.append(pix.copy())
    return out


def _same(a, b):
    if isinstance(a, tuple):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return np.array_equal(a, b)
    return a == b


def test_rtcd_entry_points_from_eight_threads(hip, oracle):
    lib = C.CDLL(hip.capi.lib._name)  # a second handle of the same library: its function objects carry their own (absent) argtypes
    lib.aomhip_status.restype = C.c_int
    lib.aomhip_last_error.restype = C.c_char_p
    lib.aomhip_status_clear.restype = None
    lib.aomhip_status_clear()
    serial = [_rtcd_work(lib, oracle, t, ROUNDS) for t in range(N_THREADS)]
    # anchor the serial results on the oracle for the cost-type calls
    rng = np.random.default_rng(1000)
    a = rng.integers(0, 256, (16, 40), dtype=np.uint8)
    r0 = rng.integers(0, 256, (16, 48), dtype=np.uint8)
    assert serial[0][0] == int(np.abs(a[:, :16].astype(np.int32) - r0[:, :16].astype(np.int32)).sum())
    got, errs = [None] * N_THREADS, []
    start = threading.Barrier(N_THREADS)

    def worker(t):
        try:
            start.wait()
            got[t] = _rtcd_work(lib, oracle, t, ROUNDS)
        except Exception as e:  # pragma: no cover
            errs.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(N_THREADS)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    assert lib.aomhip_status() == 0, lib.aomhip_last_error()
    for t in range(N_THREADS):
        assert len(got[t]) == len(serial[t])
        for i, (g, s) in enumerate(zip(got[t], serial[t])):
            assert _same(g, s), "thread %d call %d differs from the serial run" % (t, i)


def test_own_context_batched_calls_from_eight_threads(hip, oracle):
    """Every thread: its own aomhip_ctx (own stream), its own plane pair and lists; SAD x4d + single + the bucketed launch + a 16x16
    transform / quantise pass, twice.  All results equal the oracle."""
    W, H, border, bd = 640, 368, 160, 8
    errs, results = [], [None] * N_THREADS
    start = threading.Barrier(N_THREADS)
    qt = oracle.build_quantizer_y(8, 100)

    def worker(t):
        try:
            c = hip.capi.Context(0)
            s, r = hip.synth.lcg_frame(W, H, 2 * t, 0, bd), hip.synth.lcg_frame(W, H, 2 * t + 1, 1, bd)
            ps, pr = c.planes_alloc(W, H, border, bd, 1), c.planes_alloc(W, H, border, bd, 1)
            c.planes_upload(ps, 0, s); c.planes_upload(pr, 0, r)
            cands, groups = hip.synth.mode_a_worklist(W, H, 16, seed=50 + t, search=32)
            n = len(groups)
            perm, off = hip.synth.bucket_order(groups["sx"], groups["sy"], W, H, 128, 64)
            d_g, d_c = c.to_device(groups), c.to_device(cands)
            d_gs, d_cs, d_off = c.to_device(groups[perm]), c.to_device(cands[perm]), c.to_device(off)
            d_o4, d_o1, d_p4, d_p1 = c.malloc(n * 16), c.malloc(n * 4), c.malloc(n * 16), c.malloc(n * 4)
            rng = np.random.default_rng(t)
            res = ((rng.integers(0, 1 << 16, (H // 16 * 16, W)) & 511) - 256).astype(np.int16)
            nb = (W // 16) * (res.shape[0] // 16)
            d_res = c.to_device(res)
            d_q, d_dq, d_e = c.malloc(nb * 256 * 4), c.malloc(nb * 256 * 4), c.malloc(nb * 2)
            qp = hip.capi.QuantParams.from_tables(qt)
            start.wait()
            for _ in range(2):
                c.sad_x4d_batch(ps, pr, 0, 1, 16, 16, 0, d_g, n, 0, d_o4)
                c.sad_batch(ps, pr, 0, 1, 16, 16, 0, d_c, n, 0, d_o1)
                c.sad_sb_batch(ps, pr, 0, 1, 16, 16, 0, 128, 64, 32, len(off) - 1, d_gs, d_off, n, 0, d_p4, d_cs, d_off, n, 0, d_p1)
                c.xform_quant_batch(d_res, W, 2, None, nb, W // 16, 0, qp, False, None, d_q, d_dq, d_e)
            o4, o1 = c.from_device(d_o4, (n, 4), np.uint32), c.from_device(d_o1, (n,), np.uint32)
            p4, p1 = c.from_device(d_p4, (n, 4), np.uint32), c.from_device(d_p1, (n,), np.uint32)
            gq, ge = c.from_device(d_q, (nb * 256,), np.int32), c.from_device(d_e, (nb,), np.uint16)
            results[t] = (s, r, cands, groups, perm, o4, o1, p4, p1, res, nb, gq, ge, ps.stride, pr.stride)
            for d in (d_g, d_c, d_gs, d_cs, d_off, d_o4, d_o1, d_p4, d_p1, d_res, d_q, d_dq, d_e):
                c.free(d)
            c.planes_free(ps); c.planes_free(pr)
            c.close()
        except Exception as e:  # pragma: no cover
            errs.append((t, repr(e)))
            try:
                start.abort()
            except Exception:
                pass

    th = [threading.Thread(target=worker, args=(t,)) for t in range(N_THREADS)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for t in range(N_THREADS):
        s, r, cands, groups, perm, o4, o1, p4, p1, res, nb, gq, ge, ss, rs = results[t]
        sb, rb = oracle.extend_plane(s, border, ss), oracle.extend_plane(r, border, rs)
        w4 = oracle.sad_x4d_batch(sb, rb, border, 16, 16, groups, threads=4)
        w1 = oracle.sad_batch(sb, rb, border, 16, 16, cands, threads=4)
        assert np.array_equal(o4, w4) and np.array_equal(o1, w1), t
        assert np.array_equal(p4, w4[perm]) and np.array_equal(p1, w1[perm]), t
        _, wq, _, we = oracle.xform_quant_batch(res, 2, None, nb, W // 16, 0, qt, False, nb * 256, False, threads=4)
        assert np.array_equal(gq, wq) and np.array_equal(ge, we), t
