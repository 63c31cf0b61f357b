"""Deblocking oracle:
  * the box-filter formulation of filter6/8/14 == the reference's literal tap listings, by evaluating the
    reference's own `*opN = ROUND_POWER_OF_TWO(<expr>, s);` statements (aom_dsp/loopfilter.c) on random
    pixels (live when /root/reference exists);
  * known behaviour: a flat region passes unchanged; level 0 / len 0 leave the plane untouched;
  * the whole-plane driver in the reference's superblock-row order == "all vertical then all horizontal"
    on random transform partitions (the independence property of SURVEY 8(a'), kept as a permanent test);
  * edges of one pass are order-independent (left-to-right == right-to-left)."""
import re

import numpy as np
import pytest

from conftest import REFERENCE, have_reference


def _ref_flat_exprs():
    src = open(REFERENCE + "/aom_dsp/loopfilter.c").read()
    out = {}
    for name in ("filter6", "filter8", "filter14"):
        m = re.search(r"static INLINE void %s\(.*?\n\}" % name, src, re.S)
        body = m.group(0)
        stm = re.findall(r"\*o([pq]\d) = ROUND_POWER_OF_TWO\((.*?),\s*(\d)\);", body, re.S)
        out[name] = [(t, re.sub(r"\s+", " ", e), int(s)) for t, e, s in stm]
    return out


@pytest.mark.skipif(not have_reference(), reason="needs /root/reference")
def test_flat_filters_equal_reference_tap_listings(oracle):
    exprs = _ref_flat_exprs()
    assert [len(v) for v in exprs.values()] == [4, 6, 12]
    rng = np.random.default_rng(1)
    for name, length in (("filter6", 6), ("filter8", 8), ("filter14", 14)):
        for _ in range(200):
            base = int(rng.integers(2, 250))
            taps = {("p%d" % i): base + int(rng.integers(-1, 2)) for i in range(7)}
            taps.update({("q%d" % i): base + int(rng.integers(-1, 2)) for i in range(7)})
            # flat region (all within 1 of p0/q0) with generous limits -> the flat branch is taken
            px = np.zeros((4, 16), np.uint8)
            row = [taps["p%d" % i] for i in range(6, -1, -1)] + [taps["q%d" % i] for i in range(7)]
            px[:, 1:15] = row
            want = dict(taps)
            for t, e, s in exprs[name]:
                want[t] = (eval(e, {}, taps) + (1 << (s - 1))) >> s
            oracle.lpf_edge(px, 0, 8, True, length, 255, 255, 0)
            got = px[0, 1:15].tolist()
            exp = [want["p%d" % i] for i in range(6, -1, -1)] + [want["q%d" % i] for i in range(7)]
            if abs(taps["p0"] - taps["q0"]) * 2 + abs(taps["p1"] - taps["q1"]) // 2 <= 255:
                flat_ok = all(abs(taps["p%d" % i] - taps["p0"]) <= 1 and abs(taps["q%d" % i] - taps["q0"]) <= 1
                              for i in range(1, {6: 3, 8: 4, 14: 7}[length]))
                if flat_ok:
                    assert got == exp, (name, row)


def test_filter4_and_thresholds(oracle):
    import ctypes as C
    mb, lim, hev = C.c_uint8(), C.c_uint8(), C.c_uint8()
    # update_sharpness (av1_loopfilter.c:47-66): sharpness 0 -> lim = max(level,1), mblim = 2*(level+2)+lim
    oracle.lib.orc_lpf_thresholds(32, 0, C.byref(mb), C.byref(lim), C.byref(hev))
    assert (mb.value, lim.value, hev.value) == (2 * 34 + 32, 32, 2)
    oracle.lib.orc_lpf_thresholds(63, 7, C.byref(mb), C.byref(lim), C.byref(hev))
    assert (lim.value, mb.value, hev.value) == (2, 2 * 65 + 2, 3)
    oracle.lib.orc_lpf_thresholds(0, 0, C.byref(mb), C.byref(lim), C.byref(hev))
    assert lim.value == 1
    # hand-evaluated filter4 (loopfilter.c:104-134): p = 100, q = 108, no hev -> f = 24, f1 = f2 = 3, outer tap 2;
    # outside blimit the edge is left alone
    px = np.zeros((4, 8), np.uint8); px[:, :4] = 100; px[:, 4:] = 108
    oracle.lpf_edge(px, 0, 4, True, 4, 40, 20, 2)
    assert px[0].tolist() == [100, 100, 102, 103, 105, 106, 108, 108]
    px = np.zeros((4, 8), np.uint8); px[:, :4] = 10; px[:, 4:] = 200
    oracle.lpf_edge(px, 0, 4, True, 4, 40, 20, 2)
    assert px[0].tolist() == [10] * 4 + [200] * 4
    flat = np.full((8, 16), 77, np.uint8)
    for length in (4, 6, 8, 14):
        oracle.lpf_edge(flat, 0, 8, True, length, 60, 20, 1)
        oracle.lpf_edge(flat, 4, 8, False, length, 60, 20, 1) if length <= 8 else None
    assert (flat == 77).all()


@pytest.mark.parametrize("bd", [8, 10])
def test_two_pass_order_equals_reference_superblock_order(oracle, bd):
    rng = np.random.default_rng(bd)
    for trial in range(6):
        W, H = int(rng.choice([64, 128, 200, 320])), int(rng.choice([64, 136, 192]))
        mx = (1 << bd) - 1
        dt = np.uint8 if bd == 8 else np.uint16
        # smooth-ish content so that many edges actually get filtered (incl. the flat branches)
        base = rng.integers(0, mx // 4, (H // 8 + 1, W // 8 + 1))
        pix = np.kron(base, np.ones((8, 8), np.int64))[:H, :W] + rng.integers(-2, 3, (H, W)) + mx // 3
        pix = np.clip(pix, 0, mx).astype(dt)
        params = oracle.random_edge_params(rng, W, H)
        sharp = int(rng.integers(0, 8))
        a = oracle.deblock_plane(pix, params, sharp, bd, order=0)
        b = oracle.deblock_plane(pix, params, sharp, bd, order=1)
        assert np.array_equal(a, b)
        assert not np.array_equal(a, pix)  # something was filtered
        # within a pass the edge order is irrelevant: mirror the vertical pass
        p_rev = params.copy(); p_rev[..., 2:] = 0
        v_only = oracle.deblock_plane(pix, p_rev, sharp, bd, order=1)
        manual = pix.copy()
        import ctypes as C
        for uy in range(params.shape[0]):
            for ux in range(params.shape[1] - 1, -1, -1):
                l, lv = params[uy, ux, 0], params[uy, ux, 1]
                if l and lv:
                    mb, lim, hev = C.c_uint8(), C.c_uint8(), C.c_uint8()
                    oracle.lib.orc_lpf_thresholds(int(lv), sharp, C.byref(mb), C.byref(lim), C.byref(hev))
                    oracle.lpf_edge(manual, 4 * uy, 4 * ux, True, int(l), mb.value, lim.value, hev.value, bd)
        assert np.array_equal(manual, v_only)
