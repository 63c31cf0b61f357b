"""Parity of the fused forward-transform + quantise HIP kernel with the oracle (bit-exact), through
the C ABI: all 19 transform sizes x every servable TX_TYPE, low-bd and high-bd quantisers, list
mode / grid mode / fused-subtract mode, qindex sweep, extreme inputs (mirrors
test/av1_fwd_txfm2d_test.cc:243-297 input classes and test/quantize_func_test.cc:202-259)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_list(hip, oracle, ctx, residual, tx_size, blocks, q, is_hbd, want_coeff=True):
    n = len(blocks)
    nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
    total = int(blocks["out_offset"].max()) + nc if n else 0
    d_res = ctx.to_device(residual)
    d_blk = ctx.to_device(blocks)
    d_c = ctx.malloc(total * 4) if want_coeff else None
    d_q, d_dq, d_e = ctx.malloc(total * 4), ctx.malloc(total * 4), ctx.malloc(max(2 * n, 16))
    for d in (d_q, d_dq):
        hip.capi.check(hip.capi.lib.aomhip_memset(ctx.h, d, 0x5A, total * 4))
    qp = hip.capi.QuantParams.from_tables(q)
    ctx.xform_quant_batch(d_res, residual.shape[1], tx_size, d_blk, n, 0, 0, qp, is_hbd, d_c, d_q, d_dq, d_e)
    got = (ctx.from_device(d_c, (total,), np.int32) if want_coeff else None, ctx.from_device(d_q, (total,), np.int32),
           ctx.from_device(d_dq, (total,), np.int32), ctx.from_device(d_e, (n,), np.uint16))
    for d in (d_res, d_blk, d_c, d_q, d_dq, d_e):
        if d:
            ctx.free(d)
    want = oracle.xform_quant_batch(residual, tx_size, blocks, n, 0, 0, q, is_hbd, total, want_coeff, threads=4)
    return got, want


def _blocks(hip, rng, W, H, w, h, n, types, nc, shuffle=True):
    b = np.zeros(n, hip.capi.txb_dtype)
    b["x"] = rng.integers(0, W - w + 1, n)
    b["y"] = rng.integers(0, H - h + 1, n)
    b["tx_type"] = rng.choice(types, n)
    order = rng.permutation(n) if shuffle else np.arange(n)
    b["out_offset"] = order * nc  # the reference's BLOCK_OFFSET is arbitrary per block
    return b


@pytest.mark.parametrize("is_hbd", [False, True])
@pytest.mark.parametrize("tx_size", range(19))
def test_all_sizes_all_types_random(hip, oracle, ctx, tx_size, is_hbd):
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    types = [t for t in range(16) if oracle.lib.orc_txfm_valid(tx_size, t)]
    rng = np.random.default_rng(tx_size * 2 + is_hbd)
    W, H = 192, 160
    nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
    bits = 11 if is_hbd else 9  # residual of 10-bit / 8-bit video
    residual = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), (H, W)).astype(np.int16)
    n = 301 if w * h <= 256 else 67  # ragged
    blocks = _blocks(hip, rng, W, H, w, h, n, types, nc)
    for qindex in (0, 60, 255):
        q = oracle.build_quantizer_y(10 if is_hbd else 8, qindex)
        got, want = _run_list(hip, oracle, ctx, residual, tx_size, blocks, q, is_hbd)
        for g, wv, name in zip(got, want, ("coeff", "qcoeff", "dqcoeff", "eob")):
            assert np.array_equal(g, wv), (tx_size, is_hbd, qindex, name)


@pytest.mark.parametrize("tx_size", [0, 1, 2, 3, 4, 9, 12, 17])
def test_extreme_inputs(hip, oracle, ctx, tx_size):
    """all-max / all-min / alternating residuals (av1_fwd_txfm2d_test.cc:264-269 uses +-(1<<bd)-1),
    DC-only and zero blocks (quantize_func_test.cc ZeroInput / DcOnly / LargeNegativeInput)."""
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    types = [t for t in range(16) if oracle.lib.orc_txfm_valid(tx_size, t)]
    nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
    pats = []
    for v in (1023, -1023, 255, -256, 0, 4095, -4096):
        pats.append(np.full((h, w), v, np.int16))
    chk = np.indices((h, w)).sum(0) % 2
    pats.append(np.where(chk, 1023, -1023).astype(np.int16))
    one = np.zeros((h, w), np.int16); one[0, 0] = -8191; pats.append(one)
    residual = np.concatenate(pats, axis=1)
    n = len(pats) * len(types)
    blocks = np.zeros(n, hip.capi.txb_dtype)
    k = 0
    for pi in range(len(pats)):
        for t in types:
            blocks[k] = (pi * w, 0, k * nc, t, (0, 0, 0)); k += 1
    for is_hbd in (False, True):
        for qindex in (1, 100):
            q = oracle.build_quantizer_y(12 if is_hbd else 8, qindex)
            got, want = _run_list(hip, oracle, ctx, residual, tx_size, blocks, q, is_hbd)
            for g, wv, name in zip(got, want, ("coeff", "qcoeff", "dqcoeff", "eob")):
                assert np.array_equal(g, wv), (tx_size, is_hbd, qindex, name)


def test_grid_mode_and_null_coeff(hip, oracle, ctx):
    rng = np.random.default_rng(11)
    W, H = 256, 128
    residual = rng.integers(-256, 256, (H, W)).astype(np.int16)
    for tx_size, tx_type in [(0, 0), (1, 3), (2, 0), (3, 0), (2, 9), (8, 5)]:
        w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
        gc, n = W // w, (W // w) * (H // h)
        nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
        q = oracle.build_quantizer_y(8, 100)
        d_res = ctx.to_device(residual)
        d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
        ctx.xform_quant_batch(d_res, W, tx_size, None, n, gc, tx_type, hip.capi.QuantParams.from_tables(q), False, None,
                              d_q, d_dq, d_e)
        _, wq, wdq, we = oracle.xform_quant_batch(residual, tx_size, None, n, gc, tx_type, q, False, n * nc, False, 4)
        assert np.array_equal(ctx.from_device(d_q, (n * nc,), np.int32), wq)
        assert np.array_equal(ctx.from_device(d_dq, (n * nc,), np.int32), wdq)
        assert np.array_equal(ctx.from_device(d_e, (n,), np.uint16), we)
        for d in (d_res, d_q, d_dq, d_e):
            ctx.free(d)


@pytest.mark.parametrize("bd", [8, 10])
def test_fused_subtract(hip, oracle, ctx, bd):
    """aom_subtract_block + transform + quantise in one launch == oracle subtract then xform_quant."""
    rng = np.random.default_rng(bd)
    W, H, border = 128, 96, 32
    src = hip.synth.lcg_frame(W, H, 5, 0, bd)
    pred = np.clip(src.astype(np.int32) + rng.integers(-40, 41, (H, W)), 0, (1 << bd) - 1).astype(src.dtype)
    ps, pp = ctx.planes_alloc(W, H, border, bd, 2), ctx.planes_alloc(W, H, border, bd, 2)
    ctx.planes_upload(ps, 1, src)
    ctx.planes_upload(pp, 1, pred)
    residual = (src.astype(np.int32) - pred.astype(np.int32)).astype(np.int16)  # subtract.c:20-53
    for tx_size in (0, 2, 3, 7):
        w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
        types = [t for t in range(16) if oracle.lib.orc_txfm_valid(tx_size, t)]
        nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
        n = 97
        blocks = _blocks(hip, rng, W, H, w, h, n, types, nc)
        q = oracle.build_quantizer_y(bd, 80)
        d_blk = ctx.to_device(blocks)
        d_c, d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
        ctx.subtract_xform_quant_batch(ps, pp, 1, tx_size, d_blk, n, 0, 0, hip.capi.QuantParams.from_tables(q), d_c,
                                       d_q, d_dq, d_e)
        wc, wq, wdq, we = oracle.xform_quant_batch(residual, tx_size, blocks, n, 0, 0, q, bd > 8, n * nc, True, 4)
        assert np.array_equal(ctx.from_device(d_c, (n * nc,), np.int32), wc)
        assert np.array_equal(ctx.from_device(d_q, (n * nc,), np.int32), wq)
        assert np.array_equal(ctx.from_device(d_dq, (n * nc,), np.int32), wdq)
        assert np.array_equal(ctx.from_device(d_e, (n,), np.uint16), we)
        # the _ex form: same levels plus av1_block_error / av1_highbd_block_error per block (rdopt.c:635-682)
        d_err = ctx.malloc(16 * n)
        ctx.subtract_xform_quant_ex_batch(ps, pp, 1, tx_size, d_blk, n, 0, 0, hip.capi.QuantParams.from_tables(q), 0, d_c, d_q, d_dq,
                                          d_e, d_err)
        assert np.array_equal(ctx.from_device(d_q, (n * nc,), np.int32), wq)
        err = ctx.from_device(d_err, (n, 2), np.int64)
        f = oracle.lib.orc_block_error
        f.restype = C.c_int64
        for i in range(n):
            off = int(blocks["out_offset"][i])
            c, d = np.ascontiguousarray(wc[off:off + nc]), np.ascontiguousarray(wdq[off:off + nc])
            ssz = C.c_int64()
            e = f(C.c_void_p(c.ctypes.data), C.c_void_p(d.ctypes.data), C.c_ssize_t(nc), C.byref(ssz), bd if bd > 8 else 0)
            assert (int(err[i, 0]), int(err[i, 1])) == (e, ssz.value), (tx_size, i)
        for d in (d_blk, d_c, d_q, d_dq, d_e, d_err):
            ctx.free(d)
    ctx.planes_free(ps); ctx.planes_free(pp)


def test_full_size_1080p_properties(hip, oracle, ctx):
    """BASELINE configs[2] at full size (all 16x16 blocks of a 1920x1088 residual plane): exact vs the
    oracle, plus size-independent properties: dqcoeff == qcoeff*dequant>>log_scale, eob consistent with
    the scan, zero residual -> all zero."""
    rng = np.random.default_rng(3)
    W, H = 1920, 1088
    residual = ((rng.integers(0, 1 << 16, (H, W)) & 511) - 256).astype(np.int16)
    tx_size, w = 2, 16
    gc, n, nc = W // w, (W // w) * (H // w), 256
    assert n == 8160
    q = oracle.build_quantizer_y(8, 100)
    d_res = ctx.to_device(residual)
    d_c, d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
    ctx.xform_quant_batch(d_res, W, tx_size, None, n, gc, 0, hip.capi.QuantParams.from_tables(q), False, d_c, d_q, d_dq,
                          d_e)
    gq = ctx.from_device(d_q, (n, nc), np.int32)
    gdq = ctx.from_device(d_dq, (n, nc), np.int32)
    ge = ctx.from_device(d_e, (n,), np.uint16)
    gcf = ctx.from_device(d_c, (n * nc,), np.int32)
    wc, wq, wdq, we = oracle.xform_quant_batch(residual, tx_size, None, n, gc, 0, q, False, n * nc, True, threads=8)
    assert np.array_equal(gcf, wc) and np.array_equal(gq.ravel(), wq) and np.array_equal(gdq.ravel(), wdq)
    assert np.array_equal(ge, we)
    deq = np.where(np.arange(nc) == 0, int(q["dequant"][0]), int(q["dequant"][1]))
    assert np.array_equal(np.abs(gdq), np.abs(gq) * deq)
    scan, _ = oracle.get_scan(tx_size, 0)
    nz = gq[:, scan] != 0
    last = np.where(nz.any(1), nc - np.argmax(nz[:, ::-1], axis=1), 0)
    assert np.array_equal(ge, last)
    hip.capi.check(hip.capi.lib.aomhip_memset(ctx.h, d_res, 0, residual.nbytes))
    ctx.xform_quant_batch(d_res, W, tx_size, None, n, gc, 0, hip.capi.QuantParams.from_tables(q), False, d_c, d_q, d_dq,
                          d_e)
    assert not ctx.from_device(d_q, (n * nc,), np.int32).any() and not ctx.from_device(d_e, (n,), np.uint16).any()
    for d in (d_res, d_c, d_q, d_dq, d_e):
        ctx.free(d)


@pytest.mark.parametrize("hbd", [False, True])
def test_adaptive_quantiser(hip, oracle, ctx, hbd):
    """aom_[highbd_]quantize_b_adaptive on device-resident coefficients == oracle, every transform size, scan class and
    log_scale; coefficient populations that exercise each rule: dense, sparse tails inside the widened dead zone, a
    single +-1 survivor (dropped or kept), all-zero."""
    rng = np.random.default_rng(11 + hbd)
    bd = 10 if hbd else 8
    for tx_size in range(19):
        w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
        nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
        ls = int(w * h > 256) + int(w * h > 1024)
        types = [t for t in (0, 10, 11, 9, 3) if oracle.lib.orc_txfm_valid(tx_size, t)]
        n = 96
        for qindex in (20, 120, 220):
            q = oracle.build_quantizer_y(bd, qindex)
            dqs = int(q["dequant"][1])
            coeff = np.zeros((n, nc), np.int32)
            for i in range(n):
                kind = i % 6
                if kind == 0:
                    coeff[i] = rng.normal(0, dqs * 2, nc)
                elif kind == 1:  # energy only at the start of the scan, small tail
                    coeff[i] = rng.normal(0, dqs * 0.6, nc)
                elif kind == 2:  # one coefficient near the +-1 level, position random
                    coeff[i, rng.integers(0, nc)] = int(rng.choice([-1, 1])) * int(dqs * rng.uniform(0.4, 2.2))
                elif kind == 3:
                    coeff[i] = rng.integers(-dqs, dqs + 1, nc) * (rng.random(nc) < 0.05)
                elif kind == 4:
                    coeff[i, 0] = int(rng.integers(-4 * dqs, 4 * dqs))  # DC only
                # kind 5: all zero
            tt = rng.choice(types, n).astype(np.uint8)
            blocks = np.zeros(n, hip.capi.txb_dtype)
            blocks["out_offset"] = np.arange(n) * nc
            blocks["tx_type"] = tt
            d_c, d_b = ctx.to_device(coeff), ctx.to_device(blocks)
            d_q, d_dq, d_e = ctx.malloc(n * nc * 4), ctx.malloc(n * nc * 4), ctx.malloc(2 * n)
            ctx.quantize_b_adaptive_batch(d_c, tx_size, d_b, n, 0, hip.capi.QuantParams.from_tables(q), hbd, d_q, d_dq, d_e)
            gq, gdq = ctx.from_device(d_q, (n, nc), np.int32), ctx.from_device(d_dq, (n, nc), np.int32)
            ge = ctx.from_device(d_e, (n,), np.uint16)
            differs = 0
            for i in range(n):
                scan, iscan = oracle.get_scan(tx_size, int(tt[i]))
                wq, wdq, we = oracle.quantize_b_adaptive(coeff[i], q, scan, ls, hbd)
                assert np.array_equal(gq[i], wq) and np.array_equal(gdq[i], wdq) and ge[i] == we, (tx_size, qindex, i, int(tt[i]))
                pq, _, pe = oracle.quantize_b(coeff[i], q, scan, iscan, ls, hbd)
                differs += int(pe != we or not np.array_equal(pq, wq))
            assert differs > 0, (tx_size, qindex)  # the adaptive rules actually fired somewhere
            for d in (d_c, d_b, d_q, d_dq, d_e):
                ctx.free(d)


@pytest.mark.parametrize("is_hbd,bit_depth", [(False, 8), (True, 8), (True, 10), (True, 12)])
@pytest.mark.parametrize("tx_size", [0, 1, 2, 3, 4, 6, 9, 13, 18])
def test_fused_block_error(hip, oracle, ctx, tx_size, is_hbd, bit_depth):
    """aomhip_xform_quant_ex_batch with d_block_error: the same coefficients / levels as aomhip_xform_quant_batch plus av1_block_error /
    av1_highbd_block_error (rdopt.c:635-682) of every block, against the oracle's restatement (itself pinned against the
    interpreted reference, tests/test_golden_ref_eval.py)."""
    import ctypes as C
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    types = [t for t in range(16) if oracle.lib.orc_txfm_valid(tx_size, t)]
    rng = np.random.default_rng(tx_size * 5 + bit_depth + is_hbd)
    W, H = 192, 160
    nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
    bits = bit_depth + 1
    residual = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), (H, W)).astype(np.int16)
    n = 203 if w * h <= 256 else 41
    blocks = _blocks(hip, rng, W, H, w, h, n, types, nc)
    q = oracle.build_quantizer_y(bit_depth if is_hbd else 8, 120)
    total = int(blocks["out_offset"].max()) + nc
    d_res, d_blk = ctx.to_device(residual), ctx.to_device(blocks)
    d_c, d_q, d_dq, d_e, d_err = ctx.malloc(total * 4), ctx.malloc(total * 4), ctx.malloc(total * 4), ctx.malloc(max(2 * n, 16)), ctx.malloc(16 * n)
    qp = hip.capi.QuantParams.from_tables(q)
    ctx.xform_quant_ex_batch(d_res, W, tx_size, d_blk, n, 0, 0, qp, is_hbd, bit_depth, 0, d_c, d_q, d_dq, d_e, d_err)
    coeff, dq = ctx.from_device(d_c, (total,), np.int32), ctx.from_device(d_dq, (total,), np.int32)
    err = ctx.from_device(d_err, (n, 2), np.int64)
    want = oracle.xform_quant_batch(residual, tx_size, blocks, n, 0, 0, q, is_hbd, total, True, threads=4)
    assert np.array_equal(coeff, want[0]) and np.array_equal(dq, want[2])
    f = oracle.lib.orc_block_error
    f.restype = C.c_int64
    for i in range(n):
        off = int(blocks["out_offset"][i])
        c, d = np.ascontiguousarray(coeff[off:off + nc]), np.ascontiguousarray(dq[off:off + nc])
        ssz = C.c_int64()
        e = f(C.c_void_p(c.ctypes.data), C.c_void_p(d.ctypes.data), C.c_ssize_t(nc), C.byref(ssz), bit_depth if is_hbd else 0)
        assert (int(err[i, 0]), int(err[i, 1])) == (e, ssz.value), (tx_size, i)
    for d in (d_res, d_blk, d_c, d_q, d_dq, d_e, d_err):
        ctx.free(d)


@pytest.mark.parametrize("is_hbd", [False, True])
@pytest.mark.parametrize("tx_size", [0, 1, 2, 3, 4, 5, 10, 14, 17])
def test_quantize_fp_flavour(hip, oracle, ctx, tx_size, is_hbd):
    """AOMHIP_QUANT_FP: av1_[highbd_]quantize_fp{,_32x32,_64x64} fused behind the transform, with the block error."""
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    types = [t for t in range(16) if oracle.lib.orc_txfm_valid(tx_size, t)]
    rng = np.random.default_rng(tx_size * 11 + is_hbd)
    W, H = 192, 160
    nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
    bits = 11 if is_hbd else 9
    residual = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), (H, W)).astype(np.int16)
    n = 157 if w * h <= 256 else 37
    blocks = _blocks(hip, rng, W, H, w, h, n, types, nc)
    total = int(blocks["out_offset"].max()) + nc
    for qindex in (0, 40, 130, 255):
        q = oracle.build_quantizer_y(10 if is_hbd else 8, qindex)
        dq = q["dequant"].astype(np.int64)
        qfp = dict(q, round=((64 * dq) >> 7).astype(np.int16), quant=((1 << 16) // dq).astype(np.int16))   # y_round_fp / y_quant_fp
        d_res, d_blk = ctx.to_device(residual), ctx.to_device(blocks)
        d_c, d_q, d_dq, d_e = ctx.malloc(total * 4), ctx.malloc(total * 4), ctx.malloc(total * 4), ctx.malloc(max(2 * n, 16))
        for d in (d_q, d_dq):
            hip.capi.check(hip.capi.lib.aomhip_memset(ctx.h, d, 0x5A, total * 4))
        ctx.xform_quant_ex_batch(d_res, W, tx_size, d_blk, n, 0, 0, hip.capi.QuantParams.from_tables(qfp), is_hbd, 10 if is_hbd else 8, 1,
                                 d_c, d_q, d_dq, d_e, None)
        got = (ctx.from_device(d_c, (total,), np.int32), ctx.from_device(d_q, (total,), np.int32), ctx.from_device(d_dq, (total,), np.int32),
               ctx.from_device(d_e, (n,), np.uint16))
        oracle.lib.orc_xform_quant_set_kind(1)
        try:
            want = oracle.xform_quant_batch(residual, tx_size, blocks, n, 0, 0, qfp, is_hbd, total, True, threads=1)
        finally:
            oracle.lib.orc_xform_quant_set_kind(0)
        used = np.zeros(total, bool)
        for off in blocks["out_offset"]:
            used[int(off):int(off) + nc] = True
        for g, wv, name in zip(got, want, ("coeff", "qcoeff", "dqcoeff", "eob")):
            if name == "eob":
                assert np.array_equal(g, wv), (tx_size, is_hbd, qindex, name)
            else:
                assert np.array_equal(g[used], wv[used]), (tx_size, is_hbd, qindex, name)
        for d in (d_res, d_blk, d_c, d_q, d_dq, d_e):
            ctx.free(d)


def test_per_block_tx_type_that_does_not_exist_for_the_size_is_reported(hip, oracle, ctx):
    """a per-block list is in device memory, so the host cannot validate it when the call is made; a check kernel in
    front of the transform ORs a code into the context's device status word and the next aomhip_ctx_sync fails (then clears it)."""
    rng = np.random.default_rng(5)
    W = H = 64
    residual = rng.integers(-255, 256, (H, W)).astype(np.int16)
    d_res = ctx.to_device(residual)
    q = hip.capi.QuantParams.from_tables(oracle.build_quantizer_y(8, 100))
    for tx_size, bad_type, good_type in ((3, 1, 0), (4, 9, 0), (2, 16, 1), (1, 200, 3)):     # ADST on 32 points, IDTX on 64, WHT beyond 4x4, junk
        w = hip.capi.lib.aomhip_tx_size_wide(tx_size)
        nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
        blocks = np.zeros(2, hip.capi.txb_dtype)
        blocks["x"] = [0, 0]; blocks["y"] = [0, 0]; blocks["out_offset"] = [0, nc]
        d_c, d_q, d_dq, d_e = ctx.malloc(2 * nc * 4), ctx.malloc(2 * nc * 4), ctx.malloc(2 * nc * 4), ctx.malloc(16)
        for types, fails in (([good_type, bad_type], True), ([good_type, good_type], False)):
            blocks["tx_type"] = types
            d_b = ctx.to_device(blocks)
            ctx.xform_quant_batch(d_res, W, tx_size, d_b, 2, 0, 0, q, False, d_c, d_q, d_dq, d_e)
            if fails:
                with pytest.raises(hip.capi.AomHipError, match="tx_type"):
                    ctx.sync()
            ctx.sync()                                           # reported once, then clear
            ctx.free(d_b)
        for d in (d_c, d_q, d_dq, d_e):
            ctx.free(d)
    ctx.free(d_res)


@pytest.mark.parametrize("tx_size", range(19))
def test_fast_butterfly_boundary(hip, oracle, ctx, tx_size):
    """The kernels evaluate half_btf with 24-bit products and a 32-bit sum for blocks whose residual magnitude is at most
    kSafeMax[tx_size][tx_type] (csrc/txfm_safe_max.inc, from the oracle's interval analysis) and with the exact 64-bit form otherwise.
    Blocks AT the bound -- all +M, all -M, checkerboards, the sign patterns of the DCT's basis functions (the worst cases of a
    butterfly network), random signs -- and blocks one above it (exact path), interleaved in one launch so that wavefronts mix both
    paths, must equal the oracle bit for bit."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
    import gen_txfm_bounds as gb
    w, h = oracle.TX_W[tx_size], oracle.TX_H[tx_size]
    nc = hip.capi.lib.aomhip_tx_max_eob(tx_size)
    rng = np.random.default_rng(900 + tx_size)
    types = [t for t in range(16) if oracle.lib.orc_txfm_valid(tx_size, t)]
    yy, xx = np.mgrid[0:h, 0:w]
    tiles, btypes = [], []
    for t in types:
        M = gb.safe_max(tx_size, t)
        assert M > 0
        pats = [np.ones((h, w)), -np.ones((h, w)), np.where((yy + xx) & 1, -1, 1), np.where(yy & 1, -1, 1), np.where(xx & 1, -1, 1)]
        for k in (1, 2, 3, w - 1):   # sign of cos((2x + 1) k pi / 2w) x the same in y: rows / columns of the DCT matrix
            cx = np.sign(np.cos((2 * xx + 1) * k * np.pi / (2 * w)) + 1e-9)
            cy = np.sign(np.cos((2 * yy + 1) * min(k, h - 1) * np.pi / (2 * h)) + 1e-9)
            pats += [cx, cy, cx * cy]
        pats += [rng.choice([-1, 1], (h, w)) for _ in range(3)]
        for p in pats:
            for mag in ((M, min(M + 1, 32767)) if M < 32767 else (M,)):
                tiles.append((p * mag).astype(np.int16)); btypes.append(t)
            if M >= 32767:
                tiles.append(np.full((h, w), -32768, np.int16)); btypes.append(t)   # |x| = 32768 > every bound: the exact path
    n = len(tiles)
    cols = max(1, 1024 // w)
    rows = (n + cols - 1) // cols
    residual = np.zeros((rows * h, cols * w), np.int16)
    blocks = np.zeros(n, hip.capi.txb_dtype)
    for i, tl in enumerate(tiles):
        r, c = divmod(i, cols)
        residual[r * h:(r + 1) * h, c * w:(c + 1) * w] = tl
        blocks[i] = (c * w, r * h, i * nc, btypes[i], (0, 0, 0))
    q = oracle.build_quantizer_y(10, 40)
    got, want = _run_list(hip, oracle, ctx, residual, tx_size, blocks, q, True)
    for g, wv, name in zip(got, want, ("coeff", "qcoeff", "dqcoeff", "eob")):
        bad = np.nonzero(g != wv)[0]
        assert bad.size == 0, (tx_size, name, bad[:5], [btypes[i // nc] for i in bad[:5]] if name != "eob" else [btypes[i] for i in bad[:5]])
