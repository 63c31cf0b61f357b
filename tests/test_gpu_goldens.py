"""The HIP path, through the C ABI, straight against the golden vectors obtained by interpreting the reference's own C
functions (tests/golden/ref_eval_*.npz; generators gen_ref_eval_golden.py / gen_ref_eval_more.py): no oracle in between.

  SAD / skip SAD / x4d, variance, sub-pixel variance (8/10/12-bit)      rtcd-signature entry points
  av1_fwd_txfm2d (19 sizes), av1_inv_txfm2d_add (8/10/12-bit)            aomhip_xform_quant_batch (coeff) / aomhip_inv_txfm_add_batch
  aom_[highbd_]quantize_b*_adaptive                                       aomhip_quantize_b_adaptive_batch
  aom_[highbd_]lpf_{horizontal,vertical}_{4,6,8,14}                      aomhip_deblock_plane with a single edge unit
  av1_cdef_filter_fb (luma + four chroma subsamplings)                   aomhip_cdef_luma_plane / aomhip_cdef_chroma_plane
(The motion-search goldens are covered by tests/test_gpu_full_pixel_search.py.)"""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    return z, json.loads(bytes(z["cases"]).decode())


def test_sad_variance_entry_points_match_reference_goldens(hip):
    lib = hip.capi.lib
    z, rows = load("ref_eval_sadvar.npz")
    for r in rows:
        if r.get("extra"):
            a, b = np.ascontiguousarray(z["a8"].astype(np.uint8)), np.ascontiguousarray(z["b8"].astype(np.uint8))
            S = a.shape[1]
            pa, pb = a.ctypes.data + r["oy"] * S + r["ox"], b.ctypes.data + r["ry"] * S + r["rx"]
            sse, sm = C.c_uint(), C.c_int()
            for (w, h) in ((16, 16), (16, 8), (8, 16), (8, 8)):
                assert [lib.aomhip_mse(pa, S, pb, S, w, h, C.byref(sse)), sse.value] == r["mse%dx%d" % (w, h)]
            for n in (8, 16):
                lib.aomhip_get_var(pa, S, pb, S, n, n, C.byref(sse), C.byref(sm))
                assert [sse.value, sm.value] == r["get%dvar" % n]
            s8, m8, ts, tm, v8, ts0, tm0 = r["quad"]
            o_s, o_m, o_v = (C.c_uint32 * 4)(), (C.c_int * 4)(), (C.c_uint32 * 4)()
            t_s, t_m = C.c_uint(ts0), C.c_int(tm0)
            lib.aomhip_get_var_sse_sum_8x8_quad(pa, S, pb, S, o_s, o_m, C.byref(t_s), C.byref(t_m), o_v)
            assert (list(o_s), list(o_m), t_s.value, t_m.value, list(o_v)) == (s8, m8, ts, tm, v8)
            s16, ts, tm, v16, ts0, tm0 = r["dual"]
            o_s2, o_v2 = (C.c_uint32 * 2)(), (C.c_uint32 * 2)()
            t_s, t_m = C.c_uint(ts0), C.c_int(tm0)
            lib.aomhip_get_var_sse_sum_16x16_dual(pa, S, pb, S, o_s2, C.byref(t_s), C.byref(t_m), o_v2)
            assert (list(o_s2), t_s.value, t_m.value, list(o_v2)) == (s16, ts, tm, v16)
            continue
        bd, w, h = r["bd"], r["w"], r["h"]
        a = np.ascontiguousarray(z["a%d" % bd].astype(np.uint8 if bd == 8 else np.uint16))
        b = np.ascontiguousarray(z["b%d" % bd].astype(np.uint8 if bd == 8 else np.uint16))
        S = a.shape[1]
        es = a.itemsize
        pa = a.ctypes.data + (r["oy"] * S + r["ox"]) * es
        pb = b.ctypes.data + (r["ry"] * S + r["rx"]) * es
        sse = C.c_uint()
        if bd == 8:
            assert lib.aomhip_sad(pa, S, pb, S, w, h) == r["sad"], r
            assert lib.aomhip_sad_skip(pa, S, pb, S, w, h) == r["sad_skip"], r
            if "x4d" in r:
                ptrs = (C.c_void_p * 4)(*[b.ctypes.data + y * S + x for (x, y) in r["x4d_offs"]])
                out = (C.c_uint32 * 4)()
                lib.aomhip_sad_x4d(pa, S, ptrs, S, out, w, h)
                assert list(out) == r["x4d"], r
            assert (lib.aomhip_variance(pa, S, pb, S, w, h, C.byref(sse)), sse.value) == (r["var"], r["sse"]), r
            for xo, yo, v, q in r.get("subpel", []):
                assert (lib.aomhip_sub_pixel_variance(pa, S, xo, yo, pb, S, w, h, C.byref(sse)), sse.value) == (v, q), (r, xo, yo)
        else:       # CONVERT_TO_BYTEPTR: the reference passes uint16 addresses >> 1 (aom_ports/mem.h:80)
            assert lib.aomhip_highbd_sad(pa >> 1, S, pb >> 1, S, w, h, 0) == r["sad"], r
            assert lib.aomhip_highbd_sad(pa >> 1, S, pb >> 1, S, w, h, bd) == r["sad"] >> (2 if bd == 10 else 4), r
            assert (lib.aomhip_highbd_variance(pa >> 1, S, pb >> 1, S, w, h, bd, C.byref(sse)), sse.value) == (r["var"], r["sse"]), r
            for xo, yo, v, q in r.get("subpel", []):
                got = lib.aomhip_highbd_sub_pixel_variance(pa >> 1, S, xo, yo, pb >> 1, S, w, h, bd, C.byref(sse))
                assert (got, sse.value) == (v, q), (r, xo, yo)


def test_fwd_and_inv_txfm2d_match_reference_goldens(hip, oracle, ctx):
    z, cases = load("ref_eval_txfm2d.npz")
    q = oracle.build_quantizer_y(8, 100)          # the quantiser outputs are not looked at here (tables are inputs only)
    qp = hip.capi.QuantParams.from_tables(q)
    for k, c in enumerate(cases):
        w, h, ts = c["w"], c["h"], c["tx_size"]
        nc = hip.capi.lib.aomhip_tx_max_eob(ts)
        blk = np.zeros(1, hip.capi.txb_dtype)
        blk["tx_type"] = c["tx_type"]
        res = np.ascontiguousarray(z["x%d" % k].reshape(h, w))
        d_res, d_b = ctx.to_device(res), ctx.to_device(blk)
        d_c, d_q, d_dq, d_e = ctx.malloc(nc * 4), ctx.malloc(nc * 4), ctx.malloc(nc * 4), ctx.malloc(16)
        ctx.xform_quant_batch(d_res, w, ts, d_b, 1, 0, 16 if c.get("wht") else 0, qp, c["bd"] > 8, d_c, d_q, d_dq, d_e)
        assert np.array_equal(ctx.from_device(d_c, (nc,), np.int32), z["c%d" % k][:nc]), c
        for d in (d_res, d_c, d_q, d_dq, d_e):
            ctx.free(d)
        if "inv_bd" in c:
            bd = c["inv_bd"]
            P, border = 64, 32
            pred = np.zeros((P, P), np.uint8 if bd == 8 else np.uint16)
            pred[:h, :w] = z["p%d" % k].reshape(h, w)
            p = ctx.planes_alloc(P, P, border, bd, 1)
            ctx.planes_upload(p, 0, pred)
            d_dq, d_e = ctx.to_device(np.ascontiguousarray(z["dq%d" % k][:nc])), ctx.to_device(np.asarray([c.get("eob", nc)], np.uint16))
            ctx.inv_txfm_add_batch(d_dq, ts, d_b, 1, 0, 0, d_e, p, 0)
            rec = ctx.planes_download(p, 0)[border:border + h, border:border + w]
            assert np.array_equal(rec.astype(np.uint16), z["r%d" % k]), c
            ctx.planes_free(p); ctx.free(d_dq); ctx.free(d_e)
        ctx.free(d_b)


def test_adaptive_quantiser_matches_reference_goldens(hip, ctx):
    z, cases = load("ref_eval_quant.npz")
    n = 0
    for k, c in enumerate(cases):
        if not c["adaptive"]:
            continue
        nc = c["n"]
        assert hip.capi.lib.aomhip_tx_max_eob(c["tx_size"]) == nc
        qp = hip.capi.QuantParams.from_tables({m: np.asarray(v, np.int16) for m, v in c["tables"].items()})
        blk = np.zeros(1, hip.capi.txb_dtype)
        blk["tx_type"] = c["tx_type"]
        d_c, d_b = ctx.to_device(np.ascontiguousarray(z["c%d" % k])), ctx.to_device(blk)
        d_q, d_dq, d_e = ctx.malloc(nc * 4), ctx.malloc(nc * 4), ctx.malloc(16)
        ctx.quantize_b_adaptive_batch(d_c, c["tx_size"], d_b, 1, 0, qp, bool(c["hbd"]), d_q, d_dq, d_e)
        got = (ctx.from_device(d_q, (nc,), np.int32), ctx.from_device(d_dq, (nc,), np.int32), int(ctx.from_device(d_e, (1,), np.uint16)[0]))
        assert np.array_equal(got[0], z["q%d" % k]) and np.array_equal(got[1], z["d%d" % k]) and got[2] == c["eob"], c
        for d in (d_c, d_b, d_q, d_dq, d_e):
            ctx.free(d)
        n += 1
    assert n >= 250


def test_deblock_edges_match_reference_goldens(hip, ctx):
    z, cases = load("ref_eval_lpf.npz")
    N, border = 24, 32
    planes = {bd: ctx.planes_alloc(N, N, border, bd, 1) for bd in (8, 10, 12)}
    d_params = ctx.malloc(6 * 6 * 4)
    for k, c in enumerate(cases):
        bd = c["bd"]
        pix = np.ascontiguousarray(z["i%d" % k].astype(np.uint8 if bd == 8 else np.uint16))
        params = np.zeros((N // 4, N // 4, 4), np.uint8)
        uy, ux = c["y"] // 4, c["x"] // 4
        if c["vertical"]:
            params[uy, ux, 0], params[uy, ux, 1] = c["len"], c["level"]
        else:
            params[uy, ux, 2], params[uy, ux, 3] = c["len"], c["level"]
        p = planes[bd]
        ctx.planes_upload(p, 0, pix)
        hip.capi.check(hip.capi.lib.aomhip_memcpy_h2d(ctx.h, d_params, params.ctypes.data, params.nbytes))
        ctx.deblock_plane(p, 0, d_params, N // 4, c["sharp"], 3)
        out = ctx.planes_download(p, 0)[border:border + N, border:border + N]
        assert np.array_equal(out.astype(np.uint16), z["o%d" % k]), c
    for p in planes.values():
        ctx.planes_free(p)
    ctx.free(d_params)


def test_cdef_filter_blocks_match_reference_goldens(hip, ctx):
    z, cases = load("ref_eval_cdef_fb.npz")
    border = 32
    for k, c in enumerate(cases):
        bd, xdec, ydec = c["bd"], c["xdec"], c["ydec"]
        luma = z["luma%d" % bd]
        plane = np.ascontiguousarray((luma if not c["pli"] else luma[::(1 << ydec), ::(1 << xdec)]).astype(np.uint8 if bd == 8 else np.uint16))
        ph_, pw_ = plane.shape
        fby, fbx = (0, 0) if c["at_edge"] else (1, 1)
        skip = np.ones((24, 24), np.uint8)
        skip[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8] = z["s%d" % k]
        pri, sec = np.full((3, 3), c["level"], np.uint8), np.full((3, 3), c["sec"], np.uint8)
        src, dst = ctx.planes_alloc(pw_, ph_, border, bd, 1), ctx.planes_alloc(pw_, ph_, border, bd, 1)
        ctx.planes_upload(src, 0, plane)
        d_pri, d_sec, d_skip = ctx.to_device(pri), ctx.to_device(sec), ctx.to_device(skip)
        if not c["pli"]:
            d_dir, d_var = ctx.malloc(24 * 24), ctx.malloc(24 * 24 * 4)
            ctx.cdef_luma_plane(src, 0, dst, 0, d_pri, d_sec, 3, d_skip, c["damping"], d_dir, d_var)
            d = ctx.from_device(d_dir, (24, 24), np.uint8)[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8]
            v = ctx.from_device(d_var, (24, 24), np.int32)[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8]
            keep = z["s%d" % k] == 0
            assert np.array_equal(d[keep], z["d%d" % k][keep].astype(np.uint8)) and np.array_equal(v[keep], z["v%d" % k][keep]), c
            ctx.free(d_dir); ctx.free(d_var)
        else:
            ld = np.zeros((24, 24), np.uint8)
            ld[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8] = z["ld%d" % k]
            d_ld = ctx.to_device(ld)
            ctx.cdef_chroma_plane(src, 0, dst, 0, xdec, ydec, d_ld, d_pri, d_sec, 3, d_skip, c["damping"])
            ctx.free(d_ld)
        out = ctx.planes_download(dst, 0)[border + c["y0"]:border + c["y0"] + c["ph"], border + c["x0"]:border + c["x0"] + c["pw"]]
        assert np.array_equal(out.astype(np.uint16), z["o%d" % k]), c
        for d in (d_pri, d_sec, d_skip):
            ctx.free(d)
        ctx.planes_free(src); ctx.planes_free(dst)
