"""The CPU oracle against golden vectors produced by interpreting the REFERENCE'S OWN C functions.

tests/golden/gen_ref_eval_golden.py evaluated aom_dsp/quantize.c, aom_dsp/loopfilter.c, av1/common/cdef_block.c,
aom_dsp/sad.c, aom_dsp/variance.c and aom_dsp/subtract.c where they lie (tests/golden/ref_c_eval.py) and stored
inputs + outputs; here the from-scratch restatement in oracle/ must reproduce every output bit for bit.  This
is what pins the oracle for the functions whose reference gtests are only SIMD-vs-C comparisons.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name))
    return z, json.loads(bytes(z["cases"]).decode())


def test_quantize_b_matches_reference_evaluation(oracle):
    z, cases = load("ref_eval_quant.npz")
    assert len(cases) > 500
    for k, c in enumerate(cases):
        q = {m: np.asarray(v, np.int16) for m, v in c["tables"].items()}
        scan, iscan = oracle.get_scan(c["tx_size"], c["tx_type"])
        coeff = z["c%d" % k]
        if c["adaptive"]:
            qc, dq, eob = oracle.quantize_b_adaptive(coeff, q, scan, c["log_scale"], highbd=bool(c["hbd"]))
        else:
            qc, dq, eob = oracle.quantize_b(coeff, q, scan, iscan, c["log_scale"], highbd=bool(c["hbd"]))
        assert eob == c["eob"], (k, c["fn"], c["kind"], c["qindex"])
        assert np.array_equal(qc, z["q%d" % k]), (k, c["fn"], c["kind"], c["qindex"])
        assert np.array_equal(dq, z["d%d" % k]), (k, c["fn"], c["kind"], c["qindex"])


def test_block_error_matches_reference_evaluation(oracle):
    z, _ = load("ref_eval_quant.npz")
    f = oracle.lib.orc_block_error
    f.restype = C.c_int64
    rows = z["block_error"]
    assert len(rows) >= 90
    for row in rows:
        kk = int(row[0])
        c, dq = np.ascontiguousarray(z["c%d" % kk], np.int32), np.ascontiguousarray(z["d%d" % kk], np.int32)
        ssz = C.c_int64()
        for j, bd in enumerate((0, 8, 10, 12)):
            e = f(C.c_void_p(c.ctypes.data), C.c_void_p(dq.ctypes.data), C.c_ssize_t(c.size), C.byref(ssz), bd)
            assert (e, ssz.value) == (int(row[1 + 2 * j]), int(row[2 + 2 * j])), (kk, bd)


def test_quantize_fp_matches_reference_evaluation(oracle):
    z, cases = load("ref_eval_quant.npz")
    f = oracle.lib.orc_quantize_fp
    f.restype = None
    rows = z["quantize_fp"]
    assert len(rows) >= 130
    for kk, eob, r0, r1, q0, q1 in rows.tolist():
        c = cases[kk]
        coeff = np.ascontiguousarray(z["c%d" % kk], np.int32)
        scan, _ = oracle.get_scan(c["tx_size"], c["tx_type"])
        scan = np.ascontiguousarray(scan, np.int16)
        rfp, qfp, deq = np.asarray([r0, r1], np.int16), np.asarray([q0, q1], np.int16), np.asarray(c["tables"]["dequant"], np.int16)
        qc, dq = np.zeros_like(coeff), np.zeros_like(coeff)
        e = C.c_uint16()
        f(C.c_void_p(coeff.ctypes.data), C.c_ssize_t(coeff.size), C.c_void_p(rfp.ctypes.data), C.c_void_p(qfp.ctypes.data),
          C.c_void_p(qc.ctypes.data), C.c_void_p(dq.ctypes.data), C.c_void_p(deq.ctypes.data), C.byref(e), C.c_void_p(scan.ctypes.data),
          c["log_scale"], c["hbd"])
        assert e.value == eob and np.array_equal(qc, z["fq%d" % kk]) and np.array_equal(dq, z["fd%d" % kk]), (kk, c["fn"])


def test_lpf_matches_reference_evaluation(oracle):
    z, cases = load("ref_eval_lpf.npz")
    assert len(cases) >= 700
    changed = 0
    for k, c in enumerate(cases):
        inp, want = z["i%d" % k], z["o%d" % k]
        px = np.ascontiguousarray(inp.astype(np.uint8 if c["bd"] == 8 else np.uint16))
        oracle.lpf_edge(px, c["y"], c["x"], c["vertical"], c["len"], c["blimit"], c["limit"], c["thresh"], bd=c["bd"])
        assert np.array_equal(px.astype(np.uint16), want), (k, c)
        changed += int(not np.array_equal(inp, want))
    assert changed > len(cases) // 3      # the vectors do exercise the filters, not only the masks


def test_cdef_find_dir_matches_reference_evaluation(oracle):
    z, _ = load("ref_eval_cdef.npz")
    for img, (bd, d, var) in zip(z["find_dir_in"], z["find_dir_out"]):
        got = oracle.cdef_find_dir(img, coeff_shift=int(bd) - 8)
        assert got == (int(d), int(var))


def test_cdef_filter_block_matches_reference_evaluation(oracle):
    z, cases = load("ref_eval_cdef.npz")
    f = oracle.lib.orc_cdef_filter_block
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] + [C.c_void_p] + [C.c_int] * 10
    differs = 0
    for k, c in enumerate(cases):
        assert c["bstride"] == 144
        tile = np.ascontiguousarray(z["t%d" % k], np.uint16)
        want = z["f%d" % k]
        use16 = "_16_" in c["fn"]
        dst = np.zeros((c["bh"], c["bw"]), np.uint16 if use16 else np.uint8)
        src = tile.ctypes.data + (3 * 144 + 8) * 2
        f(None if use16 else dst.ctypes.data, dst.ctypes.data if use16 else None, c["bw"], src, c["pri"], c["sec"], c["dir"],
          c["pri_damping"], c["sec_damping"], c["coeff_shift"], c["bw"], c["bh"], int(c["variant"] in (0, 1)), int(c["variant"] in (0, 2)))
        assert np.array_equal(dst.astype(np.uint16), want), (k, c)
        differs += int(not np.array_equal(want, tile[3:3 + c["bh"], 8:8 + c["bw"]]))
    assert differs > len(cases) // 2


def test_sad_variance_match_reference_evaluation(oracle):
    z, rows = load("ref_eval_sadvar.npz")
    assert len(rows) >= 60
    for r in rows:
        if r.get("extra"):      # aom_mse / aom_get*var / 8x8 quad / 16x16 dual: all `variance()` with the sum kept
            a, b = np.ascontiguousarray(z["a8"].astype(np.uint8)), np.ascontiguousarray(z["b8"].astype(np.uint8))
            var_at = lambda dx, w, h: oracle.variance(a, r["oy"], r["ox"] + dx, b, r["ry"], r["rx"] + dx, w, h, bd=8)
            for (w, h) in ((16, 16), (16, 8), (8, 16), (8, 8)):
                assert r["mse%dx%d" % (w, h)] == [var_at(0, w, h)[1]] * 2
            for n in (8, 16):
                v, sse, sm = var_at(0, n, n)
                assert r["get%dvar" % n] == [sse, sm]
            s8, m8, ts, tm, v8, ts0, tm0 = r["quad"]
            got = [var_at(8 * k, 8, 8) for k in range(4)]
            assert ([g[1] for g in got], [g[2] for g in got], [g[0] for g in got]) == (s8, m8, v8)
            assert (ts, tm) == (ts0 + sum(s8), tm0 + sum(m8))
            s16, ts, tm, v16, ts0, tm0 = r["dual"]
            got = [var_at(16 * k, 16, 16) for k in range(2)]
            assert ([g[1] for g in got], [g[0] for g in got]) == (s16, v16) and (ts, tm) == (ts0 + sum(s16), tm0 + sum(g[2] for g in got))
            continue
        bd, w, h = r["bd"], r["w"], r["h"]
        a = np.ascontiguousarray(z["a%d" % bd].astype(np.uint8 if bd == 8 else np.uint16))
        b = np.ascontiguousarray(z["b%d" % bd].astype(np.uint8 if bd == 8 else np.uint16))
        # the fixture holds the raw aom_highbd_sad value; bd=8 in the oracle call selects "no wrapper shift"
        assert oracle.sad(a, r["oy"], r["ox"], b, r["ry"], r["rx"], w, h, bd=8) == r["sad"], r
        assert oracle.sad(a, r["oy"], r["ox"], b, r["ry"], r["rx"], w, h, skip=True, bd=8) == r["sad_skip"], r
        if "x4d" in r:
            for (x, y), want in zip(r["x4d_offs"], r["x4d"]):
                assert oracle.sad(a, r["oy"], r["ox"], b, y, x, w, h, bd=8) == want
        v, sse, _ = oracle.variance(a, r["oy"], r["ox"], b, r["ry"], r["rx"], w, h, bd=bd)
        assert (v, sse) == (r["var"], r["sse"]), r
        for xo, yo, want_v, want_sse in r.get("subpel", []):
            assert oracle.sub_pixel_variance(a, r["oy"], r["ox"], xo, yo, b, r["ry"], r["rx"], w, h, bd=bd) == (want_v, want_sse), (r, xo, yo)
        if "subtract_sum" in r:
            diff = np.zeros((h, w), np.int16)
            fn = oracle.lib.orc_subtract_block if bd == 8 else oracle.lib.orc_highbd_subtract_block
            fn.restype = None
            fn(h, w, C.c_void_p(diff.ctypes.data), C.c_ssize_t(w), C.c_void_p(oracle._addr(a, r["oy"], r["ox"])), C.c_ssize_t(a.shape[1]),
               C.c_void_p(oracle._addr(b, r["ry"], r["rx"])), C.c_ssize_t(b.shape[1]))
            assert int(np.sum(diff.astype(np.int64).ravel() * (np.arange(w * h) % 251 + 1))) == r["subtract_sum"]
            sp = np.random.default_rng(r["second_pred_seed"]).integers(0, (1 << bd), w * h).astype(a.dtype).reshape(1, h, w)
            cand = np.zeros(1, dtype=[("sx", "<i2"), ("sy", "<i2"), ("rx", "<i2"), ("ry", "<i2")])
            cand["sx"], cand["sy"], cand["rx"], cand["ry"] = r["ox"], r["oy"], r["rx"], r["ry"]
            got = oracle.sad_avg_batch(a, b, 0, w, h, cand, sp, [0], bd=8)
            assert int(got[0]) == r["sad_avg"], r


# ---- motion search: av1/encoder/mcomp.c interpreted (tests/golden/gen_ref_eval_mcomp.py)

def load_mcomp():
    z = np.load(os.path.join(GOLD, "ref_eval_mcomp.npz"))
    meta = json.loads(bytes(z["cases"]).decode())
    return z, meta


def blocks_of(oracle_or_dtype, blk):
    dt = np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")])
    b = np.zeros(1, dt)
    for n, v in zip(dt.names, blk):
        b[n] = v
    return b


def test_fullpel_diamond_matches_reference_evaluation(oracle):
    z, meta = load_mcomp()
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "diamond":
            continue
        mv, cost = oracle.fullpel_diamond_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], blocks_of(oracle, c["block"]),
                                                clamped=int(c["method"] == "CLAMPED_DIAMOND"), step_param=c["step_param"], cost_type=c["cost_type"],
                                                bd=c["bd"], threads=1)
        assert (list(map(int, mv[0])), int(cost[0])) == (c["mv"], c["cost"]), c
        n += 1
    assert n >= 70


def test_mesh_search_matches_reference_evaluation(oracle):
    z, meta = load_mcomp()
    n = 0
    for c in meta["cases"]:
        if c["kind"] != "mesh":
            continue
        mv, cost = oracle.mesh_search_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], blocks_of(oracle, c["block"]),
                                            c["mesh"], fine_search_interval=c["fine_interval"], cost_type=c["cost_type"], bd=c["bd"], threads=1)
        assert (list(map(int, mv[0])), int(cost[0])) == (c["mv"], c["cost"]), c
        n += 1
    assert n >= 12


def test_subpel_tree_matches_reference_evaluation(oracle):
    """av1_find_best_sub_pixel_tree{_pruned_more,_pruned,} (bilinear): with and without a cost list, every MV cost type."""
    z, meta = load_mcomp()
    n, with_cl = 0, 0
    for c in meta["cases"]:
        if c["kind"] != "subpel":
            continue
        blk = list(c["block"])
        blk[2], blk[3] = c["fullpel_mv"][0] * 8, c["fullpel_mv"][1] * 8      # starts from the full-pel optimum (1/8 pel units)
        blk[6:10] = c["subpel_limits"]                                        # SubpelMvLimits from av1_set_subpel_mv_search_range
        tree = {"av1_find_best_sub_pixel_tree_pruned_more": "pruned_more", "av1_find_best_sub_pixel_tree_pruned": "pruned",
                "av1_find_best_sub_pixel_tree": "tree"}[c["fn"]]
        args = (z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"], blocks_of(oracle, blk))
        mv, err, dist, sse = oracle.subpel_tree_batch(*args, tree=tree, cost_type=c["cost_type"], error_per_bit=c["error_per_bit"],
                                                      mvjcost=z["mvjcost"], mvcost0=z["mvcost0"], mvcost1=z["mvcost1"], iters=c["iters"],
                                                      allow_hp=c["allow_hp"], forced_stop=c["forced_stop"],
                                                      cost_lists=[c["cost_list"]] if "cost_list" in c else None, bd=c["bd"], threads=1,
                                                      subpel_search_type=c.get("subpel_search_type", 0))
        assert (list(map(int, mv[0])), int(err[0]), int(dist[0]), int(sse[0])) == (c["mv"], c["err"], c["distortion"], c["sse"]), c
        if tree == "pruned_more" and "cost_list" not in c and c["cost_type"] != 0:     # the older entry point agrees
            got = oracle.subpel_bilinear_batch(*args, cost_type=c["cost_type"], iters=c["iters"], allow_hp=c["allow_hp"],
                                               forced_stop=c["forced_stop"], bd=c["bd"], threads=1)
            assert np.array_equal(got[0], mv) and int(got[1][0]) == c["err"]
        n += 1
        with_cl += "cost_list" in c
    assert n >= 58 and with_cl >= 24
    assert sum(c.get("subpel_search_type", 0) == 3 for c in meta["cases"]) >= 10      # the 8-tap (up-sampled prediction) tree


def test_search_site_tables_match_reference_evaluation(oracle):
    _, meta = load_mcomp()
    for method, ref in meta["sites"].items():
        ns, per, rad, mv = oracle.search_sites(method)
        assert ns == ref["num_search_steps"], method
        first = ref["first_stage"]
        lo = 1 if method in ("DIAMOND", "CLAMPED_DIAMOND", "NSTEP", "NSTEP_8PT", "NSTEP_FPF") else 0
        for i in range(ns):
            st = i + first
            assert per[st] == ref["searches_per_step"][i] and rad[st] == ref["radius"][i], (method, st)
            assert mv[st, lo:lo + per[st]].tolist() == ref["mv"][i], (method, st)


def test_full_pixel_search_matches_reference_evaluation(oracle):
    """av1_full_pixel_search: all 11 methods x {entropy, L1, none} costs, cost lists, second-best MVs, mesh follow-ups
    (forced / pruned), the downsampled-SAD re-check -- and full_pixel_diamond's cost list."""
    z, meta = load_mcomp()
    n = 0
    for c in meta["cases"]:
        if c["kind"] not in ("search", "diamond"):
            continue
        q = oracle.search_params(c["method"], c["step_param"], c["cost_type"], c.get("sad_per_bit", 20), c.get("error_per_bit", 60),
                                 c.get("skip_sad", False), c.get("run_mesh", 0), c.get("prune_mesh", 0), c.get("mesh_diff_thr", 0),
                                 c.get("force_mesh_thresh", 2147483647), c.get("fine_interval", 0), c.get("mesh"))
        mv, cost, cl, sec = oracle.full_pixel_search_batch(z["src%d" % c["bd"]], z["ref%d" % c["bd"]], meta["border"], c["w"], c["h"],
                                                           blocks_of(oracle, c["block"]), q, z["mvjcost"], z["mvcost0"], z["mvcost1"],
                                                           bd=c["bd"], threads=1)
        got = (mv[0].tolist(), int(cost[0]), cl[0].tolist())
        assert got == (c["mv"], c["cost"], c["cost_list"]), (c, got)
        if c.get("second_best") is not None and c["kind"] == "search":
            assert sec[0].tolist() == c["second_best"], (c, sec[0].tolist())
        n += 1
    assert n >= 180


# ---- 2-D transforms, threshold / quantiser tables, whole CDEF filter blocks (tests/golden/gen_ref_eval_more.py)

def test_txfm2d_matches_reference_evaluation(oracle):
    z, cases = load("ref_eval_txfm2d.npz")
    assert len(cases) >= 340 and len({c["tx_size"] for c in cases}) == 19 and sum(c.get("wht", 0) for c in cases) == 24
    n_inv = 0
    for k, c in enumerate(cases):
        w, h = c["w"], c["h"]
        if c.get("wht"):            # lossless Walsh-Hadamard pair
            out = np.zeros(16, np.int32)
            xin = np.ascontiguousarray(z["x%d" % k].reshape(4, 4))
            oracle.lib.orc_fwht4x4(C.c_void_p(xin.ctypes.data), C.c_void_p(out.ctypes.data), 4)
            assert np.array_equal(out, z["c%d" % k]), c
            dst = np.ascontiguousarray(z["p%d" % k].reshape(4, 4).astype(np.uint16))
            dq = np.ascontiguousarray(z["dq%d" % k])
            oracle.lib.orc_iwht4x4_add(C.c_void_p(dq.ctypes.data), C.c_void_p(dst.ctypes.data), 4, c["eob"], c["inv_bd"])
            assert np.array_equal(dst, z["r%d" % k]), c
            n_inv += 1
            continue
        got = oracle.fwd_txfm2d(z["x%d" % k].reshape(h, w), c["tx_size"], c["tx_type"], c["bd"])
        nn = min(w, 32) * min(h, 32)      # 64-point sizes: only the re-packed 32 low frequencies are defined (av1_fwd_txfm2d.c:241-312)
        assert np.array_equal(got[:nn], z["c%d" % k][:nn]), c
        if "inv_bd" in c:
            rec = oracle.inv_txfm2d_add(z["dq%d" % k][:nn], z["p%d" % k].reshape(h, w), c["tx_size"], c["tx_type"], c["inv_bd"])
            assert np.array_equal(rec, z["r%d" % k]), c
            n_inv += 1
    assert n_inv >= 200


def test_lpf_thresholds_and_quantizer_tables_match_reference_evaluation(oracle):
    z, _ = load("ref_eval_tables.npz")
    th = z["lpf_thresholds"]
    for sharp in range(8):
        for lvl in range(64):
            a, b, c = C.c_uint8(), C.c_uint8(), C.c_uint8()
            oracle.lib.orc_lpf_thresholds(lvl, sharp, C.byref(a), C.byref(b), C.byref(c))
            assert [a.value, b.value, c.value] == th[sharp, lvl].tolist(), (sharp, lvl)
    for bd in (8, 10, 12):
        tab = z["quant_y_bd%d_0" % bd]          # [qindex, {zbin, round, quant, quant_shift, dequant}, 8 lanes]
        for qindex in range(256):
            q = oracle.build_quantizer_y(bd, qindex)
            for f, name in enumerate(("zbin", "round", "quant", "quant_shift", "dequant")):
                assert q[name].tolist() == tab[qindex, f, :2].tolist(), (bd, qindex, name)
                assert (tab[qindex, f, 2:] == tab[qindex, f, 1]).all()     # lanes 2..7 replicate the AC entry
    # the delta_q planes: dc/ac lookups at shifted indices (orc_dc_q / orc_ac_q are what a caller builds U/V tables from)
    oracle.lib.orc_dc_q.restype = oracle.lib.orc_ac_q.restype = C.c_int16
    ydc, udc, uac, vdc, vac = -7, 5, -3, 9, 12
    for plane, (ddc, dac) in (("y", (ydc, 0)), ("u", (udc, uac)), ("v", (vdc, vac))):
        tab = z["quant_%s_bd8_d" % plane]
        for qindex in (0, 1, 17, 128, 250, 255):
            assert tab[qindex, 4, 0] == oracle.lib.orc_dc_q(qindex, ddc, 8) and tab[qindex, 4, 1] == oracle.lib.orc_ac_q(qindex, dac, 8)


def test_cdef_filter_block_64x64_matches_reference_evaluation(oracle):
    z, cases = load("ref_eval_cdef_fb.npz")
    assert len(cases) >= 40
    for k, c in enumerate(cases):
        bd, xdec, ydec = c["bd"], c["xdec"], c["ydec"]
        luma = z["luma%d" % bd]
        plane = np.ascontiguousarray((luma if not c["pli"] else luma[::(1 << ydec), ::(1 << xdec)]).astype(np.uint8 if bd == 8 else np.uint16))
        fby, fbx = (0, 0) if c["at_edge"] else (1, 1)
        skip = np.ones((24, 24), np.uint8)                                  # only the filter block under test is processed
        skip[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8] = z["s%d" % k]
        pri, sec = np.full((3, 3), c["level"], np.uint8), np.full((3, 3), c["sec"], np.uint8)
        ys, xs = slice(c["y0"], c["y0"] + c["ph"]), slice(c["x0"], c["x0"] + c["pw"])
        if not c["pli"]:
            out, d, v = oracle.cdef_plane_luma(plane, pri, sec, skip, c["damping"], bd)
            keep = z["s%d" % k] == 0
            assert np.array_equal(d[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8][keep], z["d%d" % k][keep].astype(np.uint8)), c
            assert np.array_equal(v[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8][keep], z["v%d" % k][keep]), c
        else:
            ld = np.zeros((24, 24), np.uint8)
            ld[fby * 8:fby * 8 + 8, fbx * 8:fbx * 8 + 8] = z["ld%d" % k]
            out = oracle.cdef_plane_chroma(plane, xdec, ydec, ld, pri, sec, skip, c["damping"], bd)
        assert np.array_equal(out[ys, xs].astype(np.uint16), z["o%d" % k]), c


def test_compound_masked_obmc_match_reference_evaluation(oracle):
    """svaf / jsvaf / msvf / msdf / osdf / ovf / osvf (oracle/aomref_compound.c) against the interpreted
    aom_[highbd_N_]{sub_pixel_avg,dist_wtd_sub_pixel_avg,masked_sub_pixel,obmc[_sub_pixel]}_variance and
    aom_[highbd_]{masked,obmc}_sad functions (aom_dsp/variance.c, aom_dsp/sad_av1.c), 8 / 10 / 12-bit."""
    z, cases = load("ref_eval_compound.npz")
    assert len(cases) >= 25
    lib = oracle.lib
    lib.orc_compound_sub_pixel_variance.restype = C.c_uint32
    lib.orc_masked_sad.restype = lib.orc_obmc_sad.restype = C.c_uint
    lib.orc_obmc_variance.restype = C.c_uint32
    q = C.c_uint32()
    checked = 0
    for c in cases:
        bd, w, h, k = c["bd"], c["w"], c["h"], c["k"]
        e16 = int(bd > 8 or c.get("hbd8", 0))
        dt = np.uint16 if e16 else np.uint8
        a, b = np.ascontiguousarray(z["a%d" % bd], dt), np.ascontiguousarray(z["b%d" % bd], dt)
        S = a.shape[1]
        A = C.c_void_p(a.ctypes.data + (c["ay"] * S + c["ax"]) * a.itemsize)
        B = C.c_void_p(b.ctypes.data + (c["by"] * S + c["bx"]) * b.itemsize)
        sp = np.ascontiguousarray(z["sp%d" % k], dt)
        mask, ms = np.ascontiguousarray(z["mask%d" % k]), c["mask_stride"]
        ws, om = np.ascontiguousarray(z["ws%d" % k]), np.ascontiguousarray(z["om%d" % k])
        SP, M, WS, OM = (C.c_void_p(x.ctypes.data) for x in (sp, mask, ws, om))
        for xo, yo, v, sse in c.get("svaf", []):
            got = lib.orc_compound_sub_pixel_variance(A, S, xo, yo, B, S, w, h, e16, bd, 0, SP, 0, 0, None, 0, 0, C.byref(q))
            assert (got, q.value) == (v, sse), ("svaf", c)
            checked += 1
        for xo, yo, fwd, bck, v, sse in c.get("jsvaf", []):
            got = lib.orc_compound_sub_pixel_variance(A, S, xo, yo, B, S, w, h, e16, bd, 1, SP, fwd, bck, None, 0, 0, C.byref(q))
            assert (got, q.value) == (v, sse), ("jsvaf", c)
            checked += 1
        for xo, yo, inv, v, sse in c.get("msvf", []):
            got = lib.orc_compound_sub_pixel_variance(A, S, xo, yo, B, S, w, h, e16, bd, 2, SP, 0, 0, M, ms, inv, C.byref(q))
            assert (got, q.value) == (v, sse), ("msvf", c)
            checked += 1
        for inv, v in c.get("msdf", []):
            # the fixture holds the raw kernel value; bd = 8 asks the oracle for no wrapper shift
            assert lib.orc_masked_sad(B, S, A, S, SP, M, ms, inv, w, h, e16, 8) == v, ("msdf", c)
            for wbd, sh in ((10, 2), (12, 4)):
                if e16:
                    assert lib.orc_masked_sad(B, S, A, S, SP, M, ms, inv, w, h, e16, wbd) == v >> sh
            checked += 1
        if "osdf" in c:
            assert lib.orc_obmc_sad(A, S, WS, OM, w, h, e16, 8) == c["osdf"], ("osdf", c)
            checked += 1
        got = lib.orc_obmc_variance(A, S, 0, 0, 0, WS, OM, w, h, e16, bd, C.byref(q))
        assert [got, q.value] == c["ovf"], ("ovf", c)
        for xo, yo, v, sse in c["osvf"]:
            got = lib.orc_obmc_variance(A, S, 1, xo, yo, WS, OM, w, h, e16, bd, C.byref(q))
            assert (got, q.value) == (v, sse), ("osvf", c)
            checked += 1
    assert checked > 300


def test_convolve_sr_matches_reference_evaluation(oracle):
    """orc_convolve_sr (the single-reference sub-pel interpolation of av1_enc_build_inter_predictor) against the interpreted
    av1_[highbd_]convolve_2d_facade: copy / x_sr / y_sr / 2d_sr, the four interpolation filters, the 4-tap sets for a
    dimension <= 4, round_0 / round_1 of 8 / 10 / 12-bit (av1/common/convolve.c, convolve.h:63-100, filter.h)."""
    z, cases = load("ref_eval_convolve.npz")
    assert len(cases) >= 200
    f = oracle.lib.orc_convolve_sr
    f.restype = None
    kinds = set()
    for c in cases:
        bd, w, h = c["bd"], c["w"], c["h"]
        e16 = int(bd > 8)
        dt = np.uint16 if e16 else np.uint8
        p = np.ascontiguousarray(z["p%d" % bd], dt)
        S = p.shape[1]
        dst = np.zeros((h, w), dt)
        f(C.c_void_p(p.ctypes.data + (c["y0"] * S + c["x0"]) * p.itemsize), S, C.c_void_p(dst.ctypes.data), w, w, h, c["fx"], c["fy"],
          c["sx"], c["sy"], e16, bd)
        assert np.array_equal(dst.ravel(), z["d%d" % c["k"]]), c
        kinds.add((bool(c["sx"]), bool(c["sy"])))
    assert len(kinds) == 4


def test_rd_helpers_match_reference_evaluation(oracle):
    """orc_sse, orc_hadamard (+ satd) and orc_txb_init_levels against the interpreted aom_[highbd_]sse_c, aom_hadamard_*_c /
    aom_hadamard_lp_*_c / aom_highbd_hadamard_*_c + aom_satd[_lp]_c (aom_dsp/sse.c, aom_dsp/avg.c) and av1_txb_init_levels_c
    (av1/encoder/encodetxb.c), including full-range int16 residuals that wrap the 16-bit intermediates."""
    z, cases = load("ref_eval_rdhelp.npz")
    lib = oracle.lib
    lib.orc_sse.restype = C.c_int64
    lib.orc_hadamard.restype = C.c_int
    lib.orc_txb_init_levels.restype = None
    seen = set()
    for c in cases:
        seen.add(c["kind"])
        if c["kind"] == "sse":
            e16 = int(c["bd"] > 8)
            dt = np.uint16 if e16 else np.uint8
            a, b = np.ascontiguousarray(z["sa%d" % c["bd"]], dt), np.ascontiguousarray(z["sb%d" % c["bd"]], dt)
            S = a.shape[1]
            got = lib.orc_sse(C.c_void_p(a.ctypes.data + (c["oy"] * S + c["ox"]) * a.itemsize), S, C.c_void_p(b.ctypes.data + (3 * S + 2) * b.itemsize), S,
                              c["w"], c["h"], e16)
            assert got == c["value"], c
        elif c["kind"] == "hadamard":
            r = np.ascontiguousarray(z["r%d" % c["k"]])
            n = c["n"]
            out = np.zeros(n * n, np.int32)
            satd = lib.orc_hadamard(C.c_void_p(r.ctypes.data + (c["y"] * r.shape[1] + c["x"]) * 2), C.c_ssize_t(r.shape[1]), n, c["flavour"],
                                    C.c_void_p(out.ctypes.data))
            assert np.array_equal(out, z["c%d" % c["k"]]), c
            assert satd == c["satd"], c
        else:
            coeff = np.ascontiguousarray(z["tc%d" % c["k"]])
            want = z["tl%d" % c["k"]]
            lv = np.full(want.size, 0xAA, np.uint8)
            lib.orc_txb_init_levels(C.c_void_p(coeff.ctypes.data), c["w"], c["h"], C.c_void_p(lv.ctypes.data))
            assert np.array_equal(lv, want), c
    assert seen == {"sse", "hadamard", "levels"}


def test_cdef_search_distortion_matches_reference_evaluation(oracle):
    """oracle.cdef_search_sse_luma (orc_cdef_plane_luma per strength + the squared error over the non-skip 8x8 units)
    against the interpreted get_filt_error (av1/encoder/pickcdef.c:401-501): 8-bit build path (aom_sse on whole or
    partial filter blocks) and high-bit-depth path (compute_cdef_dist_highbd, >> 2 * coeff_shift)."""
    from cdef_search_fixture import load_cases, mapped_strengths, planes_of
    z, cases = load_cases()
    assert len(cases) == 6 and sum(c["pli"] for c in cases) == 2
    for c in cases:
        recon, source, skip, fb, ldir = planes_of(z, c)
        if c["pli"]:
            got = oracle.cdef_search_sse_chroma(recon, source, 1, 1, ldir, mapped_strengths(c), skip, c["damping"], c["bd"])
        else:
            got = oracle.cdef_search_sse_luma(recon, source, mapped_strengths(c), skip, c["damping"], c["bd"])
        shift = 2 * (c["bd"] - 8)
        assert [int(v) >> shift for v in got[:, fb[0], fb[1]]] == c["errors"], (c["variant"], c["pli"])
        others = got.copy()
        others[:, fb[0], fb[1]] = 0
        assert not others.any()


def test_wiener_stats_match_reference_evaluation(oracle):
    """orc_compute_stats against the interpreted av1_compute_stats_c (full and down-sampled rows) and
    av1_compute_stats_highbd_c (10 / 12-bit dividers) of av1/encoder/pickrst.c, 7x7 and 5x5 windows."""
    z, cases = load("ref_eval_lrstats.npz")
    assert len(cases) == 8
    f = oracle.lib.orc_compute_stats
    f.restype = None
    for c in cases:
        bd, win = c["bd"], c["win"]
        e16 = int(bd > 8)
        dt = np.uint16 if e16 else np.uint8
        dgd, src = np.ascontiguousarray(z["dgd%d" % bd], dt), np.ascontiguousarray(z["src%d" % bd], dt)
        hs, he, vs, ve = c["rect"]
        M, H = np.zeros(win * win, np.int64), np.zeros(win ** 4, np.int64)
        f(win, C.c_void_p(dgd.ctypes.data), C.c_void_p(src.ctypes.data), hs, he, vs, ve, dgd.shape[1], src.shape[1], e16, bd, c["downsample"],
          C.c_void_p(M.ctypes.data), C.c_void_p(H.ctypes.data))
        assert np.array_equal(M, z["M%d" % c["k"]]) and np.array_equal(H, z["H%d" % c["k"]]), c


def test_compound_convolve_matches_reference_evaluation(oracle):
    """orc_convolve_compound against the interpreted compound path of av1_[highbd_]convolve_2d_facade (first reference into
    the CONV_BUF, second averaged in; plain and distance-weighted; copy / x / y / 2-D kernels of both references mixed)."""
    z, cases = load("ref_eval_convolve_compound.npz")
    assert len(cases) >= 40
    f = oracle.lib.orc_convolve_compound
    f.restype = None
    kinds = set()
    for c in cases:
        bd, w, h = c["bd"], c["w"], c["h"]
        e16 = int(bd > 8)
        dt = np.uint16 if e16 else np.uint8
        p0, p1 = np.ascontiguousarray(z["p%d_0" % bd], dt), np.ascontiguousarray(z["p%d_1" % bd], dt)
        S = p0.shape[1]
        (x0, y0), (x1, y1) = c["pos"]
        (sx0, sy0), (sx1, sy1) = c["subs"]
        wts = c["weights"] or (0, 0)
        dst = np.zeros((h, w), dt)
        f(C.c_void_p(p0.ctypes.data + (y0 * S + x0) * p0.itemsize), S, sx0, sy0, C.c_void_p(p1.ctypes.data + (y1 * S + x1) * p1.itemsize), S, sx1, sy1,
          C.c_void_p(dst.ctypes.data), w, w, h, c["fx"], c["fy"], wts[0], wts[1], e16, bd)
        assert np.array_equal(dst.ravel(), z["d%d" % c["k"]]), c
        kinds.add((bool(sx0), bool(sy0)))
    assert len(kinds) == 4


def test_masked_compound_matches_reference_evaluation(oracle):
    """orc_convolve_compound_mask against the interpreted compound convolves + aom_[lowbd|highbd]_blend_a64_d16_mask_c
    (aom_dsp/blend_a64_mask.c), mask at plane resolution and at 2x (2x2 mean / pair average)."""
    z, cases = load("ref_eval_convolve_masked.npz")
    assert len(cases) >= 20
    f = oracle.lib.orc_convolve_compound_mask
    f.restype = None
    subs = set()
    n_diff = 0
    for c in cases:
        bd, w, h = c["bd"], c["w"], c["h"]
        e16 = int(bd > 8)
        dt = np.uint16 if e16 else np.uint8
        p0, p1 = np.ascontiguousarray(z["p%d_0" % bd], dt), np.ascontiguousarray(z["p%d_1" % bd], dt)
        S = p0.shape[1]
        (x0, y0), (x1, y1) = c["pos"]
        (sx0, sy0), (sx1, sy1) = c["subs"]
        mask = np.ascontiguousarray(z["m%d" % c["k"]])
        dst = np.zeros((h, w), dt)
        A0, A1 = C.c_void_p(p0.ctypes.data + (y0 * S + x0) * p0.itemsize), C.c_void_p(p1.ctypes.data + (y1 * S + x1) * p1.itemsize)
        if c.get("diffwtd"):     # the mask is an OUTPUT here: av1_build_compound_diffwtd_mask_d16_c (reconinter.c:296-328)
            g = oracle.lib.orc_convolve_compound_diffwtd
            g.restype = None
            mout = np.zeros((h, w), np.uint8)
            g(A0, S, sx0, sy0, A1, S, sx1, sy1, C.c_void_p(dst.ctypes.data), w, w, h, c["fx"], c["fy"], e16, bd, c["diffwtd"] - 1, C.c_void_p(mout.ctypes.data))
            assert np.array_equal(mout, mask), c
            n_diff += 1
        else:
            f(A0, S, sx0, sy0, A1, S, sx1, sy1, C.c_void_p(dst.ctypes.data), w, w, h, c["fx"], c["fy"], 0, 0, e16, bd, C.c_void_p(mask.ctypes.data),
              c["mask_stride"], c["subw"], c["subh"])
        assert np.array_equal(dst.ravel(), z["d%d" % c["k"]]), c
        subs.add((c["subw"], c["subh"]))
    assert len(subs) == 4 and n_diff >= 12


def test_obmc_blend_matches_reference_evaluation(oracle):
    """orc_blend_a64_1d against the interpreted aom_[highbd_]blend_a64_vmask_c / _hmask_c with av1_get_obmc_mask's tables."""
    z, cases = load("ref_eval_obmc_blend.npz")
    assert len(cases) == 30
    f = oracle.lib.orc_blend_a64_1d
    f.restype = None
    assert z["obmc_mask_64"][0] == 33 and z["obmc_mask_2"].tolist() == [45, 64]
    for c in cases:
        bd = c["bd"]
        dt = np.uint8 if bd == 8 else np.uint16
        pred, adj = np.ascontiguousarray(z["pred%d" % bd], dt).copy(), np.ascontiguousarray(z["adj%d" % bd], dt)
        S = pred.shape[1]
        mask = np.ascontiguousarray(z["obmc_mask_%d" % (c["h"] if c["vertical"] else c["w"])])
        off = (c["y"] * S + c["x"]) * pred.itemsize
        f(C.c_void_p(pred.ctypes.data + off), S, C.c_void_p(adj.ctypes.data + off), S, C.c_void_p(mask.ctypes.data), c["w"], c["h"], c["vertical"], int(bd > 8))
        assert np.array_equal(pred[c["y"]:c["y"] + c["h"], c["x"]:c["x"] + c["w"]], z["o%d" % c["k"]]), c
