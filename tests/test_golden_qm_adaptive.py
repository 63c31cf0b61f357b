"""The oracle's ADAPTIVE quantiser with quantisation matrices (orc_quantize_b_adaptive_qm, oracle/aomref_quant.c) against the reference's own
aom_[highbd_]quantize_b_adaptive_helper_c interpreted with the matrices of av1/common/quant_common.c (tests/golden/ref_eval_qm_adaptive.npz): 180 cases
-- the pre-scan's widened dead zone and the single-coefficient rule under per-coefficient weights -- bit for bit."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_qm_adaptive.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def orc_adaptive_qm(oracle, coeff, tables, scan, log_scale, hbd, qm, iqm):
    lib = oracle.lib
    lib.orc_quantize_b_adaptive_qm.restype = None
    lib.orc_quantize_b_adaptive_qm.argtypes = [C.c_void_p, C.c_ssize_t] + [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    c = np.ascontiguousarray(coeff, np.int32)
    qc, dq = np.zeros_like(c), np.zeros_like(c)
    eob = C.c_uint16()
    t = {m: np.array(v, np.int16) for m, v in tables.items()}
    sc = np.ascontiguousarray(scan, np.int16)
    qm_, iqm_ = np.ascontiguousarray(qm, np.uint8), np.ascontiguousarray(iqm, np.uint8)
    lib.orc_quantize_b_adaptive_qm(c.ctypes.data, c.size, t["zbin"].ctypes.data, t["round"].ctypes.data, t["quant"].ctypes.data, t["quant_shift"].ctypes.data,
                                   qc.ctypes.data, dq.ctypes.data, t["dequant"].ctypes.data, C.addressof(eob), sc.ctypes.data, log_scale, int(hbd), qm_.ctypes.data,
                                   iqm_.ctypes.data)
    return qc, dq, eob.value


def test_oracle_adaptive_quantiser_with_matrices_reproduces_the_interpreted_reference(oracle):
    z, cases = load()
    assert len(cases) == 180
    dropped = cut = 0
    for c in cases:
        k = c["k"]
        scan, _ = oracle.get_scan(c["tx_size"], 0)
        qc, dq, eob = orc_adaptive_qm(oracle, z["c%d" % k], c["tables"], scan, c["log_scale"], c["hbd"], z["qm_" + c["matrix"]], z["iqm_" + c["matrix"]])
        assert np.array_equal(qc, z["q%d" % k]) and np.array_equal(dq, z["d%d" % k]) and eob == c["eob"], c
        if c["kind"] == "single":
            dropped += eob == 0
        if c["kind"] == "tail":
            cut += eob < c["n"]
    assert dropped >= 5 and cut >= 30      # both adaptive rules fire in the fixtures
