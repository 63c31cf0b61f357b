"""The oracle's `fp` quantiser WITH quantisation matrices (orc_quantize_fp_qm, oracle/aomref_quant.c) against the reference's own quantize_fp_helper_c /
highbd_quantize_fp_helper_c interpreted with the matrices of av1/common/quant_common.c (tests/golden/ref_eval_qm_fp.npz,
tests/golden/gen_ref_eval_qm_fp.py): 140 cases, bit for bit."""
import ctypes as C
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_qm_fp.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def orc_fp_qm(oracle, coeff, tables, scan, log_scale, hbd, qm, iqm):
    lib = oracle.lib
    lib.orc_quantize_fp_qm.restype = None
    lib.orc_quantize_fp_qm.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p]
    c = np.ascontiguousarray(coeff, np.int32)
    qc, dq = np.zeros_like(c), np.zeros_like(c)
    eob = C.c_uint16()
    t = {m: np.array(v, np.int16) for m, v in tables.items()}
    sc = np.ascontiguousarray(scan, np.int16)
    qm_ = None if qm is None else np.ascontiguousarray(qm, np.uint8)
    iqm_ = None if iqm is None else np.ascontiguousarray(iqm, np.uint8)
    lib.orc_quantize_fp_qm(c.ctypes.data, c.size, t["round"].ctypes.data, t["quant"].ctypes.data, qc.ctypes.data, dq.ctypes.data, t["dequant"].ctypes.data,
                           C.addressof(eob), sc.ctypes.data, log_scale, int(hbd), None if qm_ is None else qm_.ctypes.data, None if iqm_ is None else iqm_.ctypes.data)
    return qc, dq, eob.value


def test_oracle_fp_quantiser_with_matrices_reproduces_the_interpreted_reference(oracle):
    z, cases = load()
    assert len(cases) == 140
    seen = set()
    for c in cases:
        k = c["k"]
        scan, _ = oracle.get_scan(c["tx_size"], 0)
        qm = z["qm_" + c["matrix"]] if c["which"] != "iqm_only" else None
        iqm = z["iqm_" + c["matrix"]] if c["which"] != "qm_only" else None
        qc, dq, eob = orc_fp_qm(oracle, z["c%d" % k], c["tables"], scan, c["log_scale"], c["hbd"], qm, iqm)
        assert np.array_equal(qc, z["q%d" % k]) and np.array_equal(dq, z["d%d" % k]) and eob == c["eob"], c
        seen.add((c["tx_size"], c["hbd"], c["which"]))
    assert len(seen) == 5 * 2 * 3
