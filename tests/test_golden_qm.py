"""The oracle's quantisers WITH quantisation matrices against the reference's own aom_[highbd_]quantize_b_helper_c interpreted with the
matrices of av1/common/quant_common.c (tests/golden/ref_eval_qm.npz, tests/golden/gen_ref_eval_qm.py): 240 cases, bit for bit."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load():
    z = np.load(os.path.join(HERE, "golden", "ref_eval_qm.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def test_oracle_quantisers_with_matrices_reproduce_the_interpreted_reference(oracle):
    z, cases = load()
    assert len(cases) == 240
    seen = set()
    for k, c in enumerate(cases):
        q = {m: np.array(v, np.int16) for m, v in c["tables"].items()}
        scan, iscan = oracle.get_scan(c["tx_size"], 0)
        qm, iqm = z["qm_" + c["matrix"]], z["iqm_" + c["matrix"]]
        qc, dq, eob = oracle.quantize_b(z["c%d" % k], q, scan, iscan, c["log_scale"], bool(c["hbd"]), qm=qm, iqm=iqm)
        assert np.array_equal(qc, z["q%d" % k]) and np.array_equal(dq, z["d%d" % k]) and eob == c["eob"], c
        seen.add((c["tx_size"], c["qm_level"], c["plane"], c["hbd"]))
        # the matrices matter: the flat quantiser gives another answer on these inputs (except where the matrix IS flat)
        if (qm != 32).any() and c["kind"] == "random":
            fq, fdq, _ = oracle.quantize_b(z["c%d" % k], q, scan, iscan, c["log_scale"], bool(c["hbd"]))
            assert not (np.array_equal(fq, qc) and np.array_equal(fdq, dq))
    assert len(seen) == 5 * 3 * 2 * 2


def test_flat_matrices_equal_the_plain_quantiser(oracle):
    rng = np.random.default_rng(5)
    for hbd in (False, True):
        q = oracle.build_quantizer_y(10 if hbd else 8, 90)
        scan, iscan = oracle.get_scan(2, 0)
        c = rng.integers(-20000, 20000, 256)
        a = oracle.quantize_b(c, q, scan, iscan, 0, hbd)
        b = oracle.quantize_b(c, q, scan, iscan, 0, hbd, qm=np.full(256, 32, np.uint8), iqm=np.full(256, 32, np.uint8))
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
