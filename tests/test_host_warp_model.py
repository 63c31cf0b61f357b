"""aomhip_get_shear_params (host/warp_model.c, no GPU) against the interpreted reference's av1_get_shear_params values (ref_eval_warp_error.npz) and the
restatement on random models, valid and not."""
import numpy as np

from test_golden_warp_error import load, oracle_shear


def test_host_shear_decomposition():
    import importlib
    capi = importlib.import_module("aom-av1-psy_amd.capi")
    _, cases = load()
    for c in cases:
        rec = np.zeros(1, capi.warp_model_dtype)
        rec["mat"][0] = c["mat"]
        assert capi.get_shear_params(rec)[0] == c["valid"]
        if c["mat"][2] > 0:
            assert [int(rec[f][0]) for f in ("alpha", "beta", "gamma", "delta")] == c["shear"]
    rng = np.random.default_rng(7)
    verdicts = 0
    for _ in range(4000):
        scale = int(rng.choice([8, 11, 13, 15, 17]))
        mat = [int(rng.integers(-1 << 20, 1 << 20)), int(rng.integers(-1 << 20, 1 << 20)), (1 << 16) + int(rng.integers(-(1 << scale), 1 << scale)),
               int(rng.integers(-(1 << scale), 1 << scale)), int(rng.integers(-(1 << scale), 1 << scale)), (1 << 16) + int(rng.integers(-(1 << scale), 1 << scale))]
        rec = np.zeros(1, capi.warp_model_dtype)
        rec["mat"][0] = mat
        ok = capi.get_shear_params(rec)[0]
        want_ok, want = oracle_shear(mat)
        assert ok == want_ok
        if mat[2] > 0:
            assert [int(rec[f][0]) for f in ("alpha", "beta", "gamma", "delta")] == want.tolist(), mat
        verdicts += ok
    assert 800 < verdicts < 3600
    # degenerate models (tiny mat[2], huge mat[4]): the rounded quotient does not fit an int and the reference truncates it BEFORE clamping
    # (av1/common/warped_motion.c:227-228) -- gamma wraps instead of saturating
    wrapped = 0
    for m2, m4 in ((1, 1 << 20), (1, (1 << 20) + 12345), (3, -(1 << 22) - 7), (2, (1 << 30) - 1), (5, -(1 << 31) + 1), (1, 1 << 15)):
        mat = [0, 0, m2, 17, m4, 1 << 16]
        rec = np.zeros(1, capi.warp_model_dtype)
        rec["mat"][0] = mat
        ok = capi.get_shear_params(rec)[0]
        want_ok, want = oracle_shear(mat)
        assert ok == want_ok
        assert [int(rec[f][0]) for f in ("alpha", "beta", "gamma", "delta")] == want.tolist(), mat
        wrapped += abs(int(want[2])) < 32704
    assert wrapped >= 1   # (at least one case where a saturating gamma would have been +-32704)
