"""The oracle's composition of mode_estimation's inter leg (oracle/pyoracle.py tpl_inter_estimation_batch; av1/encoder/tpl_model.c:620-770):
structural properties that follow from the reference's text -- one candidate without pruning is motion_estimation itself; a candidate whose
SAD is more than 20 % above the next better one is cut; equal candidates keep their order; a reference that does not exist is reported so."""
import numpy as np


def _setup(hip, oracle, bd=8, bs=16, n_refs=2, seed=5):
    W, H, B = 128, 96, 64
    rng = np.random.default_rng(seed)
    dt = np.uint8 if bd == 8 else np.uint16
    src, ref0 = hip.synth.shifted_smooth_pair(W, H, 3, bd, shift=(2, -3), frac8=(5, 2))
    refs = [np.clip(np.roll(ref0, (r, -r), (0, 1)).astype(np.int32) + rng.integers(-5, 6, ref0.shape), 0, (1 << bd) - 1).astype(dt) for r in range(n_refs)]
    gc = W // bs
    n = gc * (H // bs)
    blocks = np.zeros(n, hip.capi.search_block_dtype)
    blocks["bx"], blocks["by"] = (np.arange(n) % gc) * bs, (np.arange(n) // gc) * bs
    ext = B - 8
    blocks["col_min"], blocks["col_max"] = -(blocks["bx"] + ext), W - blocks["bx"] - bs + ext
    blocks["row_min"], blocks["row_max"] = -(blocks["by"] + ext), H - blocks["by"] - bs + ext
    stride = (W + 2 * B + 31) & ~31
    sb = oracle.extend_plane(src.astype(dt), B, stride)
    rbs = [oracle.extend_plane(r, B, stride) for r in refs]
    oq = oracle.search_params("NSTEP", 2, 4, sad_per_bit=20, error_per_bit=64, no_cost_list=1)
    sub = dict(tree="pruned", cost_type=4, error_per_bit=64, iters=2, allow_hp=1, forced_stop=0, subpel_search_type=1)
    return W, H, B, bs, n, blocks, sb, rbs, oq, sub, rng


def test_one_candidate_is_motion_estimation_and_a_missing_reference_is_reported(hip, oracle):
    W, H, B, bs, n, blocks, sb, rbs, oq, sub, rng = _setup(hip, oracle)
    centers = np.zeros((n, 2, 4, 2), np.int16)
    centers[:, :, 0] = rng.integers(-40, 41, (n, 2, 2))
    counts = np.ones((n, 2), np.uint8)
    counts[::4, 1] = 0
    mv, pe, rf, bc = oracle.tpl_inter_estimation_batch(sb, rbs, B, W, H, bs, blocks, centers, counts, oq, sub, 0, 0, bd=8, threads=2)
    for r in range(2):
        ent = blocks.copy()
        ent["ref_row"], ent["ref_col"] = centers[:, r, 0, 0], centers[:, r, 0, 1]
        want = oracle.motion_estimation_batch(sb, rbs[r], B, bs, bs, ent, oq, sub, 0, bd=8, threads=2)[0]
        have = counts[:, r] > 0
        assert np.array_equal(mv[have, r], want[have])
        assert (mv[~have, r] == -32768).all() and (pe[~have, r] == 2147483647).all()
    assert (pe[counts > 0] >= 1).all() and np.array_equal(bc, np.where(counts[:, 1] > 0, np.minimum(pe[:, 0], pe[:, 1]), pe[:, 0]))
    assert np.array_equal(rf, np.where((counts[:, 1] > 0) & (pe[:, 1] < pe[:, 0]), 1, 0))


def test_pruning_cuts_the_far_candidate_and_keeps_equal_ones_in_order(hip, oracle):
    W, H, B, bs, n, blocks, sb, rbs, oq, sub, rng = _setup(hip, oracle, n_refs=1)
    # candidate 1 = candidate 0 (equal SADs), candidate 2 far away (a much larger SAD): with prune 1 at most 3 survive and the far one goes
    centers = np.zeros((n, 1, 4, 2), np.int16)
    centers[:, 0, 0] = (16, -24)
    centers[:, 0, 1] = (16, -24)
    centers[:, 0, 2] = (8 * 40, 8 * 40)
    counts = np.full((n, 1), 3, np.uint8)
    a = oracle.tpl_inter_estimation_batch(sb, rbs, B, W, H, bs, blocks, centers, counts, oq, sub, 0, 1, bd=8, threads=2)
    only = np.full((n, 1), 1, np.uint8)
    b = oracle.tpl_inter_estimation_batch(sb, rbs, B, W, H, bs, blocks, centers, only, oq, sub, 0, 0, bd=8, threads=2)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])     # the two equal candidates give the same search; the far one never wins
    # without pruning the far candidate is searched as well; where it wins, the results differ
    c = oracle.tpl_inter_estimation_batch(sb, rbs, B, W, H, bs, blocks, centers, counts, oq, sub, 0, 0, bd=8, threads=2)
    assert c[0].shape == a[0].shape
