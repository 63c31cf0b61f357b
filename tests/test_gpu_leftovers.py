"""The small members of the named files on the device (csrc/dsp_misc.hip, the extreme-MV members of aomhip_subpel_tree_batch) straight against
the reference's own functions interpreted (tests/golden/ref_eval_leftovers.npz) and against the oracle on larger random inputs."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import pyoracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_eval_leftovers.npz")


def load():
    z = np.load(GOLD)
    return z, json.loads(bytes(z["cases"]))


def byteptr(a):
    """CONVERT_TO_BYTEPTR of a uint16 array's address."""
    assert a.ctypes.data % 2 == 0
    return a.ctypes.data >> 1


def test_mb_ss_and_16_bit_mse_entry_points_match_the_reference(hip):
    lib = hip.capi.lib
    z, cases = load()
    n = 0
    for c in cases:
        if c["kind"] == "mb_ss":
            a = np.ascontiguousarray(z[c["a"]])
            assert lib.aomhip_get_mb_ss(a.ctypes.data) == c["out"], c
        elif c["kind"] == "mse_wxh":
            dst, src = np.ascontiguousarray(z["mse_dst%d" % c["bd"]]), np.ascontiguousarray(z["mse_src%d" % c["bd"]])
            S = dst.shape[1]
            off = c["y"] * S + c["x"]
            f = lib.aomhip_mse_wxh_16bit if c["bd"] == 8 else lib.aomhip_mse_wxh_16bit_highbd
            assert f(dst.ctypes.data + off * dst.itemsize, S, src.ctypes.data + off * 2, S, c["w"], c["h"]) == int(c["out"]), c
        elif c["kind"] == "mse_16xh":
            dst, src = np.ascontiguousarray(z["mse_dst8"]), np.ascontiguousarray(z[c["src"]])
            S = dst.shape[1]
            assert lib.aomhip_mse_16xh_16bit(dst.ctypes.data + c["y"] * S + c["x"], S, src.ctypes.data, c["w"], c["h"]) == int(c["out"]), c
        else:
            continue
        n += 1
    assert n >= 45
    assert lib.aomhip_status() == 0
    # larger blocks than the fixtures hold, against the oracle
    rng = np.random.default_rng(5)
    dst, src = rng.integers(0, 256, (140, 150)).astype(np.uint8), rng.integers(0, 4096, (140, 170)).astype(np.uint16)
    for (w, h) in ((64, 64), (128, 128), (33, 7)):
        assert lib.aomhip_mse_wxh_16bit(dst.ctypes.data + 3, 150, src.ctypes.data + 10, 170, w, h) == orc.mse_wxh_16bit(dst[:, 3:], src[:, 5:], w, h)


def test_comp_mask_pred_entry_points_match_the_reference(hip):
    lib = hip.capi.lib
    z, cases = load()
    n = 0
    for c in cases:
        if c["kind"] != "comp_mask":
            continue
        k = c["k"]
        pred, ref, mask = (np.ascontiguousarray(z["cmp_%s%d" % (s, k)]) for s in ("pred", "ref", "mask"))
        out = np.zeros_like(pred)
        if c["bd"] == 8:
            lib.aomhip_comp_mask_pred(out.ctypes.data, pred.ctypes.data, c["w"], c["h"], ref.ctypes.data, ref.shape[1], mask.ctypes.data, mask.shape[1],
                                      c["invert"])
        else:
            lib.aomhip_highbd_comp_mask_pred(byteptr(out), byteptr(pred), c["w"], c["h"], byteptr(ref), ref.shape[1], mask.ctypes.data, mask.shape[1],
                                             c["invert"])
        assert np.array_equal(out, z["cmp_out%d" % k]), c
        n += 1
    assert n == 36 and lib.aomhip_status() == 0
    rng = np.random.default_rng(6)
    for bd, dt in ((8, np.uint8), (12, np.uint16)):
        pred, ref = rng.integers(0, 1 << bd, (128, 128)).astype(dt), rng.integers(0, 1 << bd, (128, 160)).astype(dt)
        mask = rng.integers(0, 65, (128, 128)).astype(np.uint8)
        out = np.zeros_like(pred)
        if bd == 8:
            lib.aomhip_comp_mask_pred(out.ctypes.data, pred.ctypes.data, 128, 128, ref.ctypes.data, 160, mask.ctypes.data, 128, 1)
        else:
            lib.aomhip_highbd_comp_mask_pred(byteptr(out), byteptr(pred), 128, 128, byteptr(ref), 160, mask.ctypes.data, 128, 1)
        assert np.array_equal(out, orc.comp_mask_pred(pred, ref, mask, 1))


def test_extreme_sub_pixel_mv_members_of_the_sub_pel_entry_point(hip, ctx):
    capi = hip.capi
    _, cases = load()
    rows = [c for c in cases if c["kind"] == "extreme_mv"]
    src, ref = ctx.planes_alloc(64, 64, 32, 8, 1), ctx.planes_alloc(64, 64, 32, 8, 1)
    for allow_hp in (0, 1):
        sel = [c for c in rows if c["allow_hp"] == allow_hp]
        blocks = np.zeros(len(sel) + 1, capi.search_block_dtype)
        for i, c in enumerate(sel):
            blocks["col_min"][i], blocks["col_max"][i], blocks["row_min"][i], blocks["row_max"][i] = c["limits"]
        blocks["row_min"][-1], blocks["row_max"][-1] = 5, 4     # an empty window: the block is skipped, its outputs stay
        d_b = ctx.to_device(blocks)
        for tree, key in ((3, "max"), (4, "min")):
            d_mv, d_err = ctx.malloc(4 * len(blocks)), ctx.malloc(4 * len(blocks))
            ctx.memset(d_mv, 0x55, 4 * len(blocks)); ctx.memset(d_err, 0x55, 4 * len(blocks))
            p = capi.SubpelParams(tree, capi.MV_COST_NONE, 0, 2, allow_hp, 0, 0)
            ctx.subpel_tree_batch(src, ref, 0, 16, 16, p, d_b, len(blocks), d_mv, d_err, None, None)
            mv, err = ctx.from_device(d_mv, (len(blocks), 2), np.int16), ctx.from_device(d_err, (len(blocks),), np.uint32)
            for i, c in enumerate(sel):
                assert [int(err[i]), int(mv[i, 0]), int(mv[i, 1])] == c[key], (c, tree)
            assert int(err[-1]) == 0x55555555 and int(mv[-1, 0]) == 0x5555
            ctx.free(d_mv); ctx.free(d_err)
        ctx.free(d_b)
    ctx.planes_free(src); ctx.planes_free(ref)
