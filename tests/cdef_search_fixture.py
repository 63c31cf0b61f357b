"""Turns a case of tests/golden/ref_eval_cdef_search.npz (one 64x64 filter block as get_filt_error saw it: the 16-bit
footprint with its borders, the source block, the skip flags) into whole planes for the plane-level entry points."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BS, VB, HB = 144, 2, 8     # CDEF_BSTRIDE, CDEF_VBORDER, CDEF_HBORDER


def load_cases():
    z = np.load(os.path.join(GOLD, "ref_eval_cdef_search.npz"))
    return z, json.loads(bytes(z["cases"]).decode())


def planes_of(z, c):
    """-> (recon, source, skip8x8, (fb_row, fb_col)).  "full": the block sits at (64, 64) of a 192 x 192 frame with the
    footprint's border pixels around it; "partial": at (0, 0) of a 128 x 128 frame (frame edges above and to the left).
    Every other filter block is all-skip, so only the block of interest contributes."""
    k, bd = c["k"], c["bd"]
    dt = np.uint8 if bd == 8 else np.uint16
    foot = z["in%d" % k][:(64 + 2 * VB) * BS].reshape(64 + 2 * VB, BS)[:, :64 + 2 * HB].astype(np.int64)
    src = z["src%d" % k].astype(np.int64)
    if c["variant"] == "full":
        n, oy, ox, fb = 192, 64, 64, (1, 1)
    else:
        n, oy, ox, fb = 128, 0, 0, (0, 0)
    recon, source = np.zeros((n, n), np.int64), np.zeros((n, n), np.int64)
    y0, x0 = oy - VB, ox - HB
    ys, xs = max(y0, 0), max(x0, 0)
    recon[ys:oy + 64 + VB, xs:ox + 64 + HB] = foot[ys - y0:, xs - x0:]
    source[oy:oy + 64, ox:ox + 64] = src
    skip = np.ones((n // 8, n // 8), np.uint8)
    skip[oy // 8:oy // 8 + 8, ox // 8:ox // 8 + 8] = z["skip%d" % k]
    return recon.astype(dt), source.astype(dt), skip, fb


def mapped_strengths(c):
    """(pri, sec) as the entry point takes them: sec 3 -> 4 (pickcdef.c:439)."""
    return [(p, s + (s == 3)) for p, s in c["strengths"]]
