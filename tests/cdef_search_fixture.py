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
    """-> (recon, source, skip8x8, (fb_row, fb_col), luma_dir or None).  "full": the block sits at (N, N) of a 3N x 3N
    plane with the footprint's border pixels around it; "partial": at (0, 0) of a 2N x 2N plane (frame edges above and to
    the left).  N = 64 for luma, 32 for the 4:2:0 chroma cases.  Every other filter block is all-skip, so only the block
    of interest contributes."""
    k, bd, pli = c["k"], c["bd"], c.get("pli", 0)
    N = 32 if pli else 64
    dt = np.uint8 if bd == 8 else np.uint16
    foot = z["in%d" % k][:(N + 2 * VB) * BS].reshape(N + 2 * VB, BS)[:, :N + 2 * HB].astype(np.int64)
    src = z["src%d" % k].astype(np.int64)
    if c["variant"] == "full":
        n, oy, ox, fb = 3 * N, N, N, (1, 1)
    else:
        n, oy, ox, fb = 2 * N, 0, 0, (0, 0)
    recon, source = np.zeros((n, n), np.int64), np.zeros((n, n), np.int64)
    y0, x0 = oy - VB, ox - HB
    ys, xs = max(y0, 0), max(x0, 0)
    recon[ys:oy + N + VB, xs:ox + N + HB] = foot[ys - y0:, xs - x0:]
    source[oy:oy + N, ox:ox + N] = src
    nb = n // (N // 8)
    skip = np.ones((nb, nb), np.uint8)
    by0, bx0 = oy // (N // 8), ox // (N // 8)
    skip[by0:by0 + 8, bx0:bx0 + 8] = z["skip%d" % k]
    ldir = None
    if pli:
        ldir = np.zeros((nb, nb), np.uint8)
        ldir[by0:by0 + 8, bx0:bx0 + 8] = z["ld%d" % k]
    return recon.astype(dt), source.astype(dt), skip, fb, ldir


def mapped_strengths(c):
    """(pri, sec) as the entry point takes them: sec 3 -> 4 (pickcdef.c:439)."""
    return [(p, s + (s == 3)) for p, s in c["strengths"]]
