"""Deterministic synthetic frames and work lists.

Pixels come from the 32-bit LCG x <- x*1664525 + 1013904223 run in raster order
over the visible area, seed 0xbaba + 977*frame_idx + 31*plane; an 8-bit sample is
bits 8..15 of the state, a 10-bit sample bits 8..17.  Rows are generated in
parallel by jumping the LCG ahead to each row start.
"""
import numpy as np

LCG_A = 1664525
LCG_C = 1013904223
_M = 0xFFFFFFFF


def _lcg_jump(n):
    """(A_n, C_n) with x_{k+n} = A_n * x_k + C_n (mod 2^32)."""
    a_acc, c_acc = 1, 0
    a, c = LCG_A, LCG_C
    while n:
        if n & 1:
            a_acc, c_acc = (a_acc * a) & _M, (c_acc * a + c) & _M
        a, c = (a * a) & _M, (c * a + c) & _M
        n >>= 1
    return a_acc, c_acc


def lcg_frame(width, height, frame_idx, plane=0, bit_depth=8):
    seed = (0xBABA + 977 * frame_idx + 31 * plane) & _M
    ar, cr = _lcg_jump(width)
    starts = np.empty(height, np.uint64)
    x = seed
    for y in range(height):
        starts[y] = x
        x = (ar * x + cr) & _M
    out = np.empty((height, width), np.uint16 if bit_depth > 8 else np.uint8)
    st = starts
    mask = (1 << bit_depth) - 1 if bit_depth > 8 else 0xFF
    a64, c64, m64 = np.uint64(LCG_A), np.uint64(LCG_C), np.uint64(_M)
    for c in range(width):
        st = (st * a64 + c64) & m64
        out[:, c] = ((st >> np.uint64(8)) & np.uint64(mask)).astype(out.dtype)
    return out


def shifted_smooth_pair(width, height, frame_idx, bit_depth=8, shift=(3, -2), frac8=(0, 0)):
    """Natural-like content: low-pass filtered noise; ref = src shifted by (dx, dy) [+ frac8/8 pel, bilinear] plus a
    little noise, so motion searches converge (configs 4/5)."""
    rng = np.random.default_rng(0xBABA + frame_idx)
    pad = 16
    n = rng.standard_normal((height + 2 * pad, width + 2 * pad)).astype(np.float32)
    k = 5
    for _ in range(2):  # separable box blur twice ~ gaussian
        n = np.cumsum(n, axis=0)
        n = n[k:] - n[:-k]
        n = np.pad(n, ((k - k // 2, k // 2), (0, 0)), mode="edge")
        n = np.cumsum(n, axis=1)
        n = n[:, k:] - n[:, :-k]
        n = np.pad(n, ((0, 0), (k - k // 2, k // 2)), mode="edge")
    n = (n - n.mean()) / (n.std() + 1e-6)
    mx = (1 << bit_depth) - 1
    img = np.clip(n * (mx / 6.0) + mx / 2.0, 0, mx)
    dx, dy = shift
    src = img[pad:pad + height, pad:pad + width]
    fx, fy = frac8[0] / 8.0, frac8[1] / 8.0

    def win(oy, ox):
        return img[pad - dy - oy:pad - dy - oy + height, pad - dx - ox:pad - dx - ox + width]
    ref = ((1 - fy) * ((1 - fx) * win(0, 0) + fx * win(0, 1)) + fy * ((1 - fx) * win(1, 0) + fx * win(1, 1))
           + rng.normal(0, mx / 256.0, (height, width)))
    dt = np.uint16 if bit_depth > 8 else np.uint8
    return np.clip(np.rint(src), 0, mx).astype(dt), np.clip(np.rint(ref), 0, mx).astype(dt)


def mode_a_worklist(width, height, bsize=16, seed=1, search=64, order="raster"):
    """SURVEY 8(d) Mode A: per full bsize x bsize block one candidate at mv (0,0) and one x4d
    group whose four reference positions are uniform in [-search, search]^2.
    Returns (cands[n_blocks], groups[n_blocks]) as structured arrays.  order = "sb64": superblock by
    superblock (64x64, raster), blocks in raster order inside each -- the order in which the encoder's
    per-superblock call sites (av1/encoder/encodeframe.c:1069 encode_sb_row) visit them; "raster": plain raster."""
    from .capi import sad_cand_dtype, sad_x4d_dtype
    bx = np.arange(0, width - bsize + 1, bsize, dtype=np.int16)
    by = np.arange(0, height - bsize + 1, bsize, dtype=np.int16)
    gx, gy = np.meshgrid(bx, by)
    gx, gy = gx.ravel(), gy.ravel()
    if order == "sb64":
        sbs_per_row = (width + 63) // 64
        key = ((gy // 64).astype(np.int64) * sbs_per_row + gx // 64) * 4096 + (gy % 64) * 64 + (gx % 64)
        perm = np.argsort(key, kind="stable")
        gx, gy = gx[perm], gy[perm]
    n = gx.size
    rng = np.random.default_rng(seed)
    cands = np.zeros(n, sad_cand_dtype)
    cands["sx"], cands["sy"], cands["rx"], cands["ry"] = gx, gy, gx, gy
    groups = np.zeros(n, sad_x4d_dtype)
    groups["sx"], groups["sy"] = gx, gy
    groups["rx"] = gx[:, None] + rng.integers(-search, search + 1, (n, 4), dtype=np.int16)
    groups["ry"] = gy[:, None] + rng.integers(-search, search + 1, (n, 4), dtype=np.int16)
    return cands, groups


def bucket_order(sx, sy, width, height, sb_w, sb_h):
    """Host batching for aomhip_sad_sb_batch: -> (perm, offsets) such that list[perm] is sorted by the sb_w x sb_h
    cell (raster over cells) its source block starts in, stable inside a cell, and offsets[b]..offsets[b+1]
    delimit cell b (int32, n_cells + 1)."""
    cpr = (width + sb_w - 1) // sb_w
    rows = (height + sb_h - 1) // sb_h
    cell = (np.asarray(sy, np.int64) // sb_h) * cpr + np.asarray(sx, np.int64) // sb_w
    perm = np.argsort(cell, kind="stable")
    counts = np.bincount(cell, minlength=cpr * rows)
    offsets = np.zeros(cpr * rows + 1, np.int32)
    offsets[1:] = np.cumsum(counts)
    return perm, offsets


def tile_column_bounds(width, n_cols, sb=64):
    """Uniform tile columns in superblock units (av1/common/tile_common.c:76-97):
    size_sb = ceil(sb_cols / n_cols); returns [(x0, x1), ...] in pixels (may be fewer than n_cols)."""
    sb_cols = (width + sb - 1) // sb
    size_sb = (sb_cols + n_cols - 1) // n_cols
    out = []
    s = 0
    while s < sb_cols:
        e = min(s + size_sb, sb_cols)
        out.append((s * sb, min(e * sb, width)))
        s = e
    return out
