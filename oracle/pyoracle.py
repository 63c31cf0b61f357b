"""ctypes loader for oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by
the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


if not os.path.exists(_SO):
    build()
lib = C.CDLL(_SO)

_u8 = np.ctypeslib.ndpointer(np.uint8, flags="C")
_vp, _i = C.c_void_p, C.c_int

lib.orc_sad.restype = C.c_uint
lib.orc_sad.argtypes = [_vp, _i, _vp, _i, _i, _i]
lib.orc_sad_skip.restype = C.c_uint
lib.orc_sad_skip.argtypes = [_vp, _i, _vp, _i, _i, _i]
lib.orc_highbd_sad.restype = C.c_uint
lib.orc_highbd_sad.argtypes = [_vp, _i, _vp, _i, _i, _i, _i]
lib.orc_highbd_sad_skip.restype = C.c_uint
lib.orc_highbd_sad_skip.argtypes = [_vp, _i, _vp, _i, _i, _i, _i]
lib.orc_variance.restype = C.c_uint32
lib.orc_variance.argtypes = [_vp, _i, _vp, _i, _i, _i, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
lib.orc_sub_pixel_variance.restype = C.c_uint32
lib.orc_sub_pixel_variance.argtypes = [_vp, _i, _i, _i, _vp, _i, _i, _i, C.POINTER(C.c_uint32)]
lib.orc_highbd_variance.restype = C.c_uint32
lib.orc_highbd_variance.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
lib.orc_highbd_sub_pixel_variance.restype = C.c_uint32
lib.orc_highbd_sub_pixel_variance.argtypes = [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, C.POINTER(C.c_uint32)]


def _addr(arr, y, x):
    """address of element (y, x) of a 2-D C-contiguous array (may point into a border)."""
    return arr.ctypes.data + (int(y) * arr.shape[1] + int(x)) * arr.itemsize


def sad(src, sy, sx, ref, ry, rx, w, h, skip=False, bd=None):
    """src/ref: 2-D arrays (uint8, or uint16 for highbd); (sy,sx)/(ry,rx) top-left element indices."""
    if src.dtype == np.uint8:
        f = lib.orc_sad_skip if skip else lib.orc_sad
        return f(_addr(src, sy, sx), src.shape[1], _addr(ref, ry, rx), ref.shape[1], w, h)
    f = lib.orc_highbd_sad_skip if skip else lib.orc_highbd_sad
    return f(_addr(src, sy, sx), src.shape[1], _addr(ref, ry, rx), ref.shape[1], w, h, bd or 0)


def variance(a, ay, ax, b, by, bx, w, h, bd=None):
    sse, s = C.c_uint32(), C.c_int()
    if a.dtype == np.uint8:
        v = lib.orc_variance(_addr(a, ay, ax), a.shape[1], _addr(b, by, bx), b.shape[1], w, h, C.byref(sse), C.byref(s))
    else:
        v = lib.orc_highbd_variance(_addr(a, ay, ax), a.shape[1], _addr(b, by, bx), b.shape[1], w, h, bd,
                                    C.byref(sse), C.byref(s))
    return v, sse.value, s.value


def sub_pixel_variance(a, ay, ax, xoff, yoff, b, by, bx, w, h, bd=None):
    sse = C.c_uint32()
    if a.dtype == np.uint8:
        v = lib.orc_sub_pixel_variance(_addr(a, ay, ax), a.shape[1], xoff, yoff, _addr(b, by, bx), b.shape[1], w, h,
                                       C.byref(sse))
    else:
        v = lib.orc_highbd_sub_pixel_variance(_addr(a, ay, ax), a.shape[1], xoff, yoff, _addr(b, by, bx), b.shape[1],
                                              w, h, bd, C.byref(sse))
    return v, sse.value


# ---- work-list drivers (aomref_batch.c)
lib.orc_max_threads.restype = C.c_int
lib.orc_sad_batch.restype = None
lib.orc_sad_batch.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i]
lib.orc_sad_x4d_batch.restype = None
lib.orc_sad_x4d_batch.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _i, _i]


def extend_plane(pixels, border, stride=None):
    """Host model of an HBM plane: replicate edges into `border` px on every side
    (aom_scale/generic/yv12extend.c:22-221); returns (bordered array, origin (y, x))."""
    h, w = pixels.shape
    stride = stride or (w + 2 * border)
    out = np.zeros((h + 2 * border, stride), pixels.dtype)
    core = np.pad(pixels, border, mode="edge") if border else pixels
    out[:, :w + 2 * border] = core
    if stride > w + 2 * border:
        out[:, w + 2 * border:] = core[:, -1:]
    return out


def sad_batch(src_b, ref_b, border, w, h, cands, skip=False, bd=8, threads=1, reps=1):
    """src_b/ref_b: bordered planes from extend_plane(); cands: structured array (sx,sy,rx,ry)."""
    cands = np.ascontiguousarray(cands)
    out = np.empty(len(cands), np.uint32)
    lib.orc_sad_batch(_addr(src_b, border, border), src_b.shape[1], _addr(ref_b, border, border), ref_b.shape[1],
                      int(src_b.dtype != np.uint8), bd, w, h, int(skip), cands.ctypes.data, len(cands),
                      out.ctypes.data, threads, reps)
    return out


def sad_x4d_batch(src_b, ref_b, border, w, h, groups, skip=False, bd=8, threads=1, reps=1):
    groups = np.ascontiguousarray(groups)
    out = np.empty((len(groups), 4), np.uint32)
    lib.orc_sad_x4d_batch(_addr(src_b, border, border), src_b.shape[1], _addr(ref_b, border, border), ref_b.shape[1],
                          int(src_b.dtype != np.uint8), bd, w, h, int(skip), groups.ctypes.data, len(groups),
                          out.ctypes.data, threads, reps)
    return out
