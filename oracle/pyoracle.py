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
# before libgomp loads: with OMP_PROC_BIND set its constructor binds THIS thread to the first place, after which the affinity mask
# no longer says how many CPUs the process may use
_CPUS_AT_IMPORT = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
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


def sad_avg_batch(src_b, ref_b, border, w, h, cands, preds, pred_index, fwd_offset=0, bck_offset=0, bd=8):
    """cands: structured (sx, sy, rx, ry); preds: [n_preds, h, w] contiguous blocks; pred_index: per candidate."""
    lib.orc_sad_avg_any.restype = C.c_uint
    preds = np.ascontiguousarray(preds, src_b.dtype)
    e16 = int(src_b.dtype != np.uint8)
    out = np.zeros(len(cands), np.uint32)
    for i, c in enumerate(cands):
        out[i] = lib.orc_sad_avg_any(C.c_void_p(_addr(src_b, border + int(c["sy"]), border + int(c["sx"]))), src_b.shape[1],
                                     C.c_void_p(_addr(ref_b, border + int(c["ry"]), border + int(c["rx"]))), ref_b.shape[1],
                                     C.c_void_p(preds[int(pred_index[i])].ctypes.data), w, h, e16, bd, fwd_offset, bck_offset)
    return out


def compound_batch(src_b, ref_b, border, w, h, cands, kind, subpel, bd=8, preds=None, pred_index=None, fwd_offset=0, bck_offset=0,
                   masks=None, mask_stride=0, mask_offset=None, invert_mask=0, wsrc=None, omask=None):
    """The compound / masked / OBMC table members (oracle/aomref_compound.c) over a candidate list, mirroring
    aomhip_compound_batch: returns (var, sse, sad) arrays.  cands: structured (sx, sy, rx, ry, xoff, yoff)."""
    e16 = int(src_b.dtype != np.uint8)
    n = len(cands)
    var, sse, sad = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    lib.orc_compound_sub_pixel_variance.restype = C.c_uint32
    lib.orc_masked_sad.restype = lib.orc_obmc_sad.restype = C.c_uint
    lib.orc_obmc_variance.restype = C.c_uint32
    lib.orc_sad_avg_any.restype = C.c_uint
    q = C.c_uint32()
    if preds is not None:
        preds = np.ascontiguousarray(preds, src_b.dtype)
    if masks is not None:
        masks = np.ascontiguousarray(masks, np.uint8)
    for i, c in enumerate(cands):
        a = C.c_void_p(_addr(ref_b, border + int(c["ry"]), border + int(c["rx"])))
        b = C.c_void_p(_addr(src_b, border + int(c["sy"]), border + int(c["sx"])))
        xo, yo = (int(c["xoff"]), int(c["yoff"])) if subpel else (0, 0)
        k = int(pred_index[i]) if pred_index is not None else 0
        if kind == 3:
            ws, om = np.ascontiguousarray(wsrc[k], np.int32), np.ascontiguousarray(omask[k], np.int32)
            var[i] = lib.orc_obmc_variance(a, ref_b.shape[1], int(subpel), xo, yo, C.c_void_p(ws.ctypes.data), C.c_void_p(om.ctypes.data), w, h,
                                           e16, bd, C.byref(q))
            sse[i] = q.value
            if not subpel:
                sad[i] = lib.orc_obmc_sad(a, ref_b.shape[1], C.c_void_p(ws.ctypes.data), C.c_void_p(om.ctypes.data), w, h, e16, bd)
            continue
        sp = C.c_void_p(preds[k].ctypes.data)
        mp = C.c_void_p(masks.ctypes.data + (int(mask_offset[i]) if mask_offset is not None else 0)) if kind == 2 else None
        # the reference runs both bilinear passes in the sub-pixel forms only; offset (0, 0) is the identity, which is what
        # lets one restatement serve the full-pel SAD forms' variance outputs too
        var[i] = lib.orc_compound_sub_pixel_variance(a, ref_b.shape[1], xo, yo, b, src_b.shape[1], w, h, e16, bd, kind, sp, fwd_offset,
                                                     bck_offset, mp, mask_stride, invert_mask, C.byref(q))
        sse[i] = q.value
        if not subpel:
            if kind == 2:
                sad[i] = lib.orc_masked_sad(b, src_b.shape[1], a, ref_b.shape[1], sp, mp, mask_stride, invert_mask, w, h, e16, bd)
            else:
                sad[i] = lib.orc_sad_avg_any(b, src_b.shape[1], a, ref_b.shape[1], sp, w, h, e16, bd, fwd_offset if kind == 1 else 0,
                                             bck_offset if kind == 1 else 0)
    return var, sse, sad


def build_inter_pred(ref_b, border, width, height, w, h, blocks, mvs, filter_x=0, filter_y=0, bd=8, ss_x=0, ss_y=0):
    """oracle/aomref_convolve.c over a block list: returns the visible height x width prediction plane (zeros where no
    block wrote).  ref_b: border-extended reference plane; mvs: (row, col) per block in 1/8 pel."""
    lib.orc_build_inter_pred_block_ss.restype = None
    e16 = int(ref_b.dtype != np.uint8)
    out = np.zeros((height, width), ref_b.dtype)
    origin = C.c_void_p(_addr(ref_b, border, border))
    for b, mv in zip(blocks, mvs):
        x, y = int(b["bx"]), int(b["by"])
        lib.orc_build_inter_pred_block_ss(origin, ref_b.shape[1], C.c_void_p(_addr(out, y, x)), width, x, y, w, h, int(mv[0]), int(mv[1]),
                                          filter_x, filter_y, e16, bd, ss_x, ss_y)
    return out


def tf_apply_frames(planes_b, border, width, height, filter_frame, mvs, mses, noise_levels, q_factor, filter_strength, bd=8, ss_x=0, ss_y=0,
                    present=None, threads=8, block_first=0, block_step=1):
    """oracle/aomref_tf.c orc_tf_apply_frames.  planes_b: per component (1 or 3) a list of border-extended frames of the window
    (every plane with the same `border`); mvs (F, n_blocks, 4, 2) int16, mses (F, n_blocks, 4) int32 as aomhip_tf_motion_search_frames
    writes them.  Returns the filtered planes (border-extended arrays of the same shapes; only the block-covered area is written)."""
    P, F = len(planes_b), len(planes_b[0])
    e16 = int(planes_b[0][0].dtype != np.uint8)
    origins = (C.c_void_p * (3 * F))()
    strides = (C.c_int * 3)()
    for p in range(P):
        strides[p] = planes_b[p][0].shape[1]
        for f in range(F):
            assert planes_b[p][f].shape == planes_b[p][0].shape
            if present is None or present[f]:
                origins[f * 3 + p] = _addr(planes_b[p][f], border, border)
    outs = [np.zeros_like(planes_b[p][0]) for p in range(P)]
    out_ptrs = (C.c_void_p * 3)(*[_addr(outs[p], border, border) for p in range(P)] + [None] * (3 - P))
    ostr = (C.c_int * 3)(*[outs[p].shape[1] for p in range(P)] + [0] * (3 - P))
    nl = (C.c_double * 3)(*([float(v) for v in noise_levels] + [0.0] * 3)[:3])
    mv = np.ascontiguousarray(mvs, np.int16)
    ms = np.ascontiguousarray(mses, np.int32)
    f = lib.orc_tf_apply_frames
    f.restype = None
    f.argtypes = None
    f(origins, strides, C.c_int(F), C.c_int(filter_frame), C.c_int(width), C.c_int(height), C.c_int(P), C.c_int(ss_x), C.c_int(ss_y), nl,
      C.c_void_p(mv.ctypes.data), C.c_void_p(ms.ctypes.data), C.c_int(q_factor), C.c_int(filter_strength), out_ptrs, ostr, C.c_int(e16),
      C.c_int(bd), C.c_int(threads), C.c_int(block_first), C.c_int(block_step))
    return outs


def build_compound_pred(ref0_b, ref1_b, border, width, height, w, h, blocks, mv0, mv1, filter_x=0, filter_y=0, fwd=0, bck=0, bd=8, ss_x=0, ss_y=0):
    """orc_convolve_compound over a block list (two border-extended references, two MV lists in 1/8 pel)."""
    lib.orc_convolve_compound.restype = None
    e16 = int(ref0_b.dtype != np.uint8)
    out = np.zeros((height, width), ref0_b.dtype)
    for b, a, c in zip(blocks, mv0, mv1):
        x, y = int(b["bx"]), int(b["by"])
        p = []
        for ref, mv in ((ref0_b, a), (ref1_b, c)):
            px, py = (x << 4) + int(mv[1]) * (1 << (1 - ss_x)), (y << 4) + int(mv[0]) * (1 << (1 - ss_y))
            p.append((C.c_void_p(_addr(ref, border + (py >> 4), border + (px >> 4))), ref.shape[1], px & 15, py & 15))
        lib.orc_convolve_compound(p[0][0], p[0][1], p[0][2], p[0][3], p[1][0], p[1][1], p[1][2], p[1][3], C.c_void_p(_addr(out, y, x)), width, w, h,
                                  filter_x, filter_y, fwd, bck, e16, bd)
    return out


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


def physical_cores():
    """(physical cores, logical CPUs, model name) of this host (Linux /proc/cpuinfo; SMT siblings counted once)."""
    cores, model, logical = set(), "unknown", 0
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "processor":
                logical += 1
            elif k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
                cores.add((phys, core))
    except OSError:
        pass
    import os
    logical = logical or (os.cpu_count() or 1)
    n = len(cores) or logical
    if n * 4 < logical:  # a virtualised /proc/cpuinfo that repeats core ids: assume 2-way SMT rather than report 2 cores
        n = max(1, logical // 2)
    return n, logical, model


def usable_cpus():
    """CPUs this process may actually use at once: min(scheduler affinity, cgroup CPU quota -- cpu.max of cgroup v2 or
    cfs_quota_us / cfs_period_us of v1).  A container on a shared host sees every core in /proc/cpuinfo but is throttled to its
    quota: more threads than that only time-share it.  -> (usable, quota or None)"""
    import math
    import os
    n = _CPUS_AT_IMPORT
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(math.floor(quota + 1e-9))))
    return n, quota


def bench_sad_mode_a(src_planes, ref_planes, border, cands, groups, bd, threads, avx2, seconds):
    """bench.py's CPU baseline for the Mode-A SAD workload (oracle/aomref_bench.c): F bordered frame pairs, the shared
    single-candidate list and F per-frame x4d lists; -> (candidates/s, candidates, elapsed s)."""
    F = len(src_planes)
    cands, groups = np.ascontiguousarray(cands), np.ascontiguousarray(groups)
    n = len(cands)
    assert groups.shape[0] == F * n
    so = (C.c_void_p * F)(*[_addr(p, border, border) for p in src_planes])
    ro = (C.c_void_p * F)(*[_addr(p, border, border) for p in ref_planes])
    el, ck = C.c_double(), C.c_ulonglong()
    f = lib.orc_bench_sad_mode_a
    f.restype = C.c_longlong
    f.argtypes = None
    done = f(so, ro, C.c_int(F), C.c_int(src_planes[0].shape[1]), C.c_int(ref_planes[0].shape[1]),
             C.c_int(int(src_planes[0].dtype != np.uint8)), C.c_int(bd), C.c_void_p(cands.ctypes.data),
             C.c_void_p(groups.ctypes.data), C.c_int(n), C.c_int(threads), C.c_int(int(avx2)), C.c_double(seconds),
             C.byref(el), C.byref(ck))
    return done / el.value, done, el.value


def bench_txq(planes, q, threads, avx2_quant, seconds, bd=8):
    """bench.py's CPU baseline for fwd_txfm2d + quantize_b over all 4x4..32x32 blocks of int16 residual planes
    (bd > 8: aom_highbd_quantize_b and the transform's bd stage ranges); -> (blocks/s, blocks, elapsed s)."""
    P = len(planes)
    H, W = planes[0].shape
    ptrs = (C.c_void_p * P)(*[p.ctypes.data for p in planes])
    qa = np.ascontiguousarray([q[k] for k in ("zbin", "round", "quant", "quant_shift", "dequant")], np.int16)
    el, ck = C.c_double(), C.c_ulonglong()
    f = lib.orc_bench_txq_bd
    f.restype = C.c_longlong
    f.argtypes = None
    done = f(ptrs, C.c_int(P), C.c_int(W), C.c_int(H), C.c_void_p(qa.ctypes.data), C.c_int(threads), C.c_int(int(avx2_quant)),
             C.c_int(int(bd)), C.c_double(seconds), C.byref(el), C.byref(ck))
    return done / el.value, done, el.value


def sad_x4d_batch(src_b, ref_b, border, w, h, groups, skip=False, bd=8, threads=1, reps=1):
    groups = np.ascontiguousarray(groups)
    out = np.empty((len(groups), 4), np.uint32)
    lib.orc_sad_x4d_batch(_addr(src_b, border, border), src_b.shape[1], _addr(ref_b, border, border), ref_b.shape[1],
                          int(src_b.dtype != np.uint8), bd, w, h, int(skip), groups.ctypes.data, len(groups),
                          out.ctypes.data, threads, reps)
    return out


# ---- transforms / quantize / tables (aomref_txfm.c, aomref_quant.c)
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C")
_i16 = np.ctypeslib.ndpointer(np.int16, flags="C")
lib.orc_fwd_txfm1d.argtypes = [_i, _i, _i32, _i32, _i]
lib.orc_inv_txfm1d.argtypes = [_i, _i, _i32, _i32, _i, _i]
lib.orc_txfm_valid.restype = _i
lib.orc_txfm_valid.argtypes = [_i, _i]
lib.orc_fwd_txfm2d.argtypes = [_vp, _i32, _i, _i, _i, _i]
lib.orc_inv_txfm2d_add.argtypes = [_i32, _vp, _i, _i, _i, _i]
lib.orc_get_scan.restype = _i
lib.orc_get_scan.argtypes = [_i, _i, _i16, _i16]
lib.orc_dc_q.restype = C.c_int16
lib.orc_dc_q.argtypes = [_i, _i, _i]
lib.orc_ac_q.restype = C.c_int16
lib.orc_ac_q.argtypes = [_i, _i, _i]
lib.orc_build_quantizer_y.argtypes = [_i, _i, _i16]
for _f in (lib.orc_quantize_b, lib.orc_highbd_quantize_b):
    _f.restype = None
    _f.argtypes = [_i32, C.c_ssize_t, _i16, _i16, _i16, _i16, _i32, _i32, _i16, C.POINTER(C.c_uint16), _i16, _i16, _i]

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
V_KIND = [0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3]  # per TX_TYPE: 0 DCT 1 ADST 2 FLIPADST 3 IDTX
H_KIND = [0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2]


def av1_tx_valid(tx_size, tx_type):
    """test/av1_txfm_test.h:89-100 IsTxSizeTypeValid: the (size, type) pairs AV1 actually uses."""
    m = max(TX_W[tx_size], TX_H[tx_size])
    if m > 32:
        return tx_type == 0
    if m == 32:
        return tx_type in (0, 9)
    return True


def cospi_table():
    return np.ctypeslib.as_array((C.c_int32 * (7 * 64)).in_dll(lib, "orc_cospi")).reshape(7, 64).copy()


def sinpi_table():
    return np.ctypeslib.as_array((C.c_int32 * (7 * 5)).in_dll(lib, "orc_sinpi")).reshape(7, 5).copy()


def fwd_txfm1d(kind, x, cos_bit):
    x = np.ascontiguousarray(x, np.int32)
    o = np.zeros_like(x)
    lib.orc_fwd_txfm1d(kind, x.size, x, o, cos_bit)
    return o


def inv_txfm1d(kind, x, cos_bit, clamp_bit):
    x = np.ascontiguousarray(x, np.int32)
    o = np.zeros_like(x)
    lib.orc_inv_txfm1d(kind, x.size, x, o, cos_bit, clamp_bit)
    return o


def fwd_txfm2d(block, tx_size, tx_type, bd=8):
    """block: int16 [h, w] residual -> int32 [w*h] coefficients in the reference's transposed layout."""
    block = np.ascontiguousarray(block, np.int16)
    h, w = block.shape
    assert (w, h) == (TX_W[tx_size], TX_H[tx_size])
    out = np.zeros(w * h, np.int32)
    lib.orc_fwd_txfm2d(block.ctypes.data, out, w, tx_size, tx_type, bd)
    return out


def inv_txfm2d_add(coeff, dst, tx_size, tx_type, bd):
    coeff = np.ascontiguousarray(coeff, np.int32)
    dst = np.ascontiguousarray(dst, np.uint16).copy()
    lib.orc_inv_txfm2d_add(coeff, dst.ctypes.data, dst.shape[1], tx_size, tx_type, bd)
    return dst


def get_scan(tx_size, tx_type):
    n = min(TX_W[tx_size], 32) * min(TX_H[tx_size], 32)
    s, i = np.zeros(n, np.int16), np.zeros(n, np.int16)
    assert lib.orc_get_scan(tx_size, tx_type, s, i) == n
    return s, i


def build_quantizer_y(bit_depth, qindex):
    """-> dict of int16[2] (DC, AC): zbin, round, quant, quant_shift, dequant."""
    t = np.zeros((5, 2), np.int16)
    lib.orc_build_quantizer_y(bit_depth, qindex, t)
    return dict(zip(("zbin", "round", "quant", "quant_shift", "dequant"), t))


def quantize_b(coeff, q, scan, iscan, log_scale, highbd=False, qm=None, iqm=None):
    """aom_[highbd_]quantize_b_helper_c; qm / iqm: uint8 quantisation matrices indexed like the coefficients (None: flat)."""
    coeff = np.ascontiguousarray(coeff, np.int32)
    qc, dq = np.zeros_like(coeff), np.zeros_like(coeff)
    eob = C.c_uint16()
    if qm is not None or iqm is not None:
        qm_ = np.ascontiguousarray(qm, np.uint8) if qm is not None else None
        iqm_ = np.ascontiguousarray(iqm, np.uint8) if iqm is not None else None
        f = lib.orc_highbd_quantize_b_qm if highbd else lib.orc_quantize_b_qm
        f.restype = None
        f(C.c_void_p(coeff.ctypes.data), C.c_ssize_t(coeff.size), *[C.c_void_p(np.ascontiguousarray(q[k], np.int16).ctypes.data) for k in ("zbin", "round", "quant", "quant_shift")],
          C.c_void_p(qc.ctypes.data), C.c_void_p(dq.ctypes.data), C.c_void_p(np.ascontiguousarray(q["dequant"], np.int16).ctypes.data), C.byref(eob),
          C.c_void_p(np.ascontiguousarray(scan, np.int16).ctypes.data), C.c_int(log_scale), C.c_void_p(qm_.ctypes.data if qm_ is not None else None),
          C.c_void_p(iqm_.ctypes.data if iqm_ is not None else None))
        return qc, dq, eob.value
    f = lib.orc_highbd_quantize_b if highbd else lib.orc_quantize_b
    f(coeff, coeff.size, q["zbin"], q["round"], q["quant"], q["quant_shift"], qc, dq, q["dequant"], C.byref(eob),
      np.ascontiguousarray(scan, np.int16), np.ascontiguousarray(iscan, np.int16), log_scale)
    return qc, dq, eob.value


def quantize_b_adaptive(coeff, q, scan, log_scale, highbd=False):
    coeff = np.ascontiguousarray(coeff, np.int32)
    qc, dq = np.zeros_like(coeff), np.zeros_like(coeff)
    eob = C.c_uint16()
    sc = np.ascontiguousarray(scan, np.int16)
    tabs = [np.ascontiguousarray(q[k], np.int16) for k in ("zbin", "round", "quant", "quant_shift", "dequant")]
    lib.orc_quantize_b_adaptive.restype = None
    lib.orc_quantize_b_adaptive(C.c_void_p(coeff.ctypes.data), C.c_ssize_t(coeff.size), C.c_void_p(tabs[0].ctypes.data),
                                C.c_void_p(tabs[1].ctypes.data), C.c_void_p(tabs[2].ctypes.data), C.c_void_p(tabs[3].ctypes.data),
                                C.c_void_p(qc.ctypes.data), C.c_void_p(dq.ctypes.data), C.c_void_p(tabs[4].ctypes.data),
                                C.byref(eob), C.c_void_p(sc.ctypes.data), log_scale, int(highbd))
    return qc, dq, eob.value


lib.orc_xform_quant_batch.restype = None
lib.orc_xform_quant_batch.argtypes = [_vp, _i, _i, _vp, _i, _i, _i, _i16, _i, _vp, _vp, _vp, _vp, _i, _i]


def xform_quant_batch(residual, tx_size, blocks, n, grid_cols, tx_type, q, is_hbd, total_coeffs, want_coeff=True,
                      threads=1, reps=1, qm=None, iqm=None):
    """residual: int16 2-D array; blocks: structured array (x,y,out_offset,tx_type) or None (grid mode);
    q: dict from build_quantizer_y; qm / iqm: the quantisation matrices of this transform size (uint8, coefficient order) or None.
    -> (coeff|None, qcoeff, dqcoeff, eob)"""
    residual = np.ascontiguousarray(residual, np.int16)
    if qm is not None:
        qm, iqm = np.ascontiguousarray(qm, np.uint8), np.ascontiguousarray(iqm, np.uint8)
        lib.orc_set_qm.restype = None
        lib.orc_set_qm(C.c_void_p(qm.ctypes.data), C.c_void_p(iqm.ctypes.data))
        try:
            return xform_quant_batch(residual, tx_size, blocks, n, grid_cols, tx_type, q, is_hbd, total_coeffs, want_coeff, threads, reps)
        finally:
            lib.orc_set_qm(None, None)
    qt = np.ascontiguousarray(np.stack([q[k] for k in ("zbin", "round", "quant", "quant_shift", "dequant")]), np.int16)
    coeff = np.zeros(total_coeffs, np.int32) if want_coeff else None
    qc, dq = np.zeros(total_coeffs, np.int32), np.zeros(total_coeffs, np.int32)
    eob = np.zeros(n, np.uint16)
    bl = np.ascontiguousarray(blocks) if blocks is not None else None
    lib.orc_xform_quant_batch(residual.ctypes.data, residual.shape[1], tx_size, bl.ctypes.data if bl is not None else None,
                              n, grid_cols, tx_type, qt, int(is_hbd), coeff.ctypes.data if want_coeff else None,
                              qc.ctypes.data, dq.ctypes.data, eob.ctypes.data, threads, reps)
    return coeff, qc, dq, eob


def variance_cands(src_b, ref_b, border, w, h, cands, subpel=False, bd=8):
    """Per-candidate (var, sse) for a structured candidate array (sx,sy,rx,ry,xoff,yoff).
    plain: variance(src, ref); sub-pixel: sub_pixel_variance(ref, xoff, yoff, src) -- the reference's call shapes."""
    out = np.zeros((len(cands), 2), np.uint32)
    for i, c in enumerate(cands):
        sy, sx, ry, rx = int(c["sy"]) + border, int(c["sx"]) + border, int(c["ry"]) + border, int(c["rx"]) + border
        if subpel:
            out[i] = sub_pixel_variance(ref_b, ry, rx, int(c["xoff"]), int(c["yoff"]), src_b, sy, sx, w, h, bd)
        else:
            v, sse, _ = variance(src_b, sy, sx, ref_b, ry, rx, w, h, bd)
            out[i] = (v, sse)
    return out


lib.orc_inv_txfm_add_batch.restype = None
lib.orc_inv_txfm_add_batch.argtypes = [_i32, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i]


def inv_txfm_add_batch(dqcoeff, tx_size, blocks, n, grid_cols, tx_type, eob, dst, bd):
    """dst: 2-D pixel array (uint8 or uint16); returns the reconstructed copy (same dtype)."""
    work = np.ascontiguousarray(dst, np.uint16).copy()
    bl = np.ascontiguousarray(blocks) if blocks is not None else None
    e = np.ascontiguousarray(eob, np.uint16) if eob is not None else None
    lib.orc_inv_txfm_add_batch(np.ascontiguousarray(dqcoeff, np.int32), tx_size, bl.ctypes.data if bl is not None else None,
                               n, grid_cols, tx_type, e.ctypes.data if e is not None else None, work.ctypes.data,
                               work.shape[1], bd)
    return work.astype(dst.dtype)


# ---- deblocking (aomref_lpf.c)
lib.orc_lpf.restype = None
lib.orc_lpf.argtypes = [_vp, _i, _i, _i, C.c_uint8, C.c_uint8, C.c_uint8]
lib.orc_highbd_lpf.restype = None
lib.orc_highbd_lpf.argtypes = [_vp, _i, _i, _i, C.c_uint8, C.c_uint8, C.c_uint8, _i]
lib.orc_lpf_thresholds.restype = None
lib.orc_lpf_thresholds.argtypes = [_i, _i, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
lib.orc_deblock_plane.restype = None
lib.orc_deblock_plane.argtypes = [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i]


def lpf_edge(pixels, y, x, vertical, length, blimit, limit, thresh, bd=8):
    """Filter one 4-px edge unit in place; (y, x) is q0 of the first pixel line."""
    if pixels.dtype == np.uint8:
        lib.orc_lpf(_addr(pixels, y, x), pixels.shape[1], int(vertical), length, blimit, limit, thresh)
    else:
        lib.orc_highbd_lpf(_addr(pixels, y, x), pixels.shape[1], int(vertical), length, blimit, limit, thresh, bd)


def deblock_plane(pixels, params, sharpness=0, bd=8, order=0):
    """pixels: 2-D visible plane; params: uint8 [urows, ucols, 4] -> filtered copy."""
    out = np.ascontiguousarray(pixels).copy()
    params = np.ascontiguousarray(params, np.uint8)
    h, w = out.shape
    lib.orc_deblock_plane(out.ctypes.data, out.shape[1], w, h, int(out.dtype != np.uint8), bd, params.ctypes.data,
                          params.shape[1], sharpness, order)
    return out


def random_edge_params(rng, width, height, max_log2=6, level_range=(1, 64), chroma=False):
    """Edge records consistent with a random transform partition (so that footprints never overlap, as in the
    reference): each 64x64 superblock is split quad-tree style into square transforms of 4..64 px; every
    transform carries a random level; len = tx_dim_to_filter_length[min(tx on both sides)]."""
    ucols, urows = (width + 3) // 4, (height + 3) // 4
    txs = np.zeros((urows, ucols), np.uint8)   # transform size (px) covering each 4x4 unit
    lvl = np.zeros((urows, ucols), np.uint8)
    org = np.zeros((urows, ucols, 2), np.int32)  # top-left unit of the covering transform

    def split(y0, x0, size):
        if size > 4 and (size > (1 << max_log2) or rng.random() < 0.55):
            h = size // 2
            for dy in (0, h):
                for dx in (0, h):
                    split(y0 + dy, x0 + dx, h)
            return
        u = size // 4
        uy, ux = y0 // 4, x0 // 4
        if uy >= urows or ux >= ucols:
            return
        txs[uy:uy + u, ux:ux + u] = size
        lvl[uy:uy + u, ux:ux + u] = rng.integers(level_range[0], level_range[1])
        org[uy:uy + u, ux:ux + u] = (uy, ux)
    for y0 in range(0, height, 64):
        for x0 in range(0, width, 64):
            split(y0, x0, 64)
    flen = (lambda t: 4 if t == 4 else 6) if chroma else (lambda t: 4 if t == 4 else 8 if t == 8 else 14)
    p = np.zeros((urows, ucols, 4), np.uint8)
    for uy in range(urows):
        for ux in range(ucols):
            if ux > 0 and org[uy, ux, 1] == ux:      # a transform starts here: vertical edge on its left
                t = min(txs[uy, ux], txs[uy, ux - 1])
                l = lvl[uy, ux] if lvl[uy, ux] else lvl[uy, ux - 1]  # av1_loopfilter.c:285-297: current level, else previous
                p[uy, ux, 0], p[uy, ux, 1] = flen(t), l
            if uy > 0 and org[uy, ux, 0] == uy:
                t = min(txs[uy, ux], txs[uy - 1, ux])
                l = lvl[uy, ux] if lvl[uy, ux] else lvl[uy - 1, ux]
                p[uy, ux, 2], p[uy, ux, 3] = flen(t), l
    return p


# ---- CDEF (aomref_cdef.c)
lib.orc_cdef_find_dir.restype = _i
lib.orc_cdef_find_dir.argtypes = [_vp, _i, C.POINTER(C.c_int32), _i]
lib.orc_cdef_plane_luma.restype = None
lib.orc_cdef_plane_luma.argtypes = [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp]


def cdef_find_dir(block16, coeff_shift=0):
    b = np.ascontiguousarray(block16, np.uint16)
    var = C.c_int32()
    d = lib.orc_cdef_find_dir(b.ctypes.data, b.shape[1], C.byref(var), coeff_shift)
    return d, var.value


def cdef_plane_luma(pixels, fb_pri, fb_sec, skip, damping, bd=8):
    """pixels: 2-D plane (uint8/uint16); fb_pri/fb_sec: [fb_rows, fb_cols] uint8; skip: [h/8, w/8] uint8.
    -> (filtered plane, dir [h/8, w/8] uint8, var int32)"""
    src = np.ascontiguousarray(pixels)
    dst = np.zeros_like(src)
    h, w = src.shape
    fb_pri = np.ascontiguousarray(fb_pri, np.uint8); fb_sec = np.ascontiguousarray(fb_sec, np.uint8)
    skip = np.ascontiguousarray(skip, np.uint8)
    d = np.zeros((h // 8, w // 8), np.uint8); v = np.zeros((h // 8, w // 8), np.int32)
    lib.orc_cdef_plane_luma(src.ctypes.data, dst.ctypes.data, src.shape[1], w, h, int(src.dtype != np.uint8), bd,
                            fb_pri.ctypes.data, fb_sec.ctypes.data, fb_pri.shape[1], skip.ctypes.data, damping,
                            d.ctypes.data, v.ctypes.data)
    return dst, d, v


def cdef_search_sse_luma(recon, source, strengths, skip, damping, bd=8):
    """The luma distortion table of av1_cdef_search (pickcdef.c:401-615) restated on top of cdef_plane_luma: for each
    (pri, sec) the whole plane is filtered with that strength in every filter block and the squared error against
    `source` is summed per 64x64 filter block over the non-skip 8x8 units.  -> uint64 [n_strengths, fb_rows, fb_cols] raw sums."""
    h, w = recon.shape
    fbh, fbw = (h + 63) // 64, (w + 63) // 64
    out = np.zeros((len(strengths), fbh, fbw), np.uint64)
    keep = np.kron((np.asarray(skip) == 0).astype(np.int64), np.ones((8, 8), np.int64))[:h, :w]
    for gi, (pri, sec) in enumerate(strengths):
        filt, _, _ = cdef_plane_luma(recon, np.full((fbh, fbw), pri, np.uint8), np.full((fbh, fbw), sec, np.uint8), skip, damping, bd)
        e = (filt.astype(np.int64) - source.astype(np.int64)) ** 2 * keep
        for r in range(fbh):
            for c in range(fbw):
                out[gi, r, c] = e[r * 64:(r + 1) * 64, c * 64:(c + 1) * 64].sum()
    return out


def cdef_search_sse_chroma(recon, source, xdec, ydec, luma_dir, strengths, skip, damping, bd=8):
    """cdef_search_sse_luma for a chroma plane (built on cdef_plane_chroma): raw sums [n_strengths, fb_rows, fb_cols]."""
    h, w = recon.shape
    fh, fw = 64 >> ydec, 64 >> xdec
    fbh, fbw = (h + fh - 1) // fh, (w + fw - 1) // fw
    out = np.zeros((len(strengths), fbh, fbw), np.uint64)
    keep = np.kron((np.asarray(skip) == 0).astype(np.int64), np.ones((8 >> ydec, 8 >> xdec), np.int64))[:h, :w]
    for gi, (pri, sec) in enumerate(strengths):
        filt = cdef_plane_chroma(recon, xdec, ydec, luma_dir, np.full((fbh, fbw), pri, np.uint8), np.full((fbh, fbw), sec, np.uint8), skip, damping, bd)
        e = (filt.astype(np.int64) - source.astype(np.int64)) ** 2 * keep
        for r in range(fbh):
            for c in range(fbw):
                out[gi, r, c] = e[r * fh:(r + 1) * fh, c * fw:(c + 1) * fw].sum()
    return out


def cdef_plane_chroma(pixels, xdec, ydec, luma_dir, fb_pri, fb_sec, skip, damping, bd=8):
    """pixels: chroma plane; luma_dir: [h_blocks, w_blocks] uint8 from cdef_plane_luma; fb_pri / fb_sec: uv strengths."""
    src = np.ascontiguousarray(pixels)
    dst = np.zeros_like(src)
    h, w = src.shape
    d = np.ascontiguousarray(luma_dir, np.uint8)
    fb_pri = np.ascontiguousarray(fb_pri, np.uint8); fb_sec = np.ascontiguousarray(fb_sec, np.uint8)
    skip = np.ascontiguousarray(skip, np.uint8)
    lib.orc_cdef_plane_chroma.restype = None
    lib.orc_cdef_plane_chroma(C.c_void_p(src.ctypes.data), C.c_void_p(dst.ctypes.data), src.shape[1], w, h,
                              int(src.dtype != np.uint8), bd, xdec, ydec, C.c_void_p(d.ctypes.data),
                              C.c_void_p(fb_pri.ctypes.data), C.c_void_p(fb_sec.ctypes.data), fb_pri.shape[1],
                              C.c_void_p(skip.ctypes.data), damping)
    return dst


# ---- motion search (aomref_mcomp.c)
lib.orc_fullpel_diamond_batch.restype = None
lib.orc_fullpel_diamond_batch.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _i]
lib.orc_subpel_bilinear_batch.restype = None
lib.orc_subpel_bilinear_batch.argtypes = [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i]


def fullpel_diamond_batch(src_b, ref_b, border, w, h, blocks, clamped=0, step_param=4, cost_type=3, bd=8, threads=4):
    blocks = np.ascontiguousarray(blocks)
    mv = np.zeros((len(blocks), 2), np.int16); cost = np.zeros(len(blocks), np.int32)
    lib.orc_fullpel_diamond_batch(_addr(src_b, border, border), src_b.shape[1], _addr(ref_b, border, border),
                                  ref_b.shape[1], int(src_b.dtype != np.uint8), bd, w, h, clamped, step_param, cost_type,
                                  blocks.ctypes.data, len(blocks), mv.ctypes.data, cost.ctypes.data, threads)
    return mv, cost


GOOD_QUALITY_MESH_PATTERNS = [  # av1/encoder/speed_features.c:25-33, indexed by mesh speed
    [(64, 8), (28, 4), (15, 1), (7, 1)], [(64, 8), (28, 4), (15, 1), (7, 1)], [(64, 8), (14, 2), (7, 1), (7, 1)],
    [(64, 16), (24, 8), (12, 4), (7, 1)], [(64, 16), (24, 8), (12, 4), (7, 1)], [(64, 16), (24, 8), (12, 4), (7, 1)]]


def mesh_search_batch(src_b, ref_b, border, w, h, blocks, patterns, fine_search_interval=0, cost_type=3, bd=8, threads=4):
    blocks = np.ascontiguousarray(blocks)
    pat = np.ascontiguousarray(np.asarray(patterns, np.int32).reshape(-1))
    assert pat.size == 8
    mv = np.zeros((len(blocks), 2), np.int16); cost = np.zeros(len(blocks), np.int32)
    lib.orc_mesh_search_batch.restype = None
    lib.orc_mesh_search_batch(C.c_void_p(_addr(src_b, border, border)), src_b.shape[1], C.c_void_p(_addr(ref_b, border, border)),
                              ref_b.shape[1], int(src_b.dtype != np.uint8), bd, w, h, cost_type, C.c_void_p(pat.ctypes.data),
                              fine_search_interval, C.c_void_p(blocks.ctypes.data), len(blocks), C.c_void_p(mv.ctypes.data),
                              C.c_void_p(cost.ctypes.data), threads)
    return mv, cost


def subpel_bilinear_batch(src_b, ref_b, border, w, h, blocks, cost_type=3, iters=2, allow_hp=1, forced_stop=0, bd=8,
                          threads=4):
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    mv = np.zeros((n, 2), np.int16); err = np.zeros(n, np.uint32); dist = np.zeros(n, np.int32); sse = np.zeros(n, np.uint32)
    lib.orc_subpel_bilinear_batch(_addr(src_b, border, border), src_b.shape[1], _addr(ref_b, border, border),
                                  ref_b.shape[1], int(src_b.dtype != np.uint8), bd, w, h, cost_type, iters, allow_hp,
                                  forced_stop, blocks.ctypes.data, n, mv.ctypes.data, err.ctypes.data, dist.ctypes.data,
                                  sse.ctypes.data, threads)
    return mv, err, dist, sse


SUBPEL_TREES = {"pruned_more": 0, "pruned": 1, "tree": 2}


def subpel_tree_batch(src_b, ref_b, border, w, h, blocks, tree="pruned_more", cost_type=3, error_per_bit=0, mvjcost=None,
                      mvcost0=None, mvcost1=None, iters=2, allow_hp=1, forced_stop=0, cost_lists=None, bd=8, threads=4,
                      subpel_search_type=0, mv_lists=None):
    """The three bilinear sub-pel trees (tree: pruned_more / pruned / tree) with an optional per-block cost list and an optional
    last_mv_search_list per block (mv_lists: int16 [n, 3, 2], INVALID_MV = -32768, updated in place; check_repeated_mv_and_update)."""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    mv = np.zeros((n, 2), np.int16); err = np.zeros(n, np.uint32); dist = np.zeros(n, np.int32); sse = np.zeros(n, np.uint32)
    keep = []
    def centre(t):
        if t is None:
            return None
        t = np.ascontiguousarray(t, np.int32); keep.append(t)
        return C.c_void_p(t.ctypes.data + (t.size // 2) * 4)
    j = None
    if mvjcost is not None:
        jj = np.ascontiguousarray(mvjcost, np.int32); keep.append(jj); j = C.c_void_p(jj.ctypes.data)
    cl = None
    if cost_lists is not None:
        cla = np.ascontiguousarray(cost_lists, np.int32).reshape(n, 5); keep.append(cla); cl = C.c_void_p(cla.ctypes.data)
    ml = None
    if mv_lists is not None:
        assert mv_lists.dtype == np.int16 and mv_lists.shape == (n, 3, 2) and mv_lists.flags.c_contiguous
        ml = C.c_void_p(mv_lists.ctypes.data)
    lib.orc_subpel_tree_batch_list.restype = None
    lib.orc_subpel_tree_batch_list(C.c_void_p(_addr(src_b, border, border)), src_b.shape[1], C.c_void_p(_addr(ref_b, border, border)),
                                   ref_b.shape[1], int(src_b.dtype != np.uint8), bd, w, h, SUBPEL_TREES.get(tree, tree), subpel_search_type,
                                   cost_type, error_per_bit, j, centre(mvcost0), centre(mvcost1), iters, allow_hp, forced_stop,
                                   C.c_void_p(blocks.ctypes.data), cl, n, C.c_void_p(mv.ctypes.data), C.c_void_p(err.ctypes.data),
                                   C.c_void_p(dist.ctypes.data), C.c_void_p(sse.ctypes.data), threads, ml)
    return mv, err, dist, sse


SEARCH_METHODS = ["DIAMOND", "NSTEP", "NSTEP_8PT", "CLAMPED_DIAMOND", "HEX", "BIGDIA", "SQUARE", "FAST_HEX", "FAST_DIAMOND",
                  "FAST_BIGDIA", "VFAST_DIAMOND",      # SEARCH_METHODS values, mcomp_structs.h:50-83
                  "NSTEP_FPF"]                         # + NSTEP on the first-pass site table (av1_init_motion_fpf)


class SearchParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("search_method", "step_param", "cost_type", "sad_per_bit", "error_per_bit", "skip_sad",
                                         "run_mesh_search", "prune_mesh_search", "mesh_search_mv_diff_threshold",
                                         "force_mesh_thresh", "fine_search_interval")] + [("mesh_patterns", C.c_int32 * 8), ("no_cost_list", C.c_int32)]


def search_params(method, step_param, cost_type, sad_per_bit=0, error_per_bit=0, skip_sad=0, run_mesh=0, prune_mesh=0,
                  mesh_diff_thr=0, force_mesh_thresh=2147483647, fine_interval=0, mesh=None, no_cost_list=0):
    q = SearchParams(method if isinstance(method, int) else SEARCH_METHODS.index(method), step_param, cost_type, sad_per_bit,
                     error_per_bit, int(skip_sad), run_mesh, prune_mesh, mesh_diff_thr, force_mesh_thresh, fine_interval)
    for i, v in enumerate(np.asarray(mesh if mesh is not None else [[0, 0]] * 4).reshape(-1)):
        q.mesh_patterns[i] = int(v)
    q.no_cost_list = int(no_cost_list)
    return q


def full_pixel_search_batch(src_b, ref_b, border, w, h, blocks, q, mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4):
    """av1_full_pixel_search per block.  mvcost0/1: full tables (odd length); their centres are passed on.
    -> mv [n,2], cost [n], cost_list [n,5], second_best [n,2]"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    mv = np.zeros((n, 2), np.int16); cost = np.zeros(n, np.int32); cl = np.zeros((n, 5), np.int32); sec = np.zeros((n, 2), np.int16)
    keep = []
    def centre(t):
        if t is None:
            return None
        t = np.ascontiguousarray(t, np.int32); keep.append(t)
        return C.c_void_p(t.ctypes.data + (t.size // 2) * 4)
    j = None
    if mvjcost is not None:
        jj = np.ascontiguousarray(mvjcost, np.int32); keep.append(jj); j = C.c_void_p(jj.ctypes.data)
    lib.orc_full_pixel_search_batch.restype = None
    lib.orc_full_pixel_search_batch(C.c_void_p(_addr(src_b, border, border)), src_b.shape[1], C.c_void_p(_addr(ref_b, border, border)),
                                    ref_b.shape[1], int(src_b.dtype != np.uint8), bd, w, h, C.byref(q), j, centre(mvcost0), centre(mvcost1),
                                    C.c_void_p(blocks.ctypes.data), n, C.c_void_p(mv.ctypes.data), C.c_void_p(cost.ctypes.data),
                                    C.c_void_p(cl.ctypes.data), C.c_void_p(sec.ctypes.data), threads)
    return mv, cost, cl, sec


def search_sites(method):
    """-> (num_search_steps, searches_per_step[22], radius[22], mv[22,17,2]) of the method's site table."""
    ns = C.c_int()
    per, rad, mv = np.zeros(22, np.int32), np.zeros(22, np.int32), np.zeros((22, 17, 2), np.int16)
    lib.orc_search_sites_dump(method if isinstance(method, int) else SEARCH_METHODS.index(method), C.byref(ns),
                              C.c_void_p(per.ctypes.data), C.c_void_p(rad.ctypes.data), C.c_void_p(mv.ctypes.data))
    return ns.value, per, rad, mv


def mv_limits_for_block(bx, by, w, h, width, height, border, ref_row=0, ref_col=0):
    """av1_set_mv_limits (mcomp.h:216-247; frame-relative, may reach border - 2*AOM_INTERP_EXTEND(4) outside the
    frame... here: block + interp extend stays inside the replicated border) intersected with
    av1_set_mv_search_range around ref_mv (mcomp.c:196-215, MAX_FULL_PEL_VAL 1023)."""
    ext = border - 8
    col_min, col_max = -(bx + ext), (width - bx - w) + ext
    row_min, row_max = -(by + ext), (height - by - h) + ext
    fr, fc = ref_row >> 3, ref_col >> 3
    col_min, col_max = max(col_min, fc - 1023), min(col_max, fc + 1023)
    row_min, row_max = max(row_min, fr - 1023), min(row_max, fr + 1023)
    return row_min, row_max, col_min, col_max


# ---- in-loop filter parameter planes from a mode-info grid (oracle/aomref_filtermaps.c) ----
mbmi_dtype = np.dtype([("bsize", "u1"), ("tx_size", "u1"), ("inter_tx_size", "u1", (16,)), ("skip_txfm", "u1"), ("mode", "u1"), ("segment_id", "u1"),
                       ("ref_frame0", "i1"), ("delta_lf_from_base", "i1"), ("delta_lf", "i1", (4,)), ("cdef_strength", "i1")])


class LfFrame(C.Structure):
    _fields_ = [("filter_level", C.c_int * 2), ("filter_level_u", C.c_int), ("filter_level_v", C.c_int), ("mode_ref_delta_enabled", C.c_int),
                ("ref_deltas", C.c_int8 * 8), ("mode_deltas", C.c_int8 * 2), ("delta_lf_present_flag", C.c_int), ("delta_lf_multi", C.c_int),
                ("seg_enabled", C.c_int), ("seg_feature_mask", C.c_uint8 * 8), ("seg_feature_data", (C.c_int16 * 8) * 8)]


class MiGrid:
    """blocks: structured array (mbmi_dtype); owner[mi_rows, mi_cols]: index of the block covering each 4x4 mode-info unit (-1: none)."""

    def __init__(self, blocks, owner):
        self.blocks = np.ascontiguousarray(blocks)
        self.owner = np.asarray(owner)
        self.mi_rows, self.mi_cols = self.owner.shape
        base = self.blocks.ctypes.data
        ptrs = np.where(self.owner >= 0, base + self.owner.astype(np.int64) * self.blocks.itemsize, 0).astype(np.uint64)
        # two spare rows / columns of NULL around the grid's far side (the reference's grid is allocated larger than mi_rows x mi_cols)
        self.ptrs = np.zeros((self.mi_rows + 2, self.mi_cols + 2), np.uint64)
        self.ptrs[:self.mi_rows, :self.mi_cols] = ptrs
        self.mi_stride = self.mi_cols + 2


def lf_frame_init(frame):
    lvl = np.zeros((3, 8, 2, 8, 2), np.uint8)
    lib.orc_lf_frame_init.restype = None
    lib.orc_lf_frame_init.argtypes = None
    lib.orc_lf_frame_init(C.byref(frame), C.c_void_p(lvl.ctypes.data))
    return lvl


def lf_edge_plane(grid, frame, lvl, plane, ssx, ssy):
    """set_lpf_parameters at every 4x4 unit of the plane -> int16 [rows, cols, 5]: len_v, lvl_v, len_h, lvl_h, ts."""
    w, h = (grid.mi_cols * 4) >> ssx, (grid.mi_rows * 4) >> ssy
    out = np.zeros((h // 4, w // 4, 5), np.int16)
    f = lib.orc_set_lpf_parameters
    f.restype = C.c_int
    f.argtypes = None
    fl, lv = C.c_int(), C.c_int()
    for uy in range(h // 4):
        for ux in range(w // 4):
            for d in range(2):
                ts = f(C.c_void_p(grid.ptrs.ctypes.data), C.c_int(grid.mi_stride), C.byref(frame), C.c_void_p(lvl.ctypes.data), C.c_int(d),
                       C.c_uint(4 * ux), C.c_uint(4 * uy), C.c_int(plane), C.c_int(ssx), C.c_int(ssy), C.c_uint(w), C.c_uint(h), C.byref(fl), C.byref(lv))
                out[uy, ux, 2 * d], out[uy, ux, 2 * d + 1] = fl.value, lv.value
                out[uy, ux, 4] = ts
    return out


def lf_units(grid, frame, lvl, plane, ssx, ssy):
    """The compact description aomhip_lf_build_edge_params takes: uint8 [rows, cols, 6] = aomhip_lf_unit."""
    w, h = (grid.mi_cols * 4) >> ssx, (grid.mi_rows * 4) >> ssy
    u = np.zeros((h // 4, w // 4, 6), np.uint8)
    lib.orc_lf_units.restype = None
    lib.orc_lf_units.argtypes = None
    lib.orc_lf_units(C.c_void_p(grid.ptrs.ctypes.data), C.c_int(grid.mi_stride), C.byref(frame), C.c_void_p(lvl.ctypes.data), C.c_int(plane),
                     C.c_int(ssx), C.c_int(ssy), C.c_int(w), C.c_int(h), C.c_void_p(u.ctypes.data))
    return u


def cdef_skip_map(grid):
    """1 where av1_cdef_compute_sb_list leaves an 8x8 block out (all four mode infos skip_txfm), over the whole grid."""
    skip = np.ones((grid.mi_rows // 2, grid.mi_cols // 2), np.uint8)
    f = lib.orc_cdef_compute_sb_list
    f.restype = C.c_int
    f.argtypes = None
    buf = np.zeros(512, np.uint8)
    for fr in range((grid.mi_rows + 15) // 16):
        for fc in range((grid.mi_cols + 15) // 16):
            n = f(C.c_void_p(grid.ptrs.ctypes.data), C.c_int(grid.mi_stride), C.c_int(grid.mi_rows), C.c_int(grid.mi_cols), C.c_int(fr * 16),
                  C.c_int(fc * 16), C.c_void_p(buf.ctypes.data))
            for i in range(n):
                skip[fr * 8 + buf[2 * i], fc * 8 + buf[2 * i + 1]] = 0
    return skip


# ---- temporal filter: tf_motion_search over a filter window (av1/encoder/temporal_filter.c:87-293, 849-867) ----
# A composition of the search restatements above (orc_full_pixel_search_batch = av1_full_pixel_search, orc_subpel_tree_batch
# = av1_find_best_sub_pixel_tree*) with the reference's bookkeeping between them in numpy integer arithmetic.
TF_BLOCK = 32


def tf_init_search_range(size):
    """av1_init_search_range (mcomp.c:217-226)"""
    sr, size = 0, max(16, size)
    while (size << sr) < 1023:
        sr += 1
    return min(sr, 11 - 2)


def tf_block_list(width, height, border):
    """One record per 32x32 block, raster order (get_num_blocks, encoder.h:3850); limits = mb->mv_limits from av1_set_mv_row_limits /
    av1_set_mv_col_limits (mcomp.h:216-240) with mi_rows / mi_cols = size_in_mi (encoder_utils.h:55-69)."""
    mb_rows, mb_cols = -(-height // TF_BLOCK), -(-width // TF_BLOCK)
    mi_rows, mi_cols = ((height + 7) & ~7) // 4, ((width + 7) & ~7) // 4
    b = np.zeros(mb_rows * mb_cols, np.dtype([(n, "<i2") for n in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max",
                                                                    "col_min", "col_max")]))
    r, c = np.divmod(np.arange(b.size), mb_cols)
    b["bx"], b["by"] = c * TF_BLOCK, r * TF_BLOCK
    for pos, mi_n, lo, hi in ((r * 8, mi_rows, "row_min", "row_max"), (c * 8, mi_cols, "col_min", "col_max")):
        b[lo] = np.maximum(-(pos * 4 + border - 8), -((pos + 8) * 4 + 8))
        b[hi] = np.minimum((mi_n - pos - 8) * 4 + border - 8, (mi_n - pos) * 4 + 8)
    return b


def tf_params(width, height, bit_depth, q, prune_mesh_level, mesh, subpel_tree="tree", iters_per_step=2, allow_hp=1, use_cost_list=0,
              use_downsampled_sad=0, force_integer_mv=0):
    """What tf_motion_search sets up (:118-128, :153-167, :176-185, :249-252)."""
    m = min(width, height)
    prune, thr = int(prune_mesh_level == 2), 4                                   # mcomp.c:138-140
    if prune_mesh_level == 1:
        prune, thr = int(q > 20), 2                                              # :163-167
    return dict(step_param=tf_init_search_range(max(width, height)), cost_type=3 if m >= 720 else (2 if m >= 480 else 1), prune=prune, thr=thr,
                mesh=mesh, tree=subpel_tree, iters=iters_per_step, allow_hp=allow_hp, use_cost_list=use_cost_list, skip_sad=use_downsampled_sad,
                force_integer_mv=force_integer_mv, mse_thresh=(12 if m >= 720 else 3) << (bit_depth - 8), bd=bit_depth)


def _rawpel(v):
    v = np.asarray(v, np.int32)
    return (v + 3 + (v >= 0)) >> 3                                               # GET_MV_RAWPEL (mv.h:28)


def _tf_lists(blocks, per, start, subpel):
    """search records for the block (per = 1) or its four sub-blocks (per = 4): origin, start MV, limits derived from the BLOCK's
    mb->mv_limits for the zero baseline MV (av1_set_mv_search_range, mcomp.c:196-215 / av1_set_subpel_mv_search_range, mcomp.h:344-361)."""
    out = np.repeat(blocks, per)
    if per == 4:
        k = np.tile(np.arange(4), len(blocks))
        out["bx"] += (k & 1) * 16
        out["by"] += (k >> 1) * 16
    out["ref_row"] = out["ref_col"] = 0
    out["start_row"], out["start_col"] = start[:, 0], start[:, 1]
    for lo, hi in (("row_min", "row_max"), ("col_min", "col_max")):
        if subpel:
            out[lo] = np.maximum(np.maximum(out[lo].astype(np.int32) * 8, -1023 * 8), -(1 << 14) + 1)
            out[hi] = np.minimum(np.minimum(out[hi].astype(np.int32) * 8, 1023 * 8), (1 << 14) - 1)
        else:
            out[lo] = np.maximum(out[lo], max(-1023, int(_rawpel(-(1 << 14))) + 1))
            out[hi] = np.minimum(out[hi], min(1023, int(_rawpel(1 << 14)) - 1))
    return out


def tf_motion_search_frames(frames_b, filter_frame, border, blocks, p, frame_present=None, threads=4):
    """frames_b: border-extended luma planes of the window.  -> (mvs [F, n, 4, 2] int16, mses [F, n, 4] int32, ref_mv [n, 2])."""
    F, n = len(frames_b), len(blocks)
    mvs = np.zeros((F, n, 4, 2), np.int16)
    mses = np.full((F, n, 4), 2147483647, np.int32)
    ref_mv = np.zeros((n, 2), np.int32)                                          # :855
    src = frames_b[filter_frame]
    q = search_params("NSTEP", p["step_param"], p["cost_type"], skip_sad=p["skip_sad"], run_mesh=1, prune_mesh=p["prune"], mesh_diff_thr=p["thr"],
                      mesh=p["mesh"], no_cost_list=int(not p["use_cost_list"]))
    sub = dict(tree=p["tree"], cost_type=4, iters=p["iters"], allow_hp=p["allow_hp"], forced_stop=0, bd=p["bd"], threads=threads, subpel_search_type=3)
    for f in range(F):
        if f == filter_frame:
            ref_mv = -ref_mv                                                     # :864-867
            continue
        if frame_present is not None and not frame_present[f]:
            continue
        ref = frames_b[f]
        full32, _, cl32, _ = full_pixel_search_batch(src, ref, border, 32, 32, _tf_lists(blocks, 1, _rawpel(ref_mv), False), q, bd=p["bd"], threads=threads)
        if p["force_integer_mv"]:                                                # :158-168
            block_mv = full32.astype(np.int32) * 8
            err = np.array([variance(ref, border + int(b["by"]) + int(m[0]), border + int(b["bx"]) + int(m[1]), src, border + int(b["by"]),
                                     border + int(b["bx"]), 32, 32, p["bd"] if p["bd"] > 8 else None)[0] for b, m in zip(blocks, full32)], np.uint32)
            block_mse = ((err.astype(np.uint64) + 512) // 1024).astype(np.int32)
            sub_mv = np.zeros((n, 4, 2), np.int32)
            sub_mse = np.full((n, 4), 2147483647, np.int64)
        else:
            mv32, err32, _, _ = subpel_tree_batch(src, ref, border, 32, 32, _tf_lists(blocks, 1, full32.astype(np.int32) * 8, True),
                                                  cost_lists=cl32 if p["use_cost_list"] else None, **sub)
            block_mv = mv32.astype(np.int32)
            block_mse = ((err32.astype(np.uint64) + 512) // 1024).astype(np.int32)  # DIVIDE_AND_ROUND on unsigned error (:190)
            ref_mv = block_mv.copy()                                             # :192
            full16, _, cl16, _ = full_pixel_search_batch(src, ref, border, 16, 16, _tf_lists(blocks, 4, np.repeat(_rawpel(ref_mv), 4, axis=0), False), q,
                                                         bd=p["bd"], threads=threads)
            mv16, err16, _, _ = subpel_tree_batch(src, ref, border, 16, 16, _tf_lists(blocks, 4, full16.astype(np.int32) * 8, True),
                                                  cost_lists=cl16 if p["use_cost_list"] else None, **sub)
            sub_mv = mv16.astype(np.int32).reshape(n, 4, 2)
            sub_mse = ((err16.astype(np.uint64) + 128) // 256).astype(np.int64).reshape(n, 4)
        # tf_determine_block_partition (:270-293)
        total, spread = sub_mse.sum(1), sub_mse.max(1) - sub_mse.min(1)
        bm = block_mse.astype(np.int64)
        keep = ((bm * 15 < total * 4) & (spread < 48)) | ((bm * 14 < total * 4) & (spread < 24))
        sub_mv[keep] = block_mv[keep][:, None, :]
        sub_mse[keep] = bm[keep][:, None]
        mvs[f], mses[f] = sub_mv, sub_mse
        ref_mv = np.where((block_mse > p["mse_thresh"])[:, None], 0, ref_mv)     # :249-252
    return mvs, mses, ref_mv.astype(np.int16)


# ---- first pass: first_pass_motion_search for a list of blocks (av1/encoder/firstpass.c:261-299) ----
def mv_err_cost(mrow, mcol, ref_row, ref_col, cost_type, error_per_bit=0, mvjcost=None, mvcost0=None, mvcost1=None):
    """mv_err_cost_ (mcomp.c:271-308) for an MV in 1/8 pel; entropy tables are full arrays addressed from their centres."""
    dr, dc = int(mrow) - int(ref_row), int(mcol) - int(ref_col)
    if cost_type == 0:
        bits = int(mvjcost[(dc != 0) | ((dr != 0) << 1)]) + int(mvcost0[len(mvcost0) // 2 + dr]) + int(mvcost1[len(mvcost1) // 2 + dc])
        return (bits * int(error_per_bit) + (1 << 13)) >> 14
    lam = {1: 2, 2: 0, 3: 1}.get(cost_type, 0)
    return (lam * (abs(dr) + abs(dc))) >> 3


def first_pass_motion_search_batch(src_b, ref_b, border, w, h, blocks, q, mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4):
    """av1_full_pixel_search (q: search_params, normally NSTEP_FPF + entropy costs) then av1_get_mvpred_sse + NEW_MV_MODE_PENALTY (32)
    when the search returned < INT_MAX (mcomp.c:3637-3649).  -> (mv [n, 2] full-pel, err [n] int32)"""
    mv, cost, _, _ = full_pixel_search_batch(src_b, ref_b, border, w, h, blocks, q, mvjcost, mvcost0, mvcost1, bd=bd, threads=threads)
    err = np.full(len(blocks), 2147483647, np.int32)
    for i, b in enumerate(blocks):
        if cost[i] == 2147483647:
            continue
        y, x = border + int(b["by"]), border + int(b["bx"])
        sse = variance(src_b, y, x, ref_b, y + int(mv[i, 0]), x + int(mv[i, 1]), w, h, bd if bd > 8 else None)[1]
        err[i] = sse + mv_err_cost(int(mv[i, 0]) * 8, int(mv[i, 1]) * 8, b["ref_row"], b["ref_col"], q.cost_type, q.error_per_bit, mvjcost, mvcost0,
                                   mvcost1) + 32
    return mv, err


def first_pass_inter_frame(src_b, last_b, golden_b, last_source_b, border, bs, blocks, rows, cols, q, intra_error, skip_motion_search_threshold=0,
                           skip_zeromv_motion_search=0, mvjcost=None, mvcost0=None, mvcost1=None, bd=8):
    """firstpass_inter_prediction (av1/encoder/firstpass.c:690-815) for every block of a frame in the raster order of av1_first_pass_row
    (:1148-1193): best_ref_mv starts at kZeroMv in every row (:1165) and becomes each block's *best_mv (:1190).  A scalar walk, one block and
    one search leg at a time, as the reference does it.  blocks: raster list with bx, by and the raw x->mv_limits; golden_b None when the
    frame has no golden reference (:742).  -> (best_mv [n, 2] 1/8 pel, full_mv [n, 2], motion_error, gf_motion_error, raw_motion_error)"""
    hb = bd if bd > 8 else None
    n = rows * cols
    best_mv, full_mv = np.zeros((n, 2), np.int16), np.zeros((n, 2), np.int16)
    motion_error, gf_error, raw_error = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)

    def err00(ref_b, b):                                   # get_prediction_error_bitdepth (:113-160): the mse function's sse
        y, x = border + int(b["by"]), border + int(b["bx"])
        return int(variance(src_b, y, x, ref_b, y, x, bs, bs, hb)[1])

    def leg(ref_b, b, ref_row, ref_col, mv, best_err):     # first_pass_motion_search (:261-299)
        one = np.array([b], dtype=blocks.dtype)
        one["ref_row"], one["ref_col"] = ref_row, ref_col
        one["start_row"], one["start_col"] = _rawpel(np.int32(ref_row)), _rawpel(np.int32(ref_col))
        raw = (b["row_min"], b["row_max"], b["col_min"], b["col_max"])
        one["row_min"], one["row_max"], one["col_min"], one["col_max"] = set_mv_search_range(raw, ref_row, ref_col)
        m, e = first_pass_motion_search_batch(src_b, ref_b, border, bs, bs, one, q, mvjcost, mvcost0, mvcost1, bd=bd, threads=1)
        if int(e[0]) < best_err:
            return (int(m[0, 0]), int(m[0, 1])), int(e[0])
        return mv, best_err

    for r in range(rows):
        ref_row = ref_col = 0
        for c in range(cols):
            i = r * cols + c
            b = blocks[i]
            mv, err = (0, 0), err00(last_b, b)
            raw = err00(last_source_b, b)
            gf = err
            if raw > skip_motion_search_threshold:
                mv, err = leg(last_b, b, ref_row, ref_col, mv, err)
                if not skip_zeromv_motion_search and (ref_row or ref_col):
                    tmp_mv, tmp_err = leg(last_b, b, 0, 0, (0, 0), 2147483647)
                    if tmp_err < err:
                        mv, err = tmp_mv, tmp_err
                gf = err
                if golden_b is not None:
                    _, gf = leg(golden_b, b, 0, 0, (0, 0), err00(golden_b, b))
            ref_row = ref_col = 0
            if err <= int(intra_error[i]):
                ref_row, ref_col = mv[0] * 8, mv[1] * 8
            best_mv[i] = (ref_row, ref_col)
            full_mv[i] = mv
            motion_error[i], gf_error[i], raw_error[i] = err, gf, raw
    return best_mv, full_mv, motion_error, gf_error, raw_error


# ---- full-pel + sub-pel search of a block list: tpl_model.c motion_estimation (av1/encoder/tpl_model.c:248-301) ----
def set_mv_search_range(limits, ref_row, ref_col):
    """av1_set_mv_search_range (mcomp.c:196-215): limits = (row_min, row_max, col_min, col_max) of x->mv_limits -> FullMvLimits around ref_mv."""
    out = []
    for (lo, hi), r in (((limits[0], limits[1]), int(ref_row)), ((limits[2], limits[3]), int(ref_col))):
        mn = int(_rawpel(r)) - 1023 + (1 if r & 7 else 0)
        mx = int(_rawpel(r)) + 1023
        mn, mx = max(mn, int(_rawpel(-(1 << 14))) + 1), min(mx, int(_rawpel(1 << 14)) - 1)
        out += [max(int(lo), mn), min(int(hi), mx)]
    return out


def set_subpel_mv_search_range(limits, ref_row, ref_col):
    """av1_set_subpel_mv_search_range (mcomp.h:344-361) -> (row_min, row_max, col_min, col_max) in 1/8 pel."""
    out = []
    for (lo, hi), r in (((limits[0], limits[1]), int(ref_row)), ((limits[2], limits[3]), int(ref_col))):
        out += [max(-(1 << 14) + 1, max(int(lo) * 8, r - 8184)), min((1 << 14) - 1, min(int(hi) * 8, r + 8184))]
    return out


def motion_estimation_batch(src_b, ref_b, border, w, h, blocks, q, sub, use_cost_list=0, mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4):
    """blocks: ref_* = center_mv (1/8 pel), limits = raw x->mv_limits.  sub: kwargs of subpel_tree_batch (tree, cost_type, iters, ...).
    -> (mv [n, 2] 1/8 pel, err, dist, sse, full_mv [n, 2])"""
    fl, sl = np.array(blocks, copy=True), np.array(blocks, copy=True)
    for i, b in enumerate(blocks):
        raw = (b["row_min"], b["row_max"], b["col_min"], b["col_max"])
        fl["row_min"][i], fl["row_max"][i], fl["col_min"][i], fl["col_max"][i] = set_mv_search_range(raw, b["ref_row"], b["ref_col"])
        sl["row_min"][i], sl["row_max"][i], sl["col_min"][i], sl["col_max"][i] = set_subpel_mv_search_range(raw, b["ref_row"], b["ref_col"])
    fl["start_row"], fl["start_col"] = _rawpel(blocks["ref_row"]), _rawpel(blocks["ref_col"])
    full_mv, _, cl, _ = full_pixel_search_batch(src_b, ref_b, border, w, h, fl, q, mvjcost, mvcost0, mvcost1, bd=bd, threads=threads)
    sl["start_row"], sl["start_col"] = full_mv[:, 0].astype(np.int32) * 8, full_mv[:, 1].astype(np.int32) * 8
    mv, err, dist, sse = subpel_tree_batch(src_b, ref_b, border, w, h, sl, mvjcost=mvjcost, mvcost0=mvcost0, mvcost1=mvcost1,
                                           cost_lists=cl if use_cost_list else None, bd=bd, threads=threads, **sub)
    return mv, err, dist, sse, full_mv


def tpl_prune(sads, prune_starting_mv):
    """The `if (cpi->sf.tpl_sf.prune_starting_mv)` block of mode_estimation (av1/encoder/tpl_model.c:706-731) on the candidates' SADs: the order qsort +
    compare_sad leave them in (ties keep their order: glibc's qsort is a merge sort), cut to 4 - prune_starting_mv and once more when the last SAD
    exceeds the one before it by more than 20 %.  -> the surviving candidate indices, best first.  PINNED by tests/golden/ref_eval_tpl.npz (the
    block's own statements interpreted)."""
    order = sorted(range(len(sads)), key=lambda k: int(sads[k]))
    cnt = min(4 - int(prune_starting_mv), len(order))
    if cnt > 1 and (int(sads[order[cnt - 1]]) - int(sads[order[cnt - 2]])) * 5 > int(sads[order[cnt - 2]]):
        cnt -= 1
    return order[:cnt]


def tpl_best_of(errs):
    """The loop over a reference's remaining candidates (:733-743): index of the first candidate whose motion_estimation error is below every earlier one
    and below UINT32_MAX, or None (best_rfidx_mv then stays { 0 })."""
    best, which = 0xFFFFFFFF, None
    for k, e in enumerate(errs):
        if int(e) < best:
            best, which = int(e), k
    return which


def tpl_best_ref(costs, have):
    """The per-reference tail (:755-765): pred_error[r] = max(1, cost) for the references that exist, best_rf = the first smallest cost (-1: none).
    -> (best_rf, best_cost, pred_error list with None for missing references)"""
    best_rf, best_cost, pe = -1, 2147483647, []
    for r, c in enumerate(costs):
        if not have[r]:
            pe.append(None)
            continue
        pe.append(max(1, int(c)))
        if int(c) < best_cost:
            best_rf, best_cost = r, int(c)
    return best_rf, best_cost, pe


def tpl_inter_estimation_batch(src_b, ref_bs, border, width, height, bw, blocks, centers, counts, q, sub, use_cost_list=0, prune_starting_mv=0,
                               mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4):
    """The inter leg of mode_estimation (av1/encoder/tpl_model.c:620-770) for independent blocks, as a composition of the pinned pieces:
    per reference the candidates' SADs, the ranking and the two cuts of prune_starting_mv (:706-731; qsort with compare_sad, stable on ties:
    glibc's qsort is a merge sort), motion_estimation (:248-301) from every remaining candidate with the first smallest error winning
    (:733-743), the EIGHTTAP_REGULAR predictor and tpl_get_satd_cost (:199-212: residual, DCT_DCT, aom_satd), pred_error = max(1, cost); then
    the reference with the smallest cost (:759-765).  The selection glue is tpl_prune / tpl_best_of / tpl_best_ref above, each pinned by the
    reference's own statements interpreted (tests/golden/ref_eval_tpl.npz); the pieces it strings together are pinned by their own fixtures.
    blocks: bx, by, raw mv limits; centers [n, n_refs, 4, 2] 1/8 pel; counts [n, n_refs].
    -> (best_mv [n, n_refs, 2], pred_error [n, n_refs], best_rf [n], best_cost [n])"""
    n, n_refs = len(blocks), len(ref_bs)
    tx = {8: 1, 16: 2, 32: 3}[bw]
    px = bw * bw
    best_mv = np.zeros((n, n_refs, 2), np.int16)
    pred_error = np.zeros((n, n_refs), np.int32)
    raw = np.full((n, n_refs), 2147483647, np.int64)
    qd = build_quantizer_y(bd, 100)
    for r, ref_b in enumerate(ref_bs):
        entries, owner = [], []
        for i, b in enumerate(blocks):
            cnt = int(counts[i, r])
            if cnt == 0:
                best_mv[i, r] = -32768
                pred_error[i, r] = 2147483647
                continue
            order = list(range(cnt))
            if prune_starting_mv:
                cands = np.zeros(cnt, np.dtype([("sx", "<i2"), ("sy", "<i2"), ("rx", "<i2"), ("ry", "<i2")]))
                for k in range(cnt):
                    row = min(max(int(_rawpel(int(centers[i, r, k, 0]))), int(b["row_min"])), int(b["row_max"]))     # get_fullmv_from_mv + clamp_fullmv
                    col = min(max(int(_rawpel(int(centers[i, r, k, 1]))), int(b["col_min"])), int(b["col_max"]))
                    cands[k] = (b["bx"], b["by"], b["bx"] + col, b["by"] + row)
                order = tpl_prune([int(v) for v in sad_batch(src_b, ref_b, border, bw, bw, cands, bd=bd)], prune_starting_mv)
                cnt = len(order)
            for k in order[:cnt]:
                entries.append(k)
                owner.append(i)
        if not entries:
            continue
        ent = blocks[owner].copy()   # (one entry per (block, surviving candidate): ref_mv = the candidate, raw limits)
        ent["ref_row"], ent["ref_col"] = centers[owner, r, entries, 0], centers[owner, r, entries, 1]
        mv, err, _, _, _ = motion_estimation_batch(src_b, ref_b, border, bw, bw, ent, q, sub, use_cost_list, mvjcost, mvcost0, mvcost1, bd=bd, threads=threads)
        best = {}
        owner_a = np.array(owner)
        for i in sorted(set(owner)):
            es = np.nonzero(owner_a == i)[0]
            k = tpl_best_of(err[es])
            best[i] = (int(err[es[k]]), mv[es[k]]) if k is not None else (0xFFFFFFFF, np.zeros(2, np.int16))
        idx = sorted(best)
        for i in idx:
            best_mv[i, r] = best[i][1]
        sub_blocks = blocks[idx]
        pred = build_inter_pred(ref_b, border, width, height, bw, bw, sub_blocks, np.array([best[i][1] for i in idx], np.int16), 0, 0, bd)
        src_vis = src_b[border:border + height, border:border + width]
        residual = np.zeros((len(idx) * bw, bw), np.int16)
        for j, i in enumerate(idx):
            bx, by = int(blocks["bx"][i]), int(blocks["by"][i])
            residual[j * bw:(j + 1) * bw] = src_vis[by:by + bw, bx:bx + bw].astype(np.int32) - pred[by:by + bw, bx:bx + bw].astype(np.int32)
        coeff, _, _, _ = xform_quant_batch(residual, tx, None, len(idx), 1, 0, qd, bd > 8, len(idx) * px, True, threads)
        satd = np.abs(coeff.astype(np.int64)).reshape(len(idx), px).sum(1)
        for j, i in enumerate(idx):
            raw[i, r] = int(satd[j])
            pred_error[i, r] = max(1, int(satd[j]))
    best_rf, best_cost = np.full(n, -1, np.int8), np.full(n, 2147483647, np.int32)
    for i in range(n):
        best_rf[i], best_cost[i], _ = tpl_best_ref(raw[i], counts[i])
    return best_mv, pred_error, best_rf, best_cost


def mv_bit_cost(mrow, mcol, ref_row, ref_col, mvjcost, mvcost0, mvcost1, weight=108):
    """av1_mv_bit_cost (mcomp.c:261-266): ROUND_POWER_OF_TWO(mv_cost(diff) * weight, 7); weight MV_COST_WEIGHT = 108 (rd.h:45)."""
    dr, dc = int(mrow) - int(ref_row), int(mcol) - int(ref_col)
    bits = int(mvjcost[(dc != 0) | ((dr != 0) << 1)]) + int(mvcost0[len(mvcost0) // 2 + dr]) + int(mvcost1[len(mvcost1) // 2 + dc])
    return (bits * int(weight) + 64) >> 7


INVALID_MV_ROW_COL = -32768


def estimate_txfm_yrd(residual, bw, bh, bd, q, above, left, costs, tx_type_rate, tx_size_rate, no_skip_txfm_rate, skip_txfm_rate, rdmult, lossless=0):
    """oracle/aomref_yrd.c orc_estimate_txfm_yrd (av1_estimate_txfm_yrd, tx_search.c:3204-3255) of one block's residual (int16 [bh, bw]); q = build_quantizer_y's
    tables, above / left = the block's entropy contexts (32 bytes each), costs = LV_MAP_COEFF_COST + eob_cost.  -> dict(rd, rate, skip_txfm, dist, sse)"""
    f = lib.orc_estimate_txfm_yrd
    f.restype = C.c_int64
    res = np.ascontiguousarray(residual, np.int16)
    tabs = np.ascontiguousarray(np.stack([np.asarray(q[k], np.int16)[:2] for k in ("zbin", "round", "quant", "quant_shift", "dequant")]))
    ab, lf, cs = np.ascontiguousarray(above, np.uint8), np.ascontiguousarray(left, np.uint8), np.ascontiguousarray(costs, np.int32)
    out = np.zeros(4, np.int64)
    rd = f(C.c_void_p(res.ctypes.data), res.shape[1], bw, bh, bd, int(bd > 8), C.c_void_p(tabs.ctypes.data), C.c_void_p(ab.ctypes.data), C.c_void_p(lf.ctypes.data),
           C.c_void_p(cs.ctypes.data), int(tx_type_rate), int(tx_size_rate), int(no_skip_txfm_rate), int(skip_txfm_rate), int(rdmult), int(lossless),
           C.c_void_p(out.ctypes.data))
    return dict(rd=int(rd), rate=int(out[0]), skip_txfm=int(out[1]), dist=int(out[2]), sse=int(out[3]))


def single_motion_search_batch(src_b, ref_b, border, w, h, blocks, q, sub, start2=None, use_cost_list=0, try_second_mv=0, force_integer_mv=0,
                               mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4, rd=None):
    """The SIMPLE_TRANSLATION core of av1_single_motion_search (motion_search_facade.c:120-495) as a composition of the pinned pieces, for a
    list of independent (block, reference) pairs.  blocks: ref_* = ref_mv (1/8 pel), start_* = cand[0] (FULLPEL, -32768 = skipped by
    skip_fullpel_search_using_startmv), limits = raw x->mv_limits; start2 [n, 2] = cand[1] or -32768 (:271-290: the caller knows before
    searching how many candidates the weight rule admits).  try_second_mv: use_accurate_subpel_search with disable_second_mv == 1 (:367-430, the
    second sub-pel search from second_best_mv, kept when its var is smaller).  rd (with try_second_mv): the disable_second_mv == 0 form (:378-418) --
    dict(filter_x, filter_y, q = build_quantizer_y tables, costs, tx_type_rate, rdmult, lossless, yrd_blocks = records with tx_size_rate,
    no_skip_txfm_rate, skip_txfm_rate, above_ctx, left_ctx per block): each candidate's luma predictor (build_inter_pred) and residual go through
    estimate_txfm_yrd, and the second candidate is kept when RDCOST(rdmult, mv rate + rate, dist) is smaller (orc_second_mv_rd_choice); the two RD_STATS
    come back as stats_first / stats_second (lists, None where not computed) and the two candidates as cand_mvs [n, 2, 2] (-32768 where none).
    rd["yrd_fn"] (tests of the sequencing only): called as yrd_fn(i, (row, col)) -> dict(rate, dist) INSTEAD of predictor + estimate_txfm_yrd.  -> dict(best_mv [n,2] 1/8 pel or -32768, bestsme, rate_mv, pred_sse,
    full_mv, second_best)"""
    n = len(blocks)
    fl, sl = np.array(blocks, copy=True), np.array(blocks, copy=True)
    for i, b in enumerate(blocks):
        raw = (b["row_min"], b["row_max"], b["col_min"], b["col_max"])
        fl["row_min"][i], fl["row_max"][i], fl["col_min"][i], fl["col_max"][i] = set_mv_search_range(raw, b["ref_row"], b["ref_col"])
        sl["row_min"][i], sl["row_max"][i], sl["col_min"][i], sl["col_max"][i] = set_subpel_mv_search_range(raw, b["ref_row"], b["ref_col"])
    bestsme = np.full(n, 2147483647, np.int64)
    full_mv = np.full((n, 2), INVALID_MV_ROW_COL, np.int16)
    second = np.full((n, 2), INVALID_MV_ROW_COL, np.int16)
    cost_list = np.zeros((n, 5), np.int32)
    starts = [np.stack([blocks["start_row"], blocks["start_col"]], 1).astype(np.int16)]
    if start2 is not None:
        starts.append(np.asarray(start2, np.int16))
    for st in starts:                                      # "Perform a search with the top 2 candidates" (:271-290)
        idx = np.flatnonzero(st[:, 0] != INVALID_MV_ROW_COL)
        if not len(idx):
            continue
        l = fl[idx].copy()
        l["start_row"], l["start_col"] = st[idx, 0], st[idx, 1]
        mv, cost, cl, sec = full_pixel_search_batch(src_b, ref_b, border, w, h, l, q, mvjcost, mvcost0, mvcost1, bd=bd, threads=threads)
        for k, i in enumerate(idx):
            cost_list[i] = cl[k]                            # ONE cost_list array for all candidates (:247, :279): the last search's stays
            if int(cost[k]) < bestsme[i]:
                bestsme[i], full_mv[i], second[i] = int(cost[k]), mv[k], sec[k]
    out = dict(best_mv=np.full((n, 2), INVALID_MV_ROW_COL, np.int16), bestsme=bestsme.astype(np.int32), rate_mv=np.zeros(n, np.int32),
               pred_sse=np.zeros(n, np.uint32), full_mv=full_mv, second_best=second)
    if rd is not None:
        out["stats_first"], out["stats_second"], out["cand_mvs"] = [None] * n, [None] * n, np.full((n, 2, 2), INVALID_MV_ROW_COL, np.int16)
        height, width = src_b.shape[0] - 2 * border, src_b.shape[1] - 2 * border

        def yrd_at(i, mv):   # av1_enc_build_inter_predictor + av1_subtract_plane + av1_estimate_txfm_yrd (:381-388, :404-411)
            if rd.get("yrd_fn") is not None:
                return rd["yrd_fn"](i, (int(mv[0]), int(mv[1])))
            b, yb = blocks[i], rd["yrd_blocks"][i]
            pred = build_inter_pred(ref_b, border, width, height, w, h, blocks[i:i + 1], [mv], rd.get("filter_x", 0), rd.get("filter_y", 0), bd=bd)
            x0, y0 = int(b["bx"]), int(b["by"])
            res = src_b[border + y0:border + y0 + h, border + x0:border + x0 + w].astype(np.int32) - pred[y0:y0 + h, x0:x0 + w].astype(np.int32)
            return estimate_txfm_yrd(res, w, h, bd, rd["q"], yb["above_ctx"], yb["left_ctx"], rd["costs"], rd["tx_type_rate"], yb["tx_size_rate"],
                                     yb["no_skip_txfm_rate"], yb["skip_txfm_rate"], rd["rdmult"], rd.get("lossless", 0))
    live = np.flatnonzero(full_mv[:, 0] != INVALID_MV_ROW_COL)     # if (best_mv->as_int == INVALID_MV) return (:298)
    if not len(live):
        return out
    if force_integer_mv:                                    # convert_fullmv_to_mv, no fractional search (:343-349)
        out["best_mv"][live] = full_mv[live] * 8
    else:
        l = sl[live].copy()
        l["start_row"], l["start_col"] = full_mv[live, 0].astype(np.int32) * 8, full_mv[live, 1].astype(np.int32) * 8
        lists = np.full((len(live), 3, 2), INVALID_MV_ROW_COL, np.int16) if try_second_mv else None     # av1_set_fractional_mv
        cls = cost_list[live] if use_cost_list else None
        mv, err, dist, sse = subpel_tree_batch(src_b, ref_b, border, w, h, l, mvjcost=mvjcost, mvcost0=mvcost0, mvcost1=mvcost1, cost_lists=cls, bd=bd,
                                               threads=threads, mv_lists=lists, **sub)
        out["best_mv"][live], out["pred_sse"][live] = mv, sse
        if try_second_mv:
            for k, i in enumerate(live):
                s2 = second[i]
                if s2[0] == INVALID_MV_ROW_COL or (s2 == full_mv[i]).all():                  # try_second (:370-372)
                    continue
                st = (int(s2[0]) * 8, int(s2[1]) * 8)
                if not (l["row_min"][k] <= st[0] <= l["row_max"][k] and l["col_min"][k] <= st[1] <= l["col_max"][k]):   # av1_is_subpelmv_in_range
                    continue
                one = l[k:k + 1].copy()
                one["start_row"], one["start_col"] = st
                mv2, err2, _, sse2 = subpel_tree_batch(src_b, ref_b, border, w, h, one, mvjcost=mvjcost, mvcost0=mvcost0, mvcost1=mvcost1,
                                                       cost_lists=cls[k:k + 1] if cls is not None else None, bd=bd, threads=1,
                                                       mv_lists=lists[k:k + 1], **sub)
                if rd is not None:
                    b = blocks[i]
                    s0, s1 = yrd_at(i, mv[k]), yrd_at(i, mv2[0])
                    out["stats_first"][i], out["stats_second"][i] = s0, s1
                    out["cand_mvs"][i, 0], out["cand_mvs"][i, 1] = mv[k], mv2[0]
                    r0 = mv_bit_cost(mv[k][0], mv[k][1], b["ref_row"], b["ref_col"], mvjcost, mvcost0, mvcost1)
                    r1 = mv_bit_cost(mv2[0][0], mv2[0][1], b["ref_row"], b["ref_col"], mvjcost, mvcost0, mvcost1)
                    lib.orc_second_mv_rd_choice.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int64]
                    if lib.orc_second_mv_rd_choice(int(rd["rdmult"]), r0, s0["rate"], s0["dist"], r1, s1["rate"], s1["dist"]):   # tmp_rd < rd (:414-418)
                        out["best_mv"][i], out["pred_sse"][i] = mv2[0], sse2[0]
                elif int(np.int32(err2[0])) < int(np.int32(err[k])):                            # this_var < best_mv_var (int compare, :421)
                    out["best_mv"][i], out["pred_sse"][i] = mv2[0], sse2[0]
    for i in live:
        b = blocks[i]
        out["rate_mv"][i] = mv_bit_cost(out["best_mv"][i, 0], out["best_mv"][i, 1], b["ref_row"], b["ref_col"], mvjcost, mvcost0, mvcost1)
    return out


def tpl_is_alike_mv(cand, centers, skip_alike_starting_mv):
    """is_alike_mv (av1/encoder/tpl_model.c:317-331): both components closer than the threshold (1/8 pel: 1, 8 << 3, 16 << 3) to one of the centres"""
    thr = (1, 8 << 3, 16 << 3)[int(skip_alike_starting_mv)]
    return any(abs(int(c[1]) - int(cand[1])) < thr and abs(int(c[0]) - int(cand[0])) < thr for c in centers)


def tpl_gather_candidates(above, left, above_right, skip_alike_starting_mv):
    """The starting MVs of one reference for one TPL block (mode_estimation, tpl_model.c:643-683): the zero MV, then the MVs the TPL stats hold for the
    block above, the block to the left and the block above-right (None where xd->up_available / left_available / `mi_col + mi_width < tile.mi_col_end`
    fails), each taken unless is_alike_mv finds it among those already taken.  -> list of (row, col), 1 to 4 entries"""
    centers = [(0, 0)]
    for mv in (above, left, above_right):
        if mv is not None and not tpl_is_alike_mv(mv, centers, skip_alike_starting_mv):
            centers.append((int(mv[0]), int(mv[1])))
    return centers


def tpl_mode_decision(intra_costs, best_rf, best_inter_cost, best_mv):
    """mode_estimation's decisions around the inter leg with allow_compound_pred == 0 (tpl_model.c:552-566, :766-770, :912-915, :990-994): the intra mode
    with the first smallest cost (INT32_MAX start, DC_PRED = 0 first), NEWMV (16) when a reference exists and its cost is SMALLER than the best intra
    cost; intra_cost = max(best, 1), inter_cost = min(intra_cost, best inter cost); ref_frame_index = (best_rf, -1) for NEWMV, (-1, -1) otherwise.
    -> dict(best_mode, intra_cost, inter_cost, ref_frame_index)"""
    best_intra, mode = 2147483647, 0
    for m, c in enumerate(intra_costs):
        if int(c) < best_intra:
            best_intra, mode = int(c), m
    newmv = int(best_rf) != -1 and int(best_inter_cost) < best_intra
    intra_cost = max(best_intra, 1)
    return dict(best_mode=16 if newmv else mode, intra_cost=intra_cost, inter_cost=min(intra_cost, int(best_inter_cost)),
                ref_frame_index=[int(best_rf), -1] if newmv else [-1, -1])


def tpl_mode_estimation_rows(src_b, ref_bs, border, width, height, bw, positions, limits, intra_costs, q, sub, use_cost_list=0, prune_starting_mv=0,
                             skip_alike_starting_mv=0, mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=1):
    """mode_estimation for the blocks of whole rows of a frame in raster order, as av1_mc_flow_dispenser_row calls it (tpl_model.c:1257-1330) with
    tpl_model_store between blocks: per block and reference tpl_gather_candidates on the stored stats of the neighbours, tpl_inter_estimation_batch for
    the block, tpl_mode_decision.  ref_bs: list of border-extended reference planes (every reference exists); positions: (bx, by) in raster order,
    complete rows from row 0; limits: raw x->mv_limits per block; intra_costs: per block the costs of the intra modes searched.
    -> list of dict(mv [n_refs, 2], pred_error [n_refs], best_rf, best_inter_cost, candidates [n_refs] lists, **tpl_mode_decision)"""
    n_refs = len(ref_bs)
    stored, out = {}, []
    cols = width // bw
    for (bx, by), lim, ic in zip(positions, limits, intra_costs):
        r, c = by // bw, bx // bw
        blk = np.zeros(1, np.dtype([(k, "<i2") for k in ("bx", "by", "start_row", "start_col", "ref_row", "ref_col", "row_min", "row_max", "col_min", "col_max")]))
        blk["bx"], blk["by"] = bx, by
        blk["row_min"], blk["row_max"], blk["col_min"], blk["col_max"] = lim
        centers = np.zeros((1, n_refs, 4, 2), np.int16)
        counts = np.zeros((1, n_refs), np.uint8)
        cands = []
        for k in range(n_refs):
            nb = lambda rr, cc: stored[(rr, cc)]["mv"][k] if (rr, cc) in stored else None
            cl = tpl_gather_candidates(nb(r - 1, c) if r > 0 else None, nb(r, c - 1) if c > 0 else None,
                                       nb(r - 1, c + 1) if r > 0 and c + 1 < cols else None, skip_alike_starting_mv)
            cands.append(cl)
            centers[0, k, :len(cl)] = cl
            counts[0, k] = len(cl)
        mv, pe, rf, bc = tpl_inter_estimation_batch(src_b, ref_bs, border, width, height, bw, blk, centers, counts, q, sub, use_cost_list, prune_starting_mv,
                                                    mvjcost, mvcost0, mvcost1, bd=bd, threads=threads)
        rec = dict(mv=mv[0].copy(), pred_error=pe[0].copy(), best_rf=int(rf[0]), best_inter_cost=int(bc[0]), candidates=cands)
        rec.update(tpl_mode_decision(ic, rf[0], bc[0], mv[0, int(rf[0])] if rf[0] >= 0 else None))
        stored[(r, c)] = rec
        out.append(rec)
    return out


def warp_block_pred(ref_vis, bd, mat, shear, bx, by, bw, bh):
    """The luma predictor of a WARPED_CAUSAL block: av1_warp_plane as av1_make_inter_predictor calls it (reconinter.c: the block's rectangle, the frame's
    visible size, get_conv_params(0, 0, bd)) through oracle/aomref_warp.c.  ref_vis: the visible reference plane.  -> [bh, bw]"""
    lib.orc_warp_affine.restype = None
    lib.orc_warp_affine.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p] + [C.c_int] * 13
    ref = np.ascontiguousarray(ref_vis)
    h, w = ref.shape
    m = np.ascontiguousarray(mat, np.int32)
    out = np.zeros((bh, bw), ref.dtype)
    lib.orc_warp_affine(m.ctypes.data, ref.ctypes.data, int(ref.dtype != np.uint8), w, h, w, out.ctypes.data, int(bx), int(by), bw, bh, bw, 0, 0, bd,
                        5 if bd == 12 else 3, int(shear[0]), int(shear[1]), int(shear[2]), int(shear[3]))
    return out


def find_projection(n, pts, pts_inref, bw, bh, mv, mi_row, mi_col, mat0=None):
    """orc_find_projection: -> (ok, mat [6] int32, shear [4] int16); mat0 = the model before the call (left alone by a singular system)"""
    p, q = np.ascontiguousarray(pts, np.int32), np.ascontiguousarray(pts_inref, np.int32)
    mat = np.array(mat0 if mat0 is not None else [0, 0, 1 << 16, 0, 0, 1 << 16], np.int32)
    sh = np.zeros(4, np.int16)
    bad = lib.orc_find_projection(int(n), p.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p), bw, bh, int(mv[0]), int(mv[1]), mat.ctypes.data_as(C.c_void_p),
                                  sh.ctypes.data_as(C.c_void_p), int(mi_row), int(mi_col))
    return (not bad), mat, sh


def refine_warped_mv(src_vis, ref_vis, bd, bw, bh, b, allow_hp, cost_type, error_per_bit=0, mvjcost=None, mvcost0=None, mvcost1=None, pred_fn=None):
    """av1_refine_warped_mv (av1/encoder/mcomp.c:3224-3293) for ONE block, statement by statement, over the pinned pieces: orc_select_samples /
    orc_find_projection, the warped predictor (warp_block_pred; pred_fn(mat, shear) replaces it in tests of the sequencing), vf(pred, src) and
    mv_err_cost_.  b: a record with bx, by, mv_row / mv_col (the start), ref_row / ref_col, the four SubpelMvLimits, total_samples, num_proj_ref, pts,
    pts_inref, model (mat + alpha .. delta).  -> dict(mv, mat, shear, num_proj_ref, bestmse, measured = the candidate MVs whose cost was computed)"""
    bx, by = int(b["bx"]), int(b["by"])
    src_blk = np.ascontiguousarray(src_vis[by:by + bh, bx:bx + bw])

    def motion_cost(mv, mat, shear):   # compute_motion_cost (:3197-3221)
        pred = pred_fn(mat, shear) if pred_fn else warp_block_pred(ref_vis, bd, mat, shear, bx, by, bw, bh)
        v = variance(np.ascontiguousarray(pred), 0, 0, src_blk, 0, 0, bw, bh, bd)[0]      # vf(dst, dst_stride, src, src_stride, &sse)
        return (int(v) + mv_err_cost(mv[0], mv[1], b["ref_row"], b["ref_col"], cost_type, error_per_bit, mvjcost, mvcost0, mvcost1)) & 0xFFFFFFFF
    nb = [(0, -1), (1, 0), (0, 1), (-1, 0), (0, -2), (2, 0), (0, 2), (-2, 0)]
    best_mv = [int(b["mv_row"]), int(b["mv_col"])]
    model = b["model"]
    best_mat, best_sh = np.array(model["mat"], np.int32).reshape(6), np.array([model["alpha"], model["beta"], model["gamma"], model["delta"]], np.int16).reshape(4)
    cur_mat, cur_np = best_mat.copy(), int(b["num_proj_ref"])
    best_np = cur_np
    total = int(b["total_samples"])
    start = 0 if allow_hp else 4
    measured = [tuple(best_mv)]
    bestmse = motion_cost(best_mv, best_mat, best_sh)
    for _ in range(2):
        best_idx = -1
        for idx in range(start, start + 4):
            mv = (best_mv[0] + nb[idx][0], best_mv[1] + nb[idx][1])
            if not (b["col_min"] <= mv[1] <= b["col_max"] and b["row_min"] <= mv[0] <= b["row_max"]):     # av1_is_subpelmv_in_range
                continue
            pts, pin = np.array(b["pts"], np.int32).reshape(-1).copy(), np.array(b["pts_inref"], np.int32).reshape(-1).copy()
            if total > 1:
                cur_np = lib.orc_select_samples(mv[0], mv[1], pts.ctypes.data_as(C.c_void_p), pin.ctypes.data_as(C.c_void_p), total, bw, bh)
            ok, cur_mat, sh = find_projection(cur_np, pts, pin, bw, bh, mv, by >> 2, bx >> 2, cur_mat)
            if ok:
                measured.append(mv)
                mse = motion_cost(mv, cur_mat, sh)
                if mse < bestmse:
                    best_idx, best_mat, best_sh, best_np, bestmse = idx, cur_mat.copy(), sh.copy(), cur_np, mse
        if best_idx == -1:
            break
        best_mv = [best_mv[0] + nb[best_idx][0], best_mv[1] + nb[best_idx][1]]
    return dict(mv=best_mv, mat=best_mat.tolist(), shear=best_sh.tolist(), num_proj_ref=int(best_np), bestmse=int(bestmse), measured=measured)


def simple_motion_search_batch(src_b, ref_b, border, width, height, w, h, blocks, q, sub=None, use_cost_list=0, mvjcost=None, mvcost0=None, mvcost1=None, bd=8,
                               threads=4):
    """av1_simple_motion_search + av1_simple_motion_sse_var (motion_search_facade.c:925-1060) as a composition of the pinned pieces:
    blocks: start_* = FULLPEL start_mv, limits = raw x->mv_limits, ref_mv = 0.  sub: kwargs of subpel_tree_batch or None (no sub-pel
    stage).  -> (mv [n, 2] 1/8 pel, sse, var, pred plane)"""
    fl, sl = np.array(blocks, copy=True), np.array(blocks, copy=True)
    fl["ref_row"] = fl["ref_col"] = sl["ref_row"] = sl["ref_col"] = 0
    for i, b in enumerate(blocks):
        raw = (b["row_min"], b["row_max"], b["col_min"], b["col_max"])
        fl["row_min"][i], fl["row_max"][i], fl["col_min"][i], fl["col_max"][i] = set_mv_search_range(raw, 0, 0)
        sl["row_min"][i], sl["row_max"][i], sl["col_min"][i], sl["col_max"][i] = set_subpel_mv_search_range(raw, 0, 0)
    full_mv, cost, cl, _ = full_pixel_search_batch(src_b, ref_b, border, w, h, fl, q, mvjcost, mvcost0, mvcost1, bd=bd, threads=threads)
    mv = full_mv.astype(np.int32) * 8
    if sub is not None:
        sl["start_row"], sl["start_col"] = mv[:, 0], mv[:, 1]
        smv, _, _, _ = subpel_tree_batch(src_b, ref_b, border, w, h, sl, mvjcost=mvjcost, mvcost0=mvcost0, mvcost1=mvcost1,
                                         cost_lists=cl if use_cost_list else None, bd=bd, threads=threads, **sub)
        ok = np.asarray(cost) != 2147483647
        mv[ok] = smv[ok]
    mv = mv.astype(np.int16)
    pred = build_inter_pred(ref_b, border, width, height, w, h, blocks, mv, 0, 0, bd=bd)
    sse, var = np.zeros(len(blocks), np.uint32), np.zeros(len(blocks), np.uint32)
    src_vis = src_b[border:border + height, border:border + width]
    for i, b in enumerate(blocks):
        x, y = int(b["bx"]), int(b["by"])
        var[i], sse[i], _ = variance(np.ascontiguousarray(src_vis), y, x, pred, y, x, w, h, bd=bd)
    return mv, sse, var, pred


def _cost_tables(mvjcost, mvcost0, mvcost1, keep):
    def centre(t):
        if t is None:
            return None
        t = np.ascontiguousarray(t, np.int32); keep.append(t)
        return C.c_void_p(t.ctypes.data + (t.size // 2) * 4)
    j = None
    if mvjcost is not None:
        jj = np.ascontiguousarray(mvjcost, np.int32); keep.append(jj); j = C.c_void_p(jj.ctypes.data)
    return j, centre(mvcost0), centre(mvcost1)


def refining_search_8p_batch(src_b, ref_b, border, w, h, blocks, second_pred, mask=None, invert_mask=0, cost_type=3, sad_per_bit=0, error_per_bit=0,
                             mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4):
    """av1_refining_search_8p_c + av1_get_mvpred_compound_var per block (oracle/aomref_mcomp.c).  second_pred [n, h, w] (plane dtype), mask
    [n, h, w] uint8 or None.  -> mv [n, 2], sad [n], var [n]"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    sp = np.ascontiguousarray(second_pred, src_b.dtype).reshape(n, h * w)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(n, h * w)
    mv = np.zeros((n, 2), np.int16); sad = np.zeros(n, np.int32); var = np.zeros(n, np.int32)
    keep = []
    j, c0, c1 = _cost_tables(mvjcost, mvcost0, mvcost1, keep)
    lib.orc_refining_search_8p_batch.restype = None
    lib.orc_refining_search_8p_batch(C.c_void_p(_addr(src_b, border, border)), src_b.shape[1], C.c_void_p(_addr(ref_b, border, border)), ref_b.shape[1],
                                     int(src_b.dtype != np.uint8), bd, w, h, C.c_void_p(blocks.ctypes.data), n, cost_type, sad_per_bit, error_per_bit, j, c0, c1,
                                     C.c_void_p(sp.ctypes.data), None if mk is None else C.c_void_p(mk.ctypes.data), int(invert_mask),
                                     C.c_void_p(mv.ctypes.data), C.c_void_p(sad.ctypes.data), C.c_void_p(var.ctypes.data), threads)
    return mv, sad, var


def compound_full_pixel_search_batch(src_b, ref_b, border, w, h, blocks, q, second_pred, mask=None, invert_mask=0, mvjcost=None, mvcost0=None,
                                     mvcost1=None, bd=8, threads=4):
    """av1_full_pixel_search with ms_buffers.second_pred [/ mask] set (the extensive full-pel step of av1_joint_motion_search): the diamond
    runs on sdaf / msdf and svaf / msvf, the mesh passes on the plain SAD as in the reference.  q: search params (cost_list NULL).
    -> mv [n, 2], cost [n], second_best [n, 2]"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    sp = np.ascontiguousarray(second_pred, src_b.dtype).reshape(n, h * w)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(n, h * w)
    mv = np.zeros((n, 2), np.int16); cost = np.zeros(n, np.int32); sec = np.zeros((n, 2), np.int16)
    keep = []
    j, c0, c1 = _cost_tables(mvjcost, mvcost0, mvcost1, keep)
    lib.orc_compound_full_pixel_search_batch.restype = None
    lib.orc_compound_full_pixel_search_batch(C.c_void_p(_addr(src_b, border, border)), src_b.shape[1], C.c_void_p(_addr(ref_b, border, border)),
                                             ref_b.shape[1], int(src_b.dtype != np.uint8), bd, w, h, C.byref(q), j, c0, c1,
                                             C.c_void_p(blocks.ctypes.data), n, C.c_void_p(sp.ctypes.data),
                                             None if mk is None else C.c_void_p(mk.ctypes.data), int(invert_mask), C.c_void_p(mv.ctypes.data),
                                             C.c_void_p(cost.ctypes.data), C.c_void_p(sec.ctypes.data), threads)
    return mv, cost, sec


def obmc_full_pixel_search_batch(ref_b, border, w, h, blocks, wsrc, obmc_mask, method="NSTEP", step_param=0, fast_obmc_search=0, cost_type=3, sad_per_bit=0,
                                 error_per_bit=0, mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4):
    """av1_obmc_full_pixel_search per block.  wsrc / obmc_mask [n, h, w] int32.  -> mv [n, 2], cost [n]"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    ws = np.ascontiguousarray(wsrc, np.int32).reshape(n, h * w); om = np.ascontiguousarray(obmc_mask, np.int32).reshape(n, h * w)
    mv = np.zeros((n, 2), np.int16); cost = np.zeros(n, np.int32)
    keep = []
    j, c0, c1 = _cost_tables(mvjcost, mvcost0, mvcost1, keep)
    lib.orc_obmc_full_pixel_search_batch.restype = None
    lib.orc_obmc_full_pixel_search_batch(C.c_void_p(_addr(ref_b, border, border)), ref_b.shape[1], int(ref_b.dtype != np.uint8), bd, w, h,
                                         C.c_void_p(blocks.ctypes.data), n, method if isinstance(method, int) else SEARCH_METHODS.index(method), step_param,
                                         int(fast_obmc_search), cost_type, sad_per_bit, error_per_bit, j, c0, c1, C.c_void_p(ws.ctypes.data),
                                         C.c_void_p(om.ctypes.data), C.c_void_p(mv.ctypes.data), C.c_void_p(cost.ctypes.data), threads)
    return mv, cost


def obmc_subpel_tree_batch(ref_b, border, w, h, blocks, wsrc, obmc_mask, cost_type=4, error_per_bit=0, mvjcost=None, mvcost0=None, mvcost1=None,
                           iters_per_step=2, allow_hp=1, forced_stop=0, subpel_search_type=0, bd=8, threads=4):
    """av1_find_best_obmc_sub_pixel_tree_up per block (blocks: subpel_block_dtype, start / limits in 1/8 pel; subpel_search_type 0 =
    USE_2_TAPS_ORIG, 3 = USE_8_TAPS).  wsrc / obmc_mask [n, h, w] int32.  -> mv [n, 2], err [n], distortion [n], sse [n]"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    ws = np.ascontiguousarray(wsrc, np.int32).reshape(n, h * w); om = np.ascontiguousarray(obmc_mask, np.int32).reshape(n, h * w)
    mv = np.zeros((n, 2), np.int16); err = np.zeros(n, np.uint32); dist = np.zeros(n, np.int32); sse = np.zeros(n, np.uint32)
    keep = []
    j, c0, c1 = _cost_tables(mvjcost, mvcost0, mvcost1, keep)
    lib.orc_obmc_subpel_tree_batch.restype = None
    lib.orc_obmc_subpel_tree_batch(C.c_void_p(_addr(ref_b, border, border)), ref_b.shape[1], int(ref_b.dtype != np.uint8), bd, w, h,
                                   C.c_void_p(blocks.ctypes.data), n, cost_type, error_per_bit, j, c0, c1, iters_per_step, allow_hp, forced_stop,
                                   int(subpel_search_type), C.c_void_p(ws.ctypes.data), C.c_void_p(om.ctypes.data), C.c_void_p(mv.ctypes.data),
                                   C.c_void_p(err.ctypes.data), C.c_void_p(dist.ctypes.data), C.c_void_p(sse.ctypes.data), threads)
    return mv, err, dist, sse


def compound_subpel_tree_batch(src_b, ref_b, border, w, h, blocks, second_pred, mask=None, invert_mask=0, tree=2, subpel_search_type=0, cost_type=4, error_per_bit=0,
                               mvjcost=None, mvcost0=None, mvcost1=None, iters_per_step=2, allow_hp=1, forced_stop=0, bd=8, threads=4):
    """The sub-pel trees on a compound prediction (ms_buffers.second_pred [/ mask]): second_pred [n, h, w] pixels, mask [n, h, w] uint8 or None.
    -> mv [n, 2], err [n], distortion [n], sse [n]"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    sp = np.ascontiguousarray(second_pred, src_b.dtype).reshape(n, h * w)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(n, h * w)
    mv = np.zeros((n, 2), np.int16); err = np.zeros(n, np.uint32); dist = np.zeros(n, np.int32); sse = np.zeros(n, np.uint32)
    keep = []
    j, c0, c1 = _cost_tables(mvjcost, mvcost0, mvcost1, keep)
    lib.orc_compound_subpel_tree_batch.restype = None
    lib.orc_compound_subpel_tree_batch(C.c_void_p(_addr(src_b, border, border)), src_b.shape[1], C.c_void_p(_addr(ref_b, border, border)), ref_b.shape[1],
                                       int(src_b.dtype != np.uint8), bd, w, h, tree, subpel_search_type, cost_type, error_per_bit, j, c0, c1, iters_per_step,
                                       allow_hp, forced_stop, C.c_void_p(blocks.ctypes.data), n, C.c_void_p(sp.ctypes.data),
                                       None if mk is None else C.c_void_p(mk.ctypes.data), int(invert_mask), C.c_void_p(mv.ctypes.data),
                                       C.c_void_p(err.ctypes.data), C.c_void_p(dist.ctypes.data), C.c_void_p(sse.ctypes.data), threads)
    return mv, err, dist, sse


def joint_motion_search_batch(src_b, ref0_b, ref1_b, border, width, height, w, h, blocks, ref_mv, cur_mv, mask=None, cost_type=0, sad_per_bit=0, sub=None,
                              force_integer_mv=0, mvjcost=None, mvcost0=None, mvcost1=None, bd=8, threads=4, full=None, allow_second_mv=0):
    """av1_joint_motion_search (motion_search_facade.c:496-702), composed of the pinned pieces (the sequencing itself is read from the reference,
    not interpreted).  full None: the refining-search branch; full = search params (search_params(..)): the extensive branch -- av1_full_pixel_search
    on the compound prediction, and with allow_second_mv the second sub-pel start (:621-623, :664-676).  blocks = bx, by + raw x->mv_limits; ref_mv / cur_mv [n, 2, 2] in 1/8 pel; sub = kwargs of
    compound_subpel_tree_batch (tree, subpel_search_type, error_per_bit, iters_per_step, allow_hp).  -> (cur_mv [n, 2, 2], rate_mv [n], best_err [n], iterations [n])"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    cur = np.array(cur_mv, np.int32).reshape(n, 2, 2).copy()
    init = cur.copy()
    refmv = np.asarray(ref_mv, np.int32).reshape(n, 2, 2)
    last = np.full((n, 2), 2**31 - 1, np.int64)
    live = np.ones(n, bool)
    iters = np.zeros(n, np.int32)
    refs = (ref0_b, ref1_b)
    sub = dict(sub or {})
    epb = sub.get("error_per_bit", 0)
    for ite in range(4):
        i_d = ite & 1
        for i in range(n):
            if live[i] and ite >= 2 and (cur[i, 1 - i_d] == init[i, 1 - i_d]).all():
                if (cur[i, i_d] == init[i, i_d]).all() or ((cur[i, i_d] >> 3) == (init[i, i_d] >> 3)).all():
                    live[i] = False
        plane = build_inter_pred(refs[1 - i_d], border, width, height, w, h, blocks, cur[:, 1 - i_d], 0, 0, bd=bd)
        sp = np.stack([plane[b["by"]:b["by"] + h, b["bx"]:b["bx"] + w] for b in blocks])
        fl, sl = np.array(blocks, copy=True), np.array(blocks, copy=True)
        for i, b in enumerate(blocks):
            raw = (b["row_min"], b["row_max"], b["col_min"], b["col_max"])
            rr, rc = int(refmv[i, i_d, 0]), int(refmv[i, i_d, 1])
            fl["row_min"][i], fl["row_max"][i], fl["col_min"][i], fl["col_max"][i] = set_mv_search_range(raw, rr, rc)
            sl["row_min"][i], sl["row_max"][i], sl["col_min"][i], sl["col_max"][i] = set_subpel_mv_search_range(raw, rr, rc)
            for l_ in (fl, sl):
                l_["ref_row"][i], l_["ref_col"][i] = rr, rc
        fl["start_row"], fl["start_col"] = _rawpel(cur[:, i_d, 0]), _rawpel(cur[:, i_d, 1])
        sec = None
        if full is not None:
            fmv, fsad, sec = compound_full_pixel_search_batch(src_b, refs[i_d], border, w, h, fl, full, sp, mask, i_d, mvjcost=mvjcost, mvcost0=mvcost0,
                                                              mvcost1=mvcost1, bd=bd, threads=threads)
        else:
            fmv, fsad, _ = refining_search_8p_batch(src_b, refs[i_d], border, w, h, fl, sp, mask, i_d, cost_type=cost_type, sad_per_bit=sad_per_bit,
                                                    error_per_bit=epb, mvjcost=mvjcost, mvcost0=mvcost0, mvcost1=mvcost1, bd=bd, threads=threads)
        best = fmv.astype(np.int32) * 8
        sme = fsad.astype(np.int64)
        if not force_integer_mv:
            sl["start_row"], sl["start_col"] = best[:, 0], best[:, 1]
            smv, serr, _, _ = compound_subpel_tree_batch(src_b, refs[i_d], border, w, h, sl, sp, mask, i_d, cost_type=cost_type, mvjcost=mvjcost, mvcost0=mvcost0,
                                                         mvcost1=mvcost1, forced_stop=0, bd=bd, threads=threads, **sub)
            ok = sme < 2**31 - 1
            best = np.where(ok[:, None], smv.astype(np.int32), best)
            sme = np.where(ok, serr.astype(np.int64).astype(np.int32).astype(np.int64), sme)   # (int)besterr
            if sec is not None and allow_second_mv:   # try_second: valid, != best_mv, inside the sub-pel limits
                s8 = sec.astype(np.int32) * 8
                use = ok & ~((sec[:, 0] == -32768) & (sec[:, 1] == -32768)) & (sec != fmv).any(1) & (s8[:, 0] >= sl["row_min"]) & (s8[:, 0] <= sl["row_max"]) & \
                    (s8[:, 1] >= sl["col_min"]) & (s8[:, 1] <= sl["col_max"])
                if use.any():
                    idx = np.flatnonzero(use)
                    sl2 = np.array(sl[idx], copy=True)
                    sl2["start_row"], sl2["start_col"] = s8[idx, 0], s8[idx, 1]
                    smv2, serr2, _, _ = compound_subpel_tree_batch(src_b, refs[i_d], border, w, h, sl2, sp[idx], None if mask is None else np.asarray(mask)[idx],
                                                                   i_d, cost_type=cost_type, mvjcost=mvjcost, mvcost0=mvcost0, mvcost1=mvcost1, forced_stop=0,
                                                                   bd=bd, threads=threads, **sub)
                    e2 = serr2.astype(np.int64).astype(np.int32).astype(np.int64)
                    for j, i in enumerate(idx):
                        if e2[j] < sme[i]:
                            sme[i] = e2[j]; best[i] = smv2[j]
        for i in range(n):
            if not live[i]:
                continue
            if sme[i] < last[i, i_d]:
                cur[i, i_d] = best[i]; last[i, i_d] = sme[i]; iters[i] = ite + 1
            else:
                live[i] = False
    rate = np.array([sum(mv_bit_cost(cur[i, r, 0], cur[i, r, 1], refmv[i, r, 0], refmv[i, r, 1], mvjcost, mvcost0, mvcost1) for r in range(2)) for i in range(n)], np.int32)
    return cur.astype(np.int16), rate, last.min(1).astype(np.int32), iters


def compound_single_motion_search_batch(src_b, ref_b, border, width, height, w, h, blocks, ref_mv, this_mv, full, sub=None, other_b=None, other_mv=None,
                                        filter_x=0, filter_y=0, second_pred=None, mask=None, ref_idx=0, force_integer_mv=0, mvjcost=None, mvcost0=None,
                                        mvcost1=None, bd=8, threads=4):
    """av1_compound_single_motion_search[_interinter] (motion_search_facade.c:703-853), composed of the pinned pieces: [build_second_inter_pred ->]
    av1_full_pixel_search on the compound prediction (full = search_params(..), step_param 5 in the reference) -> the compound sub-pel tree
    (forced_stop EIGHTH_PEL) -> *this_mv, *rate_mv.  blocks = bx, by + raw x->mv_limits; ref_mv / this_mv / other_mv [n, 2] in 1/8 pel.
    -> (this_mv [n, 2], rate_mv [n], bestsme [n])"""
    blocks = np.ascontiguousarray(blocks)
    n = len(blocks)
    this = np.array(this_mv, np.int32).reshape(n, 2).copy()
    refmv = np.asarray(ref_mv, np.int32).reshape(n, 2)
    if second_pred is None:
        plane = build_inter_pred(other_b, border, width, height, w, h, blocks, np.asarray(other_mv).reshape(n, 2), filter_x, filter_y, bd=bd)
        second_pred = np.stack([plane[b["by"]:b["by"] + h, b["bx"]:b["bx"] + w] for b in blocks])
    fl, sl = np.array(blocks, copy=True), np.array(blocks, copy=True)
    for i, b in enumerate(blocks):
        raw = (b["row_min"], b["row_max"], b["col_min"], b["col_max"])
        rr, rc = int(refmv[i, 0]), int(refmv[i, 1])
        fl["row_min"][i], fl["row_max"][i], fl["col_min"][i], fl["col_max"][i] = set_mv_search_range(raw, rr, rc)
        sl["row_min"][i], sl["row_max"][i], sl["col_min"][i], sl["col_max"][i] = set_subpel_mv_search_range(raw, rr, rc)
        for l_ in (fl, sl):
            l_["ref_row"][i], l_["ref_col"][i] = rr, rc
    fl["start_row"], fl["start_col"] = _rawpel(this[:, 0]), _rawpel(this[:, 1])
    fmv, fvar, _ = compound_full_pixel_search_batch(src_b, ref_b, border, w, h, fl, full, second_pred, mask, ref_idx, mvjcost=mvjcost, mvcost0=mvcost0,
                                                    mvcost1=mvcost1, bd=bd, threads=threads)
    best = fmv.astype(np.int32) * 8
    sme = fvar.astype(np.int64)
    if not force_integer_mv:
        sl["start_row"], sl["start_col"] = best[:, 0], best[:, 1]
        smv, serr, _, _ = compound_subpel_tree_batch(src_b, ref_b, border, w, h, sl, second_pred, mask, ref_idx, cost_type=full.cost_type, mvjcost=mvjcost,
                                                     mvcost0=mvcost0, mvcost1=mvcost1, forced_stop=0, bd=bd, threads=threads, **dict(sub or {}))
        ok = sme < 2**31 - 1
        best = np.where(ok[:, None], smv.astype(np.int32), best)
        sme = np.where(ok, serr.astype(np.int64).astype(np.int32).astype(np.int64), sme)
    this = np.where((sme < 2**31 - 1)[:, None], best, this)
    rate = np.array([mv_bit_cost(this[i, 0], this[i, 1], refmv[i, 0], refmv[i, 1], mvjcost, mvcost0, mvcost1) for i in range(n)], np.int32)
    return this.astype(np.int16), rate, sme.astype(np.int32)


# ---- the small members of the named files (aomref_misc.c)
lib.orc_get_mb_ss.restype = C.c_uint32
lib.orc_get_mb_ss.argtypes = [_vp]
lib.orc_mse_wxh_16bit.restype = C.c_uint64
lib.orc_mse_wxh_16bit.argtypes = [_vp, _i, _i, _vp, _i, _i, _i]
lib.orc_mse_16xh_16bit.restype = C.c_uint64
lib.orc_mse_16xh_16bit.argtypes = [_vp, _i, _vp, _i, _i]
lib.orc_comp_mask_pred.restype = None
lib.orc_comp_mask_pred.argtypes = [_vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i]
lib.orc_return_extreme_sub_pixel_mv.restype = _i
lib.orc_return_extreme_sub_pixel_mv.argtypes = [_vp, _i, _i, _vp]


def get_mb_ss(a):
    a = np.ascontiguousarray(a, np.int16)
    assert a.size == 256
    return int(lib.orc_get_mb_ss(a.ctypes.data))


def mse_wxh_16bit(dst, src, w, h):
    """dst: 2-D uint8 / uint16 view (its row pitch is used), src: 2-D uint16 view."""
    assert dst.strides[1] == dst.itemsize and src.strides[1] == 2
    return int(lib.orc_mse_wxh_16bit(dst.ctypes.data, dst.strides[0] // dst.itemsize, int(dst.itemsize == 2), src.ctypes.data, src.strides[0] // 2, w, h))


def mse_16xh_16bit(dst, src, w, h):
    src = np.ascontiguousarray(src, np.uint16)
    return int(lib.orc_mse_16xh_16bit(dst.ctypes.data, dst.strides[0], src.ctypes.data, w, h))


def comp_mask_pred(pred, ref, mask, invert_mask):
    """pred: (h, w) contiguous; ref / mask: 2-D views; -> (h, w) of pred's dtype."""
    pred = np.ascontiguousarray(pred)
    h, w = pred.shape
    out = np.empty_like(pred)
    lib.orc_comp_mask_pred(out.ctypes.data, pred.ctypes.data, w, h, ref.ctypes.data, ref.strides[0] // ref.itemsize, mask.ctypes.data, mask.strides[0],
                           int(invert_mask), int(pred.itemsize == 2))
    return out


def return_extreme_sub_pixel_mv(limits, allow_hp, want_max):
    """limits = (col_min, col_max, row_min, row_max) -> (besterr, (row, col))."""
    lim = np.asarray(limits, np.int32)
    mv = np.zeros(2, np.int16)
    e = lib.orc_return_extreme_sub_pixel_mv(lim.ctypes.data, int(allow_hp), int(want_max), mv.ctypes.data)
    return int(e), (int(mv[0]), int(mv[1]))
