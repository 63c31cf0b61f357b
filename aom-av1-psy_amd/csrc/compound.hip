// libaomhip -- the compound / masked / OBMC members of the encoder's kernel table on gfx950
// (aom_variance_fn_ptr_t: sdaf, svaf, jsdaf, jsvaf, msdf, msvf, osdf, ovf, osvf; aom_dsp/variance.h:84-103).
//
// Reference, per member:
//   svaf / jsvaf  aom_[highbd_N_][dist_wtd_]sub_pixel_avg_varianceWxH_c   aom_dsp/variance.c:165-200,563-690
//   sdaf / jsdaf  aom_[highbd_][dist_wtd_]sadWxH_avg_c                    aom_dsp/sad.c:50-64,282-297
//   msvf          aom_[highbd_N_]masked_sub_pixel_varianceWxH_c           aom_dsp/variance.c:793-811,863-928
//   msdf          aom_[highbd_]masked_sadWxH_c                            aom_dsp/sad_av1.c:20-52
//   ovf / osvf    aom_[highbd_N_]obmc_[sub_pixel_]varianceWxH_c           aom_dsp/variance.c:957-1000,1064-1192
//   osdf          aom_[highbd_]obmc_sadWxH_c                              aom_dsp/sad_av1.c:163-186,215-239
// All of them are one shape: [bilinear-interpolate the reference block] -> [blend it with a second predictor:
// rounded average, distance weights, or a 6-bit mask; or weigh it against the OBMC target] -> sum / sum of squares /
// sum of absolute differences against the source.  One kernel template does that per candidate with the lane
// mapping of variance.hip (row units of 16 bytes over TPC lanes, nothing materialised: the reference's fdata3 /
// temp2 / temp3 scratch blocks stay in registers), and the final formulas of the 8-bit / highbd 8,10,12 families.
#include "variance_device.h"

namespace aomhip {

enum { kCompWeights = 0, kCompMask = 2, kCompObmc = 3 };

struct CompoundArgs {
  const void *second_pred;      // pixel blocks, bw * bh contiguous
  const uint8_t *mask;          // kCompMask: 0..64 weights
  const int32_t *wsrc, *omask;  // kCompObmc: target and weights, both scaled by 4096, bw * bh contiguous
  const uint32_t *pred_index;   // per candidate: which second_pred / (wsrc, omask) block; null = block 0
  const uint32_t *mask_offset;  // per candidate: byte offset of its mask inside `mask`; null = 0
  int fwd, bck;                 // kCompWeights: comp = (pred * bck + ref * fwd + 8) >> 4  (8 / 8 == the rounded average)
  int mask_stride, invert;
  int bit_depth;
  uint32_t *out_var, *out_sse, *out_sad;
};

template <typename T, int W, int H, int KIND, bool SUBPEL>
__global__ __launch_bounds__(kVarThreads) void compound_kernel(PlaneView<T> src, PlaneView<T> ref, int first_frame,
                                                               const aomhip_var_cand *__restrict__ cands, int n_cands,
                                                               int64_t cand_frame_stride, int bpf8, CompoundArgs g) {
  using G = VarGeom<T, W, H>;
  constexpr int kCpb = kVarThreads / G::kTpc;
  constexpr int E = G::kUnitElems;
  const unsigned b = blockIdx.x;
  const unsigned f_rel = b / bpf8;
  const unsigned x = xcd_chunked_index(b % bpf8, bpf8);
  const int lane_in_cand = threadIdx.x % G::kTpc;
  const int ci = x * kCpb + threadIdx.x / G::kTpc;
  if (ci >= n_cands) return;
  const int64_t slot = (int64_t)f_rel * n_cands + ci;
  const aomhip_var_cand c = cands[(int64_t)f_rel * cand_frame_stride + ci];
  const int64_t fo = (int64_t)(first_frame + f_rel);
  const T *sp = src.origin + fo * src.frame_stride + (int64_t)c.sy * src.stride + c.sx;
  const T *rp = ref.origin + fo * ref.frame_stride + (int64_t)c.ry * ref.stride + c.rx;
  const int64_t blk = (g.pred_index ? (int64_t)g.pred_index[slot] : 0) * (W * H);
  const T *pp = KIND == kCompObmc ? nullptr : static_cast<const T *>(g.second_pred) + blk;
  const uint8_t *mp = KIND == kCompMask ? g.mask + (g.mask_offset ? g.mask_offset[slot] : 0) : nullptr;
  static_assert(kBilinear[3][0] == 128 - 16 * 3 && kBilinear[3][1] == 16 * 3 && kBilinear[7][0] == 16, "bilinear taps are 128 - 16 i, 16 i");
  const int fx1 = (c.xoff & 7) << 4, fx0 = 128 - fx1, fy1 = (c.yoff & 7) << 4, fy0 = 128 - fy1;   // (by arithmetic: a table index is a memory load)
  int64_t sum = 0;
  uint64_t sse = 0, sad = 0;
#pragma unroll
  for (int k = 0; k < G::kUnitsPerLane; ++k) {
    const int u = lane_in_cand + k * G::kTpc;
    const int row = u / G::kUnitsPerRow;
    const int col = (u % G::kUnitsPerRow) * E;
    int a[E];
    if constexpr (!SUBPEL) {
      load_elems<T, E>(rp + (int64_t)row * ref.stride + col, a);
    } else {
      // aom_var_filter_block2d_bil_first_pass / _second_pass (variance.c:91-139; highbd :475-520): +64 >> 7 each,
      // the intermediate kept in 16 bits, the result stored in the pixel type
      int r0[E], r0n[E], r1[E], r1n[E];
      const T *a0 = rp + (int64_t)row * ref.stride + col;
      const T *a1 = a0 + ref.stride;
      load_elems<T, E>(a0, r0);
      load_elems<T, E>(a0 + 1, r0n);
      load_elems<T, E>(a1, r1);
      load_elems<T, E>(a1 + 1, r1n);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int h0 = (r0[i] * fx0 + r0n[i] * fx1 + 64) >> 7;
        const int h1 = (r1[i] * fx0 + r1n[i] * fx1 + 64) >> 7;
        a[i] = ((__mul24(h0, fy0) + __mul24(h1, fy1) + 64) >> 7) & (sizeof(T) == 1 ? 0xFF : 0xFFFF);
      }
    }
    int32_t us = 0;
    uint32_t uq = 0, ua = 0;
    if constexpr (KIND == kCompObmc) {
      // obmc_variance / obmc_sad: the target already holds src * 4096 minus the neighbours' share
      const int32_t *wp = g.wsrc + blk + row * W + col;
      const int32_t *op = g.omask + blk + row * W + col;
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int v = wp[i] - a[i] * op[i];
        const int mag = (abs(v) + 2048) >> 12;  // ROUND_POWER_OF_TWO(abs(v), 12); _SIGNED puts the sign back
        const int d = v < 0 ? -mag : mag;
        us += d;
        uq += (uint32_t)__mul24(d, d);
        ua += (uint32_t)mag;
      }
    } else {
      int p[E], s[E];
      load_elems<T, E>(pp + row * W + col, p);
      load_elems<T, E>(sp + (int64_t)row * src.stride + col, s);
      int m[E];
      if constexpr (KIND == kCompMask) load_elems<uint8_t, E>(mp + row * g.mask_stride + col, m);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        int comp;
        if constexpr (KIND == kCompMask) {
          // AOM_BLEND_A64 (aom_dsp/blend.h:24-28); not inverted: the mask weighs the reference (variance.c:773-791)
          const int w0 = g.invert ? 64 - m[i] : m[i];
          comp = (w0 * a[i] + (64 - w0) * p[i] + 32) >> 6;
        } else {
          comp = (p[i] * g.bck + a[i] * g.fwd + 8) >> 4;  // variance.c:306-339
        }
        const int d = comp - s[i];
        us += d;
        uq += (uint32_t)__mul24(d, d);
        ua += (uint32_t)abs(d);
      }
    }
    sum += us;
    sse += uq;
    sad += ua;
  }
  sum = (int64_t)gsum64<G::kTpc>((uint64_t)sum);
  sse = gsum64<G::kTpc>(sse);
  sad = gsum64<G::kTpc>(sad);
  if (lane_in_cand == 0) {
    if (g.out_var || g.out_sse) {
      uint32_t v, q;
      finish<0, ilog2v(W * H)>(sum, sse, g.bit_depth, &v, &q);
      if (g.out_var) g.out_var[slot] = v;
      if (g.out_sse) g.out_sse[slot] = q;
    }
    // the encoder's _bits10 / _bits12 SAD wrappers (av1/encoder/encoder_utils.h:210-262,363-387,527-542)
    if (g.out_sad) g.out_sad[slot] = (uint32_t)sad >> (sizeof(T) == 1 ? 0 : g.bit_depth == 10 ? 2 : g.bit_depth == 12 ? 4 : 0);
  }
}

struct CompoundLaunch {
  hipStream_t stream;
  int first_frame, n_frames;
  const aomhip_var_cand *cands;
  int n;
  int64_t cfs;
  CompoundArgs g;
};

template <typename T, int W, int H, int KIND, bool SUBPEL>
static int launch_compound(const CompoundLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r) {
  using G = VarGeom<T, W, H>;
  constexpr int kCpb = kVarThreads / G::kTpc;
  const int bpf = (l.n + kCpb - 1) / kCpb;
  const int bpf8 = (bpf + 7) & ~7;
  hipLaunchKernelGGL((compound_kernel<T, W, H, KIND, SUBPEL>), dim3((unsigned)bpf8 * l.n_frames), dim3(kVarThreads), 0, l.stream, s,
                     r, l.first_frame, l.cands, l.n, l.cfs, bpf8, l.g);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

template <typename T, int KIND, bool SUBPEL>
static int dispatch_size(const CompoundLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, int bw, int bh) {
#define X(W, H) \
  if (bw == W && bh == H) return launch_compound<T, W, H, KIND, SUBPEL>(l, s, r);
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

template <typename T>
static int dispatch_compound(int kind, bool subpel, const CompoundLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, int bw, int bh) {
  if (kind == kCompMask)
    return subpel ? dispatch_size<T, kCompMask, true>(l, s, r, bw, bh) : dispatch_size<T, kCompMask, false>(l, s, r, bw, bh);
  if (kind == kCompObmc)
    return subpel ? dispatch_size<T, kCompObmc, true>(l, s, r, bw, bh) : dispatch_size<T, kCompObmc, false>(l, s, r, bw, bh);
  return subpel ? dispatch_size<T, kCompWeights, true>(l, s, r, bw, bh) : dispatch_size<T, kCompWeights, false>(l, s, r, bw, bh);
}

static bool params_ok(const aomhip_compound_params *p) {
  if (!p || p->kind < AOMHIP_COMP_AVG || p->kind > AOMHIP_COMP_OBMC) return false;
  if (p->kind == AOMHIP_COMP_DIST_WTD && (p->fwd_offset < 0 || p->bck_offset < 0 || p->fwd_offset + p->bck_offset != 16)) return false;
  if (p->kind == AOMHIP_COMP_MASK && p->mask_stride <= 0) return false;
  return true;
}

static void fill_args(const aomhip_compound_params *p, CompoundArgs *g) {
  g->fwd = p->kind == AOMHIP_COMP_DIST_WTD ? p->fwd_offset : 8;
  g->bck = p->kind == AOMHIP_COMP_DIST_WTD ? p->bck_offset : 8;
  g->mask_stride = p->mask_stride;
  g->invert = p->invert_mask != 0;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_compound_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame, int n_frames, int bw,
                          int bh, const aomhip_var_cand *d_cands, int n_cands, int64_t cand_frame_stride,
                          const aomhip_compound_params *p, const void *d_second_pred, const uint8_t *d_mask, const int32_t *d_obmc_wsrc,
                          const int32_t *d_obmc_mask, const uint32_t *d_pred_index, const uint32_t *d_mask_offset, uint32_t *d_var,
                          uint32_t *d_sse, uint32_t *d_sad) {
  if (!ctx || !src || !ref || !src->base || !ref->base || (n_cands > 0 && !d_cands) || !valid_block(bw, bh) ||
      (src->bit_depth == 8) != (ref->bit_depth == 8) || n_cands < 0 || n_frames < 0 || first_frame < 0 ||
      first_frame + n_frames > src->n_frames || first_frame + n_frames > ref->n_frames || !params_ok(p) || (!d_var && !d_sse && !d_sad)) {
    set_error("aomhip_compound_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if ((p->kind == AOMHIP_COMP_OBMC && (!d_obmc_wsrc || !d_obmc_mask)) || (p->kind != AOMHIP_COMP_OBMC && !d_second_pred) ||
      (p->kind == AOMHIP_COMP_MASK && !d_mask)) {
    set_error("aomhip_compound_batch: kind %d is missing one of its operand buffers", p->kind);
    return AOMHIP_ERR_INVALID;
  }
  if (n_cands == 0 || n_frames == 0) return AOMHIP_OK;
  CompoundLaunch l{ ctx->stream, first_frame, n_frames, d_cands, n_cands, cand_frame_stride, {} };
  l.g = CompoundArgs{ d_second_pred, d_mask, d_obmc_wsrc, d_obmc_mask, d_pred_index, d_mask_offset, 8, 8, 0, 0, src->bit_depth, d_var, d_sse,
                      d_sad };
  fill_args(p, &l.g);
  const int kind = p->kind == AOMHIP_COMP_MASK ? kCompMask : p->kind == AOMHIP_COMP_OBMC ? kCompObmc : kCompWeights;
  if (src->bit_depth == 8)
    return dispatch_compound<uint8_t>(kind, p->subpel != 0, l, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), bw, bh);
  return dispatch_compound<uint16_t>(kind, p->subpel != 0, l, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), bw, bh);
}

}  // extern "C"

namespace aomhip {

// rtcd-signature path (host memory, one launch per call -- the conformance functions behind the vtable entries).
// `a` is the operand that is interpolated / blended (the reference or predictor), `b` the source block.
template <typename T>
static uint32_t host_compound(const aomhip_compound_params *p, const T *a, int a_stride, int xoff, int yoff, const T *b, int b_stride,
                              const T *second_pred, const uint8_t *mask, const int32_t *wsrc, const int32_t *omask, int bw, int bh,
                              int bit_depth, bool want_sad, uint32_t *sse_out) {
  if (sse_out) *sse_out = kFailedVarCost;  // the defined results of a failed call: LOSING scores (0 would win every search)
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return kFailedVarCost;
  if (!valid_block(bw, bh) || !params_ok(p)) {
    set_error("aomhip_compound: unsupported block size %dx%d or parameters", bw, bh);
    note_failure("aomhip_compound", AOMHIP_ERR_INVALID);
    return kFailedVarCost;
  }
  const bool subpel = p->subpel != 0, obmc = p->kind == AOMHIP_COMP_OBMC;
  const int aw = bw + (subpel ? 1 : 0), ah = bh + (subpel ? 1 : 0);
  const int astr = (aw + 15) & ~15;  // padded so the kernel's trailing wide load stays inside the staging area
  auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t n = (size_t)bw * bh;
  const size_t a_bytes = (size_t)astr * (ah + 1) * sizeof(T) + 64;
  const size_t b_off = up16(a_bytes), p_off = up16(b_off + n * sizeof(T)), m_off = up16(p_off + n * sizeof(T) + 16);
  const size_t w_off = up16(m_off + n + 16), o_off = up16(w_off + (obmc ? n * 4 : 0)), c_off = up16(o_off + (obmc ? n * 4 : 0));
  const size_t r_off = c_off + 16, total = r_off + 16;
  char *h = static_cast<char *>(pinned(ctx, total));
  char *d = static_cast<char *>(scratch(ctx, total));
  if (!h || !d) { note_failure("aomhip_compound scratch", AOMHIP_ERR_NOMEM); return kFailedVarCost; }
  memset(h, 0, total);
  for (int r = 0; r < ah; ++r) memcpy(reinterpret_cast<T *>(h) + (size_t)r * astr, a + (size_t)r * a_stride, (size_t)aw * sizeof(T));
  if (!obmc) {
    for (int r = 0; r < bh; ++r) memcpy(reinterpret_cast<T *>(h + b_off) + (size_t)r * bw, b + (size_t)r * b_stride, (size_t)bw * sizeof(T));
    memcpy(h + p_off, second_pred, n * sizeof(T));
    if (p->kind == AOMHIP_COMP_MASK)
      for (int r = 0; r < bh; ++r) memcpy(h + m_off + (size_t)r * bw, mask + (size_t)r * p->mask_stride, (size_t)bw);
  } else {
    memcpy(h + w_off, wsrc, n * 4);
    memcpy(h + o_off, omask, n * 4);
  }
  *reinterpret_cast<aomhip_var_cand *>(h + c_off) = aomhip_var_cand{ 0, 0, 0, 0, (uint8_t)xoff, (uint8_t)yoff, { 0, 0 } };
  if (hipMemcpyAsync(d, h, r_off, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { note_failure("aomhip_compound H2D"); return kFailedVarCost; }
  PlaneView<T> pa{ reinterpret_cast<const T *>(d), 0, astr };
  PlaneView<T> pb{ reinterpret_cast<const T *>(d + b_off), 0, bw };
  uint32_t *res = reinterpret_cast<uint32_t *>(d + r_off);
  CompoundLaunch l{ ctx->stream, 0, 1, reinterpret_cast<const aomhip_var_cand *>(d + c_off), 1, 0, {} };
  l.g = CompoundArgs{ d + p_off, reinterpret_cast<const uint8_t *>(d + m_off), reinterpret_cast<const int32_t *>(d + w_off),
                      reinterpret_cast<const int32_t *>(d + o_off), nullptr, nullptr, 8, 8, bw, 0, bit_depth, res, res + 1, res + 2 };
  fill_args(p, &l.g);
  l.g.mask_stride = bw;  // the staged mask is packed
  const int kind = p->kind == AOMHIP_COMP_MASK ? kCompMask : obmc ? kCompObmc : kCompWeights;
  if (dispatch_compound<T>(kind, subpel, l, pb, pa, bw, bh) != AOMHIP_OK) { note_failure("aomhip_compound launch"); return kFailedVarCost; }
  if (hipMemcpyAsync(h + r_off, d + r_off, 12, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    note_failure("aomhip_compound D2H");
    return kFailedVarCost;
  }
  const uint32_t *out = reinterpret_cast<const uint32_t *>(h + r_off);
  if (sse_out) *sse_out = out[1];
  return want_sad ? out[2] : out[0];
}

}  // namespace aomhip

extern "C" unsigned int aomhip_compound(const aomhip_compound_params *p, const uint8_t *a, int a_stride, int xoffset, int yoffset,
                                        const uint8_t *b, int b_stride, const uint8_t *second_pred, const uint8_t *mask,
                                        const int32_t *obmc_wsrc, const int32_t *obmc_mask, int bw, int bh, int bd, int is_hbd,
                                        int want_sad, unsigned int *sse) {
  if (!is_hbd)
    return host_compound<uint8_t>(p, a, a_stride, xoffset, yoffset, b, b_stride, second_pred, mask, obmc_wsrc, obmc_mask, bw, bh, 8,
                                  want_sad != 0, sse);
  auto dec = [](const uint8_t *q) { return reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(q) << 1); };  // CONVERT_TO_SHORTPTR
  return host_compound<uint16_t>(p, dec(a), a_stride, xoffset, yoffset, dec(b), b_stride, dec(second_pred), mask, obmc_wsrc, obmc_mask, bw,
                                 bh, bd, want_sad != 0, sse);
}
