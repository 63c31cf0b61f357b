// Batched variance / sub-pixel (bilinear) variance on gfx950.
// Reference: aom_dsp/variance.c:56-73,141-163 (8-bit), :342-429,475-561 (highbd);
// bilinear taps aom_dsp/aom_filter.h:43-50 (FILTER_BITS 7).
//
// Same lane mapping as the SAD kernels (sad.hip): a W x H block is cut into row units of up to
// 16 bytes spread over TPC adjacent lanes; each lane accumulates sum(d) and sum(d*d) for its
// units, TPC partials are folded with DPP / lane permutes, one lane applies the reference's
// final formula (including the 10/12-bit rounding and the max(0, .) clamp).
// Differences are formed element-wise in 32 bits; 8-bit block totals fit 32 bits (variance.c:56-73),
// highbd totals are 64-bit (variance.c:342-362).
// The sub-pixel form filters (W+1) x (H+1) reference pixels in registers (horizontal 2-tap to
// 16 bits, vertical 2-tap to pixel range, +64 >> 7 each) and never materialises the reference's
// fdata3 / temp2 scratch blocks.
#include "variance_device.h"

namespace aomhip {

// SUBPEL = false: variance(src block at (sx,sy), ref block at (rx,ry)), diff = src - ref.
// SUBPEL = true : the ref block is interpolated at (rx + xoff/8, ry + yoff/8) first and diff = ref' - src
//                 (variance.c:150-163; `a` is the reference, `b` the source, av1/encoder/mcomp.c:2327).
template <typename T, int W, int H, bool SUBPEL>
__global__ __launch_bounds__(kVarThreads) void variance_kernel(PlaneView<T> src, PlaneView<T> ref, int first_frame,
                                                               const aomhip_var_cand *__restrict__ cands, int n_cands,
                                                               int64_t cand_frame_stride, uint32_t *__restrict__ out_var,
                                                               uint32_t *__restrict__ out_sse, int bpf8, int bit_depth,
                                                               int32_t *__restrict__ out_sum) {
  using G = VarGeom<T, W, H>;
  constexpr int kCpb = kVarThreads / G::kTpc;
  constexpr int E = G::kUnitElems;
  const unsigned b = blockIdx.x;
  const unsigned f_rel = b / bpf8;
  const unsigned x = xcd_chunked_index(b % bpf8, bpf8);
  const int lane_in_cand = threadIdx.x % G::kTpc;
  const int ci = x * kCpb + threadIdx.x / G::kTpc;
  if (ci >= n_cands) return;
  const aomhip_var_cand c = cands[(int64_t)f_rel * cand_frame_stride + ci];
  const int64_t fo = (int64_t)(first_frame + f_rel);
  const T *sp = src.origin + fo * src.frame_stride + (int64_t)c.sy * src.stride + c.sx;
  const T *rp = ref.origin + fo * ref.frame_stride + (int64_t)c.ry * ref.stride + c.rx;
  static_assert(kBilinear[3][0] == 128 - 16 * 3 && kBilinear[3][1] == 16 * 3 && kBilinear[7][0] == 16, "bilinear taps are 128 - 16 i, 16 i");
  const int fx1 = (c.xoff & 7) << 4, fx0 = 128 - fx1, fy1 = (c.yoff & 7) << 4, fy0 = 128 - fy1;   // (by arithmetic: a table index is a memory load)
  int64_t sum = 0;
  uint64_t sse = 0;
#pragma unroll
  for (int k = 0; k < G::kUnitsPerLane; ++k) {
    const int u = lane_in_cand + k * G::kTpc;
    const int row = u / G::kUnitsPerRow;
    const int col = (u % G::kUnitsPerRow) * E;
    int bpx[E], apx[E];
    load_elems<T, E>(sp + (int64_t)row * src.stride + col, bpx);
    if constexpr (!SUBPEL) {
      load_elems<T, E>(rp + (int64_t)row * ref.stride + col, apx);
    } else {
      int r0[E], r0n[E], r1[E], r1n[E];
      const T *a0 = rp + (int64_t)row * ref.stride + col;
      const T *a1 = a0 + ref.stride;
      load_elems<T, E>(a0, r0);
      load_elems<T, E>(a0 + 1, r0n);
      load_elems<T, E>(a1, r1);
      load_elems<T, E>(a1 + 1, r1n);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const int h0 = (r0[i] * fx0 + r0n[i] * fx1 + 64) >> 7;  // first pass, uint16 range
        const int h1 = (r1[i] * fx0 + r1n[i] * fx1 + 64) >> 7;
        apx[i] = (__mul24(h0, fy0) + __mul24(h1, fy1) + 64) >> 7;               // second pass
        if constexpr (sizeof(T) == 1) apx[i] &= 0xFF;           // stored to uint8_t temp2 (variance.c:155)
        else apx[i] &= 0xFFFF;
      }
    }
    int32_t us = 0;
    uint32_t uq = 0;
#pragma unroll
    for (int i = 0; i < E; ++i) {
      // plain variance is called as vf(src, ref): d = src - ref (av1/encoder/mcomp.c get_mvpred_var_cost);
      // the sub-pixel form as svf(ref, xoff, yoff, src): d = interpolated ref - src (mcomp.c:2327).  The sign
      // matters for the 10/12-bit rounding of the (possibly negative) sum.
      const int d = SUBPEL ? apx[i] - bpx[i] : bpx[i] - apx[i];
      us += d;
      uq += (uint32_t)__mul24(d, d);
    }
    sum += us;
    sse += uq;  // <= 8 * 4095^2 per unit: no 32-bit overflow inside a unit
  }
  if constexpr (sizeof(T) == 1) {
    sum = gsum32<G::kTpc>((int32_t)sum);
    sse = (uint32_t)gsum32<G::kTpc>((int32_t)(uint32_t)sse);  // 8-bit: totals fit 32 bits (variance.c:56-73)
  } else {
    sum = (int64_t)gsum64<G::kTpc>((uint64_t)sum);
    sse = gsum64<G::kTpc>(sse);
  }
  if (lane_in_cand == 0) {
    uint32_t v, q;
    finish<0, ilog2v(W * H)>(sum, sse, bit_depth, &v, &q);
    out_var[(int64_t)f_rel * n_cands + ci] = v;
    out_sse[(int64_t)f_rel * n_cands + ci] = q;
    if (out_sum) out_sum[(int64_t)f_rel * n_cands + ci] = (int32_t)sum;  // the raw sum of differences (aom_get*var, 8-bit)
  }
}

struct VarLaunch {
  hipStream_t stream;
  int first_frame, n_frames, bit_depth;
  int32_t *out_sum = nullptr;
};

template <typename T, int W, int H, bool SUBPEL>
static int launch_var(const VarLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, const aomhip_var_cand *c, int n,
                      int64_t cfs, uint32_t *var, uint32_t *sse) {
  using G = VarGeom<T, W, H>;
  constexpr int kCpb = kVarThreads / G::kTpc;
  const int bpf = (n + kCpb - 1) / kCpb;
  const int bpf8 = (bpf + 7) & ~7;
  hipLaunchKernelGGL((variance_kernel<T, W, H, SUBPEL>), dim3((unsigned)bpf8 * l.n_frames), dim3(kVarThreads), 0,
                     l.stream, s, r, l.first_frame, c, n, cfs, var, sse, bpf8, l.bit_depth, l.out_sum);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

template <typename T>
static int dispatch_var(bool subpel, const VarLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, int bw, int bh,
                        const aomhip_var_cand *c, int n, int64_t cfs, uint32_t *var, uint32_t *sse) {
#define X(W, H)                                                                                  \
  if (bw == W && bh == H)                                                                        \
    return subpel ? launch_var<T, W, H, true>(l, s, r, c, n, cfs, var, sse)                      \
                  : launch_var<T, W, H, false>(l, s, r, c, n, cfs, var, sse);
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

static int var_batch(bool subpel, aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame,
                     int n_frames, int bw, int bh, const aomhip_var_cand *d_cands, int n_cands,
                     int64_t cand_frame_stride, uint32_t *d_var, uint32_t *d_sse) {
  if (!ctx || !src || !ref || !src->base || !ref->base || !d_var || !d_sse || (n_cands > 0 && !d_cands) ||
      !valid_block(bw, bh) || (src->bit_depth == 8) != (ref->bit_depth == 8) || n_cands < 0 || n_frames < 0 ||
      first_frame < 0 || first_frame + n_frames > src->n_frames || first_frame + n_frames > ref->n_frames) {
    set_error("variance batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_cands == 0 || n_frames == 0) return AOMHIP_OK;
  VarLaunch l{ ctx->stream, first_frame, n_frames, src->bit_depth };
  if (src->bit_depth == 8)
    return dispatch_var<uint8_t>(subpel, l, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), bw, bh, d_cands, n_cands,
                                 cand_frame_stride, d_var, d_sse);
  return dispatch_var<uint16_t>(subpel, l, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), bw, bh, d_cands, n_cands,
                                cand_frame_stride, d_var, d_sse);
}

// rtcd-signature path: a = (bw+1) x (bh+1) pixels at a_ptr, b = bw x bh at b_ptr, host memory.
template <typename T>
static uint32_t host_variance(bool subpel, const T *a, int a_stride, int xoff, int yoff, const T *b, int b_stride, int bw,
                              int bh, int bit_depth, uint32_t *sse_out, int *sum_out = nullptr) {
  if (sse_out) *sse_out = kFailedVarCost;  // the defined results of a failed call: LOSING scores (0 would win every search)
  if (sum_out) *sum_out = 0;
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return kFailedVarCost;
  if (!valid_block(bw, bh)) {
    set_error("unsupported block size %dx%d", bw, bh);
    note_failure("aomhip_variance", AOMHIP_ERR_INVALID);
    return kFailedVarCost;
  }
  const int aw = bw + (subpel ? 1 : 0), ah = bh + (subpel ? 1 : 0);
  const int astr = (aw + 15) & ~15;  // padded so the kernel's trailing wide load stays inside the staging area
  const size_t a_bytes = (size_t)astr * (ah + 1) * sizeof(T) + 64, b_bytes = (size_t)bw * bh * sizeof(T);
  const size_t b_off = (a_bytes + 15) & ~(size_t)15, c_off = (b_off + b_bytes + 15) & ~(size_t)15;
  const size_t o_off = c_off + 16, total = o_off + 16;
  char *h = static_cast<char *>(pinned(ctx, total));
  char *d = static_cast<char *>(scratch(ctx, total));
  if (!h || !d) { note_failure("aomhip_variance scratch", AOMHIP_ERR_NOMEM); return kFailedVarCost; }
  memset(h, 0, total);
  for (int r = 0; r < ah; ++r)
    memcpy(reinterpret_cast<T *>(h) + (size_t)r * astr, a + (size_t)r * a_stride, (size_t)aw * sizeof(T));
  for (int r = 0; r < bh; ++r)
    memcpy(reinterpret_cast<T *>(h + b_off) + (size_t)r * bw, b + (size_t)r * b_stride, (size_t)bw * sizeof(T));
  aomhip_var_cand *hc = reinterpret_cast<aomhip_var_cand *>(h + c_off);
  *hc = aomhip_var_cand{ 0, 0, 0, 0, (uint8_t)xoff, (uint8_t)yoff, { 0, 0 } };
  if (hipMemcpyAsync(d, h, o_off, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { note_failure("aomhip_variance H2D"); return kFailedVarCost; }
  // sub-pixel: a is the (interpolated) "ref" operand, b the "src" operand; plain: variance(a, b) = a - b, so a
  // takes the src slot.
  PlaneView<T> pa{ reinterpret_cast<const T *>(d), 0, astr };
  PlaneView<T> pb{ reinterpret_cast<const T *>(d + b_off), 0, bw };
  const PlaneView<T> &sv = subpel ? pb : pa;
  const PlaneView<T> &rv = subpel ? pa : pb;
  VarLaunch l{ ctx->stream, 0, 1, bit_depth };
  uint32_t *dv = reinterpret_cast<uint32_t *>(d + o_off);
  l.out_sum = reinterpret_cast<int32_t *>(dv + 2);
  if (dispatch_var<T>(subpel, l, sv, rv, bw, bh, reinterpret_cast<const aomhip_var_cand *>(d + c_off), 1, 0, dv,
                      dv + 1) != AOMHIP_OK) {
    note_failure("aomhip_variance launch");
    return kFailedVarCost;
  }
  if (hipMemcpyAsync(h + o_off, d + o_off, 12, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    note_failure("aomhip_variance D2H");
    return kFailedVarCost;
  }
  const uint32_t *res = reinterpret_cast<const uint32_t *>(h + o_off);
  if (sse_out) *sse_out = res[1];
  if (sum_out) *sum_out = (int)(int32_t)res[2];
  return res[0];
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_variance_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame,
                          int n_frames, int bw, int bh, const aomhip_var_cand *d_cands, int n_cands,
                          int64_t cand_frame_stride, uint32_t *d_var, uint32_t *d_sse) {
  return var_batch(false, ctx, src, ref, first_frame, n_frames, bw, bh, d_cands, n_cands, cand_frame_stride, d_var,
                   d_sse);
}

int aomhip_sub_pixel_variance_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref,
                                    int first_frame, int n_frames, int bw, int bh, const aomhip_var_cand *d_cands,
                                    int n_cands, int64_t cand_frame_stride, uint32_t *d_var, uint32_t *d_sse) {
  return var_batch(true, ctx, src, ref, first_frame, n_frames, bw, bh, d_cands, n_cands, cand_frame_stride, d_var,
                   d_sse);
}

unsigned int aomhip_variance(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int bw, int bh,
                             unsigned int *sse) {
  return host_variance<uint8_t>(false, a, a_stride, 0, 0, b, b_stride, bw, bh, 8, sse);
}

unsigned int aomhip_sub_pixel_variance(const uint8_t *a, int a_stride, int xoffset, int yoffset, const uint8_t *b,
                                       int b_stride, int bw, int bh, unsigned int *sse) {
  return host_variance<uint8_t>(true, a, a_stride, xoffset, yoffset, b, b_stride, bw, bh, 8, sse);
}

unsigned int aomhip_mse(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int bw, int bh, unsigned int *sse) {
  uint32_t q = 0;
  host_variance<uint8_t>(false, a, a_stride, 0, 0, b, b_stride, bw, bh, 8, &q);
  if (sse) *sse = q;
  return q;
}

void aomhip_get_var(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int bw, int bh, unsigned int *sse,
                    int *sum) {
  uint32_t q = 0;
  host_variance<uint8_t>(false, a, a_stride, 0, 0, b, b_stride, bw, bh, 8, &q, sum);
  if (sse) *sse = q;
}

void aomhip_get_var_sse_sum_8x8_quad(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, uint32_t *sse8x8,
                                     int *sum8x8, unsigned int *tot_sse, int *tot_sum, uint32_t *var8x8) {
  for (int k = 0; k < 4; ++k) {
    uint32_t q = 0;
    var8x8[k] = host_variance<uint8_t>(false, a + 8 * k, a_stride, 0, 0, b + 8 * k, b_stride, 8, 8, 8, &q, &sum8x8[k]);
    sse8x8[k] = q;
    *tot_sse += q;
    *tot_sum += sum8x8[k];
  }
}

void aomhip_get_var_sse_sum_16x16_dual(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, uint32_t *sse16x16,
                                       unsigned int *tot_sse, int *tot_sum, uint32_t *var16x16) {
  for (int k = 0; k < 2; ++k) {
    uint32_t q = 0;
    int sum = 0;
    var16x16[k] = host_variance<uint8_t>(false, a + 16 * k, a_stride, 0, 0, b + 16 * k, b_stride, 16, 16, 8, &q, &sum);
    sse16x16[k] = q;
    *tot_sse += q;
    *tot_sum += sum;
  }
}

unsigned int aomhip_variance16x16(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, unsigned int *sse) {
  return aomhip_variance(a, a_stride, b, b_stride, 16, 16, sse);
}

unsigned int aomhip_highbd_variance(const uint8_t *a8, int a_stride, const uint8_t *b8, int b_stride, int bw, int bh,
                                    int bd, unsigned int *sse) {
  const uint16_t *a = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(a8) << 1);
  const uint16_t *b = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(b8) << 1);
  return host_variance<uint16_t>(false, a, a_stride, 0, 0, b, b_stride, bw, bh, bd, sse);
}

unsigned int aomhip_highbd_sub_pixel_variance(const uint8_t *a8, int a_stride, int xoffset, int yoffset,
                                              const uint8_t *b8, int b_stride, int bw, int bh, int bd,
                                              unsigned int *sse) {
  const uint16_t *a = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(a8) << 1);
  const uint16_t *b = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(b8) << 1);
  return host_variance<uint16_t>(true, a, a_stride, xoffset, yoffset, b, b_stride, bw, bh, bd, sse);
}

}  // extern "C"
