// Integer-MV motion-compensated prediction: for a full-pel motion vector the reference's inter predictor
// (av1_build_inter_predictor -> the convolve with zero sub-pel phase, av1/common/reconinter.c) degenerates to
// aom_convolve_copy, i.e. pred block = reference block at (bx + mv.col, by + mv.row).  This is only the
// glue the frame-level pipeline (search -> residual -> transform) needs.
//
// Sub-pel MVs: inter_pred_kernel below is the single-reference, unscaled luma predictor behind
// av1_enc_build_inter_predictor (av1/encoder/reconinter_enc.c:47-51 -> av1_make_inter_predictor ->
// [highbd_]inter_predictor, av1/common/reconinter.h:252-296 -> av1_[highbd_]convolve_2d_facade,
// av1/common/convolve.c:495-567,982-1058): 8-tap separable interpolation at 1/16-pel phases.
#include "common.h"
#include "pred_device.h"

namespace aomhip {

// One LANE owns one column of one block and walks down its H + 7 input rows: per row one 8-pixel load (load8_pairs) and
// the horizontal 8-tap (av1_convolve_2d_sr_c's first loop, convolve.c:92-106), the last 8 intermediates stay in
// registers, and from row 7 on the vertical 8-tap + the two rounding stages + clip produce one output pixel per row
// (:108-125).  Pixels, intermediates and taps travel as packed 16-bit pairs, so each 8-tap is four v_dot2c_i32_i16.
// A wavefront therefore holds 64 / W blocks side by side (W <= 64; two column passes for W = 128), every
// global access is a run of W adjacent pixels, and nothing goes through LDS or scratch.
// The facade's other three cases need no code of their own: with round_0 + round_1 == 14 (non-compound,
// convolve.h:72-81) the 2-D pipeline with an identity kernel (phase 0 = {0,0,0,128,0,0,0,0}) in one direction is
// bit-identical to av1_convolve_x_sr / _y_sr / aom_convolve_copy -- the offsets cancel exactly and
// (128 a + 2^(13 - r0)) >> (14 - r0) == (a + 2^(6 - r0)) >> (7 - r0).  tests/test_gpu_inter_pred.py checks all four
// cases against the oracle, which restates them separately, and against the interpreted reference.
template <typename T, int W, int H>
__global__ __launch_bounds__(256) void inter_pred_kernel(PlaneView<T> ref, int ref_frame, T *dst_origin, int dst_stride,
                                                         const aomhip_search_block *__restrict__ blocks,
                                                         const int16_t *__restrict__ mv, int n_blocks, int set_x, int set_y, int bit_depth,
                                                         int x_lo, int x_hi, int y_lo, int y_hi, int mvx_mul, int mvy_mul) {
  constexpr int LPB = W < 64 ? W : 64;  // lanes per block
  constexpr int BPW = 64 / LPB;         // blocks per wavefront
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = (blockIdx.x * 4 + wave) * BPW + lane / LPB;
  if (bi >= n_blocks) return;
  const int col0 = lane % LPB;
  const int bx = blocks[bi].bx, by = blocks[bi].by;
  // init_subpel_params (reconinter.h:130-165), unscaled: position in 1/16 pel, mv * (1 << (1 - subsampling))
  int pos_x = (bx << 4) + mv[2 * bi + 1] * mvx_mul, pos_y = (by << 4) + mv[2 * bi] * mvy_mul;
  // keeps every access inside the allocation; the identity for MVs within av1_set_mv_limits (mcomp.h:216-247)
  pos_x = min(max(pos_x, x_lo), x_hi);
  pos_y = min(max(pos_y, y_lo), y_hi);
  const int tbd = sizeof(T) == 1 ? 8 : bit_depth;
  const int r0 = tbd == 12 ? 5 : 3, r1 = 14 - r0;  // get_conv_params_no_round (convolve.h:72-81)
  const int ob = tbd + 14 - r0;
  const int hoff = (1 << (tbd + 6)) + ((1 << r0) >> 1);
  const int voff = (1 << ob) + ((1 << r1) >> 1);
  const int vsub = (1 << (ob - r1)) + (1 << (ob - r1 - 1));
  const int pmax = (1 << tbd) - 1;
  // the kernels as four (tap 2k, tap 2k + 1) pairs each: exactly the table's memory layout
  const PU128 fx = *reinterpret_cast<const PU128 *>(&kInterp[set_x][pos_x & 15][0]);
  const PU128 fy = *reinterpret_cast<const PU128 *>(&kInterp[set_y][pos_y & 15][0]);
  const T *base = ref.origin + (int64_t)ref_frame * ref.frame_stride + (int64_t)((pos_y >> 4) - 3) * ref.stride + (pos_x >> 4) - 3;
  // dst_stride 0: block i contiguous at dst_origin + i * W * H (the second_pred layout of the compound searches)
  T *dbase = dst_stride ? dst_origin + (int64_t)by * dst_stride + bx : dst_origin + (int64_t)bi * (W * H);
  if (!dst_stride) dst_stride = W;
#pragma unroll 1
  for (int col = col0; col < W; col += 64) {
    const T *p = base + col;
    T *d = dbase + col;
    // the last 8 intermediates as four 16-bit pairs (they fit 15 bits: convolve.h:76-81 sizes round_0 for that)
    uint32_t win[4] = { 0, 0, 0, 0 };
#pragma unroll 8
    for (int r = 0; r < H + 7; ++r) {
      uint32_t px[4];
      load8_pairs<T>(p, px);
      p += ref.stride;
      int hs = hoff;
#pragma unroll
      for (int k = 0; k < 4; ++k) hs = dot2(px[k], fx.v[k], hs);
#pragma unroll
      for (int k = 0; k < 3; ++k) win[k] = __builtin_amdgcn_alignbit(win[k + 1], win[k], 16);
      win[3] = __builtin_amdgcn_alignbit((uint32_t)(hs >> r0), win[3], 16);
      if (r >= 7) {
        int vs = voff;
#pragma unroll
        for (int k = 0; k < 4; ++k) vs = dot2(win[k], fy.v[k], vs);
        int res = (vs >> r1) - vsub;
        if constexpr (sizeof(T) == 1) res = (int16_t)res;  // convolve.c:119 keeps it in an int16_t
        // bits = 14 - round_0 - round_1 == 0: ROUND_POWER_OF_TWO(res, 0) is res
        *d = (T)min(max(res, 0), pmax);
        d += dst_stride;
      }
    }
  }
}

// Compound prediction (two references): convolve_2d_facade_compound (convolve.c:471-493) -> av1_[highbd_]dist_wtd_convolve_*
// with get_conv_params_no_round's compound rounding (round_1 = COMPOUND_ROUND1_BITS 7).  Same lane-per-column walk as
// inter_pred_kernel, with both references' windows side by side in registers; the CONV_BUF intermediate of the first
// reference never exists in memory.  As in the single-reference case the copy / x / y kernels are the 2-D pipeline with an
// identity kernel: with round_1 = 7 every one of them produces exactly round_offset + the scaled value the 2-D formula
// gives (the offsets 2^offset_bits + 2^(offset_bits - 1) are multiples of 2^7 and the identity tap is 2^7).
template <typename T, int W, int H>
__global__ __launch_bounds__(256) void compound_pred_kernel(PlaneView<T> ref0, int frame0, PlaneView<T> ref1, int frame1, T *dst_origin,
                                                            int dst_stride, const aomhip_search_block *__restrict__ blocks,
                                                            const int16_t *__restrict__ mv0, const int16_t *__restrict__ mv1, int n_blocks,
                                                            int set_x, int set_y, int bit_depth, int x_lo, int x_hi, int y_lo, int y_hi,
                                                            int mvx_mul, int mvy_mul, int fwd, int bck, const uint8_t *__restrict__ mask,
                                                            const uint32_t *__restrict__ mask_offset, int mask_stride, int subw, int subh, int diffwtd,
                                                            uint8_t *__restrict__ mask_out) {
  constexpr int LPB = W < 64 ? W : 64;
  constexpr int BPW = 64 / LPB;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = (blockIdx.x * 4 + wave) * BPW + lane / LPB;
  if (bi >= n_blocks) return;
  const int col0 = lane % LPB;
  const int bx = blocks[bi].bx, by = blocks[bi].by;
  const uint8_t *mk = mask ? mask + (mask_offset ? mask_offset[bi] : 0) : nullptr;
  int px0 = (bx << 4) + mv0[2 * bi + 1] * mvx_mul, py0 = (by << 4) + mv0[2 * bi] * mvy_mul;
  int px1 = (bx << 4) + mv1[2 * bi + 1] * mvx_mul, py1 = (by << 4) + mv1[2 * bi] * mvy_mul;
  px0 = min(max(px0, x_lo), x_hi); py0 = min(max(py0, y_lo), y_hi);
  px1 = min(max(px1, x_lo), x_hi); py1 = min(max(py1, y_lo), y_hi);
  const int tbd = sizeof(T) == 1 ? 8 : bit_depth;
  const int r0 = tbd == 12 ? 5 : 3, r1 = 7;           // get_conv_params_no_round, is_compound (convolve.h:72-81)
  const int ob = tbd + 14 - r0;
  const int hoff = (1 << (tbd + 6)) + ((1 << r0) >> 1);
  const int voff = (1 << ob) + ((1 << r1) >> 1);
  const int round_offset = (1 << (ob - r1)) + (1 << (ob - r1 - 1));
  const int rb = 14 - r0 - r1;                         // round_bits: 4, or 2 for 12-bit
  const int pmax = (1 << tbd) - 1;
  const PU128 fx0 = *reinterpret_cast<const PU128 *>(&kInterp[set_x][px0 & 15][0]);
  const PU128 fy0 = *reinterpret_cast<const PU128 *>(&kInterp[set_y][py0 & 15][0]);
  const PU128 fx1 = *reinterpret_cast<const PU128 *>(&kInterp[set_x][px1 & 15][0]);
  const PU128 fy1 = *reinterpret_cast<const PU128 *>(&kInterp[set_y][py1 & 15][0]);
  const T *b0 = ref0.origin + (int64_t)frame0 * ref0.frame_stride + (int64_t)((py0 >> 4) - 3) * ref0.stride + (px0 >> 4) - 3;
  const T *b1 = ref1.origin + (int64_t)frame1 * ref1.frame_stride + (int64_t)((py1 >> 4) - 3) * ref1.stride + (px1 >> 4) - 3;
  T *dbase = dst_origin + (int64_t)by * dst_stride + bx;
#pragma unroll 1
  for (int col = col0; col < W; col += 64) {
    const T *p0 = b0 + col, *p1 = b1 + col;
    T *d = dbase + col;
    uint32_t w0[4] = { 0, 0, 0, 0 }, w1[4] = { 0, 0, 0, 0 };
#pragma unroll 4
    for (int r = 0; r < H + 7; ++r) {
      uint32_t a[4], b[4];
      load8_pairs<T>(p0, a);
      load8_pairs<T>(p1, b);
      p0 += ref0.stride;
      p1 += ref1.stride;
      int h0 = hoff, h1 = hoff;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        h0 = dot2(a[k], fx0.v[k], h0);
        h1 = dot2(b[k], fx1.v[k], h1);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        w0[k] = __builtin_amdgcn_alignbit(w0[k + 1], w0[k], 16);
        w1[k] = __builtin_amdgcn_alignbit(w1[k + 1], w1[k], 16);
      }
      w0[3] = __builtin_amdgcn_alignbit((uint32_t)(h0 >> r0), w0[3], 16);
      w1[3] = __builtin_amdgcn_alignbit((uint32_t)(h1 >> r0), w1[3], 16);
      if (r >= 7) {
        int v0 = voff, v1 = voff;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v0 = dot2(w0[k], fy0.v[k], v0);
          v1 = dot2(w1[k], fy1.v[k], v1);
        }
        const int res0 = v0 >> r1, res1 = v1 >> r1;   // the two CONV_BUF values (fit 16 bits)
        // convolve.c:222-233: distance weights (sum 16) or the plain average, then the offset comes out and the result is rounded
        int tmp;
        if (diffwtd) {  // av1_build_compound_diffwtd_mask_d16 (reconinter.c:296-328): the mask comes from the two predictors
          const int rnd = rb + tbd - 8;
          int diff = res0 > res1 ? res0 - res1 : res1 - res0;
          diff = (diff + ((1 << rnd) >> 1)) >> rnd;
          int m = min(38 + (diff >> 4), 64);   // DIFF_FACTOR 16; never below 38
          m = diffwtd == 2 ? 64 - m : m;
          if (mask_out) mask_out[((int64_t)bi * H + (r - 7)) * W + col] = (uint8_t)m;
          tmp = (m * res0 + (64 - m) * res1) >> 6;
        } else if (mk) {  // aom_[lowbd|highbd]_blend_a64_d16_mask (aom_dsp/blend_a64_mask.c): the mask weighs reference 0
          const int y = r - 7;
          const uint8_t *mr = mk + (y << subh) * mask_stride + (col << subw);
          int m = mr[0];
          if (subw & subh) m = (m + mr[1] + mr[mask_stride] + mr[mask_stride + 1] + 2) >> 2;
          else if (subw) m = (m + mr[1] + 1) >> 1;
          else if (subh) m = (m + mr[mask_stride] + 1) >> 1;
          tmp = (m * res0 + (64 - m) * res1) >> 6;
        } else {
          tmp = fwd | bck ? (res0 * fwd + res1 * bck) >> 4 : (res0 + res1) >> 1;
        }
        tmp -= round_offset;
        tmp = (tmp + ((1 << rb) >> 1)) >> rb;
        *d = (T)min(max(tmp, 0), pmax);
        d += dst_stride;
      }
    }
  }
}

// aom_[highbd_]blend_a64_vmask / _hmask in place (aom_dsp/blend_a64_vmask.c, blend_a64_hmask.c): the OBMC blends of
// build_obmc_inter_pred_above / _left (av1/common/reconinter.c:844-920).  One wavefront per rectangle.
template <typename T>
__global__ __launch_bounds__(256) void blend_1d_kernel(T *dst, int dst_stride, const T *__restrict__ src1, int src1_stride,
                                                       const aomhip_blend_item *__restrict__ items, int n, const uint8_t *__restrict__ masks) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int ii = blockIdx.x * 4 + wave;
  if (ii >= n) return;
  const aomhip_blend_item it = items[ii];
  const uint8_t *mask = masks + it.mask_offset;
  T *d = dst + (int64_t)it.y * dst_stride + it.x;
  const T *s = src1 + (int64_t)it.y * src1_stride + it.x;
  for (int i = lane; i < it.w * it.h; i += 64) {
    const int r = i / it.w, c = i - r * it.w;
    const int m = mask[it.vertical ? r : c];
    const int a = d[(int64_t)r * dst_stride + c], b = s[(int64_t)r * src1_stride + c];
    d[(int64_t)r * dst_stride + c] = (T)((m * a + (64 - m) * b + 32) >> 6);   // AOM_BLEND_A64 (aom_dsp/blend.h:24-28)
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pred_copy_kernel(PlaneView<T> ref, int ref_frame, T *dst_origin, int dst_stride,
                                                        const aomhip_search_block *__restrict__ blocks,
                                                        const int16_t *__restrict__ mv, int n_blocks, int bw, int bh) {
  // one wavefront per block; lanes sweep the block in 16-byte pieces
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + wave;
  if (bi >= n_blocks) return;
  const int bx = blocks[bi].bx, by = blocks[bi].by;
  const int row = mv[2 * bi], col = mv[2 * bi + 1];
  const T *src = ref.origin + (int64_t)ref_frame * ref.frame_stride + (int64_t)(by + row) * ref.stride + bx + col;
  T *dst = dst_origin + (int64_t)by * dst_stride + bx;
  const int total = bw * bh;
  for (int i = lane; i < total; i += 64) {
    const int r = i / bw, c = i - r * bw;
    dst[(int64_t)r * dst_stride + c] = src[(int64_t)r * ref.stride + c];
  }
}


#define AOMHIP_PRED_SIZES(X)                                                                                     \
  X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) \
  X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)

template <typename T>
static int launch_inter_pred(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred, int pred_frame, int bw,
                             int bh, const aomhip_search_block *d_blocks, const int16_t *d_mv, int n_blocks, int fx, int fy, int ss_x, int ss_y,
                             void *d_contiguous = nullptr) {
  // av1_get_interp_filter_params_with_block_size (filter.h:247-253): a dimension <= 4 takes the 4-tap sets
  auto set_of = [](int f, int dim) { return dim <= 4 ? (f == 1 ? 5 : f == 3 ? 3 : 4) : f; };
  T *d = d_contiguous ? static_cast<T *>(d_contiguous)
                      : reinterpret_cast<T *>(pred->base) + (size_t)pred_frame * pred->frame_stride + (size_t)pred->border * pred->stride + pred->border;
  const int dst_stride = d_contiguous ? 0 : pred->stride;
  // the 8-pixel row load starts 3 left of the integer position and the walk covers rows -3 .. bh + 3
  const int x_lo = (-ref->border + 3) << 4, x_hi = ((ref->width + ref->border - bw - 5) << 4) | 15;
  const int y_lo = (-ref->border + 3) << 4, y_hi = ((ref->height + ref->border - bh - 5) << 4) | 15;
  const int lpb = bw < 64 ? bw : 64, bpw = 64 / lpb;
  const dim3 grid((n_blocks + 4 * bpw - 1) / (4 * bpw)), block(256);
#define X(W, H)                                                                                                                    \
  if (bw == W && bh == H) {                                                                                                        \
    hipLaunchKernelGGL((inter_pred_kernel<T, W, H>), grid, block, 0, ctx->stream, view_of<T>(*ref), ref_frame, d, dst_stride,     \
                       d_blocks, d_mv, n_blocks, set_of(fx, W), set_of(fy, H), ref->bit_depth, x_lo, x_hi, y_lo, y_hi, 2 >> ss_x,  \
                       2 >> ss_y);                                                                                                 \
    AOMHIP_LAUNCH_CHECK();                                                                                                         \
    return AOMHIP_OK;                                                                                                              \
  }
  AOMHIP_PRED_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_build_inter_pred_ex_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred,
                                                int pred_frame, int bw, int bh, const aomhip_search_block *d_blocks, const int16_t *d_mv,
                                                int n_blocks, int interp_filter_x, int interp_filter_y, int subsampling_x, int subsampling_y) {
  if (!ctx || !ref || !pred || !ref->base || !pred->base || (n_blocks > 0 && (!d_blocks || !d_mv)) || n_blocks < 0 || ref_frame < 0 ||
      ref_frame >= ref->n_frames || pred_frame < 0 || pred_frame >= pred->n_frames || !valid_block(bw, bh) ||
      ref->bit_depth != pred->bit_depth || interp_filter_x < 0 || interp_filter_x > 3 || interp_filter_y < 0 || interp_filter_y > 3 ||
      subsampling_x < 0 || subsampling_x > 1 || subsampling_y < 0 || subsampling_y > 1) {
    set_error("aomhip_build_inter_pred_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (ref->border < 8) {
    set_error("aomhip_build_inter_pred_batch: the reference planes need a border of at least 8 pixels for the 8-tap kernels");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (ref->bit_depth == 8)
    return launch_inter_pred<uint8_t>(ctx, ref, ref_frame, pred, pred_frame, bw, bh, d_blocks, d_mv, n_blocks, interp_filter_x, interp_filter_y,
                                      subsampling_x, subsampling_y);
  return launch_inter_pred<uint16_t>(ctx, ref, ref_frame, pred, pred_frame, bw, bh, d_blocks, d_mv, n_blocks, interp_filter_x, interp_filter_y,
                                     subsampling_x, subsampling_y);
}

extern "C" int aomhip_build_inter_pred_contiguous_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, void *d_pred, int bw, int bh,
                                                        const aomhip_search_block *d_blocks, const int16_t *d_mv, int n_blocks, int interp_filter_x,
                                                        int interp_filter_y) {
  if (!ctx || !ref || !ref->base || !d_pred || (n_blocks > 0 && (!d_blocks || !d_mv)) || n_blocks < 0 || ref_frame < 0 || ref_frame >= ref->n_frames ||
      !valid_block(bw, bh) || interp_filter_x < 0 || interp_filter_x > 3 || interp_filter_y < 0 || interp_filter_y > 3) {
    set_error("aomhip_build_inter_pred_contiguous_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (ref->border < 8) {
    set_error("aomhip_build_inter_pred_contiguous_batch: the reference planes need a border of at least 8 pixels for the 8-tap kernels");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (ref->bit_depth == 8)
    return launch_inter_pred<uint8_t>(ctx, ref, ref_frame, nullptr, 0, bw, bh, d_blocks, d_mv, n_blocks, interp_filter_x, interp_filter_y, 0, 0, d_pred);
  return launch_inter_pred<uint16_t>(ctx, ref, ref_frame, nullptr, 0, bw, bh, d_blocks, d_mv, n_blocks, interp_filter_x, interp_filter_y, 0, 0, d_pred);
}

extern "C" int aomhip_build_inter_pred_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred,
                                             int pred_frame, int bw, int bh, const aomhip_search_block *d_blocks, const int16_t *d_mv,
                                             int n_blocks, int interp_filter_x, int interp_filter_y) {
  return aomhip_build_inter_pred_ex_batch(ctx, ref, ref_frame, pred, pred_frame, bw, bh, d_blocks, d_mv, n_blocks, interp_filter_x,
                                          interp_filter_y, 0, 0);
}

template <typename T>
static int launch_compound_pred(aomhip_ctx *ctx, const aomhip_planes *r0, int f0, const aomhip_planes *r1, int f1, const aomhip_planes *pred,
                                int pred_frame, int bw, int bh, const aomhip_search_block *d_blocks, const int16_t *mv0, const int16_t *mv1,
                                int n_blocks, int fx, int fy, int fwd, int bck, int ss_x, int ss_y, const uint8_t *mask = nullptr,
                                const uint32_t *mask_offset = nullptr, int mask_stride = 0, int subw = 0, int subh = 0, int diffwtd = 0,
                                uint8_t *mask_out = nullptr) {
  auto set_of = [](int f, int dim) { return dim <= 4 ? (f == 1 ? 5 : f == 3 ? 3 : 4) : f; };
  T *d = reinterpret_cast<T *>(pred->base) + (size_t)pred_frame * pred->frame_stride + (size_t)pred->border * pred->stride + pred->border;
  const int border = r0->border < r1->border ? r0->border : r1->border;
  const int x_lo = (-border + 3) << 4, x_hi = ((r0->width + border - bw - 5) << 4) | 15;
  const int y_lo = (-border + 3) << 4, y_hi = ((r0->height + border - bh - 5) << 4) | 15;
  const int lpb = bw < 64 ? bw : 64, bpw = 64 / lpb;
  const dim3 grid((n_blocks + 4 * bpw - 1) / (4 * bpw)), block(256);
#define X(W, H)                                                                                                                        \
  if (bw == W && bh == H) {                                                                                                            \
    hipLaunchKernelGGL((compound_pred_kernel<T, W, H>), grid, block, 0, ctx->stream, view_of<T>(*r0), f0, view_of<T>(*r1), f1, d,      \
                       pred->stride, d_blocks, mv0, mv1, n_blocks, set_of(fx, W), set_of(fy, H), r0->bit_depth, x_lo, x_hi, y_lo, y_hi, \
                       2 >> ss_x, 2 >> ss_y, fwd, bck, mask, mask_offset, mask_stride, subw, subh, diffwtd, mask_out);                  \
    AOMHIP_LAUNCH_CHECK();                                                                                                             \
    return AOMHIP_OK;                                                                                                                  \
  }
  AOMHIP_PRED_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

extern "C" int aomhip_build_compound_pred_batch(aomhip_ctx *ctx, const aomhip_planes *ref0, int ref0_frame, const aomhip_planes *ref1,
                                                int ref1_frame, const aomhip_planes *pred, int pred_frame, int bw, int bh,
                                                const aomhip_search_block *d_blocks, const int16_t *d_mv0, const int16_t *d_mv1, int n_blocks,
                                                int interp_filter_x, int interp_filter_y, int fwd_offset, int bck_offset, int subsampling_x,
                                                int subsampling_y) {
  if (!ctx || !ref0 || !ref1 || !pred || !ref0->base || !ref1->base || !pred->base || (n_blocks > 0 && (!d_blocks || !d_mv0 || !d_mv1)) ||
      n_blocks < 0 || ref0_frame < 0 || ref0_frame >= ref0->n_frames || ref1_frame < 0 || ref1_frame >= ref1->n_frames || pred_frame < 0 ||
      pred_frame >= pred->n_frames || !valid_block(bw, bh) || ref0->bit_depth != pred->bit_depth || ref1->bit_depth != pred->bit_depth ||
      ref0->width != ref1->width || ref0->height != ref1->height || interp_filter_x < 0 || interp_filter_x > 3 || interp_filter_y < 0 ||
      interp_filter_y > 3 || subsampling_x < 0 || subsampling_x > 1 || subsampling_y < 0 || subsampling_y > 1 || fwd_offset < 0 ||
      bck_offset < 0 || ((fwd_offset | bck_offset) && fwd_offset + bck_offset != 16) || ref0->border < 8 || ref1->border < 8) {
    set_error("aomhip_build_compound_pred_batch: invalid argument (weights 0 / 0 or summing to 16; reference borders >= 8)");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (pred->bit_depth == 8)
    return launch_compound_pred<uint8_t>(ctx, ref0, ref0_frame, ref1, ref1_frame, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n_blocks,
                                         interp_filter_x, interp_filter_y, fwd_offset, bck_offset, subsampling_x, subsampling_y);
  return launch_compound_pred<uint16_t>(ctx, ref0, ref0_frame, ref1, ref1_frame, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n_blocks,
                                        interp_filter_x, interp_filter_y, fwd_offset, bck_offset, subsampling_x, subsampling_y);
}

extern "C" int aomhip_build_masked_compound_pred_batch(aomhip_ctx *ctx, const aomhip_planes *ref0, int ref0_frame, const aomhip_planes *ref1,
                                                       int ref1_frame, const aomhip_planes *pred, int pred_frame, int bw, int bh,
                                                       const aomhip_search_block *d_blocks, const int16_t *d_mv0, const int16_t *d_mv1,
                                                       int n_blocks, int interp_filter_x, int interp_filter_y, const uint8_t *d_mask,
                                                       const uint32_t *d_mask_offset, int mask_stride, int mask_subw, int mask_subh,
                                                       int subsampling_x, int subsampling_y) {
  if (!ctx || !ref0 || !ref1 || !pred || !ref0->base || !ref1->base || !pred->base || (n_blocks > 0 && (!d_blocks || !d_mv0 || !d_mv1)) ||
      n_blocks < 0 || ref0_frame < 0 || ref0_frame >= ref0->n_frames || ref1_frame < 0 || ref1_frame >= ref1->n_frames || pred_frame < 0 ||
      pred_frame >= pred->n_frames || !valid_block(bw, bh) || ref0->bit_depth != pred->bit_depth || ref1->bit_depth != pred->bit_depth ||
      ref0->width != ref1->width || ref0->height != ref1->height || interp_filter_x < 0 || interp_filter_x > 3 || interp_filter_y < 0 ||
      interp_filter_y > 3 || subsampling_x < 0 || subsampling_x > 1 || subsampling_y < 0 || subsampling_y > 1 || !d_mask || mask_stride <= 0 ||
      mask_subw < 0 || mask_subw > 1 || mask_subh < 0 || mask_subh > 1 || ref0->border < 8 || ref1->border < 8) {
    set_error("aomhip_build_masked_compound_pred_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (pred->bit_depth == 8)
    return launch_compound_pred<uint8_t>(ctx, ref0, ref0_frame, ref1, ref1_frame, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n_blocks,
                                         interp_filter_x, interp_filter_y, 0, 0, subsampling_x, subsampling_y, d_mask, d_mask_offset, mask_stride,
                                         mask_subw, mask_subh);
  return launch_compound_pred<uint16_t>(ctx, ref0, ref0_frame, ref1, ref1_frame, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n_blocks,
                                        interp_filter_x, interp_filter_y, 0, 0, subsampling_x, subsampling_y, d_mask, d_mask_offset, mask_stride,
                                        mask_subw, mask_subh);
}

extern "C" int aomhip_build_diffwtd_compound_pred_batch(aomhip_ctx *ctx, const aomhip_planes *ref0, int ref0_frame, const aomhip_planes *ref1,
                                                        int ref1_frame, const aomhip_planes *pred, int pred_frame, int bw, int bh,
                                                        const aomhip_search_block *d_blocks, const int16_t *d_mv0, const int16_t *d_mv1,
                                                        int n_blocks, int interp_filter_x, int interp_filter_y, int mask_type,
                                                        uint8_t *d_mask_out) {
  if (!ctx || !ref0 || !ref1 || !pred || !ref0->base || !ref1->base || !pred->base || (n_blocks > 0 && (!d_blocks || !d_mv0 || !d_mv1)) ||
      n_blocks < 0 || ref0_frame < 0 || ref0_frame >= ref0->n_frames || ref1_frame < 0 || ref1_frame >= ref1->n_frames || pred_frame < 0 ||
      pred_frame >= pred->n_frames || !valid_block(bw, bh) || ref0->bit_depth != pred->bit_depth || ref1->bit_depth != pred->bit_depth ||
      ref0->width != ref1->width || ref0->height != ref1->height || interp_filter_x < 0 || interp_filter_x > 3 || interp_filter_y < 0 ||
      interp_filter_y > 3 || mask_type < 0 || mask_type > 1 || ref0->border < 8 || ref1->border < 8) {
    set_error("aomhip_build_diffwtd_compound_pred_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (pred->bit_depth == 8)
    return launch_compound_pred<uint8_t>(ctx, ref0, ref0_frame, ref1, ref1_frame, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n_blocks,
                                         interp_filter_x, interp_filter_y, 0, 0, 0, 0, nullptr, nullptr, 0, 0, 0, mask_type + 1, d_mask_out);
  return launch_compound_pred<uint16_t>(ctx, ref0, ref0_frame, ref1, ref1_frame, pred, pred_frame, bw, bh, d_blocks, d_mv0, d_mv1, n_blocks,
                                        interp_filter_x, interp_filter_y, 0, 0, 0, 0, nullptr, nullptr, 0, 0, 0, mask_type + 1, d_mask_out);
}

extern "C" int aomhip_blend_a64_1d_batch(aomhip_ctx *ctx, const aomhip_planes *pred, int pred_frame, const aomhip_planes *adjacent,
                                         int adjacent_frame, const aomhip_blend_item *d_items, int n_items, const uint8_t *d_masks) {
  if (!ctx || !pred || !adjacent || !pred->base || !adjacent->base || (n_items > 0 && (!d_items || !d_masks)) || n_items < 0 || pred_frame < 0 ||
      pred_frame >= pred->n_frames || adjacent_frame < 0 || adjacent_frame >= adjacent->n_frames || pred->bit_depth != adjacent->bit_depth ||
      pred->width != adjacent->width || pred->height != adjacent->height) {
    set_error("aomhip_blend_a64_1d_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_items == 0) return AOMHIP_OK;
  const size_t esz = pred->bit_depth == 8 ? 1 : 2;
  char *d = static_cast<char *>(pred->base) + ((size_t)pred_frame * pred->frame_stride + (size_t)pred->border * pred->stride + pred->border) * esz;
  const char *s = static_cast<const char *>(adjacent->base) +
                  ((size_t)adjacent_frame * adjacent->frame_stride + (size_t)adjacent->border * adjacent->stride + adjacent->border) * esz;
  const dim3 grid((n_items + 3) / 4), block(256);
  if (esz == 1)
    hipLaunchKernelGGL(blend_1d_kernel<uint8_t>, grid, block, 0, ctx->stream, reinterpret_cast<uint8_t *>(d), pred->stride,
                       reinterpret_cast<const uint8_t *>(s), adjacent->stride, d_items, n_items, d_masks);
  else
    hipLaunchKernelGGL(blend_1d_kernel<uint16_t>, grid, block, 0, ctx->stream, reinterpret_cast<uint16_t *>(d), pred->stride,
                       reinterpret_cast<const uint16_t *>(s), adjacent->stride, d_items, n_items, d_masks);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_build_pred_fullpel(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame,
                                         const aomhip_planes *pred, int pred_frame, int bw, int bh,
                                         const aomhip_search_block *d_blocks, const int16_t *d_fullpel_mv, int n_blocks) {
  if (!ctx || !ref || !pred || !ref->base || !pred->base || !d_blocks || !d_fullpel_mv || n_blocks < 0 || ref_frame < 0 ||
      ref_frame >= ref->n_frames || pred_frame < 0 || pred_frame >= pred->n_frames || !valid_block(bw, bh) ||
      ref->bit_depth != pred->bit_depth) {
    set_error("aomhip_build_pred_fullpel: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const size_t esz = pred->bit_depth == 8 ? 1 : 2;
  char *d = static_cast<char *>(pred->base) +
            ((size_t)pred_frame * pred->frame_stride + (size_t)pred->border * pred->stride + pred->border) * esz;
  const dim3 grid((n_blocks + 3) / 4), block(256);
  if (esz == 1)
    hipLaunchKernelGGL(pred_copy_kernel<uint8_t>, grid, block, 0, ctx->stream, view_of<uint8_t>(*ref), ref_frame,
                       reinterpret_cast<uint8_t *>(d), pred->stride, d_blocks, d_fullpel_mv, n_blocks, bw, bh);
  else
    hipLaunchKernelGGL(pred_copy_kernel<uint16_t>, grid, block, 0, ctx->stream, view_of<uint16_t>(*ref), ref_frame,
                       reinterpret_cast<uint16_t *>(d), pred->stride, d_blocks, d_fullpel_mv, n_blocks, bw, bh);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
