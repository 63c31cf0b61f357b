// The temporal filter after its motion search, on the device: aomhip_tf_apply_frames.
// Reference (av1/encoder/temporal_filter.c), per 32x32 block and window frame in av1_tf_do_filtering_row's order (:849-905):
//   tf_build_predictor (:331-392): the plane's four sub-blocks through av1_enc_build_one_inter_predictor with MULTITAP_SHARP2 --
//       av1_[highbd_]convolve_2d_facade -> copy / x_sr / y_sr / 2d_sr with the 12-tap set (av1/common/convolve.c:76-174,495-515,
//       569-668; get_conv_params rounding, convolve.h:63-100; position arithmetic init_subpel_params, reconinter.h:130-165 -- WITHOUT its
//       clamp of the block position into the frame (:155-158): the planes' borders are replicated, so a read outside the frame returns
//       the pixel the clamp would have selected; the precondition is stated at aomhip_tf_apply_frames),
//   tf_apply_temporal_filter_self (:407-442) for the frame to filter itself,
//   av1_apply_temporal_filter_c (:557-712; rtcd proto av1/common/av1_rtcd_defs.pl:405-406): compute_square_diff, the 5x5 window sums
//       with coordinates clamped to the block, compute_luma_sq_error_sum for the chroma planes, the weight
//       (int)(exp(-min(scaled_error, 7)) * TF_WEIGHT_SCALE) in double precision, accum += weight * pred, count += weight,
//   tf_normalize_filtered_frame (:740-775): (accum + count / 2) / count, and the optional FRAME_DIFF sums (:892-904).
// One workgroup owns one block for the whole window: accum / count live in registers, predictor, squared differences and the separable
// filter's intermediate rows in LDS, so the search results (MVs, errors) and the window's pixels are all that is read and the
// filtered frame is all that is written.  Floating point: the reference's expression order is kept operation by operation (no
// contraction: every product and sum is an explicit rounded operation); decay_factor (pow / log of the parameters) comes from the
// host's libm like the reference's; the only library call on the device is exp(), so a weight can differ from the reference's by one
// unit where exp(x) * 1000 lies within an ulp of an integer (tests/test_gpu_tf_apply.py counts those: none on its inputs).
#include <cmath>

#include "common.h"

namespace aomhip {
namespace {

__constant__ int16_t k_interp12[16][12] = {
#include "interp12_table.inc"
};

constexpr int kMb = 32, kThreads = 256, kTaps = 12, kFo = 5;
constexpr int kRegW = 16 + kTaps;       // source columns staged per sub-block row (16 + 11, padded to 28)
constexpr int kRegH = 16 + kTaps - 1;   // source rows staged per sub-block (27)

struct TfApplyArgs {
  const void *frames[3];   // base of each component's plane ring (element type T)
  void *out[3];
  int64_t frame_stride[3], out_frame_stride[3];
  int stride[3], out_stride[3], border[3], out_border[3];
  int rows_alloc[3];       // rows of a frame's allocation (aligned height + 2 borders): predictor reads are clamped into it
  int n_frames, filter_frame, out_frame, num_planes, ss_x, ss_y, bd, mb_cols, n_blocks, frame_w, frame_h;
  unsigned present_mask;   // bit f: window frame f exists
  double decay_factor[3], inv_factor, weight_factor, distance_threshold;
  const int16_t *mvs;      // [(f * n_blocks + b) * 8]
  const int32_t *mses;     // [(f * n_blocks + b) * 4]
  long long *diff;         // FRAME_DIFF { sum, sse } or nullptr
};

__device__ __forceinline__ int rpot(int v, int n) { return (v + ((1 << n) >> 1)) >> n; }

template <typename T>
__global__ __launch_bounds__(kThreads) void tf_apply_kernel(TfApplyArgs a) {
  __shared__ uint16_t s_src[4][kRegH][kRegW];  // the four sub-blocks' source regions
  __shared__ int16_t s_im[4][kRegH][16];        // 2-D case: horizontally filtered rows
  __shared__ uint16_t s_pred[kMb * kMb];
  __shared__ uint32_t s_sq[kMb * kMb];          // square_diff of the current plane
  __shared__ uint32_t s_lsum[kMb * kMb];        // luma_sse_sum (chroma planes)
  __shared__ uint32_t s_hs[kMb * kMb];          // the window sum's row pass
  __shared__ uint16_t s_self3[kMb * kMb + 2 * (kMb * kMb)];   // the frame-to-filter's own block of every plane (chroma at most 32 x 32 each)
  __shared__ unsigned long long s_red[kThreads / 64];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  if (b >= a.n_blocks) return;
  const int mb_row = b / a.mb_cols, mb_col = b % a.mb_cols;
  const int tbd = sizeof(T) == 1 ? 8 : a.bd;
  const int pix_max = (1 << tbd) - 1;
  int round_0 = 3, round_1 = 11;
  if (sizeof(T) == 2 && a.bd + 7 - round_0 + 2 > 16) {
    const int extra = a.bd + 7 - round_0 + 2 - 16;
    round_0 += extra;
    round_1 -= extra;
  }
  // per-thread accumulators: pixel tid + 256 * j of each plane (luma 4; chroma up to 4 each)
  uint32_t accum[3][4];
  uint32_t count[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < 4; ++j) accum[p][j] = count[p][j] = 0;

  // the frame-to-filter's own pixels of every plane's block: the same for every window frame, staged once
  for (int p = 0; p < a.num_planes; ++p) {
    const int sx = p ? a.ss_x : 0, sy = p ? a.ss_y : 0;
    const int lw = 5 - sx, w = 1 << lw, npix = (kMb >> sy) << lw;
    const int plane_y = (kMb * mb_row) >> sy, plane_x = (kMb * mb_col) >> sx;
    const T *self_org = static_cast<const T *>(a.frames[p]) + (int64_t)a.filter_frame * a.frame_stride[p] + (int64_t)a.border[p] * a.stride[p] + a.border[p];
    for (int i = tid; i < npix; i += kThreads)
      s_self3[p * (kMb * kMb) + i] = (uint16_t)self_org[(int64_t)(plane_y + (i >> lw)) * a.stride[p] + plane_x + (i & (w - 1))];
  }
  __syncthreads();

  // Barriers per (frame, plane): after the staging, after the horizontal stage (only when some sub-block needs it), after predictor +
  // squared difference, after the window sum's row pass.  Nothing an early phase writes is read by the previous iteration's last phase
  // (that one reads s_pred, s_hs, s_lsum: rewritten two or more barriers later), so no barrier separates two iterations.
  for (int f = 0; f < a.n_frames; ++f) {
    if (!((a.present_mask >> f) & 1u)) continue;
    const bool self = f == a.filter_frame;
    const int16_t *mvp = a.mvs + ((int64_t)f * a.n_blocks + b) * 8;
    const int32_t *msep = a.mses + ((int64_t)f * a.n_blocks + b) * 4;
    int mv_row[4], mv_col[4];
    double d_factor[4], block_error[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      mv_row[s] = self ? 0 : mvp[2 * s];
      mv_col[s] = self ? 0 : mvp[2 * s + 1];
      // sqrt(pow(row, 2) + pow(col, 2)) / distance_threshold, at least 1 (:606-614)
      const double r = (double)mv_row[s], c = (double)mv_col[s];
      const double distance = __dsqrt_rn(__dadd_rn(__dmul_rn(r, r), __dmul_rn(c, c)));
      const double df = __ddiv_rn(distance, a.distance_threshold);
      d_factor[s] = df > 1.0 ? df : 1.0;
      block_error[s] = self ? 0.0 : (double)msep[s];
    }
    // (block constants of the weight: the same operation on the same operands as the reference's per-pixel expression, done once)
    double be_inv[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) be_inv[s] = __dmul_rn(block_error[s], a.inv_factor);
    for (int p = 0; p < a.num_planes; ++p) {
      const int sx = p ? a.ss_x : 0, sy = p ? a.ss_y : 0;
      // every extent is a power of two: rows / columns by shift and mask (h, w runtime values made every `/ w`, `% reg_w` ... a 20-30
      // instruction division sequence -- most of this kernel's instructions, profiles/r05_tf_apply.md)
      const int lw = 5 - sx, lh = 5 - sy, h = 1 << lh, w = 1 << lw, npix = h * w;
      const int sub_h = h >> 1, sub_w = w >> 1, lsw = lw - 1;
      const int plane_y = (kMb * mb_row) >> sy, plane_x = (kMb * mb_col) >> sx;
      const uint16_t *s_self = s_self3 + p * (kMb * kMb);
      if (self) {  // tf_apply_temporal_filter_self: weight TF_WEIGHT_SCALE everywhere
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = tid + kThreads * j;
          if (i < npix) { accum[p][j] += 1000u * s_self[i]; count[p][j] += 1000u; }
        }
        continue;
      }
      // ---- tf_build_predictor for this plane: stage each sub-block's source region, then the separable 12-tap filter
      const T *ref_org = static_cast<const T *>(a.frames[p]) + (int64_t)f * a.frame_stride[p] + (int64_t)a.border[p] * a.stride[p] + a.border[p];
      int pos_x[4], pos_y[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        pos_x[s] = ((plane_x + (s & 1) * sub_w) << 4) + mv_col[s] * (1 << (1 - sx));
        pos_y[s] = ((plane_y + (s >> 1) * sub_h) << 4) + mv_row[s] * (1 << (1 - sy));
      }
      const int reg_h = sub_h + kTaps - 1, reg_w = sub_w + kTaps - 1;
      const int ymin = -a.border[p], ymax = a.rows_alloc[p] - a.border[p] - 1, xmin = -a.border[p], xmax = a.stride[p] - a.border[p] - 1;
      {  // a pass = 8 rows x 32 columns of one sub-block's region (reg_w <= 27)
        const int c = tid & 31, r0 = tid >> 5;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int by0 = (pos_y[s] >> 4) - kFo, bx0 = (pos_x[s] >> 4) - kFo;
          const int x = min(max(bx0 + c, xmin), xmax);   // (memory safety only: MVs inside the mv limits never get here)
          for (int r = r0; r < reg_h; r += 8) {
            const int y = min(max(by0 + r, ymin), ymax);
            if (c < reg_w) s_src[s][r][c] = (uint16_t)ref_org[(int64_t)y * a.stride[p] + x];
          }
        }
      }
      __syncthreads();
      // horizontal stage of the 2-D case (both fractions non-zero): im rows -5 .. sub_h + 5
      bool any_2d = false;
#pragma unroll
      for (int s = 0; s < 4; ++s) any_2d = any_2d || ((pos_x[s] & 15) && (pos_y[s] & 15));
      if (any_2d) {   // (uniform over the workgroup: the barrier below is taken by all or by none)
        const int c = tid & (sub_w - 1), r0 = tid >> lsw, rpp = kThreads >> lsw;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int fxq = pos_x[s] & 15, fyq = pos_y[s] & 15;
          if (fxq && fyq) {   // (uniform over the workgroup)
            int tap[kTaps];
#pragma unroll
            for (int k = 0; k < kTaps; ++k) tap[k] = k_interp12[fxq][k];
            for (int r = r0; r < reg_h; r += rpp) {
              int sum = 1 << (tbd + 6);
#pragma unroll
              for (int k = 0; k < kTaps; ++k) sum += tap[k] * (int)s_src[s][r][c + k];
              s_im[s][r][c] = (int16_t)rpot(sum, round_0);
            }
          }
        }
        __syncthreads();
      }
      if (p == 1) {  // compute_luma_sq_error_sum: from the LUMA plane's square_diff (still in s_sq), once for both chroma planes
        for (int q = tid; q < npix; q += kThreads) {
          const int i = q >> lw, j = q & (w - 1);
          uint32_t t = 0;
          for (int ii = 0; ii < (1 << sy); ++ii)
            for (int jj = 0; jj < (1 << sx); ++jj) t += s_sq[(((i << sy) + ii) << (lw + sx)) + (j << sx) + jj];
          s_lsum[q] = t;
        }
        __syncthreads();   // (s_sq is overwritten below)
      }
      for (int q = tid; q < npix; q += kThreads) {
        const int i = q >> lw, j = q & (w - 1);
        const int s = (i >= sub_h) * 2 + (j >= sub_w), r = i & (sub_h - 1), c = j & (sub_w - 1);
        const int fxq = pos_x[s] & 15, fyq = pos_y[s] & 15;
        int v;
        if (!fxq && !fyq) {
          v = s_src[s][r + kFo][c + kFo];
        } else if (!fyq) {  // convolve_x_sr
          int res = 0;
#pragma unroll
          for (int k = 0; k < kTaps; ++k) res += k_interp12[fxq][k] * (int)s_src[s][r + kFo][c + k];
          res = rpot(res, round_0);
          v = rpot(res, 7 - round_0);
        } else if (!fxq) {  // convolve_y_sr
          int res = 0;
#pragma unroll
          for (int k = 0; k < kTaps; ++k) res += k_interp12[fyq][k] * (int)s_src[s][r + k][c + kFo];
          v = rpot(res, 7);
        } else {  // convolve_2d_sr, vertical stage
          const int offset_bits = tbd + 14 - round_0, bits = 14 - round_0 - round_1;
          int sum = 1 << offset_bits;
#pragma unroll
          for (int k = 0; k < kTaps; ++k) sum += k_interp12[fyq][k] * (int)s_im[s][r + k][c];
          int res = rpot(sum, round_1) - ((1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1)));
          if (sizeof(T) == 1) res = (int16_t)res;
          v = rpot(res, bits);
        }
        v = min(max(v, 0), pix_max);
        s_pred[q] = (uint16_t)v;
        // ---- av1_apply_temporal_filter_c for this plane: compute_square_diff of the pixel this lane just predicted
        const int d = (int)s_self[q] - v;
        s_sq[q] = (uint32_t)__mul24(d, d);
      }
      __syncthreads();
      // the 5 x 5 window sum (coordinates clamped to the block) as a row pass and a column pass: sum over rows of (sum over columns), the
      // same 25 terms; a term is < 2^24 (12-bit difference squared), 25 of them + the 4 luma terms of a chroma pixel < 2^29: 32-bit sums
      for (int q = tid; q < npix; q += kThreads) {
        const int i = q >> lw, j = q & (w - 1);
        const uint32_t *row = s_sq + (i << lw);
        s_hs[q] = row[max(j - 2, 0)] + row[max(j - 1, 0)] + row[j] + row[min(j + 1, w - 1)] + row[min(j + 2, w - 1)];
      }
      __syncthreads();
      const double inv_num_ref_pixels = __ddiv_rn(1.0, (double)(25 + (p ? (1 << (sx + sy)) : 0)));
      const int err_shift = a.bd > 8 ? (a.bd - 8) * 2 : 0;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        const int q = tid + kThreads * jq;
        if (q < npix) {
          const int i = q >> lw, j = q & (w - 1);
          uint32_t sum = s_hs[(max(i - 2, 0) << lw) + j] + s_hs[(max(i - 1, 0) << lw) + j] + s_hs[q] + s_hs[(min(i + 1, h - 1) << lw) + j] +
                         s_hs[(min(i + 2, h - 1) << lw) + j];
          if (p) sum += s_lsum[q];
          sum >>= err_shift;
          const double window_error = __dmul_rn((double)sum, inv_num_ref_pixels);
          const int sidx = (i >= sub_h) * 2 + (j >= sub_w);
          const double combined = __dadd_rn(__dmul_rn(a.weight_factor, window_error), be_inv[sidx]);
          double scaled = __dmul_rn(__dmul_rn(combined, d_factor[sidx]), a.decay_factor[p]);
          scaled = scaled < 7.0 ? scaled : 7.0;
          const int weight = (int)__dmul_rn(exp(-scaled), 1000.0);
          accum[p][jq] += (uint32_t)(weight * (int)s_pred[q]);
          count[p][jq] = (count[p][jq] + (uint32_t)weight) & 0xFFFFu;  // (uint16_t count, :702)
        }
      }
    }
  }
  // ---- tf_normalize_filtered_frame + FRAME_DIFF
  unsigned long long sse = 0;
  for (int p = 0; p < a.num_planes; ++p) {
    const int sx = p ? a.ss_x : 0, sy = p ? a.ss_y : 0;
    const int lw = 5 - sx, h = kMb >> sy, w = kMb >> sx, npix = h * w;
    const int plane_y = (kMb * mb_row) >> sy, plane_x = (kMb * mb_col) >> sx;
    T *out_org = static_cast<T *>(a.out[p]) + (int64_t)a.out_frame * a.out_frame_stride[p] + (int64_t)a.out_border[p] * a.out_stride[p] + a.out_border[p];
    const T *self_org = static_cast<const T *>(a.frames[p]) + (int64_t)a.filter_frame * a.frame_stride[p] + (int64_t)a.border[p] * a.stride[p] + a.border[p];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + kThreads * j;
      if (q < npix) {
        const uint32_t c = count[p][j];
        const uint32_t v = c ? (accum[p][j] + (c >> 1)) / c : 0u;
        const int64_t o = (int64_t)(plane_y + (q >> lw)) * a.out_stride[p] + plane_x + (q & (w - 1));
        out_org[o] = (T)v;
        if (p == 0 && a.diff) {
          const int d = (int)self_org[(int64_t)(plane_y + (q >> lw)) * a.stride[p] + plane_x + (q & (w - 1))] - (int)v;
          sse += (unsigned)__mul24(d, d);
        }
      }
    }
  }
  if (a.diff) {  // fn_ptr[BLOCK_32X32].vf(source, filtered): its sse (highbd: the _10 / _12 variance forms round it down to 8-bit scale)
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sse += __shfl_xor(sse, m, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = sse;
    __syncthreads();
    if (tid == 0) {
      unsigned long long t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
      if (sizeof(T) == 2 && a.bd == 10) t = (t + 8) >> 4;
      if (sizeof(T) == 2 && a.bd == 12) t = (t + 128) >> 8;
      const long long s32 = (long long)(unsigned)t;  // (unsigned int sse, :897)
      atomicAdd(reinterpret_cast<unsigned long long *>(a.diff), (unsigned long long)s32);
      atomicAdd(reinterpret_cast<unsigned long long *>(a.diff + 1), (unsigned long long)(s32 * s32));
    }
  }
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_tf_apply_frames(aomhip_ctx *ctx, const aomhip_planes *frames_y, const aomhip_planes *frames_u, const aomhip_planes *frames_v,
                                      int filter_frame, const uint8_t *frame_present, const aomhip_tf_apply_params *p, int n_blocks,
                                      const int16_t *d_subblock_mvs, const int32_t *d_subblock_mses, const aomhip_planes *out_y,
                                      const aomhip_planes *out_u, const aomhip_planes *out_v, int out_frame, int64_t *d_frame_diff) {
  if (!ctx || !frames_y || !frames_y->base || !p || !out_y || !out_y->base || !d_subblock_mvs || !d_subblock_mses || filter_frame < 0 ||
      filter_frame >= frames_y->n_frames || frames_y->n_frames > 32 || out_frame < 0 || out_frame >= out_y->n_frames ||
      (p->num_planes != 1 && p->num_planes != 3) || p->ss_x < 0 || p->ss_x > 1 || p->ss_y < 0 || p->ss_y > 1) {
    set_error("aomhip_tf_apply_frames: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const aomhip_planes *fr[3] = { frames_y, frames_u, frames_v }, *out[3] = { out_y, out_u, out_v };
  const int mb_rows = (frames_y->height + 31) / 32, mb_cols = (frames_y->width + 31) / 32;
  if (n_blocks != mb_rows * mb_cols) {
    set_error("aomhip_tf_apply_frames: n_blocks %d != %d x %d blocks of 32x32", n_blocks, mb_rows, mb_cols);
    return AOMHIP_ERR_INVALID;
  }
  TfApplyArgs a;
  memset(&a, 0, sizeof(a));
  for (int k = 0; k < p->num_planes; ++k) {
    const int sx = k ? p->ss_x : 0, sy = k ? p->ss_y : 0;
    if (!fr[k] || !out[k] || !fr[k]->base || !out[k]->base || fr[k]->bit_depth != frames_y->bit_depth || out[k]->bit_depth != frames_y->bit_depth ||
        fr[k]->n_frames != frames_y->n_frames || out_frame >= out[k]->n_frames || fr[k]->width != (frames_y->width + sx) >> sx ||
        fr[k]->height != (frames_y->height + sy) >> sy || out[k]->width != fr[k]->width || out[k]->height != fr[k]->height) {
      set_error("aomhip_tf_apply_frames: plane %d does not match the luma ring (geometry, bit depth, frames)", k);
      return AOMHIP_ERR_INVALID;
    }
    // the blocks of the last row / column reach to the next multiple of 32 (>> subsampling): the planes must hold them
    const int need_w = (mb_cols * 32) >> sx, need_h = (mb_rows * 32) >> sy;
    if (fr[k]->width + fr[k]->border < need_w || fr[k]->height + fr[k]->border < need_h || out[k]->width + out[k]->border < need_w ||
        out[k]->height + out[k]->border < need_h) {
      set_error("aomhip_tf_apply_frames: plane %d's border does not cover the 32-aligned frame", k);
      return AOMHIP_ERR_INVALID;
    }
    a.frames[k] = fr[k]->base; a.out[k] = out[k]->base;
    a.frame_stride[k] = fr[k]->frame_stride; a.out_frame_stride[k] = out[k]->frame_stride;
    a.stride[k] = fr[k]->stride; a.out_stride[k] = out[k]->stride; a.border[k] = fr[k]->border; a.out_border[k] = out[k]->border;
    a.rows_alloc[k] = ((fr[k]->height + 7) & ~7) + 2 * fr[k]->border;
  }
  a.n_frames = frames_y->n_frames; a.filter_frame = filter_frame; a.out_frame = out_frame; a.num_planes = p->num_planes;
  a.ss_x = p->ss_x; a.ss_y = p->ss_y; a.bd = frames_y->bit_depth; a.mb_cols = mb_cols; a.n_blocks = n_blocks;
  a.frame_w = frames_y->width; a.frame_h = frames_y->height;
  for (int f = 0; f < a.n_frames; ++f)
    if (!frame_present || frame_present[f]) a.present_mask |= 1u << f;
  // the per-call factors of av1_apply_temporal_filter_c (:571-603), with the host's libm like the reference
  a.inv_factor = 1.0 / ((5 + 1) * 20);
  a.weight_factor = (double)5 * a.inv_factor;
  double q_decay = pow((double)p->q_factor / 20, 2);
  q_decay = q_decay < 1e-5 ? 1e-5 : q_decay > 1 ? 1 : q_decay;
  if (p->q_factor >= 128) q_decay = 0.5 * pow((double)p->q_factor / 64, 2);
  double s_decay = pow((double)p->filter_strength / 4, 2);
  s_decay = s_decay < 1e-5 ? 1e-5 : s_decay > 1 ? 1 : s_decay;
  for (int k = 0; k < p->num_planes; ++k) {
    const double n_decay = 0.5 + log(2 * p->noise_levels[k] + 5.0);
    a.decay_factor[k] = 1 / (n_decay * q_decay * s_decay);
  }
  const int min_frame_size = frames_y->height < frames_y->width ? frames_y->height : frames_y->width;
  a.distance_threshold = min_frame_size * 0.1;
  if (!(a.distance_threshold > 1)) a.distance_threshold = 1;
  a.mvs = d_subblock_mvs; a.mses = d_subblock_mses;
  a.diff = reinterpret_cast<long long *>(d_frame_diff);
  AOMHIP_TRY(hipSetDevice(ctx->device));
  if (d_frame_diff) AOMHIP_TRY(hipMemsetAsync(d_frame_diff, 0, 16, ctx->stream));
  if (frames_y->bit_depth == 8)
    hipLaunchKernelGGL(tf_apply_kernel<uint8_t>, dim3((unsigned)n_blocks), dim3(kThreads), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL(tf_apply_kernel<uint16_t>, dim3((unsigned)n_blocks), dim3(kThreads), 0, ctx->stream, a);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
