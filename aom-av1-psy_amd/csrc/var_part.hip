// The leaves of the variance-based partitioning's tree (av1_choose_var_based_partitioning, av1/encoder/var_based_part.c): what fill_variance_8x8avg,
// compute_minmax_8x8 and fill_variance_4x4avg (:255-430) compute per 16 x 16 / 8 x 8 block with aom_avg_8x8 / aom_avg_4x4 / aom_minmax_8x8
// (aom_dsp/avg.c:18-100), for a whole plane in one launch: per 8 x 8 block the difference of the source's and the prediction's rounded averages
// (its square is the leaf's sum_square_error), per 16 x 16 block the spread of its 8 x 8 blocks' (max - min) of |source - prediction|, per 4 x 4
// block of a key frame the source's rounded average - 128.  The tree's sums above the leaves and the threshold tests stay with the host: they are
// a few hundred additions per superblock on these numbers.  8 x 8: a lane per (block, row) -- 8 pixels each of source and prediction --, three
// cross-lane steps per block, two more for the 16 x 16 spread; both planes are read exactly once.
#include "common.h"

namespace aomhip {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void vbp_8x8_kernel(PlaneView<T> src, int src_frame, PlaneView<T> ref, int ref_frame, int vis_w, int vis_h, int n16x, int n16,
                                                      int16_t *__restrict__ sum8, int sum_stride, int32_t *__restrict__ minmax16, int minmax_stride) {
  const int lane = threadIdx.x & 63;
  const int b16 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const int k = (lane >> 3) & 3, row = lane & 7;
  const int by16 = b16 / n16x, bx16 = b16 - by16 * n16x;
  const int x8 = 16 * bx16 + ((k & 1) << 3), y8 = 16 * by16 + ((k >> 1) << 3);
  const bool valid = b16 < n16 && x8 < vis_w && y8 < vis_h;
  int ssum = 0, dsum = 0, mn = 255, mx = 0;   // (*min = 255 also above 8 bits: avg.c:90)
  if (valid) {
    const T *s = src.origin + (int64_t)src_frame * src.frame_stride + (int64_t)(y8 + row) * src.stride + x8;
    const T *d = ref.origin + (int64_t)ref_frame * ref.frame_stride + (int64_t)(y8 + row) * ref.stride + x8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int a = (int)s[j], b = (int)d[j], e = abs(a - b);
      ssum += a;
      dsum += b;
      mn = min(mn, e);
      mx = max(mx, e);
    }
  }
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) {
    ssum += __shfl_xor(ssum, m, 64);
    dsum += __shfl_xor(dsum, m, 64);
    mn = min(mn, __shfl_xor(mn, m, 64));
    mx = max(mx, __shfl_xor(mx, m, 64));
  }
  // (every leaf of every launched 16 x 16 block is written -- fill_variance_8x8avg gives a leaf outside the visible area sum = sse = 0 --, so a host
  // tree builder may walk all four leaves of an edge block without a pre-zeroed array)
  if (row == 0 && b16 < n16)
    sum8[(int64_t)(y8 >> 3) * sum_stride + (x8 >> 3)] = valid ? (int16_t)(((ssum + 32) >> 6) - ((dsum + 32) >> 6)) : (int16_t)0;
  if (minmax16) {
    int hi = valid ? mx - mn : 0, lo = valid ? min(mx - mn, 255) : 255;   // minmax_max starts at 0, minmax_min at 255 (also above 8 bits, where a block's spread can pass it)
#pragma unroll
    for (int m = 8; m < 32; m <<= 1) {
      hi = max(hi, __shfl_xor(hi, m, 64));
      lo = min(lo, __shfl_xor(lo, m, 64));
    }
    if ((lane & 31) == 0 && b16 < n16) minmax16[(int64_t)by16 * minmax_stride + bx16] = hi - lo;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void vbp_4x4_kernel(PlaneView<T> src, int src_frame, int vis_w, int vis_h, int border_offset, int n4x, int n4,
                                                      int16_t *__restrict__ sum4, int sum_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int y = i / n4x, x = i - y * n4x;
  int v = 0;
  if (4 * x < vis_w - border_offset && 4 * y < vis_h - border_offset) {
    const T *s = src.origin + (int64_t)src_frame * src.frame_stride + (int64_t)(4 * y) * src.stride + 4 * x;
    int sum = 0;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += (int)s[(int64_t)r * src.stride + c];
    v = ((sum + 8) >> 4) - 128;
  }
  sum4[(int64_t)y * sum_stride + x] = (int16_t)v;
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_vbp_8x8_stats_plane(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *ref, int ref_frame, int visible_width,
                                          int visible_height, int16_t *d_sum8x8, int sum_stride, int32_t *d_minmax16x16, int minmax_stride) {
  if (!ctx || !src || !ref || !src->base || !ref->base || src_frame < 0 || src_frame >= src->n_frames || ref_frame < 0 || ref_frame >= ref->n_frames ||
      src->bit_depth != ref->bit_depth || visible_width < 1 || visible_height < 1 || visible_width > src->width || visible_height > src->height ||
      visible_width > ref->width || visible_height > ref->height || src->border < 8 || ref->border < 8 || !d_sum8x8 || sum_stride < 2 * ((visible_width + 15) / 16) ||
      (d_minmax16x16 && minmax_stride < (visible_width + 15) / 16)) {
    set_error("aomhip_vbp_8x8_stats_plane: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  AOMHIP_TRY(hipSetDevice(ctx->device));
  const int n16x = (visible_width + 15) / 16, n16 = n16x * ((visible_height + 15) / 16);
  const dim3 grid((unsigned)((n16 + 7) / 8)), block(256);
  if (src->bit_depth == 8)
    hipLaunchKernelGGL(vbp_8x8_kernel<uint8_t>, grid, block, 0, ctx->stream, view_of<uint8_t>(*src), src_frame, view_of<uint8_t>(*ref), ref_frame, visible_width,
                       visible_height, n16x, n16, d_sum8x8, sum_stride, d_minmax16x16, minmax_stride);
  else
    hipLaunchKernelGGL(vbp_8x8_kernel<uint16_t>, grid, block, 0, ctx->stream, view_of<uint16_t>(*src), src_frame, view_of<uint16_t>(*ref), ref_frame, visible_width,
                       visible_height, n16x, n16, d_sum8x8, sum_stride, d_minmax16x16, minmax_stride);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_vbp_4x4_avg_plane(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, int visible_width, int visible_height, int border_offset_4x4,
                                        int16_t *d_sum4x4, int sum_stride) {
  if (!ctx || !src || !src->base || src_frame < 0 || src_frame >= src->n_frames || visible_width < 1 || visible_height < 1 || visible_width > src->width ||
      visible_height > src->height || src->border < 4 || border_offset_4x4 < 0 || !d_sum4x4 || sum_stride < (visible_width + 3) / 4) {
    set_error("aomhip_vbp_4x4_avg_plane: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  AOMHIP_TRY(hipSetDevice(ctx->device));
  const int n4x = (visible_width + 3) / 4, n4 = n4x * ((visible_height + 3) / 4);
  const dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
  if (src->bit_depth == 8)
    hipLaunchKernelGGL(vbp_4x4_kernel<uint8_t>, grid, block, 0, ctx->stream, view_of<uint8_t>(*src), src_frame, visible_width, visible_height, border_offset_4x4, n4x,
                       n4, d_sum4x4, sum_stride);
  else
    hipLaunchKernelGGL(vbp_4x4_kernel<uint16_t>, grid, block, 0, ctx->stream, view_of<uint16_t>(*src), src_frame, visible_width, visible_height, border_offset_4x4,
                       n4x, n4, d_sum4x4, sum_stride);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
