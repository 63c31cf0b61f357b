// The warped-motion predictor of a single reference: av1_warp_affine / av1_highbd_warp_affine (av1/common/warped_motion.c:264-393,538-675; AV1 spec
// 7.11.3.5) as av1_warp_plane runs it for a WARPED_CAUSAL block or a global-motion reference whose prediction is not a compound
// (conv_params->is_compound == 0).  One wavefront per 8 x 8 tile of a block:
//   horizontal pass  15 rows x 8 columns, one value per lane in two steps: the row is clamped to the frame (top / bottom), every sample
//                    column to [0, width - 1] (left / right) -- the reference reads no border, neither does this -- the 8 taps of
//                    Warped_Filters[round(sx / 1024) + 64] with sx = sx4 + alpha * l + beta * k, rounded by round_0 (3; 5 at 12 bits:
//                    get_conv_params_no_round, plus the high-bit-depth form's extra bits) into the wavefront's 120 int32 of LDS;
//   vertical pass    one output pixel per lane: the 8 taps of the kernel of sy = sy4 + gamma * l + delta * k on eight rows of that block,
//                    rounded, offsets removed, clipped.
// The tile's centre, its integer / fractional source position and the two phases' starting values are scalars (64-bit affine products
// included).  Bytes per tile: ~15 x 15 reference pixels in, 64 out -- a latency-bound gather; nothing is staged twice.
// Compound (conv_params->is_compound): MODE 1 = the first reference, its vertical sums rounded by round_1 = COMPOUND_ROUND1_BITS into the
// CONV_BUF (uint16, addressed like the plane); MODE 2 = the second reference blended with it -- (a + b) >> 1 or the distance weights
// (a fwd + b bck) >> DIST_PRECISION_BITS -- offsets removed, rounded by round_bits, clipped into the prediction.
#include "common.h"

namespace aomhip {
namespace {

__device__ const int16_t kWarpedFilter[193][8] __attribute__((aligned(16))) = {
#include "warp_table.inc"
};
#include "warp_error_table.inc"

struct WarpCompound {
  uint16_t *conv;      // CONV_BUF: element (row, col) of the plane at conv[row * conv_stride + col]
  int conv_stride;
  int use_dist_wtd, fwd_offset, bck_offset;
};

// One 8 x 8 tile at (i, j) of a block (p_col, p_row, p_width, p_height) through the model: both passes, returning this lane's vertical sum rounded by
// reduce_bits_vert (offsets still in) and whether the lane's pixel (i + k + 4, j + l + 4), k = lane / 8 - 4, l = lane % 8 - 4, lies inside the block.
// ROUND1: the vertical rounding (14 - reduce_bits_horiz for a single prediction, COMPOUND_ROUND1_BITS for the CONV_BUF).
struct WarpModel {
  int32_t mat[6];
  int alpha, beta, gamma, delta;
};
template <typename T>
__device__ __forceinline__ bool warp_tile(const T *__restrict__ rp, int ref_stride, int width, int height, const WarpModel &m, int ssx, int ssy, int bd, bool compound,
                                          int i, int j, int p_col, int p_row, int p_width, int p_height, int32_t *tmp, int lane, int32_t *out) {
  const bool hbd = sizeof(T) == 2;
  const int round_0 = bd == 12 ? 5 : 3;   // ROUND0_BITS (+ 2 at 12 bits): get_conv_params_no_round (av1/common/convolve.h)
  const int extra = hbd ? max(bd + 7 - round_0 - 14, 0) : 0;
  const int reduce_bits_horiz = round_0 + extra, reduce_bits_vert = compound ? 7 : 14 - reduce_bits_horiz;   // COMPOUND_ROUND1_BITS
  const int offset_bits_horiz = bd + 6, offset_bits_vert = bd + 14 - reduce_bits_horiz;
  // the centre of the tile in luma coordinates, through the model, back to this plane's coordinates
  const int32_t src_x = (j + 4) << ssx, src_y = (i + 4) << ssy;
  const int64_t dst_x = (int64_t)m.mat[2] * src_x + (int64_t)m.mat[3] * src_y + (int64_t)m.mat[0];
  const int64_t dst_y = (int64_t)m.mat[4] * src_x + (int64_t)m.mat[5] * src_y + (int64_t)m.mat[1];
  const int64_t x4 = dst_x >> ssx, y4 = dst_y >> ssy;
  const int ix4 = (int)(x4 >> 16), iy4 = (int)(y4 >> 16);   // WARPEDMODEL_PREC_BITS
  int sx4 = (int)(x4 & 0xffff), sy4 = (int)(y4 & 0xffff);
  sx4 += m.alpha * (-4) + m.beta * (-4);
  sy4 += m.gamma * (-4) + m.delta * (-4);
  sx4 &= ~63;   // WARP_PARAM_REDUCE_BITS
  sy4 &= ~63;
  for (int t = lane; t < 15 * 8; t += 64) {   // horizontal filter: tmp[(k + 7) * 8 + (l + 4)], k = -7 .. 7, l = -4 .. 3
    const int k = (t >> 3) - 7, l = (t & 7) - 4;
    const int iy = min(max(iy4 + k, 0), height - 1);
    const int sx = sx4 + m.beta * (k + 4) + m.alpha * (l + 4);
    const int offs = ((sx + 512) >> 10) + 64;   // ROUND_POWER_OF_TWO(sx, WARPEDDIFF_PREC_BITS) + WARPEDPIXEL_PREC_SHIFTS
    const int4 cw = *reinterpret_cast<const int4 *>(kWarpedFilter[offs]);
    const int c[8] = { (int)(int16_t)(cw.x & 0xffff), cw.x >> 16, (int)(int16_t)(cw.y & 0xffff), cw.y >> 16,
                       (int)(int16_t)(cw.z & 0xffff), cw.z >> 16, (int)(int16_t)(cw.w & 0xffff), cw.w >> 16 };
    const int ix = ix4 + l - 3;
    const T *row = rp + (int64_t)iy * ref_stride;
    int32_t sum = 1 << offset_bits_horiz;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += (int)row[min(max(ix + q, 0), width - 1)] * c[q];
    tmp[t] = (sum + ((1 << reduce_bits_horiz) >> 1)) >> reduce_bits_horiz;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // vertical filter: lane -> (k, l) = (-4 .. 3, -4 .. 3)
  const int k = (lane >> 3) - 4, l = (lane & 7) - 4;
  const int kmax = min(4, p_row + p_height - i - 4), lmax = min(4, p_col + p_width - j - 4);
  const bool inside = k < kmax && l < lmax;
  if (inside) {
    const int sy = sy4 + m.delta * (k + 4) + m.gamma * (l + 4);
    const int offs = ((sy + 512) >> 10) + 64;
    const int4 cw = *reinterpret_cast<const int4 *>(kWarpedFilter[offs]);
    const int c[8] = { (int)(int16_t)(cw.x & 0xffff), cw.x >> 16, (int)(int16_t)(cw.y & 0xffff), cw.y >> 16,
                       (int)(int16_t)(cw.z & 0xffff), cw.z >> 16, (int)(int16_t)(cw.w & 0xffff), cw.w >> 16 };
    int32_t sum = 1 << offset_bits_vert;
#pragma unroll
    for (int q = 0; q < 8; ++q) sum += tmp[(k + q + 4) * 8 + (l + 4)] * c[q];
    *out = (sum + ((1 << reduce_bits_vert) >> 1)) >> reduce_bits_vert;
  }
  return inside;
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void warp_affine_kernel(PlaneView<T> ref, int ref_frame, int width, int height, T *__restrict__ pred_origin, int64_t pred_frame_off,
                                                          int pred_stride, int ssx, int ssy, int bd, const aomhip_warp_block *__restrict__ blocks, int n_blocks,
                                                          WarpCompound cm) {
  __shared__ int32_t tmp_all[4][15 * 8];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x;
  const aomhip_warp_block b = blocks[bi];
  const int p_col = __builtin_amdgcn_readfirstlane(b.p_col), p_row = __builtin_amdgcn_readfirstlane(b.p_row);
  const int p_width = __builtin_amdgcn_readfirstlane(b.p_width), p_height = __builtin_amdgcn_readfirstlane(b.p_height);
  const int tiles_x = (p_width + 7) >> 3, tiles_y = (p_height + 7) >> 3;
  const int tile = blockIdx.y * 4 + wave;
  if (tile >= tiles_x * tiles_y) return;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int i = p_row + 8 * ty, j = p_col + 8 * tx;
  WarpModel m;
#pragma unroll
  for (int k = 0; k < 6; ++k) m.mat[k] = __builtin_amdgcn_readfirstlane(b.mat[k]);
  m.alpha = __builtin_amdgcn_readfirstlane((int)b.alpha), m.beta = __builtin_amdgcn_readfirstlane((int)b.beta);
  m.gamma = __builtin_amdgcn_readfirstlane((int)b.gamma), m.delta = __builtin_amdgcn_readfirstlane((int)b.delta);
  const int round_0 = bd == 12 ? 5 : 3, round_1 = MODE ? 7 : 14 - round_0;
  [[maybe_unused]] const int round_bits = 14 - round_0 - round_1, offset_bits = bd + 14 - round_0;
  int32_t sum;
  if (!warp_tile<T>(ref.origin + (int64_t)ref_frame * ref.frame_stride, ref.stride, width, height, m, ssx, ssy, bd, MODE != 0, i, j, p_col, p_row, p_width, p_height,
                    tmp_all[wave], lane, &sum))
    return;
  const int k = (lane >> 3) - 4, l = (lane & 7) - 4;
  if constexpr (MODE == 1) {
    cm.conv[(int64_t)(i + k + 4) * cm.conv_stride + (j + l + 4)] = (uint16_t)sum;   // CONV_BUF_TYPE
  } else {
    int v;
    if constexpr (MODE == 2) {
      int t32 = (int)cm.conv[(int64_t)(i + k + 4) * cm.conv_stride + (j + l + 4)];
      t32 = cm.use_dist_wtd ? (t32 * cm.fwd_offset + sum * cm.bck_offset) >> 4 : (t32 + sum) >> 1;   // DIST_PRECISION_BITS
      t32 = t32 - (1 << (offset_bits - round_1)) - (1 << (offset_bits - round_1 - 1));
      v = (t32 + ((1 << round_bits) >> 1)) >> round_bits;
    } else {
      v = sum - (1 << (bd - 1)) - (1 << bd);
    }
    pred_origin[pred_frame_off + (int64_t)(i + k + 4) * pred_stride + (j + l + 4)] = (T)min(max(v, 0), (1 << bd) - 1);
  }
}

// The global-motion search's model error (av1_warp_error, av1/encoder/global_motion.c:128-224): for every candidate model, the prediction of every
// 32 x 32 tile (WARP_ERROR_BLOCK) whose segment-map entry is set against the frame, each pixel's difference through error_measure_lut
// (av1/common/warped_motion.h:42-146; above 8 bits interpolated between neighbouring entries, warped_motion.c:248-259).  The reference warps a tile
// into a scratch block and then measures it; here the pixel never leaves the lane.  One workgroup per (tile, model), a wavefront per row of 8 x 8
// blocks; a tile's total (< 2^28) goes to work memory and a second small kernel adds a model's tiles in 64 bits (one atomic per wavefront on the
// model's total instead measured 0.32 ms per model and 4K frame, all of it contention on the one address).
// WARP = false: av1_segmented_frame_error (warped_motion.c:400-460,687-760), the same metric on the reference as it is.
__device__ const int kErrorMeasureLut[512] = AOMHIP_ERROR_MEASURE_LUT;

template <typename T, bool WARP>
__global__ __launch_bounds__(256) void warp_error_kernel(PlaneView<T> ref, int ref_frame, int width, int height, PlaneView<T> cur, int cur_frame, int ssx, int ssy,
                                                         int bd, const aomhip_warp_model *__restrict__ models, int p_col, int p_row, int p_width, int p_height,
                                                         const uint8_t *__restrict__ segment_map, int segment_map_stride, int tiles_x,
                                                         int32_t *__restrict__ tile_err) {
  __shared__ int32_t tmp_all[4][15 * 8];
  __shared__ int32_t wave_err[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int i0 = p_row + 32 * ty, j0 = p_col + 32 * tx;
  int32_t *out = tile_err + (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (!segment_map[(i0 >> 5) * segment_map_stride + (j0 >> 5)]) {   // no inliers of the motion model in this tile
    if (threadIdx.x == 0) *out = 0;
    return;
  }
  const int warp_w = min(min(p_width, 32), p_col + p_width - j0), warp_h = min(min(p_height, 32), p_row + p_height - i0);
  WarpModel m;
  if constexpr (WARP) {
    const aomhip_warp_model b = models[blockIdx.y];
#pragma unroll
    for (int k = 0; k < 6; ++k) m.mat[k] = __builtin_amdgcn_readfirstlane(b.mat[k]);
    m.alpha = __builtin_amdgcn_readfirstlane((int)b.alpha), m.beta = __builtin_amdgcn_readfirstlane((int)b.beta);
    m.gamma = __builtin_amdgcn_readfirstlane((int)b.gamma), m.delta = __builtin_amdgcn_readfirstlane((int)b.delta);
  }
  const T *rp = ref.origin + (int64_t)ref_frame * ref.frame_stride, *cp = cur.origin + (int64_t)cur_frame * cur.frame_stride;
  const int k = (lane >> 3) - 4, l = (lane & 7) - 4;
  const int i = i0 + 8 * wave;
  const int b = bd - 8, bmask = (1 << b) - 1;
  int32_t acc = 0;
  for (int j = j0; j < j0 + warp_w && 8 * wave < warp_h; j += 8) {
    int v;
    bool inside;
    if constexpr (WARP) {
      int32_t sum = 0;
      inside = warp_tile<T>(rp, ref.stride, width, height, m, ssx, ssy, bd, false, i, j, j0, i0, warp_w, warp_h, tmp_all[wave], lane, &sum);
      v = min(max(sum - (1 << (bd - 1)) - (1 << bd), 0), (1 << bd) - 1);
      __builtin_amdgcn_wave_barrier();   // (the next tile's horizontal pass overwrites tmp)
    } else {
      inside = k + 4 < i0 + warp_h - i && l + 4 < j0 + warp_w - j;
      v = inside ? (int)rp[(int64_t)(i + k + 4) * ref.stride + (j + l + 4)] : 0;
    }
    if (inside) {
      const int e = abs((int)cp[(int64_t)(i + k + 4) * cur.stride + (j + l + 4)] - v);
      const int e1 = e >> b, e2 = e & bmask;
      acc += kErrorMeasureLut[255 + e1] * ((1 << b) - e2) + kErrorMeasureLut[256 + e1] * e2;   // (8 bits: e2 = 0, the entry itself)
    }
  }
  for (int q = 1; q < 64; q <<= 1) acc += __shfl_xor(acc, q, 64);
  if (lane == 0) wave_err[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = wave_err[0] + wave_err[1] + wave_err[2] + wave_err[3];   // < 2^28: 1024 pixels of at most 2^18 each
}

// the tiles' totals of one model -> its error (one workgroup per model; 64-bit integer sums, so the order does not matter)
__global__ __launch_bounds__(256) void warp_error_sum_kernel(const int32_t *__restrict__ tile_err, int n_tiles, int64_t *__restrict__ err) {
  __shared__ int64_t part[4];
  const int32_t *p = tile_err + (int64_t)blockIdx.x * n_tiles;
  int64_t acc = 0;
  for (int t = threadIdx.x; t < n_tiles; t += 256) acc += p[t];
  for (int q = 1; q < 64; q <<= 1) acc += __shfl_xor((unsigned long long)acc, q, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) err[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

static int warp_launch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred, int pred_frame, int subsampling_x, int subsampling_y,
                       const aomhip_warp_block *d_blocks, int n_blocks, int max_block_width, int max_block_height, int mode, WarpCompound cm, const char *who) {
  if (!ctx || !ref || !ref->base || n_blocks < 0 || (n_blocks > 0 && !d_blocks) || ref_frame < 0 || ref_frame >= ref->n_frames ||
      (mode != 1 && (!pred || !pred->base || pred_frame < 0 || pred_frame >= pred->n_frames || (ref->bit_depth == 8) != (pred->bit_depth == 8))) ||
      (subsampling_x | subsampling_y) < 0 || subsampling_x > 1 || subsampling_y > 1 || max_block_width < 1 || max_block_height < 1 || max_block_width > 128 ||
      max_block_height > 128 || (mode != 0 && (!cm.conv || cm.conv_stride <= 0))) {
    set_error("%s: invalid argument", who);
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  const int tiles = ((max_block_width + 7) / 8) * ((max_block_height + 7) / 8);
  const dim3 grid((unsigned)n_blocks, (unsigned)((tiles + 3) / 4)), block(256);
  const int64_t poff = pred ? (int64_t)pred_frame * pred->frame_stride + (int64_t)pred->border * pred->stride + pred->border : 0;
  void *pbase = pred ? pred->base : nullptr;
  const int pstride = pred ? pred->stride : 0;
#define LAUNCH(T, M)                                                                                                                                   \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(warp_affine_kernel<T, M>), grid, block, 0, ctx->stream, view_of<T>(*ref), ref_frame, ref->width, ref->height,    \
                     static_cast<T *>(pbase), poff, pstride, subsampling_x, subsampling_y, ref->bit_depth == 8 ? 8 : ref->bit_depth, d_blocks, n_blocks, cm)
  if (ref->bit_depth == 8) {
    if (mode == 0) LAUNCH(uint8_t, 0); else if (mode == 1) LAUNCH(uint8_t, 1); else LAUNCH(uint8_t, 2);
  } else {
    if (mode == 0) LAUNCH(uint16_t, 0); else if (mode == 1) LAUNCH(uint16_t, 1); else LAUNCH(uint16_t, 2);
  }
#undef LAUNCH
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_warp_affine_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred, int pred_frame, int subsampling_x,
                                        int subsampling_y, const aomhip_warp_block *d_blocks, int n_blocks, int max_block_width, int max_block_height) {
  return warp_launch(ctx, ref, ref_frame, pred, pred_frame, subsampling_x, subsampling_y, d_blocks, n_blocks, max_block_width, max_block_height, 0,
                     WarpCompound{ nullptr, 0, 0, 0, 0 }, "aomhip_warp_affine_batch");
}

extern "C" int aomhip_warp_affine_compound_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred, int pred_frame,
                                                 int subsampling_x, int subsampling_y, const aomhip_warp_block *d_blocks, int n_blocks, int max_block_width,
                                                 int max_block_height, uint16_t *d_conv, int conv_stride, int do_average, int use_dist_wtd_comp_avg,
                                                 int fwd_offset, int bck_offset) {
  return warp_launch(ctx, ref, ref_frame, do_average ? pred : nullptr, pred_frame, subsampling_x, subsampling_y, d_blocks, n_blocks, max_block_width,
                     max_block_height, do_average ? 2 : 1, WarpCompound{ d_conv, conv_stride, use_dist_wtd_comp_avg, fwd_offset, bck_offset },
                     "aomhip_warp_affine_compound_batch");
}

static int warp_error_launch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *cur, int cur_frame, int subsampling_x, int subsampling_y,
                             const aomhip_warp_model *d_models, int n_models, int p_col, int p_row, int p_width, int p_height, const uint8_t *d_segment_map,
                             int segment_map_stride, int64_t *d_error, bool warp, const char *who) {
  if (!ctx || !ref || !ref->base || !cur || !cur->base || ref_frame < 0 || ref_frame >= ref->n_frames || cur_frame < 0 || cur_frame >= cur->n_frames ||
      (ref->bit_depth == 8) != (cur->bit_depth == 8) || ref->bit_depth != cur->bit_depth || (subsampling_x | subsampling_y) < 0 || subsampling_x > 1 ||
      subsampling_y > 1 || n_models < 0 || (warp && n_models > 0 && !d_models) || p_col < 0 || p_row < 0 || p_width < 1 || p_height < 1 ||
      p_col + p_width > cur->width || p_row + p_height > cur->height || (!warp && (p_width > ref->width || p_height > ref->height)) || !d_segment_map ||
      segment_map_stride < ((p_col + p_width + 31) >> 5) || !d_error) {
    set_error("%s: invalid argument", who);
    return AOMHIP_ERR_INVALID;
  }
  if (n_models == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  const int tiles_x = (p_width + 31) >> 5, tiles_y = (p_height + 31) >> 5;
  int32_t *d_tiles = static_cast<int32_t *>(work(ctx, sizeof(int32_t) * (size_t)n_models * tiles_x * tiles_y));
  if (!d_tiles) return AOMHIP_ERR_NOMEM;
  const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)n_models), block(256);
#define LAUNCH(T, W)                                                                                                                                   \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(warp_error_kernel<T, W>), grid, block, 0, ctx->stream, view_of<T>(*ref), ref_frame, ref->width, ref->height,     \
                     view_of<T>(*cur), cur_frame, subsampling_x, subsampling_y, ref->bit_depth, d_models, p_col, p_row, p_width, p_height, d_segment_map, \
                     segment_map_stride, tiles_x, d_tiles)
  if (ref->bit_depth == 8) { if (warp) LAUNCH(uint8_t, true); else LAUNCH(uint8_t, false); }
  else { if (warp) LAUNCH(uint16_t, true); else LAUNCH(uint16_t, false); }
#undef LAUNCH
  AOMHIP_LAUNCH_CHECK();
  hipLaunchKernelGGL(warp_error_sum_kernel, dim3((unsigned)n_models), dim3(256), 0, ctx->stream, d_tiles, tiles_x * tiles_y, d_error);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_warp_error_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *cur, int cur_frame, int subsampling_x,
                                       int subsampling_y, const aomhip_warp_model *d_models, int n_models, int p_col, int p_row, int p_width, int p_height,
                                       const uint8_t *d_segment_map, int segment_map_stride, int64_t *d_error) {
  return warp_error_launch(ctx, ref, ref_frame, cur, cur_frame, subsampling_x, subsampling_y, d_models, n_models, p_col, p_row, p_width, p_height, d_segment_map,
                           segment_map_stride, d_error, true, "aomhip_warp_error_batch");
}

extern "C" int aomhip_segmented_frame_error(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *cur, int cur_frame, int p_width,
                                            int p_height, const uint8_t *d_segment_map, int segment_map_stride, int64_t *d_error) {
  return warp_error_launch(ctx, ref, ref_frame, cur, cur_frame, 0, 0, nullptr, 1, 0, 0, p_width, p_height, d_segment_map, segment_map_stride, d_error, false,
                           "aomhip_segmented_frame_error");
}

