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

struct WarpCompound {
  uint16_t *conv;      // CONV_BUF: element (row, col) of the plane at conv[row * conv_stride + col]
  int conv_stride;
  int use_dist_wtd, fwd_offset, bck_offset;
};

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
  int32_t mat[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) mat[k] = __builtin_amdgcn_readfirstlane(b.mat[k]);
  const int alpha = __builtin_amdgcn_readfirstlane((int)b.alpha), beta = __builtin_amdgcn_readfirstlane((int)b.beta);
  const int gamma = __builtin_amdgcn_readfirstlane((int)b.gamma), delta = __builtin_amdgcn_readfirstlane((int)b.delta);
  const bool hbd = sizeof(T) == 2;
  const int round_0 = bd == 12 ? 5 : 3;   // ROUND0_BITS (+ 2 at 12 bits): get_conv_params_no_round (av1/common/convolve.h)
  const int extra = hbd ? max(bd + 7 - round_0 - 14, 0) : 0;
  const int round_1 = MODE ? 7 : 14 - round_0;   // COMPOUND_ROUND1_BITS
  const int reduce_bits_horiz = round_0 + extra, reduce_bits_vert = MODE ? round_1 : 14 - reduce_bits_horiz;
  [[maybe_unused]] const int round_bits = 14 - round_0 - round_1, offset_bits = bd + 14 - round_0;
  const int offset_bits_horiz = bd + 6, offset_bits_vert = bd + 14 - reduce_bits_horiz;
  // the centre of the tile in luma coordinates, through the model, back to this plane's coordinates
  const int32_t src_x = (j + 4) << ssx, src_y = (i + 4) << ssy;
  const int64_t dst_x = (int64_t)mat[2] * src_x + (int64_t)mat[3] * src_y + (int64_t)mat[0];
  const int64_t dst_y = (int64_t)mat[4] * src_x + (int64_t)mat[5] * src_y + (int64_t)mat[1];
  const int64_t x4 = dst_x >> ssx, y4 = dst_y >> ssy;
  const int ix4 = (int)(x4 >> 16), iy4 = (int)(y4 >> 16);   // WARPEDMODEL_PREC_BITS
  int sx4 = (int)(x4 & 0xffff), sy4 = (int)(y4 & 0xffff);
  sx4 += alpha * (-4) + beta * (-4);
  sy4 += gamma * (-4) + delta * (-4);
  sx4 &= ~63;   // WARP_PARAM_REDUCE_BITS
  sy4 &= ~63;
  int32_t *tmp = tmp_all[wave];
  const T *rp = ref.origin + (int64_t)ref_frame * ref.frame_stride;
  for (int t = lane; t < 15 * 8; t += 64) {   // horizontal filter: tmp[(k + 7) * 8 + (l + 4)], k = -7 .. 7, l = -4 .. 3
    const int k = (t >> 3) - 7, l = (t & 7) - 4;
    const int iy = min(max(iy4 + k, 0), height - 1);
    const int sx = sx4 + beta * (k + 4) + alpha * (l + 4);
    const int offs = ((sx + 512) >> 10) + 64;   // ROUND_POWER_OF_TWO(sx, WARPEDDIFF_PREC_BITS) + WARPEDPIXEL_PREC_SHIFTS
    const int4 cw = *reinterpret_cast<const int4 *>(kWarpedFilter[offs]);
    const int c[8] = { (int)(int16_t)(cw.x & 0xffff), cw.x >> 16, (int)(int16_t)(cw.y & 0xffff), cw.y >> 16,
                       (int)(int16_t)(cw.z & 0xffff), cw.z >> 16, (int)(int16_t)(cw.w & 0xffff), cw.w >> 16 };
    const int ix = ix4 + l - 3;
    const T *row = rp + (int64_t)iy * ref.stride;
    int32_t sum = 1 << offset_bits_horiz;
#pragma unroll
    for (int m = 0; m < 8; ++m) sum += (int)row[min(max(ix + m, 0), width - 1)] * c[m];
    tmp[t] = (sum + ((1 << reduce_bits_horiz) >> 1)) >> reduce_bits_horiz;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  {   // vertical filter: lane -> (k, l) = (-4 .. 3, -4 .. 3)
    const int k = (lane >> 3) - 4, l = (lane & 7) - 4;
    const int kmax = min(4, p_row + p_height - i - 4), lmax = min(4, p_col + p_width - j - 4);
    if (k < kmax && l < lmax) {
      const int sy = sy4 + delta * (k + 4) + gamma * (l + 4);
      const int offs = ((sy + 512) >> 10) + 64;
      const int4 cw = *reinterpret_cast<const int4 *>(kWarpedFilter[offs]);
      const int c[8] = { (int)(int16_t)(cw.x & 0xffff), cw.x >> 16, (int)(int16_t)(cw.y & 0xffff), cw.y >> 16,
                         (int)(int16_t)(cw.z & 0xffff), cw.z >> 16, (int)(int16_t)(cw.w & 0xffff), cw.w >> 16 };
      int32_t sum = 1 << offset_bits_vert;
#pragma unroll
      for (int m = 0; m < 8; ++m) sum += tmp[(k + m + 4) * 8 + (l + 4)] * c[m];
      sum = (sum + ((1 << reduce_bits_vert) >> 1)) >> reduce_bits_vert;
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
  }
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
