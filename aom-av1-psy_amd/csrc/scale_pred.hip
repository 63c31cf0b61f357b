// The predictor of a SCALED reference: av1_convolve_2d_scale / av1_highbd_convolve_2d_scale (av1/common/convolve.c; av1_make_inter_predictor's
// `is_scaled` branch -- reference scaling, frame resizing, super-resolution's scaled motion compensation).  Every output sample has its own
// source position in 1/1024 pel: x_qn = subpel_x_qn + x * x_step_qn (integer part x_qn >> 10, kernel phase (x_qn & 1023) >> 6), likewise down
// the rows, so neither pass has the fixed phase of the unscaled kernels (pred.hip).  One 256-lane workgroup per 32 x 32 tile of a block: the
// horizontal pass fills the intermediate rows the tile's outputs read (at most 31 * 2 + 1 + 8 = 71 at the largest step, 2:1) in LDS as int16
// -- the reference's im_block, rounded by round_0 --, the vertical pass one output per lane and step.  conv_params as get_conv_params_no_round
// gives them (round_0 = 3, 5 at 12 bits; round_1 = 7 for a compound, else 14 - round_0); MODE as in warp.hip: 0 single, 1 the first
// reference of a compound into the CONV_BUF, 2 the second blended with it.
#include "pred_device.h"

namespace aomhip {
namespace {

struct ScaleCompound {
  uint16_t *conv;      // CONV_BUF: element (row, col) of the prediction plane at conv[row * conv_stride + col]
  int conv_stride;
  int use_dist_wtd, fwd_offset, bck_offset;
};

constexpr int kScaleTile = 32, kScaleRows = 72;

template <typename T, int MODE>
__global__ __launch_bounds__(256) void scale_pred_kernel(PlaneView<T> ref, int ref_frame, T *__restrict__ pred_origin, int64_t pred_frame_off, int pred_stride,
                                                         int bw, int bh, int set_x, int set_y, int x_step_qn, int y_step_qn, int bd,
                                                         const aomhip_scaled_block *__restrict__ blocks, int tiles_x, ScaleCompound cm) {
  __shared__ int16_t im[kScaleRows * kScaleTile];
  const aomhip_scaled_block b = blocks[blockIdx.x];
  const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
  const int x0 = tx * kScaleTile, y0 = ty * kScaleTile;
  const int tw = min(kScaleTile, bw - x0), th = min(kScaleTile, bh - y0);
  const int round_0 = bd == 12 ? 5 : 3, round_1 = MODE ? 7 : 14 - round_0, bits = 14 - round_0 - round_1;
  const int r_first = (b.subpel_y_qn + y0 * y_step_qn) >> 10;                        // im_block row of the tile's first output's tap 0
  const int r_count = ((b.subpel_y_qn + (y0 + th - 1) * y_step_qn) >> 10) + 8 - r_first;
  const T *src = ref.origin + (int64_t)ref_frame * ref.frame_stride + (int64_t)b.src_y * ref.stride + b.src_x;
  for (int t = threadIdx.x; t < r_count * kScaleTile; t += 256) {   // horizontal pass: im_block row r_first + rr is source row r_first + rr - 3
    const int rr = t >> 5, xx = t & 31;
    if (xx >= tw) continue;
    const int x_qn = b.subpel_x_qn + (x0 + xx) * x_step_qn;
    const T *p = src + (int64_t)(r_first + rr - 3) * ref.stride + (x_qn >> 10) - 3;
    const int16_t *f = kInterp[set_x][(x_qn & 1023) >> 6];
    int sum = 1 << (bd + 6);
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)f[k] * (int)p[k];
    im[t] = (int16_t)((sum + ((1 << round_0) >> 1)) >> round_0);
  }
  __syncthreads();
  const int offset_bits = bd + 14 - round_0;
  const int off = (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
  for (int t = threadIdx.x; t < th * kScaleTile; t += 256) {
    const int yy = t >> 5, xx = t & 31;
    if (xx >= tw) continue;
    const int y_qn = b.subpel_y_qn + (y0 + yy) * y_step_qn;
    const int16_t *col = im + ((y_qn >> 10) - r_first) * kScaleTile + xx;
    const int16_t *f = kInterp[set_y][(y_qn & 1023) >> 6];
    int sum = 1 << offset_bits;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)f[k] * (int)col[k * kScaleTile];
    const int res = (sum + ((1 << round_1) >> 1)) >> round_1;
    const int64_t at = (int64_t)(b.dst_y + y0 + yy);
    const int ax = b.dst_x + x0 + xx;
    if constexpr (MODE == 1) {
      cm.conv[at * cm.conv_stride + ax] = (uint16_t)res;   // CONV_BUF_TYPE
    } else {
      int v = res;
      if constexpr (MODE == 2) {
        const int t32 = (int)cm.conv[at * cm.conv_stride + ax];
        v = cm.use_dist_wtd ? (t32 * cm.fwd_offset + res * cm.bck_offset) >> 4 : (t32 + res) >> 1;   // DIST_PRECISION_BITS
      }
      v = (v - off + ((1 << bits) >> 1)) >> bits;
      pred_origin[pred_frame_off + at * pred_stride + ax] = (T)min(max(v, 0), (1 << bd) - 1);
    }
  }
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

static int scale_launch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred, int pred_frame, int bw, int bh, int filter_x,
                        int filter_y, int x_step_qn, int y_step_qn, const aomhip_scaled_block *d_blocks, int n_blocks, int mode, ScaleCompound cm, const char *who) {
  if (!ctx || !ref || !ref->base || n_blocks < 0 || (n_blocks > 0 && !d_blocks) || ref_frame < 0 || ref_frame >= ref->n_frames ||
      (mode != 1 && (!pred || !pred->base || pred_frame < 0 || pred_frame >= pred->n_frames || (ref->bit_depth == 8) != (pred->bit_depth == 8))) || bw < 1 ||
      bh < 1 || bw > 128 || bh > 128 || filter_x < 0 || filter_x > 3 || filter_y < 0 || filter_y > 3 || x_step_qn < 64 || x_step_qn > 2048 || y_step_qn < 64 ||
      y_step_qn > 2048 || (mode != 0 && (!cm.conv || cm.conv_stride <= 0))) {
    set_error("%s: invalid argument (steps are 1/1024 pel per output sample, 64 .. 2048: 16:1 up-scaling to 2:1 down-scaling)", who);
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  // av1_get_interp_filter_params_with_block_size: a dimension <= 4 takes the 4-tap sets (sharp -> regular)
  auto set_of = [](int f, int dim) { return dim <= 4 ? (f == 1 ? 5 : f == 3 ? 3 : 4) : f; };
  const int sx = set_of(filter_x, bw), sy = set_of(filter_y, bh);
  const int tiles_x = (bw + kScaleTile - 1) / kScaleTile, tiles_y = (bh + kScaleTile - 1) / kScaleTile;
  const dim3 grid((unsigned)n_blocks, (unsigned)(tiles_x * tiles_y)), block(256);
  const int64_t poff = pred ? (int64_t)pred_frame * pred->frame_stride + (int64_t)pred->border * pred->stride + pred->border : 0;
  void *pbase = pred ? pred->base : nullptr;
  const int pstride = pred ? pred->stride : 0;
#define LAUNCH(T, M)                                                                                                                                       \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(scale_pred_kernel<T, M>), grid, block, 0, ctx->stream, view_of<T>(*ref), ref_frame, static_cast<T *>(pbase), poff, pstride, \
                     bw, bh, sx, sy, x_step_qn, y_step_qn, ref->bit_depth == 8 ? 8 : ref->bit_depth, d_blocks, tiles_x, cm)
  if (ref->bit_depth == 8) {
    if (mode == 0) LAUNCH(uint8_t, 0); else if (mode == 1) LAUNCH(uint8_t, 1); else LAUNCH(uint8_t, 2);
  } else {
    if (mode == 0) LAUNCH(uint16_t, 0); else if (mode == 1) LAUNCH(uint16_t, 1); else LAUNCH(uint16_t, 2);
  }
#undef LAUNCH
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_scaled_pred_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred, int pred_frame, int bw, int bh,
                                        int filter_x, int filter_y, int x_step_qn, int y_step_qn, const aomhip_scaled_block *d_blocks, int n_blocks) {
  return scale_launch(ctx, ref, ref_frame, pred, pred_frame, bw, bh, filter_x, filter_y, x_step_qn, y_step_qn, d_blocks, n_blocks, 0,
                      ScaleCompound{ nullptr, 0, 0, 0, 0 }, "aomhip_scaled_pred_batch");
}

extern "C" int aomhip_scaled_pred_compound_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int ref_frame, const aomhip_planes *pred, int pred_frame, int bw,
                                                 int bh, int filter_x, int filter_y, int x_step_qn, int y_step_qn, const aomhip_scaled_block *d_blocks,
                                                 int n_blocks, uint16_t *d_conv, int conv_stride, int do_average, int use_dist_wtd_comp_avg, int fwd_offset,
                                                 int bck_offset) {
  return scale_launch(ctx, ref, ref_frame, do_average ? pred : nullptr, pred_frame, bw, bh, filter_x, filter_y, x_step_qn, y_step_qn, d_blocks, n_blocks,
                      do_average ? 2 : 1, ScaleCompound{ d_conv, conv_stride, use_dist_wtd_comp_avg, fwd_offset, bck_offset }, "aomhip_scaled_pred_compound_batch");
}
