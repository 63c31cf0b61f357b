// Integer-MV motion-compensated prediction: for a full-pel motion vector the reference's inter predictor
// (av1_build_inter_predictor -> the convolve with zero sub-pel phase, av1/common/reconinter.c) degenerates to
// aom_convolve_copy, i.e. pred block = reference block at (bx + mv.col, by + mv.row).  This is only the
// glue the frame-level pipeline (search -> residual -> transform) needs; sub-pel interpolation is out of scope.
#include "common.h"

namespace aomhip {

template <typename T>
__global__ __launch_bounds__(256) void pred_copy_kernel(PlaneView<T> ref, int ref_frame, T *dst_origin, int dst_stride,
                                                        const aomhip_search_block *__restrict__ blocks,
                                                        const int16_t *__restrict__ mv, int n_blocks, int bw, int bh) {
  // one wavefront per block; lanes sweep the block in 16-byte pieces
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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

}  // namespace aomhip

using namespace aomhip;

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
