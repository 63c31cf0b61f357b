// libaomhip -- host entry points of the sub-pel motion search (kernels: subpel_search.inc, compiled once per pixel
// type in subpel_search_u8.hip / subpel_search_u16.hip).
#include <climits>

#include "common.h"
#include "search_device.h"

namespace aomhip {

struct SubpelCostTables {
  const int *mvjcost, *mvcost0, *mvcost1;
  int error_per_bit;
  int upsampled;  // tree 2 with subpel_search_type != USE_2_TAPS_ORIG: errors from the up-sampled prediction; the value is the type (1 / 2 / 3 = 2 / 4 / 8 taps)
  int16_t *mv_lists;  // last_mv_search_list per block (3 x (row, col), read and updated; general instantiation only) or null
  const void *second_pred;  // compound search: n x (W * H) pixels of the other reference's predictor, or null (general instantiation only)
  const uint8_t *cmask;     // ... and n x (W * H) blend weights for a masked compound, or null
  int invert_mask;
};

#define AOMHIP_DECL_SUBPEL(NAME)                                                                                              \
  int NAME(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh, int mv_cost_type, \
           int iters_per_step, int allow_hp, int forced_stop, int tree, const int32_t *d_cost_lists, SubpelCostTables ct,     \
           const aomhip_search_block *d_blocks, int n_blocks, int16_t *d_best_mv, uint32_t *d_best_err, int32_t *d_distortion, \
           uint32_t *d_sse);
AOMHIP_DECL_SUBPEL(launch_subpel_u8)
AOMHIP_DECL_SUBPEL(launch_subpel_u16)
#undef AOMHIP_DECL_SUBPEL

static int launch_subpel(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                         int mv_cost_type, int iters_per_step, int allow_hp, int forced_stop, int tree,
                         const int32_t *d_cost_lists, SubpelCostTables ct, const aomhip_search_block *d_blocks, int n_blocks,
                         int16_t *d_best_mv, uint32_t *d_best_err, int32_t *d_distortion, uint32_t *d_sse) {
  return (src->bit_depth == 8 ? launch_subpel_u8 : launch_subpel_u16)(ctx, src, ref, frame, bw, bh, mv_cost_type, iters_per_step,
                                                                     allow_hp, forced_stop, tree, d_cost_lists, ct, d_blocks,
                                                                     n_blocks, d_best_mv, d_best_err, d_distortion, d_sse);
}

// av1_return_max_sub_pixel_mv / av1_return_min_sub_pixel_mv (mcomp.c:3139-3190): the extreme MV the block's SubpelMvLimits allow, with
// lower_mv_precision (av1/common/mvref_common.h:88-97: an odd component moves one step towards zero when high precision is off); besterr 0.
// A block with an empty window (row_min > row_max) is skipped like everywhere else.
__global__ __launch_bounds__(256) void extreme_mv_kernel(const aomhip_search_block *__restrict__ blocks, int n_blocks, int allow_hp, int want_max,
                                                         int16_t *__restrict__ best_mv, uint32_t *__restrict__ best_err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_blocks) return;
  const aomhip_search_block b = blocks[i];
  if (b.row_min > b.row_max) return;
  int row = want_max ? b.row_max : b.row_min, col = want_max ? b.col_max : b.col_min;
  if (!allow_hp) {
    if (row & 1) row += row > 0 ? -1 : 1;
    if (col & 1) col += col > 0 ? -1 : 1;
  }
  best_mv[2 * i] = (int16_t)row;
  best_mv[2 * i + 1] = (int16_t)col;
  best_err[i] = 0;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_subpel_bilinear_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw,
                                 int bh, int mv_cost_type, int iters_per_step, int allow_hp, int forced_stop,
                                 const aomhip_search_block *d_blocks, int n_blocks, int16_t *d_best_mv,
                                 uint32_t *d_best_err, int32_t *d_distortion, uint32_t *d_sse) {
  int rc = check_common(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, mv_cost_type);
  if (rc != AOMHIP_OK) return rc;
  if (!d_best_mv || !d_best_err || !d_distortion || !d_sse || forced_stop < 0 || forced_stop > 3) {
    set_error("aomhip_subpel_bilinear_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  return launch_subpel(ctx, src, ref, frame, bw, bh, mv_cost_type, iters_per_step, allow_hp, forced_stop, /*tree=*/0, nullptr,
                       SubpelCostTables{ nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0 }, d_blocks, n_blocks, d_best_mv, d_best_err,
                       d_distortion, d_sse);
}

int aomhip_subpel_tree_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                             const aomhip_subpel_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                             const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, const int32_t *d_cost_list,
                             int n_blocks, int16_t *d_best_mv, uint32_t *d_best_err, int32_t *d_distortion, uint32_t *d_sse) {
  return aomhip_subpel_tree_list_batch(ctx, src, ref, frame, bw, bh, p, d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, d_cost_list, n_blocks, d_best_mv,
                                       d_best_err, d_distortion, d_sse, nullptr);
}

int aomhip_subpel_tree_list_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                  const aomhip_subpel_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                                  const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, const int32_t *d_cost_list,
                                  int n_blocks, int16_t *d_best_mv, uint32_t *d_best_err, int32_t *d_distortion, uint32_t *d_sse,
                                  int16_t *d_mv_lists) {
  if (!p) {
    set_error("aomhip_subpel_tree_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const bool entropy = p->mv_cost_type == kCostEntropy;
  int rc = check_common(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, entropy ? kCostNone : p->mv_cost_type);
  if (rc != AOMHIP_OK) return rc;
  if (p->tree == 3 || p->tree == 4) {   // av1_return_max_sub_pixel_mv / av1_return_min_sub_pixel_mv (mcomp.c:3139-3190)
    if (!d_best_mv || !d_best_err) {
      set_error("aomhip_subpel_tree_batch: invalid argument");
      return AOMHIP_ERR_INVALID;
    }
    if (n_blocks == 0) return AOMHIP_OK;
    AOMHIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(extreme_mv_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, ctx->stream, d_blocks, n_blocks, p->allow_hp,
                       p->tree == 3 ? 1 : 0, d_best_mv, d_best_err);
    AOMHIP_LAUNCH_CHECK();
    return AOMHIP_OK;
  }
  if (!d_best_mv || !d_best_err || !d_distortion || !d_sse || p->forced_stop < 0 || p->forced_stop > 3 || p->tree < 0 ||
      p->tree > 2 || (p->subpel_search_type < 0 || p->subpel_search_type > 3) ||
      (entropy && (!d_mvjcost || !d_mvcost_row || !d_mvcost_col))) {
    set_error("aomhip_subpel_tree_batch: invalid argument (tree 0..4, forced_stop 0..3, MV_COST_ENTROPY needs its tables)");
    return AOMHIP_ERR_INVALID;
  }
  return launch_subpel(ctx, src, ref, frame, bw, bh, p->mv_cost_type, p->iters_per_step, p->allow_hp, p->forced_stop, p->tree,
                       d_cost_list, SubpelCostTables{ d_mvjcost, d_mvcost_row, d_mvcost_col, p->error_per_bit, p->tree == 2 ? p->subpel_search_type : 0, d_mv_lists, nullptr, nullptr, 0 }, d_blocks,
                       n_blocks, d_best_mv, d_best_err, d_distortion, d_sse);
}

int aomhip_compound_subpel_tree_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                      const aomhip_subpel_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                      const aomhip_search_block *d_blocks, int n_blocks, const void *d_second_pred, const uint8_t *d_mask, int invert_mask,
                                      int16_t *d_best_mv, uint32_t *d_best_err, int32_t *d_distortion, uint32_t *d_sse) {
  if (!p || !d_second_pred) {
    set_error("aomhip_compound_subpel_tree_batch: invalid argument (a compound search needs the other reference's predictor)");
    return AOMHIP_ERR_INVALID;
  }
  const bool entropy = p->mv_cost_type == kCostEntropy;
  int rc = check_common(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, entropy ? kCostNone : p->mv_cost_type);
  if (rc != AOMHIP_OK) return rc;
  if (!d_best_mv || !d_best_err || !d_distortion || !d_sse || p->forced_stop < 0 || p->forced_stop > 3 || p->tree < 0 || p->tree > 2 ||
      (p->subpel_search_type < 0 || p->subpel_search_type > 3) || (entropy && (!d_mvjcost || !d_mvcost_row || !d_mvcost_col))) {
    set_error("aomhip_compound_subpel_tree_batch: invalid argument (tree 0..2, forced_stop 0..3, MV_COST_ENTROPY needs its tables)");
    return AOMHIP_ERR_INVALID;
  }
  return launch_subpel(ctx, src, ref, frame, bw, bh, p->mv_cost_type, p->iters_per_step, p->allow_hp, p->forced_stop, p->tree, nullptr,
                       SubpelCostTables{ d_mvjcost, d_mvcost_row, d_mvcost_col, p->error_per_bit, p->tree == 2 ? p->subpel_search_type : 0, nullptr,
                                         d_second_pred, d_mask, invert_mask != 0 },
                       d_blocks, n_blocks, d_best_mv, d_best_err, d_distortion, d_sse);
}

}  // extern "C"
