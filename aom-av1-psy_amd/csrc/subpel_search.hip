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
  if (!d_best_mv || !d_best_err || !d_distortion || !d_sse || p->forced_stop < 0 || p->forced_stop > 3 || p->tree < 0 ||
      p->tree > 2 || (p->subpel_search_type < 0 || p->subpel_search_type > 3) ||
      (entropy && (!d_mvjcost || !d_mvcost_row || !d_mvcost_col))) {
    set_error("aomhip_subpel_tree_batch: invalid argument (tree 0..2, forced_stop 0..3, MV_COST_ENTROPY needs its tables)");
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
