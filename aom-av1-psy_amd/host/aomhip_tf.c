/* Host side of the temporal filter's motion search (plain C99, no GPU call): the block list of a frame and the search
 * parameters as tf_motion_search derives them (av1/encoder/temporal_filter.c:87-160).  The device pass is
 * aomhip_tf_motion_search_frames (csrc/tf_search.hip). */
#include <string.h>

#include "aomhip.h"

enum { kTfBlock = 32, kMiSize = 4, kInterpExtend = 4 /* AOM_INTERP_EXTEND */, kMaxFullPelVal = 1023, kMaxMvSearchSteps = 11 };

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* av1_init_search_range (av1/encoder/mcomp.c:217-226) */
static int init_search_range(int size) {
  int sr = 0;
  size = imax(16, size);
  while ((size << sr) < kMaxFullPelVal) sr++;
  return imin(sr, kMaxMvSearchSteps - 2);
}

void aomhip_tf_default_params(int width, int height, int bit_depth, int q, int prune_mesh_level, const int mesh_patterns[8], int subpel_tree,
                              int subpel_iters_per_step, int allow_hp, int use_cost_list, int use_downsampled_sad, int force_integer_mv,
                              aomhip_tf_params *out) {
  const int min_frame_size = imin(width, height);
  memset(out, 0, sizeof(*out));
  out->full.search_method = AOMHIP_SEARCH_NSTEP;                         /* :118 */
  out->full.step_param = init_search_range(imax(width, height));         /* :121-122 */
  out->full.mv_cost_type = min_frame_size >= 720 ? AOMHIP_MV_COST_L1_HDRES
                                                 : (min_frame_size >= 480 ? AOMHIP_MV_COST_L1_MIDRES : AOMHIP_MV_COST_L1_LOWRES); /* :125-128 */
  out->full.use_downsampled_sad = use_downsampled_sad != 0;              /* mcomp.c:122-133; both block sizes are >= 16 rows */
  out->full.run_mesh_search = 1;                                         /* :160 */
  /* av1_make_default_fullpel_ms_params (mcomp.c:138-140), then tf_motion_search's LVL_1 rule (:163-167) */
  out->full.prune_mesh_search = prune_mesh_level == 2;
  out->full.mesh_search_mv_diff_threshold = 4;
  if (prune_mesh_level == 1) {
    out->full.prune_mesh_search = q <= 20 ? 0 : 1;
    out->full.mesh_search_mv_diff_threshold = 2;
  }
  out->full.force_mesh_thresh = 0x7fffffff;                              /* irrelevant once run_mesh_search is set */
  out->full.fine_search_interval = 0;
  if (mesh_patterns) memcpy(out->full.mesh_patterns, mesh_patterns, sizeof(out->full.mesh_patterns));
  out->sub.tree = subpel_tree;
  out->sub.mv_cost_type = AOMHIP_MV_COST_NONE;                           /* :183-185 */
  out->sub.error_per_bit = 0;
  out->sub.iters_per_step = subpel_iters_per_step;
  out->sub.allow_hp = allow_hp;
  out->sub.forced_stop = 0;                                              /* EIGHTH_PEL, :181 */
  out->sub.subpel_search_type = 3;                                       /* USE_8_TAPS, :123 */
  out->use_cost_list = use_cost_list != 0;
  out->force_integer_mv = force_integer_mv != 0;
  out->mse_thresh = (min_frame_size >= 720 ? 12 : 3) << (bit_depth - 8); /* :249-252 */
}

int aomhip_tf_block_list(int width, int height, int border, aomhip_search_block *blocks) {
  if (width <= 0 || height <= 0) return 0;
  const int mb_rows = (height + kTfBlock - 1) / kTfBlock, mb_cols = (width + kTfBlock - 1) / kTfBlock; /* get_num_blocks */
  if (!blocks) return mb_rows * mb_cols;
  /* mi_params->mi_rows / mi_cols of the frame (size_in_mi, av1/encoder/encoder_utils.h:55-69: 8-aligned size in 4x4 units) */
  const int mi_rows = ((height + 7) & ~7) / kMiSize, mi_cols = ((width + 7) & ~7) / kMiSize;
  const int mi_n = kTfBlock / kMiSize;
  for (int r = 0; r < mb_rows; ++r) {
    /* av1_set_mv_row_limits (mcomp.h:216-227) */
    const int mi_row = r * mi_n;
    const int row_min = imax(-(mi_row * kMiSize + border - 2 * kInterpExtend), -((mi_row + mi_n) * kMiSize + 2 * kInterpExtend));
    const int row_max = imin((mi_rows - mi_row - mi_n) * kMiSize + border - 2 * kInterpExtend, (mi_rows - mi_row) * kMiSize + 2 * kInterpExtend);
    for (int c = 0; c < mb_cols; ++c) {
      /* av1_set_mv_col_limits (mcomp.h:229-240) */
      const int mi_col = c * mi_n;
      const int col_min = imax(-(mi_col * kMiSize + border - 2 * kInterpExtend), -((mi_col + mi_n) * kMiSize + 2 * kInterpExtend));
      const int col_max = imin((mi_cols - mi_col - mi_n) * kMiSize + border - 2 * kInterpExtend, (mi_cols - mi_col) * kMiSize + 2 * kInterpExtend);
      aomhip_search_block *b = &blocks[r * mb_cols + c];
      memset(b, 0, sizeof(*b));
      b->bx = (int16_t)(c * kTfBlock);
      b->by = (int16_t)(r * kTfBlock);
      b->row_min = (int16_t)row_min; b->row_max = (int16_t)row_max;
      b->col_min = (int16_t)col_min; b->col_max = (int16_t)col_max;
    }
  }
  return mb_rows * mb_cols;
}
