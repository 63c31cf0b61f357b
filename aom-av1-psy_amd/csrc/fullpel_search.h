// Types shared by the full-pel search host code (fullpel_search.hip) and its kernels (fullpel_search.inc).
#ifndef AOMHIP_CSRC_FULLPEL_SEARCH_H_
#define AOMHIP_CSRC_FULLPEL_SEARCH_H_

#include "common.h"
#include "search_device.h"

namespace aomhip {

// search_site_config (mcomp_structs.h:36-48) without the stride-dependent offsets
struct SiteTable {
  int num_search_steps;
  int searches_per_step[22];
  int radius[22];
  int16_t mv[22][17][2];
};

enum { kDiamond, kNstep, kNstep8, kClamped, kHex, kBigdia, kSquare, kFastHex, kFastDiamond, kFastBigdia, kVfastDiamond, kNstepFpf, kMethods };

struct SearchArgs {
  int method, step_param, cost_type, sad_per_bit, error_per_bit, skip_sad;
  int run_mesh, prune_mesh, mesh_diff_thr, force_mesh_thresh, fine_interval;
  int mesh[8];
  const int *mvjcost, *mvcost0, *mvcost1;
  int bit_depth, want_cl;  // want_cl: the caller keeps a cost_list (changes pattern_search's last scale, :1077)
};

#define AOMHIP_DECL_FPS(NAME)                                                                                                 \
  int NAME(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,                   \
           const aomhip_search_block *d_blocks, int n_blocks, const SiteTable *d_sites, SearchArgs q, int reach,          \
           int16_t *d_best_mv,                                                                                                  \
           int32_t *d_best_cost, int32_t *d_cost_list, int16_t *d_second_best_mv);
AOMHIP_DECL_FPS(launch_fps_u8)
AOMHIP_DECL_FPS(launch_fps_u16)
#undef AOMHIP_DECL_FPS

}  // namespace aomhip

#endif  // AOMHIP_CSRC_FULLPEL_SEARCH_H_
