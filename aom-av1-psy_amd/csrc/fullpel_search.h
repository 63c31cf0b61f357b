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
  int resume;  // the method's own search ran elsewhere (the compound diamond, mcomp_compound.hip): out_mv / out_cost / out_second hold its result,
               // only the follow-up of av1_full_pixel_search (:1756-1830: mesh rules, full_pixel_exhaustive) runs here
};

#define AOMHIP_DECL_FPS(NAME)                                                                                                 \
  int NAME(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,                   \
           const aomhip_search_block *d_blocks, int n_blocks, const SiteTable *d_sites, SearchArgs q, int reach,          \
           int16_t *d_best_mv,                                                                                                  \
           int32_t *d_best_cost, int32_t *d_cost_list, int16_t *d_second_best_mv);
AOMHIP_DECL_FPS(launch_fps_u8)
AOMHIP_DECL_FPS(launch_fps_u16)
#undef AOMHIP_DECL_FPS

// The per-block record as scalars (every member wave-uniform; BlockScalars::of applies v_readfirstlane: the compiler packs pairs of the
// record's 16-bit fields into v_pk_min / max_i16 -- VALU only -- and everything derived from a VGPR, the whole search state, then
// follows it into the vector unit and under exec masks; mcomp.hip).
struct BlockScalars {
  int row_min, row_max, col_min, col_max, ref_row, ref_col, start_row, start_col;
  static __device__ __forceinline__ BlockScalars of(const aomhip_search_block &b) {
    return BlockScalars{ __builtin_amdgcn_readfirstlane((int)b.row_min), __builtin_amdgcn_readfirstlane((int)b.row_max),
                         __builtin_amdgcn_readfirstlane((int)b.col_min), __builtin_amdgcn_readfirstlane((int)b.col_max),
                         __builtin_amdgcn_readfirstlane((int)b.ref_row), __builtin_amdgcn_readfirstlane((int)b.ref_col),
                         __builtin_amdgcn_readfirstlane((int)b.start_row), __builtin_amdgcn_readfirstlane((int)b.start_col) };
  }
};
// shared by the first-pass composite (tf_search.hip) and its row-persistent kernel (fp_row.hip)
const SiteTable *fps_device_sites(int device, int method);   // the per-(device, method) table in device memory; nullptr on failure
SearchArgs fps_search_args(const aomhip_search_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                           int bit_depth, bool want_cost_list);
struct FpfLegs {
  const int16_t *zmv; const int32_t *zerr;   // zero-MV leg on the last frame, per block
  const int16_t *gmv; const int32_t *gerr;   // zero-MV leg on the golden frame, per block (null without one)
  const int16_t *cmv; const int32_t *cerr;   // chained leg of this column, per row (the column-at-a-time form only)
  const uint32_t *err0, *raw, *gf0;           // get_prediction_error_bitdepth at 0,0: last frame, last source, golden
};
struct FpfCost { int type, error_per_bit; const int32_t *mvjcost, *mvcost0, *mvcost1; };
struct FpfOut { int16_t *best_mv, *full_mv; int32_t *motion_error, *gf_motion_error, *raw_motion_error; };
// the chained leg of a whole frame in one launch (fp_row.hip); AOMHIP_ERR_INVALID for block sizes it is not built for
int launch_fp_rows(aomhip_ctx *ctx, const aomhip_planes *src1, const aomhip_planes *last1, int bw, int bh, const aomhip_search_params *p,
                   const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks,
                   const FpfLegs &L, const int32_t *d_intra, int rows, int cols, int thr, int skip_zeromv, const FpfOut &out);
bool fp_rows_supported(int bw, int bh);
bool fp_rows_supported(int bw, int bh, int method);   // ... and the search method (the row kernel has the diamond / n-step body only)

}  // namespace aomhip

#endif  // AOMHIP_CSRC_FULLPEL_SEARCH_H_
