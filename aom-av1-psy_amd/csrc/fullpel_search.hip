// libaomhip -- av1_full_pixel_search on the device: every SEARCH_METHODS value, gfx950.
//
// What the reference does per block (av1/encoder/mcomp.c:1693-1832): dispatch on search_method to
// pattern_search (:998-1226; HEX / BIGDIA / SQUARE and the FAST_ forms) or full_pixel_diamond (:1421-1470; DIAMOND,
// CLAMPED_DIAMOND, NSTEP, NSTEP_8PT site tables), then the mesh follow-up rules, the downsampled-SAD re-check
// (:1777-1810) and full_pixel_exhaustive (:1547-1617), returning best_mv, the variance + MV cost, cost_list[5]
// (calc_int_sad_list, :768-821) and second_best_mv.
//
// Mapping: ONE WAVEFRONT owns one block for the whole (greedy, data-dependent) search.  Its 64 lanes form 8 groups of
// 8; a "batch" evaluates up to 8 candidate positions at once (one group per candidate: 16-byte row units,
// v_sad_u8 / v_sad_u16, DPP fold), the 8 SADs are broadcast with v_readlane, and every lane then replays the
// reference's scalar comparisons IN CANDIDATE ORDER -- so ties, the second-best MV, the cost list and the
// reference's own quirks (e.g. the 6-candidate HEX scan that only visits 4 candidates inside the limits,
// :1046-1066) come out exactly as in the scalar loops.  Site tables are built on the host with the reference's
// rules and read through the scalar cache.  MV cost: NONE, the three L1 forms, and MV_COST_ENTROPY from device tables.
#include <climits>
#include <mutex>

#include "fullpel_search.h"

namespace aomhip {

static void put(SiteTable *s, int stage, int idx, int row, int col) {
  s->mv[stage][idx][0] = (int16_t)row;
  s->mv[stage][idx][1] = (int16_t)col;
}
// centre, four axis points, then the (+-radius, +-t) / (+-t, +-radius) pairs (mcomp.c:366-370,405-419,452-466)
static void ring(SiteTable *s, int stage, int radius, int t, int npts) {
  const int m[13][2] = { { 0, 0 },        { -radius, 0 }, { radius, 0 },  { 0, -radius }, { 0, radius },
                         { -radius, -t }, { radius, t },  { -t, radius }, { t, -radius }, { -radius, t },
                         { radius, -t },  { t, radius },  { -t, -radius } };
  for (int i = 0; i <= npts; ++i) put(s, stage, i, m[i][0], m[i][1]);
  s->searches_per_step[stage] = npts;
  s->radius[stage] = radius;
}

static void build_sites(int method, SiteTable *s) {
  static const int lookup[kMethods] = { kDiamond, kNstep, kNstep8, kClamped, kHex, kBigdia, kSquare, kHex, kBigdia, kBigdia, kBigdia, kNstepFpf };
  memset(s, 0, sizeof(*s));
  const int shape = lookup[method];
  if (shape == kDiamond || shape == kClamped) {  // av1_init_dsmotion_compensation (:350-389)
    const int level = shape == kClamped;
    int stage = 10, n = 0;
    for (int radius = level ? 256 : 1024; radius > 0; --stage, ++n) {
      ring(s, stage, radius, radius, 8);
      if (!level || stage < 9) radius /= 2;
    }
    s->num_search_steps = n;
  } else if (shape == kNstepFpf) {  // av1_init_motion_fpf (:391-431): the first-pass table
    int stage = 10, n = 0;
    for (int radius = 1024; radius > 0; radius /= 2, --stage, ++n) {
      int t = (int)(0.41 * radius);
      if (t < 1) t = 1;
      ring(s, stage, radius, t, radius == 1 ? 8 : 12);
    }
    s->num_search_steps = n;
  } else if (shape == kNstep || shape == kNstep8) {  // av1_init_motion_compensation_nstep (:436-474)
    const int level = shape == kNstep8, stages = level ? 16 : 15;
    int radius = 1;
    for (int stage = 0; stage < stages; ++stage) {
      int t = (int)(0.41 * radius), npts = 12;
      if (t < 1) t = 1;
      if (radius <= 5 || level) {
        t = radius;
        npts = 8;
      }
      ring(s, stage, radius, t, npts);
      if (stage < 12) {
        const double grown = radius * 1.5 + 0.5;
        radius = (int)(grown > radius + 1 ? grown : radius + 1);
      }
    }
    s->num_search_steps = stages;
  } else {  // the pattern shapes (:476-633): scale i reaches 2^i
    int r = 1;
    for (int i = 0; i < 11; ++i, r *= 2) {
      const int h = r / 2;
      int n = 8;
      if (shape == kSquare || (shape == kHex && i == 0)) {
        const int m[8][2] = { { -r, -r }, { 0, -r }, { r, -r }, { r, 0 }, { r, r }, { 0, r }, { -r, r }, { -r, 0 } };
        for (int j = 0; j < 8; ++j) put(s, i, j, m[j][0], m[j][1]);
      } else if (shape == kHex) {
        const int m[6][2] = { { -h, -r }, { h, -r }, { r, 0 }, { h, r }, { -h, r }, { -r, 0 } };
        for (int j = 0; j < 6; ++j) put(s, i, j, m[j][0], m[j][1]);
        n = 6;
      } else if (i == 0) {
        const int m[4][2] = { { 0, -1 }, { 1, 0 }, { 0, 1 }, { -1, 0 } };
        for (int j = 0; j < 4; ++j) put(s, i, j, m[j][0], m[j][1]);
        n = 4;
      } else {
        const int m[8][2] = { { -h, -h }, { 0, -r }, { h, -h }, { r, 0 }, { h, h }, { 0, r }, { -h, h }, { -r, 0 } };
        for (int j = 0; j < 8; ++j) put(s, i, j, m[j][0], m[j][1]);
      }
      s->searches_per_step[i] = n;
      s->radius[i] = r;
    }
    s->num_search_steps = 11;
  }
}


// Site tables live in device memory for the life of the process: one copy per (device, method), built on first use.
static const SiteTable *device_sites(int device, int method) {
  static std::mutex mu;
  static SiteTable *tab[16][kMethods] = {};
  if (device < 0 || device >= 16) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!tab[device][method]) {
    SiteTable h;
    build_sites(method, &h);
    SiteTable *d = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d), sizeof(SiteTable)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, &h, sizeof(SiteTable), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(d);
      return nullptr;
    }
    tab[device][method] = d;
  }
  return tab[device][method];
}

const SiteTable *fps_device_sites(int device, int method) { return device_sites(device, method); }
SearchArgs fps_search_args(const aomhip_search_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                           int bit_depth, bool want_cost_list) {
  SearchArgs q;
  q.method = p->search_method; q.step_param = p->step_param; q.cost_type = p->mv_cost_type;
  q.sad_per_bit = p->sad_per_bit; q.error_per_bit = p->error_per_bit; q.skip_sad = p->use_downsampled_sad != 0;
  q.run_mesh = p->run_mesh_search; q.prune_mesh = p->prune_mesh_search; q.mesh_diff_thr = p->mesh_search_mv_diff_threshold;
  q.force_mesh_thresh = p->force_mesh_thresh; q.fine_interval = p->fine_search_interval;
  for (int i = 0; i < 8; ++i) q.mesh[i] = p->mesh_patterns[i];
  q.mvjcost = d_mvjcost; q.mvcost0 = d_mvcost_row; q.mvcost1 = d_mvcost_col;
  q.bit_depth = bit_depth; q.want_cl = want_cost_list;
  q.resume = 0;
  return q;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_search_sites(int search_method, int *num_search_steps, int searches_per_step[22], int radius[22],
                        int16_t sites[22][17][2]) {
  if (search_method < 0 || search_method >= kMethods || !num_search_steps || !searches_per_step || !radius || !sites) {
    set_error("aomhip_search_sites: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  SiteTable s;
  build_sites(search_method, &s);
  *num_search_steps = s.num_search_steps;
  memcpy(searches_per_step, s.searches_per_step, sizeof(s.searches_per_step));
  memcpy(radius, s.radius, sizeof(s.radius));
  memcpy(sites, s.mv, sizeof(s.mv));
  return AOMHIP_OK;
}

int aomhip_full_pixel_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw,
                                   int bh, const aomhip_search_params *p, const int32_t *d_mvjcost,
                                   const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                   const aomhip_search_block *d_blocks, int n_blocks, int16_t *d_best_mv,
                                   int32_t *d_best_cost, int32_t *d_cost_list, int16_t *d_second_best_mv) {
  if (!ctx || !src || !ref || !p || !src->base || !ref->base || (n_blocks > 0 && !d_blocks) || n_blocks < 0 || frame < 0 ||
      frame >= src->n_frames || frame >= ref->n_frames || !valid_block(bw, bh) || (src->bit_depth == 8) != (ref->bit_depth == 8) ||
      !d_best_mv || !d_best_cost) {
    set_error("aomhip_full_pixel_search_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (p->search_method < 0 || p->search_method >= kMethods || p->step_param < 0 || p->mv_cost_type < 0 ||
      p->mv_cost_type > kCostNone) {
    set_error("aomhip_full_pixel_search_batch: invalid search method / step / cost type");
    return AOMHIP_ERR_INVALID;
  }
  if (p->mv_cost_type == kCostEntropy && (!d_mvjcost || !d_mvcost_row || !d_mvcost_col)) {
    set_error("aomhip_full_pixel_search_batch: MV_COST_ENTROPY needs the three cost tables");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const SiteTable *d_sites = device_sites(ctx->device, p->search_method);
  if (!d_sites) {
    set_error("aomhip_full_pixel_search_batch: could not place the site table on device %d", ctx->device);
    return AOMHIP_ERR_HIP;
  }
  int reach;   // of the whole search around its start MV: the sum of the radii from step_param down (the LDS window takes what it can)
  {
    SiteTable h;
    build_sites(p->search_method, &h);
    // (step_param == num_search_steps is what av1_single_motion_search passes for search_range < 1, motion_search_facade.c:234-236: the diamond
    // family then measures the start position only; pattern_search asserts search_step < num_search_steps)
    const bool diamond_family = p->search_method < kHex || p->search_method == kNstepFpf;
    if (p->step_param > h.num_search_steps || (p->step_param == h.num_search_steps && !diamond_family)) {
      set_error("aomhip_full_pixel_search_batch: step_param %d >= %d search steps", p->step_param, h.num_search_steps);
      return AOMHIP_ERR_INVALID;
    }
    reach = 0;
    for (int st = h.num_search_steps - 1 - p->step_param; st >= 0; --st) reach += h.radius[st];
  }
  const SearchArgs q = fps_search_args(p, d_mvjcost, d_mvcost_row, d_mvcost_col, src->bit_depth, d_cost_list != nullptr);
  return (src->bit_depth == 8 ? launch_fps_u8 : launch_fps_u16)(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, d_sites, q, reach, d_best_mv,
                                                               d_best_cost, d_cost_list, d_second_best_mv);
}

}  // extern "C"
