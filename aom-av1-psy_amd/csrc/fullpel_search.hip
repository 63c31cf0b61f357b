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

enum { kDiamond, kNstep, kNstep8, kClamped, kHex, kBigdia, kSquare, kFastHex, kFastDiamond, kFastBigdia, kVfastDiamond, kMethods };

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
  static const int lookup[kMethods] = { kDiamond, kNstep, kNstep8, kClamped, kHex, kBigdia, kSquare, kHex, kBigdia, kBigdia, kBigdia };
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

struct SearchArgs {
  int method, step_param, cost_type, sad_per_bit, error_per_bit, skip_sad;
  int run_mesh, prune_mesh, mesh_diff_thr, force_mesh_thresh, fine_interval;
  int mesh[8];
  const int *mvjcost, *mvcost0, *mvcost1;
  int bit_depth, want_cl;  // want_cl: the caller keeps a cost_list (changes pattern_search's last scale, :1077)
};

// SAD of the W x HH block, both operands from memory (used for the row-skipping form, whose geometry differs from
// the register-resident source units of the plain form)
template <typename T, int W, int HH>
__device__ __forceinline__ uint32_t group8_sad_mem(const T *sp, int sstride, const T *rp, int rstride, int l, bool active) {
  using G = G8<T, W, HH>;
  using L = typename G::L;
  uint32_t acc = 0;
  if (active) {
    for (int u = l; u < G::U; u += 8) {
      const int row = u / G::UPR, col = (u % G::UPR) * G::UE;
      const L a = *reinterpret_cast<const L *>(sp + (int64_t)row * sstride + col);
      const L b = *reinterpret_cast<const L *>(rp + (int64_t)row * rstride + col);
#pragma unroll
      for (int i = 0; i < G::UB / 4; ++i) acc = sadw<T>(a.v[i], b.v[i], acc);
    }
  }
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0xB1, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x4E, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x141, 0xf, 0xf, false);
  return acc;
}

constexpr int kFpsThreads = 256;  // 4 blocks (wavefronts) per workgroup
constexpr int kInvalidMv = -32768;  // INVALID_MV_ROW_COL (av1/common/mv.h:27)

#ifndef AOMHIP_FPS_WAVES
#define AOMHIP_FPS_WAVES 3   // waves per SIMD the register allocation aims at (profiles/r01_search_variants.md)
#endif

template <typename T, int W, int H>
__global__ __launch_bounds__(kFpsThreads, AOMHIP_FPS_WAVES) void full_pixel_search_kernel(
    PlaneView<T> src, PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks, int n_blocks,
    const SiteTable *__restrict__ sites, SearchArgs q, int16_t *__restrict__ out_mv, int32_t *__restrict__ out_cost,
    int32_t *__restrict__ out_cost_list, int16_t *__restrict__ out_second) {
  // the site table is read on the dependent chain of every step: keep it in LDS (1.7 KB), not behind a global load
  __shared__ SiteTable sS;
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(sites);
    uint32_t *d = reinterpret_cast<uint32_t *>(&sS);
    for (int i = threadIdx.x; i < (int)(sizeof(SiteTable) / 4); i += kFpsThreads) d[i] = g[i];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bi = blockIdx.x * (kFpsThreads / 64) + wave;
  if (bi >= n_blocks) return;
  const aomhip_search_block b = blocks[bi];
  const int sstride = src.stride, rstride = ref.stride;
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)b.by * sstride + b.bx;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)b.by * rstride + b.bx;
  const int shift = q.bit_depth == 10 ? 2 : q.bit_depth == 12 ? 4 : 0;  // _bits10 / _bits12 vtable wrappers
  const int g = lane >> 3, l = lane & 7;
  const int row_min = b.row_min, row_max = b.row_max, col_min = b.col_min, col_max = b.col_max;
  const int ref_row = b.ref_row, ref_col = b.ref_col;
  const int frr = (ref_row + 3 + (ref_row >= 0)) >> 3, frc = (ref_col + 3 + (ref_col >= 0)) >> 3;  // get_fullmv_from_mv
  const SiteTable &S = sS;
  bool skip = q.skip_sad != 0;

  typename G8<T, W, H>::L srcu[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1];
  group8_load_src<T, W, H>(sp, sstride, l, srcu);

  auto in_range = [&](int r, int c) { return c >= col_min && c <= col_max && r >= row_min && r <= row_max; };
  auto check_bounds = [&](int r, int c, int range) {
    return (r - range) >= row_min && (r + range) <= row_max && (c - range) >= col_min && (c + range) <= col_max;
  };
  auto mv_bits = [&](int dr, int dc) -> int {  // mv_cost (:250-254): joint + the two component tables, centre-addressed
    return q.mvjcost[(dc != 0) | ((dr != 0) << 1)] + q.mvcost0[dr] + q.mvcost1[dc];
  };
  auto sad_cost = [&](int row, int col) -> int {  // mvsad_err_cost_ (:310-339)
    const int dr = (row - frr) * 8, dc = (col - frc) * 8;
    if (q.cost_type == kCostEntropy) return (int)(((unsigned)mv_bits(dr, dc) * (unsigned)q.sad_per_bit + 256u) >> 9);
    const int lambda = q.cost_type == kCostL1Low ? 32 : q.cost_type == kCostL1Mid ? 15 : q.cost_type == kCostL1Hd ? 8 : 0;
    return (lambda * (iabsm(dr) + iabsm(dc))) >> 3;
  };
  auto var_cost = [&](int mrow, int mcol) -> int {  // mv_err_cost_ (:271-308)
    const int dr = mrow - ref_row, dc = mcol - ref_col;
    if (q.cost_type == kCostEntropy) return (int)(((int64_t)mv_bits(dr, dc) * q.error_per_bit + (1 << 13)) >> 14);
    const int lambda = q.cost_type == kCostL1Low ? 2 : q.cost_type == kCostL1Mid ? 0 : q.cost_type == kCostL1Hd ? 1 : 0;
    return (lambda * (iabsm(dr) + iabsm(dc))) >> 3;
  };
  // ms_params->sdf at (row, col), evaluated by this lane's group; inactive groups return 0
  auto sad_mine = [&](int row, int col, bool active) -> uint32_t {
    const T *rp = rbase + (int64_t)row * rstride + col;
    uint32_t v;
    if (skip)
      v = 2u * group8_sad_mem<T, W, (H / 2)>(sp, 2 * sstride, rp, 2 * rstride, l, active);   // sad_skip (sad.c:65-69)
    else
      v = group8_sad<T, W, H>(sp, sstride, rp, rstride, l, active, srcu);
    return v >> shift;
  };
  auto sad_one = [&](int row, int col) -> uint32_t {
    const uint32_t v = sad_mine(row, col, g == 0);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
  };
  auto var_cost_at = [&](int row, int col) -> int {  // get_mvpred_var_cost (:645-664): vf(src, ref) + mv_err_cost_
    uint32_t sse;
    uint32_t v = group16_variance<T, W, H, false>(rbase + (int64_t)row * rstride + col, rstride, 0, 0, sp, sstride,
                                                  /*a_minus_b=*/false, q.bit_depth, lane & 15, lane < 16, &sse);
    v = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    return (int)v + var_cost(row * 8, col * 8);
  };

  int second_row = kInvalidMv, second_col = kInvalidMv;
  int cl[5];
  auto set_cl = [&](int i, int v) {
#pragma unroll
    for (int k = 0; k < 5; ++k) cl[k] = (k == i) ? v : cl[k];
  };

  // ---- diamond_search_sad (:1299-1416) on the site table
  auto diamond = [&](int search_step, int *num00, int *orow, int *ocol) -> int {
    int row = min(max((int)b.start_row, row_min), row_max), col = min(max((int)b.start_col, col_min), col_max);
    const int tot_steps = S.num_search_steps - search_step;
    *num00 = 0;
    uint32_t bestsad = sad_one(row, col) + (uint32_t)sad_cost(row, col);
    int is_off_center = 0;
    int next_step_size = tot_steps > 2 ? S.radius[tot_steps - 2] : 1;
    for (int step = tot_steps - 1; step >= 0; --step) {
      const int nper = S.searches_per_step[step];
      int best_site = 0;
      if (step > 0) next_step_size = S.radius[step - 1];
      for (int base = 1; base <= nper; base += 8) {
        const int n = min(8, nper - base + 1);
        const int mi = min(base + g, 16);
        const int mr = row + S.mv[step][mi][0], mc = col + S.mv[step][mi][1];
        const bool inr = g < n && in_range(mr, mc);
        const uint32_t mine = sad_mine(mr, mc, inr);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j < n) {
            const uint32_t sad = (uint32_t)__builtin_amdgcn_readlane((int)mine, j * 8);
            const int ok = __builtin_amdgcn_readlane((int)inr, j * 8);
            if (ok && sad < bestsad) {
              const uint32_t thissad = sad + (uint32_t)sad_cost(row + S.mv[step][base + j][0], col + S.mv[step][base + j][1]);
              if (thissad < bestsad) {
                bestsad = thissad;
                best_site = base + j;
              }
            }
          }
        }
      }
      if (best_site != 0) {
        second_row = row;  // *second_best_mv = *best_mv
        second_col = col;
        row += S.mv[step][best_site][0];
        col += S.mv[step][best_site][1];
        is_off_center = 1;
      }
      if (is_off_center == 0) (*num00)++;
      if (best_site == 0) {
        while (next_step_size == S.radius[step] && step > 2) {
          ++(*num00);
          --step;
          next_step_size = S.radius[step - 1];
        }
      }
    }
    *orow = row;
    *ocol = col;
    return (int)bestsad;
  };

  // ---- calc_int_sad_list (:768-821): centre, left, bottom, right, top
  auto sad_cost_list = [&](int br, int bc, bool has_sad) {
    const int nr = g == 2 ? 1 : g == 4 ? -1 : 0, nc = g == 1 ? -1 : g == 3 ? 1 : 0;
    if (!has_sad) {
      const bool inr = g < 5 && in_range(br + nr, bc + nc);
      const uint32_t mine = sad_mine(br + nr, bc + nc, inr);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int ok = __builtin_amdgcn_readlane((int)inr, j * 8);
        const int sad = __builtin_amdgcn_readlane((int)mine, j * 8);
        cl[j] = (j == 0 || ok) ? sad : INT_MAX;
      }
    }
    cl[0] += sad_cost(br, bc);
    if (cl[1] != INT_MAX) cl[1] += sad_cost(br, bc - 1);
    if (cl[2] != INT_MAX) cl[2] += sad_cost(br + 1, bc);
    if (cl[3] != INT_MAX) cl[3] += sad_cost(br, bc + 1);
    if (cl[4] != INT_MAX) cl[4] += sad_cost(br - 1, bc);
  };

  // ---- full_pixel_diamond (:1421-1470): the first search at step_param, then restarts at step_param + n that are skipped
  //      while the previous one reported it would have stayed on the centre (num00).  One loop = one inlined copy of
  //      the search body.
  auto full_pixel_diamond = [&](int step_param, int *obr, int *obc) -> int {
    int n = 0, num00 = 0, br = 0, bc = 0, bestsme = INT_MAX;
    const int further_steps = S.num_search_steps - 1 - step_param;
    bool first = true;
    for (;;) {
      bool run_it = true;
      int sstep = step_param;
      if (!first) {
        if (n >= further_steps) break;
        ++n;
        if (num00) {
          --num00;
          run_it = false;
        } else {
          sstep = step_param + n;
        }
      }
      if (run_it) {
        int t00, tr, tc;
        int sme = diamond(sstep, &t00, &tr, &tc);
        if (sme < INT_MAX) sme = var_cost_at(tr, tc);
        if (first) {
          bestsme = sme;
          br = tr;
          bc = tc;
          n = t00;
        } else {
          num00 = t00;
          if (sme < bestsme) {
            bestsme = sme;
            br = tr;
            bc = tc;
          }
        }
      }
      first = false;
    }
    sad_cost_list(br, bc, false);
    *obr = br;
    *obc = bc;
    return bestsme;
  };

  // ---- pattern_search (:998-1226).  `packed`: up to 8 candidate indices, 4 bits each, in visiting order.
  uint32_t p_bestsad, p_raw;
  auto pat_eval = [&](int br, int bc, int stage, uint32_t packed, int n, bool report_pos, bool use_cl) -> int {
    const int mi = (int)((packed >> (4 * g)) & 15u);
    const int mr = br + S.mv[stage][mi][0], mc = bc + S.mv[stage][mi][1];
    const bool inr = g < n && in_range(mr, mc);
    const uint32_t mine = sad_mine(mr, mc, inr);
    int best = -1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j < n) {
        const int index = (int)((packed >> (4 * j)) & 15u);
        const int ok = __builtin_amdgcn_readlane((int)inr, j * 8);
        const uint32_t sad = (uint32_t)__builtin_amdgcn_readlane((int)mine, j * 8);
        if (!ok) {
          if (use_cl) set_cl(index + 1, INT_MAX);
        } else {
          if (use_cl) set_cl(index + 1, (int)sad);
          if (sad < p_bestsad) {  // update_mvs_and_sad (:839-858)
            const uint32_t tot = sad + (uint32_t)sad_cost(br + S.mv[stage][index][0], bc + S.mv[stage][index][1]);
            if (tot < p_bestsad) {
              p_raw = sad;
              p_bestsad = tot;
              best = report_pos ? j : index;
            }
          }
        }
      }
    }
    return best;
  };
  // candidates a full scan of one scale visits: inside the limits the reference's 4-at-a-time path drops the tail of
  // a 6-candidate scale (calc_sad_update_bestmv is called with num_candidates = n % 4 and cand_start = 4, :1056-1060)
  auto scan_count = [&](int br, int bc, int stage, int n) -> int {
    if (!check_bounds(br, bc, 1 << stage)) return n;
    const int groups = 4 * (n >> 2);
    return (n % 4) > groups ? (n % 4) : groups;
  };
  auto chk_of = [](int k, int n) -> uint32_t {  // next_chkpts_indices: k - 1, k, k + 1 (cyclic)
    const int a = (k == 0) ? n - 1 : k - 1, c = (k == n - 1) ? 0 : k + 1;
    return (uint32_t)a | ((uint32_t)k << 4) | ((uint32_t)c << 8);
  };
  auto pattern_search = [&](int search_step, bool do_init_search, int *obr, int *obc) -> int {
    constexpr uint32_t kAll = 0x76543210u;
    const bool last_is_4 = S.searches_per_step[0] == 4;
    int k = -1, st;
    search_step = min(search_step, 10);
    int best_init_s = 10 - search_step;
    int br = min(max((int)b.start_row, row_min), row_max), bc = min(max((int)b.start_col, col_min), col_max);
#pragma unroll
    for (int i = 0; i < 5; ++i) cl[i] = INT_MAX;
    bool has_sad = false;
    p_raw = sad_one(br, bc);
    p_bestsad = p_raw + (uint32_t)sad_cost(br, bc);
    if (do_init_search) {
      st = best_init_s;
      best_init_s = -1;
      for (int t = 0; t <= st; ++t) {
        const int best_site = pat_eval(br, bc, t, kAll, scan_count(br, bc, t, S.searches_per_step[t]), false, false);
        if (best_site == -1) continue;
        best_init_s = t;
        k = best_site;
      }
      if (best_init_s != -1) {
        br += S.mv[best_init_s][k][0];
        bc += S.mv[best_init_s][k][1];
      }
    }
    if (best_init_s != -1) {
      const int last_s = (last_is_4 && q.want_cl) ? 1 : 0;
      int best_site = -1;
      st = best_init_s;
      for (; st >= last_s; st--) {
        const int nc = S.searches_per_step[st];
        if (!do_init_search || st != best_init_s) {
          best_site = pat_eval(br, bc, st, kAll, scan_count(br, bc, st, nc), false, false);
          if (best_site == -1) continue;
          br += S.mv[st][best_site][0];
          bc += S.mv[st][best_site][1];
          k = best_site;
        }
        do {
          const uint32_t chk = chk_of(k, nc);
          best_site = pat_eval(br, bc, st, chk, 3, true, false);
          if (best_site != -1) {
            k = (int)((chk >> (4 * best_site)) & 15u);
            br += S.mv[st][k][0];
            bc += S.mv[st][k][1];
          }
        } while (best_site != -1);
      }
      if (st == 0) {
        cl[0] = (int)p_raw;
        has_sad = true;
        if (!do_init_search || st != best_init_s) {
          best_site = pat_eval(br, bc, 0, kAll, 4, false, true);
          if (best_site != -1) {
            br += S.mv[0][best_site][0];
            bc += S.mv[0][best_site][1];
            k = best_site;
          }
        }
        while (best_site != -1) {
          const uint32_t chk = chk_of(k, 4);
          const int carried = cl[0];
          cl[1] = cl[2] = cl[3] = cl[4] = INT_MAX;
          set_cl(((k + 2) % 4) + 1, carried);
          cl[0] = (int)p_raw;
          best_site = pat_eval(br, bc, 0, chk, 3, true, true);
          if (best_site != -1) {
            k = (int)((chk >> (4 * best_site)) & 15u);
            br += S.mv[0][k][0];
            bc += S.mv[0][k][1];
          }
        }
      }
    }
    *obr = br;
    *obc = bc;
    sad_cost_list(br, bc, has_sad);
    return var_cost_at(br, bc);
  };

  // ---- exhaustive_mesh_search (:1474-1543), 8 raster-consecutive candidates per batch
  auto mesh_pass = [&](int *row0, int *col0, int range, int step) -> int {
    const int srow = min(max(*row0, row_min), row_max), scol = min(max(*col0, col_min), col_max);
    int best_row = srow, best_col = scol;
    uint32_t best_sad = sad_one(srow, scol) + (uint32_t)sad_cost(srow, scol);
    const int start_row = max(-range, row_min - srow), start_col = max(-range, col_min - scol);
    const int end_row = min(range, row_max - srow), end_col = min(range, col_max - scol);
    const int nrows = end_row >= start_row ? (end_row - start_row) / step + 1 : 0;
    int ncols;
    if (step > 1) {
      ncols = end_col >= start_col ? (end_col - start_col) / step + 1 : 0;
    } else {  // four columns per sdx4df call; the tail group `for (i = 0; i < end_col - c; ++i)` stops before end_col
      const int cnt = end_col - start_col + 1;
      ncols = cnt <= 0 ? 0 : (cnt % 4 == 0 ? cnt : cnt - 1);
    }
    const int total = nrows * ncols;
    for (int q0 = 0; q0 < total; q0 += 8) {
      const int n = min(8, total - q0);
      const int mq = min(q0 + g, total - 1);
      const int mri = mq / ncols, mci = mq - mri * ncols;
      const uint32_t mine = sad_mine(srow + start_row + mri * step, scol + start_col + mci * step, g < n);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < n) {
          const uint32_t sad = (uint32_t)__builtin_amdgcn_readlane((int)mine, j * 8);
          if (sad < best_sad) {
            const int ri = (q0 + j) / ncols, ci = (q0 + j) - ri * ncols;
            const int row = srow + start_row + ri * step, col = scol + start_col + ci * step;
            const uint32_t tot = sad + (uint32_t)sad_cost(row, col);
            if (tot < best_sad) {
              best_sad = tot;
              second_row = best_row;
              second_col = best_col;
              best_row = row;
              best_col = col;
            }
          }
        }
      }
    }
    *row0 = best_row;
    *col0 = best_col;
    return (int)best_sad;
  };
  // ---- full_pixel_exhaustive (:1547-1617)
  auto full_pixel_exhaustive = [&](int srow, int scol, int *obr, int *obc) -> int {
    int interval = q.mesh[1], range = q.mesh[0];
    int br = srow, bc = scol;
    *obr = br;
    *obc = bc;
    if (range < 7 || range > 256 || interval < 1 || interval > range) return INT_MAX;
    const int div = range / interval;
    range = max(range, (5 * max(iabsm(br), iabsm(bc))) / 4);
    range = min(range, 256);
    interval = max(interval, range / div);
    if (q.fine_interval) interval = min(interval, 4);
    int bestsme = INT_MAX;
    const bool more = interval > 1 && range > 7;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {  // pass 0 with the grown first pattern, then (if it was coarse) the narrowing passes
      if (k > 0) {
        range = q.mesh[2 * k];
        interval = q.mesh[2 * k + 1];
      }
      bestsme = mesh_pass(&br, &bc, range, interval);
      if (!more || (k > 0 && interval == 1)) break;
    }
    if (bestsme < INT_MAX) bestsme = var_cost_at(br, bc);
    sad_cost_list(br, bc, false);
    *obr = br;
    *obc = bc;
    return bestsme;
  };

  // ---- av1_full_pixel_search (:1693-1832); the downsampled-SAD re-check restarts it once with the full SAD
  constexpr int kLog2AreaMi = (W == 4 ? 0 : __builtin_ctz(W) - 2) + (H == 4 ? 0 : __builtin_ctz(H) - 2);
  int var = 0, br = kInvalidMv, bc = kInvalidMv;
#pragma unroll 1
  for (int attempt = 0; attempt < 2; ++attempt) {
    second_row = second_col = kInvalidMv;
    const int m = q.method, sp_ = q.step_param;
    const bool is_pattern = m >= kHex;  // HEX, BIGDIA, SQUARE and the four FAST_ forms
    if (is_pattern) {
      const int floor_step = m == kFastBigdia ? 8 : m == kVfastDiamond ? 10 : (m == kFastDiamond || m == kFastHex) ? 9 : 0;
      var = pattern_search(max(floor_step, sp_), /*do_init_search=*/m == kHex || m == kSquare || m == kBigdia, &br, &bc);
    } else {
      var = full_pixel_diamond(sp_, &br, &bc);
    }

    int run_mesh = q.run_mesh;
    if (!run_mesh && (m == kNstep || m == kNstep8)) {
      const int thr = q.force_mesh_thresh >> (10 - kLog2AreaMi);
      if (var > thr) run_mesh = 1;
    }
    if (q.prune_mesh) {
      const int d = max(iabsm((int)b.start_row - br), iabsm((int)b.start_col - bc));
      if (d <= q.mesh_diff_thr) run_mesh = 0;
    }
    if (skip) {  // ms_params->sdf != vfp->sdf
      const int skip_sad = (int)sad_one(br, bc);
      skip = false;
      const int sad = (int)sad_one(br, bc);
      if (sad > (1 << kLog2AreaMi) && iabsm(skip_sad - sad) * 10 >= max(sad, 1) * 9) continue;  // redo with the full SAD
      skip = true;
    }
    if (run_mesh) {
      int er, ec;
      const int var_ex = full_pixel_exhaustive(br, bc, &er, &ec);
      if (var_ex < var) {
        var = var_ex;
        br = er;
        bc = ec;
      }
    }
    break;
  }
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)br;
    out_mv[2 * bi + 1] = (int16_t)bc;
    out_cost[bi] = var;
    if (out_cost_list) {
#pragma unroll
      for (int k = 0; k < 5; ++k) out_cost_list[5 * bi + k] = cl[k];
    }
    if (out_second) {
      out_second[2 * bi] = (int16_t)second_row;
      out_second[2 * bi + 1] = (int16_t)second_col;
    }
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
  {
    SiteTable h;
    build_sites(p->search_method, &h);
    if (p->step_param >= h.num_search_steps) {
      set_error("aomhip_full_pixel_search_batch: step_param %d >= %d search steps", p->step_param, h.num_search_steps);
      return AOMHIP_ERR_INVALID;
    }
  }
  SearchArgs q;
  q.method = p->search_method; q.step_param = p->step_param; q.cost_type = p->mv_cost_type;
  q.sad_per_bit = p->sad_per_bit; q.error_per_bit = p->error_per_bit; q.skip_sad = p->use_downsampled_sad != 0;
  q.run_mesh = p->run_mesh_search; q.prune_mesh = p->prune_mesh_search; q.mesh_diff_thr = p->mesh_search_mv_diff_threshold;
  q.force_mesh_thresh = p->force_mesh_thresh; q.fine_interval = p->fine_search_interval;
  for (int i = 0; i < 8; ++i) q.mesh[i] = p->mesh_patterns[i];
  q.mvjcost = d_mvjcost; q.mvcost0 = d_mvcost_row; q.mvcost1 = d_mvcost_col;
  q.bit_depth = src->bit_depth; q.want_cl = d_cost_list != nullptr;
  const dim3 grid((n_blocks + 3) / 4), block(kFpsThreads);
#define X(W, H)                                                                                                       \
  if (bw == W && bh == H) {                                                                                           \
    if (src->bit_depth == 8)                                                                                          \
      hipLaunchKernelGGL((full_pixel_search_kernel<uint8_t, W, H>), grid, block, 0, ctx->stream, view_of<uint8_t>(*src), \
                         view_of<uint8_t>(*ref), frame, d_blocks, n_blocks, d_sites, q, d_best_mv, d_best_cost,       \
                         d_cost_list, d_second_best_mv);                                                              \
    else                                                                                                              \
      hipLaunchKernelGGL((full_pixel_search_kernel<uint16_t, W, H>), grid, block, 0, ctx->stream,                      \
                         view_of<uint16_t>(*src), view_of<uint16_t>(*ref), frame, d_blocks, n_blocks, d_sites, q,     \
                         d_best_mv, d_best_cost, d_cost_list, d_second_best_mv);                                      \
    AOMHIP_LAUNCH_CHECK();                                                                                            \
    return AOMHIP_OK;                                                                                                 \
  }
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

}  // extern "C"
