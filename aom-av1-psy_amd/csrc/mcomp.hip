// Device-side motion search on gfx950: full-pel diamond search + bilinear sub-pel refinement over a batch
// of blocks (av1/encoder/mcomp.c).  One wavefront owns one block for the whole (sequential, greedy) search;
// the data parallelism is across blocks and across the 8 sites of a diamond step.
//
//   fullpel_diamond_kernel : full_pixel_diamond (mcomp.c:1421-1470) = diamond_search_sad (:1299-1416) with the
//       restart loop and the final get_mvpred_var_cost (:645-664).  The 64 lanes form 8 groups of 8; each
//       group evaluates one site of the step (aom_sadWxH through the vtable's sdf / sdx4df, with the 10/12-bit
//       >>2 / >>4 wrappers of encoder_utils.h:155-208), the 8 results are broadcast and every lane replays the
//       reference's two-stage comparison `if (sad < best) { sad += cost; if (sad < best) ... }` in site order,
//       so ties and cost effects resolve exactly as in the scalar loop.
//   subpel_bilinear_kernel : av1_find_best_sub_pixel_tree_pruned_more (:2844-2929) with cost_list == NULL on an
//       unscaled reference: setup_center_error (:2718-2778), two_level_checks_fast (:2503-2624) at 1/2, 1/4,
//       1/8 pel, each candidate = one aom_sub_pixel_varianceWxH evaluated by all 64 lanes.
// MV cost: MV_COST_NONE and the three L1 types (mcomp.c:236-244,271-339); the entropy-table type is refused.
#include <climits>

#include "common.h"
#include "search_device.h"
#include "search_window.h"

namespace aomhip {

// Phase timing of the cell kernel (kernel experiments only: -DAOMHIP_CELL_PROF; profiles/r04_search_cell.md): shader cycles per wavefront
// summed over the launch -- [0] waves, [1] prologue + staging, [2] LDS rounds, [3] their count, [4] global rounds, [5] their count,
// [6] variances, [7] their count, [8] whole wavefront
#ifdef AOMHIP_CELL_PROF
__device__ unsigned int g_cell_prof[40960 * 16];   // one record of plain stores per block (atomics distort the memory phases)
#define CELL_T() __builtin_readcyclecounter()
#define CELL_ADD(i, v) do { if (lane == 0 && bi < 40960) g_cell_prof[bi * 16 + (i)] = (unsigned int)(v); } while (0)
#else
#define CELL_T() 0ull
#define CELL_ADD(i, v) do { } while (0)
#endif

// WAVES blocks (wavefronts) per workgroup.  CELL: the workgroup's blocks share a reference window staged once in LDS
// (search_window.h); a round whose 8 sites lie inside it reads LDS, any other round reads the plane.
//
// A round of the search is a dependent chain -- addresses, 8 site SADs, reduction, costs, arg-min, new centre -- and the kernel is
// bound by the ISSUE of that chain's instructions (profiles/r04_search_cell.md: a round costs the same 1 300 - 1 500 cycles per wavefront
// from LDS as from L2, at 4 wavefronts per SIMD), so the round is written for instruction count: radius products on the 24-bit
// multiplier, the limits tested once per round on the scalar unit when the whole diamond lies inside them, |d| through v_sad_u32 on
// biased coordinates, two SAD accumulators, the winning site decoded from two packed tables instead of compare chains, the start
// position's SAD kept across the restarts (every run of full_pixel_diamond starts at the same clamped MV) and the final variance
// kept while consecutive runs end on the same MV.
// CLAMPED: the CLAMPED_DIAMOND table (radii capped at 256, equal consecutive radii skipped) -- a template argument: as a run-time `level` its
// tests sat in every round of the plain diamond (about 10 of the round's ~65 scalar instructions; the kernel's scalar and vector counts are
// about even, profiles/r05_inner_loop_pmc.json)
template <typename T, int W, int H, int WAVES, bool CELL, bool CLAMPED>
__global__ __launch_bounds__(WAVES * 64, CELL ? 8 : 5) void fullpel_diamond_kernel(
    PlaneView<T> src, PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks, int n_blocks, CellMap cm,
    int level, int step_param, int cost_type, int bit_depth, int16_t *__restrict__ out_mv, int32_t *__restrict__ out_cost) {
  extern __shared__ uint32_t cell_lds[];
  using G = G8<T, W, H>;
  constexpr int ES = (int)sizeof(T);
  constexpr int NU = G::KEEP ? G::PER_LANE : 1;
  [[maybe_unused]] const unsigned long long t_begin = CELL_T();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (uniform: the block record and everything derived from it -- centre, limits, step state -- then lives in SGPRs and the search loops branch on the scalar unit)
  int bi;
  if constexpr (CELL) bi = cell_block_index<WAVES>((int)xcd_chunked_index(blockIdx.x, (unsigned)cm.n_cells), wave, n_blocks);
  else bi = (int)blockIdx.x * WAVES + wave < n_blocks ? (int)blockIdx.x * WAVES + wave : -1;
  const bool have = bi >= 0;
  if constexpr (!CELL) {
    if (!have) return;
  }
  const aomhip_search_block b = blocks[have ? bi : 0];
  const int row_min = __builtin_amdgcn_readfirstlane((int)b.row_min), row_max = __builtin_amdgcn_readfirstlane((int)b.row_max);
  const int col_min = __builtin_amdgcn_readfirstlane((int)b.col_min), col_max = __builtin_amdgcn_readfirstlane((int)b.col_max);
  // clamp_fullmv (readfirstlane: the compiler packs the two 16-bit clamps into v_pk_min / max_i16 -- VALU only -- and everything derived
  // from a VGPR, the whole search state, then follows it into the vector unit and under exec masks)
  const int start_row = __builtin_amdgcn_readfirstlane(min(max((int)b.start_row, row_min), row_max));
  const int start_col = __builtin_amdgcn_readfirstlane(min(max((int)b.start_col, col_min), col_max));
  const T *rframe = ref.origin + (int64_t)frame * ref.frame_stride;
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)b.by * src.stride + b.bx;
  const T *rbase = rframe + (int64_t)b.by * ref.stride + b.bx;
  [[maybe_unused]] CellWin cw{ 0, 0, 0, 0, 0 };
  if constexpr (CELL) {
    if (cm.win_r >= 0) cw = stage_cell_window<T, W, H, WAVES>(cm, rframe, ref.stride, have, b.bx + start_col, b.by + start_row, wave, cell_lds, t_begin, bi);
    if (!have) return;   // (behind the window's barriers)
  }
  [[maybe_unused]] unsigned long long t_lds = 0, t_glob = 0, t_var = 0, n_lds = 0, n_glob = 0, n_var = 0;
  CELL_ADD(0, 1);
  CELL_ADD(1, CELL_T() - t_begin);
  const CostCtx cc{ cost_type, b.ref_row, b.ref_col };
  const int shift = bit_depth == 10 ? 2 : bit_depth == 12 ? 4 : 0;  // vtable wrappers for highbd SAD
  const int lambda = cost_type == kCostL1Low ? 32 : cost_type == kCostL1Mid ? 15 : cost_type == kCostL1Hd ? 8 : 0;  // mvsad_err_cost_: (lambda * 8 d) >> 3
  const int frr = __builtin_amdgcn_readfirstlane((b.ref_row + 3 + (b.ref_row >= 0)) >> 3);  // GET_MV_RAWPEL (readfirstlane: as for the start MV)
  const int frc = __builtin_amdgcn_readfirstlane((b.ref_col + 3 + (b.ref_col >= 0)) >> 3);
  const int g = lane >> 3, l = lane & 7;
  const int dr = (g == 0 || g == 4 || g == 6) ? -1 : (g == 1 || g == 5 || g == 7) ? 1 : 0;   // site order of
  const int dc = (g == 2 || g == 4 || g == 7) ? -1 : (g == 3 || g == 5 || g == 6) ? 1 : 0;   // mcomp.c:366-370
  // site g + 1 -> (dr + 1, dc + 1), two bits each: the centre's move when that site wins
  constexpr uint32_t kSiteDr = (0u << 2) | (2u << 4) | (1u << 6) | (1u << 8) | (0u << 10) | (2u << 12) | (0u << 14) | (2u << 16);
  constexpr uint32_t kSiteDc = (1u << 2) | (1u << 4) | (0u << 6) | (2u << 8) | (0u << 10) | (2u << 12) | (2u << 14) | (0u << 16);

  typename G::L srcu[NU];
  group8_load_src<T, W, H>(sp, src.stride, l, srcu);
  // candidates as uniform base + 32-bit lane offset; the base sits at the smallest legal MV.  The lane's unit k is unit 0 moved down by
  // k * (8 / UPR) rows (u = l + 8 k): one offset register per address space and a uniform step, not one register per unit
  static_assert(!G::KEEP || 8 % G::UPR == 0 || G::PER_LANE == 1, "unit k = unit 0 + k * step");
  constexpr int kRowStep = G::UPR <= 8 ? 8 / G::UPR : 0;
  [[maybe_unused]] const uint32_t uoff0 = (uint32_t)(((min(l, G::U - 1) / G::UPR) * ref.stride + (min(l, G::U - 1) % G::UPR) * G::UE) * ES);
  [[maybe_unused]] const uint32_t ustep = (uint32_t)(kRowStep * ref.stride * ES);
  [[maybe_unused]] uint32_t loff0 = 0, lstep = 0;   // the same inside the window
  if constexpr (CELL && G::KEEP) {
    loff0 = (uint32_t)((min(l, G::U - 1) / G::UPR) * cw.pitch + (min(l, G::U - 1) % G::UPR) * G::UB);
    lstep = (uint32_t)(kRowStep * cw.pitch);
  }
  [[maybe_unused]] const int site_goff = (dr * ref.stride + dc) * ES;                       // this lane's site per unit of radius: plane ...
  [[maybe_unused]] const int site_loff = CELL ? dr * cw.pitch + dc * ES : 0;               // ... and window
  [[maybe_unused]] const char *ubase = reinterpret_cast<const char *>(rbase + (int64_t)row_min * ref.stride + col_min);

  // SAD of this lane's group's site (dr, dc) * r around (row, col) -- or of (row, col) itself with r = 0 -- by the group's 8 lanes
  auto round_sad = [&](int row, int col, int r, bool active, bool in_win) -> uint32_t {
    uint32_t a0 = 0, a1 = 0;
    if constexpr (G::KEEP) {
      if (in_win) {
        if constexpr (CELL) {
          const int base = (b.by + row - cw.y0) * cw.pitch + (b.bx + col - cw.x0) * ES;   // (scalar)
          const uint32_t so = (uint32_t)(__mul24(site_loff, r) + base);
          if (active) {
            uint32_t d[NU][G::UB / 4 + 1];
            // (the window's pitch is a multiple of 4 bytes: every unit of the lane has the byte phase of its first one, and its dword address
            // is the first one's + k * lstep -- one mask and NU - 1 adds instead of a shift, a mask and a base add per unit)
            const uint32_t o0 = so + loff0;
            const unsigned sh = o0 & 3;
            const uint32_t *p0 = cell_lds + (o0 >> 2);
#pragma unroll
            for (int k = 0; k < NU; ++k) {
              const uint32_t *p = p0 + k * (int)(lstep >> 2);
#pragma unroll
              for (int i = 0; i <= G::UB / 4; ++i) d[k][i] = p[i];
            }
#pragma unroll
            for (int k = 0; k < NU; ++k) {
              if (l + 8 * k < G::U) {
#pragma unroll
                for (int i = 0; i < G::UB / 4; ++i) {
                  const uint32_t v = __builtin_amdgcn_alignbyte(d[k][i + 1], d[k][i], sh);
                  if ((k * (G::UB / 4) + i) & 1) a1 = sadw<T>(srcu[k].v[i], v, a1); else a0 = sadw<T>(srcu[k].v[i], v, a0);
                }
              }
            }
          }
        }
      } else {
        const int base = ((row - row_min) * ref.stride + (col - col_min)) * ES;   // (scalar)
        const uint32_t so = (uint32_t)(__mul24(site_goff, r) + base);
        if (active) {
          typename G::L v[NU];
#pragma unroll
          for (int k = 0; k < NU; ++k) v[k] = *reinterpret_cast<const typename G::L *>(ubase + (size_t)(uint32_t)(so + uoff0 + (uint32_t)k * ustep));
#pragma unroll
          for (int k = 0; k < NU; ++k) {
            if (l + 8 * k < G::U) {
#pragma unroll
              for (int i = 0; i < G::UB / 4; ++i) {
                if ((k * (G::UB / 4) + i) & 1) a1 = sadw<T>(srcu[k].v[i], v[k].v[i], a1); else a0 = sadw<T>(srcu[k].v[i], v[k].v[i], a0);
              }
            }
          }
        }
      }
      uint32_t acc = a0 + a1;
      acc += __builtin_amdgcn_update_dpp(0u, acc, 0xB1, 0xf, 0xf, false);
      acc += __builtin_amdgcn_update_dpp(0u, acc, 0x4E, 0xf, 0xf, false);
      acc += __builtin_amdgcn_update_dpp(0u, acc, 0x141, 0xf, 0xf, false);
      return acc;
    } else {
      return group8_sad<T, W, H>(sp, src.stride, rbase + (int64_t)(row + dr * r) * ref.stride + (col + dc * r), ref.stride, l, active, srcu);
    }
  };
  // does the window hold every pixel of the blocks at (row +- r, col +- r)?  (scalar)
  auto win_covers = [&](int row, int col, int r) -> bool {
    if constexpr (CELL && G::KEEP) return cw.covers(b.bx + col - r, b.by + row - r, b.bx + col + r + W, b.by + row + r + H);
    else return false;
  };

  // radius of stage k (av1_init_dsmotion_compensation): DIAMOND 2^k, CLAMPED_DIAMOND min(2^k, 256); 11 stages
  auto radius = [](int k) { const int r = 1 << k; return (CLAMPED && r > 256) ? 256 : r; };

  // every run starts at the clamped start MV: its SAD once (group 0 evaluates it, the other seven would only repeat its loads)
  uint32_t start_sad;
  {
    const uint32_t s0 = round_sad(start_row, start_col, 0, g == 0, win_covers(start_row, start_col, 0)) >> shift;
    start_sad = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)__builtin_amdgcn_readlane((int)s0, 0) + (uint32_t)(lambda * (iabsm(start_row - frr) + iabsm(start_col - frc)))));
  }

  // How far the centre is from the nearest MV limit / from the nearest edge of the window (scalars): a round of radius r lies inside the
  // limits iff r <= lim_margin and inside the window iff r <= win_margin -- one compare each per round instead of eight; the margins change
  // only when the centre moves (a round in four), and every run starts at the same centre.
  auto lim_margin_of = [&](int row, int col) { return min(min(row - row_min, row_max - row), min(col - col_min, col_max - col)); };
  auto win_margin_of = [&](int row, int col) -> int {
    if constexpr (CELL && G::KEEP)
      return min(min(b.bx + col - cw.x0, cw.x1 - (b.bx + col + W)), min(b.by + row - cw.y0, cw.y1 - (b.by + row + H)));
    else return -1;
  };
  const int lim_margin0 = lim_margin_of(start_row, start_col), win_margin0 = win_margin_of(start_row, start_col);

  auto run_diamond = [&](int search_step, int *num00, int *orow, int *ocol) -> int {
    int row = start_row, col = start_col;
    int lim_margin = lim_margin0, win_margin = win_margin0;
    const int tot_steps = 11 - search_step;
    *num00 = 0;
    uint32_t bestsad = start_sad;
    int is_off_center = 0;
    for (int step = tot_steps - 1; step >= 0; --step) {
      const int r = radius(step);
      [[maybe_unused]] const unsigned long long t_r0 = CELL_T();
      // the whole diamond inside the limits (the usual case): no per-site test
      const bool all_in = r <= lim_margin;   // (scalar)
      bool inr = true;
      if (!all_in) {
        const int srow = row + dr * r, scol = col + dc * r;
        inr = scol >= col_min && scol <= col_max && srow >= row_min && srow <= row_max;
      }
      const bool in_win = r <= win_margin;
      const uint32_t mine = round_sad(row, col, r, inr, in_win) >> shift;
      // The reference walks the 8 sites in order with `if (sad < best) { sad += cost; if (sad < best) take it }` (mcomp.c:1350-1395): since
      // the L1 costs of this kernel are never negative that is "the FIRST site that attains the smallest sad + cost, if that is below the
      // best so far".  Every group adds its own site's cost, the 8 keys (sad + cost) << 4 | site are min-reduced -- one DPP step inside
      // the 16-lane rows, four v_readlane, three s_min -- instead of 16 v_readlane and eight dependent scalar compare / branch sequences.
      // cost = lambda * (|srow - frr| + |scol - frc|) (mvsad_err_cost_ of an L1 type: (lambda * 8 d) >> 3); |x| as v_sad_u32 against a bias
      uint32_t my_this = mine;
      if (lambda) {   // (scalar)
        constexpr int kBias = 1 << 16;
        const uint32_t tr = (uint32_t)(__mul24(dr, r) + (row - frr + kBias)), tc = (uint32_t)(__mul24(dc, r) + (col - frc + kBias));
        uint32_t d;   // (no builtin for v_sad_u32)
        asm("v_sad_u32 %0, %1, %2, 0\n\tv_sad_u32 %0, %3, %2, %0" : "=&v"(d) : "v"(tc), "s"(kBias), "v"(tr));
        my_this += (uint32_t)__mul24((int)d, lambda);
      }
      uint32_t key = inr ? ((my_this << 4) | (uint32_t)(g + 1)) : 0xFFFFFFFFu;
      const uint32_t kb = groups8_min_u32(key);   // (search_device.h: three v_min_u32_dpp + one v_readlane; four v_readlane + three s_min before)
      int best_site = 0;
      if (kb != 0xFFFFFFFFu && (kb >> 4) < bestsad) {
        bestsad = kb >> 4;
        best_site = (int)(kb & 15u);
        row += ((int)((kSiteDr >> (2 * best_site)) & 3u) - 1) * r;
        col += ((int)((kSiteDc >> (2 * best_site)) & 3u) - 1) * r;
        is_off_center = 1;
        lim_margin = lim_margin_of(row, col);
        win_margin = win_margin_of(row, col);
      }
      if (is_off_center == 0) (*num00)++;
      if (CLAMPED && best_site == 0) {   // (equal consecutive radii exist only in the clamped table: radius(k) = 2^k otherwise)
        while (step > 2 && radius(step - 1) == radius(step)) {
          ++(*num00);
          --step;
        }
      }
#ifdef AOMHIP_CELL_PROF
      if (in_win) { t_lds += CELL_T() - t_r0; ++n_lds; } else { t_glob += CELL_T() - t_r0; ++n_glob; }
#endif
    }
    *orow = row;
    *ocol = col;
    return (int)bestsad;
  };

  auto var_cost_at = [&](int row, int col) -> int {  // get_mvpred_var_cost: vf(src, ref) + mv_err_cost_
    [[maybe_unused]] const unsigned long long t_v0 = CELL_T();
    uint32_t v = 0;
    bool done = false;
    if constexpr (CELL && G::KEEP) {
      if (win_covers(row, col, 0)) {  // group 0 out of the window: no global-memory round trip at the end of every run
        v = group8_variance_lds<T, W, H>(cell_lds, (unsigned)((b.by + row - cw.y0) * cw.pitch + (b.bx + col - cw.x0) * ES), cw.pitch, l,
                                         g == 0, bit_depth, srcu);
        done = true;
      }
    }
    if (!done) {
      // lanes 0..15 (one DPP row) evaluate it: the sums reduce with four DPP steps instead of twelve 64-bit shuffles,
      // which made one variance as expensive as five diamond steps
      uint32_t sse;
      v = group16_variance<T, W, H, false>(rbase + (int64_t)row * ref.stride + col, ref.stride, 0, 0, sp, src.stride,
                                           /*a_minus_b=*/false, bit_depth, lane & 15, lane < 16, &sse);
    }
    v = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
#ifdef AOMHIP_CELL_PROF
    t_var += CELL_T() - t_v0; ++n_var;
#endif
    return (int)v + cc.var_cost(row * 8, col * 8);
  };

  // full_pixel_diamond (mcomp.c:1421-1470): the first search at step_param, then restarts at step_param + n that are
  // skipped while the previous search reported it would have stayed on the centre (num00).  One loop, one inlined
  // copy of the search body (two copies cost 30 VGPRs = one wave per SIMD of occupancy).
  int n = 0, num00 = 0, br = 0, bc = 0, bestsme = INT_MAX;
  int last_r = INT_MIN, last_c = INT_MIN, last_var = 0;   // the variance at the MV the previous run ended on
  const int further_steps = 11 - 1 - step_param;
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
      int sme = run_diamond(sstep, &t00, &tr, &tc);
      if (sme < INT_MAX) {
        if (tr != last_r || tc != last_c) {
          last_var = var_cost_at(tr, tc);
          last_r = tr;
          last_c = tc;
        }
        sme = last_var;
      }
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
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)br;
    out_mv[2 * bi + 1] = (int16_t)bc;
    out_cost[bi] = bestsme;
  }
  CELL_ADD(2, t_lds); CELL_ADD(3, n_lds); CELL_ADD(4, t_glob); CELL_ADD(5, n_glob); CELL_ADD(6, t_var); CELL_ADD(7, n_var);
  CELL_ADD(8, CELL_T() - t_begin);
}

// ---- exhaustive mesh search: full_pixel_exhaustive (mcomp.c:1547-1617) over exhaustive_mesh_search (:1474-1543) ----
// One 256-lane workgroup per block.  A pass visits a (2*range/step + 1)^2 mesh around the current best MV; every
// candidate's SAD + MV cost is independent of the others, and the reference's running update (update_mvs_and_sad,
// :839-858: skip on this_sad >= best, else add the cost and take it on strict <) is, because the cost is never
// negative, exactly "arg-min of sad + cost, first in raster order wins ties, the start position wins ties against
// everything".  So: the pass's reference window (all rows and columns any candidate touches) and the source block
// are staged in LDS, lane t evaluates candidates t, t + 256, ... entirely out of LDS (source rows are broadcast
// reads, neighbouring lanes read neighbouring columns of the same window rows: conflict-free), and the workgroup
// reduces the 64-bit keys (total << 32 | raster index + 1).  The step-1 column rule of the reference is kept: columns
// are taken four at a time and the tail group `for (i = 0; i < end_col - c; ++i)` never visits column end_col.
// A pass whose window does not fit the LDS budget (range grown to 5/4 |start mv|) reads the reference from global
// memory instead -- same arithmetic.
// LDS reads: a ds_read_b128 whose address is not 16-byte aligned runs at 1/12 of the aligned rate on gfx950
// (tools/lds_unaligned_probe.hip: 0.61 vs 7.4 T lane-reads/s), so window rows are read as aligned dwords and
// realigned with v_alignbyte; the packed source rows are 16-byte aligned.
constexpr int kMeshThreads = 256;

template <typename T, int W, int H, bool FROM_LDS>
__device__ __forceinline__ uint32_t mesh_sad(const uint32_t *lds_src, const char *win, int64_t pitch) {
  constexpr int RB = W * (int)sizeof(T);
  constexpr int UB = RB < 16 ? RB : 16;
  constexpr int UPR = RB / UB;
  using L = typename MLoad<UB>::type;
  uint32_t acc = 0;
#pragma unroll H <= 16 ? H : 4
  for (int r = 0; r < H; ++r) {
#pragma unroll
    for (int u = 0; u < UPR; ++u) {
      L a, b;
      if constexpr (UB == 16) {  // packed source rows are 16-byte aligned in LDS: one ds_read_b128
        const uint4 t = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(lds_src) + r * RB + u * UB);
        a.v[0] = t.x; a.v[1] = t.y; a.v[2] = t.z; a.v[3] = t.w;
      } else {
        a = *reinterpret_cast<const L *>(reinterpret_cast<const char *>(lds_src) + r * RB + u * UB);
      }
      if constexpr (FROM_LDS) {
        const char *p = win + r * pitch + u * UB;
        const unsigned sh = (unsigned)(uintptr_t)p & 3u;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p - sh);
        uint32_t d[UB / 4 + 1];
#pragma unroll
        for (int i = 0; i <= UB / 4; ++i) d[i] = q[i];
#pragma unroll
        for (int i = 0; i < UB / 4; ++i) b.v[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
      } else {
        b = *reinterpret_cast<const L *>(win + r * pitch + u * UB);
      }
#pragma unroll
      for (int i = 0; i < UB / 4; ++i) acc = sadw<T>(a.v[i], b.v[i], acc);
    }
  }
  return acc;
}

template <typename T, int W, int H>
__global__ __launch_bounds__(kMeshThreads) void mesh_search_kernel(
    PlaneView<T> src, PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks, int n_blocks,
    int cost_type, int bit_depth, int4 pat_range, int4 pat_interval, int fine, int lds_window_bytes,
    int16_t *__restrict__ out_mv, int32_t *__restrict__ out_cost) {
  extern __shared__ uint32_t mesh_lds[];
  constexpr int ES = (int)sizeof(T);
  constexpr int kSrcBytes = W * H * ES;
  __shared__ unsigned long long wg_key[kMeshThreads / 64];
  uint32_t *lds_src = mesh_lds;
  char *lds_win = reinterpret_cast<char *>(mesh_lds) + ((kSrcBytes + 15) & ~15);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bi = blockIdx.x;
  if (bi >= n_blocks) return;
  const aomhip_search_block b = blocks[bi];
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)b.by * src.stride + b.bx;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)b.by * ref.stride + b.bx;
  const CostCtx cc{ cost_type, b.ref_row, b.ref_col };
  const int shift = bit_depth == 10 ? 2 : bit_depth == 12 ? 4 : 0;

  // source block -> LDS, packed rows
  for (int q = tid; q < kSrcBytes / 4; q += kMeshThreads) {
    const int byte = q * 4, r = byte / (W * ES), c = byte % (W * ES);
    lds_src[q] = reinterpret_cast<const MU32 *>(reinterpret_cast<const char *>(sp + (int64_t)r * src.stride) + c)->v[0];
  }

  int br = b.start_row, bc = b.start_col;
  int bestsme = INT_MAX;
  const int ranges[4] = { pat_range.x, pat_range.y, pat_range.z, pat_range.w };
  const int intervals[4] = { pat_interval.x, pat_interval.y, pat_interval.z, pat_interval.w };
  int range = ranges[0], interval = intervals[0];
  const bool legal = !(range < 7 || range > 256 || interval < 1 || interval > range);
  if (legal) {
    const int div = range / interval;
    const int m = max(iabsm(br), iabsm(bc));
    range = min(max(range, (5 * m) / 4), 256);
    interval = max(interval, range / div);
    if (fine) interval = min(interval, 4);
    const bool progressive = interval > 1 && range > 7;
    for (int pass = 0; pass < 4; ++pass) {
      if (pass > 0) {
        if (!progressive) break;
        range = ranges[pass];
        interval = intervals[pass];
      }
      // ---- one exhaustive_mesh_search(start = (br, bc), range, interval)
      const int srow = min(max(br, (int)b.row_min), (int)b.row_max), scol = min(max(bc, (int)b.col_min), (int)b.col_max);
      const int start_row = max(-range, b.row_min - srow), start_col = max(-range, b.col_min - scol);
      const int end_row = min(range, b.row_max - srow), end_col = min(range, b.col_max - scol);
      const int step = interval;
      const int nr = end_row >= start_row ? (end_row - start_row) / step + 1 : 0;
      int nc;
      if (step > 1) {
        nc = end_col >= start_col ? (end_col - start_col) / step + 1 : 0;
      } else {
        const int span = end_col - start_col + 1;  // may be <= 0
        const int g4 = span > 0 ? span / 4 : 0, rem = span > 0 ? span - 4 * g4 : 0;
        nc = 4 * g4 + (rem > 0 ? rem - 1 : 0);
      }
      const int n_cand = nr * nc;
      // window: rows srow + start_row .. srow + end_row + H - 1, cols scol + start_col .. scol + end_col + W - 1
      const int wrows = (end_row - start_row) + H, wcols = (end_col - start_col) + W;
      const int wpitch = ((wcols * ES + 15) & ~15) + 16;
      const bool fits = n_cand > 0 && wrows > 0 && (int64_t)wrows * wpitch + 16 <= lds_window_bytes;
      const T *worg = rbase + (int64_t)(srow + start_row) * ref.stride + (scol + start_col);
      __syncthreads();  // previous pass done with the window (and the source block is in place)
      if (fits) {
        const int cpr = wpitch / 16 - 1;  // chunks that carry pixels (the tail chunk may over-read <= 15 bytes: in-row)
        const int total = wrows * cpr;
        for (int q = tid; q < total; q += kMeshThreads) {
          const int r = q / cpr, c = q - r * cpr;
          const MU128 v = *reinterpret_cast<const MU128 *>(reinterpret_cast<const char *>(worg + (int64_t)r * ref.stride) + c * 16);
          *reinterpret_cast<uint4 *>(lds_win + r * wpitch + c * 16) = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
        }
      }
      __syncthreads();
      // start position (clamped start): best so far
      unsigned long long key;
      {
        uint32_t s0;
        if (fits)
          s0 = mesh_sad<T, W, H, true>(lds_src, lds_win + (-start_row) * wpitch + (-start_col) * ES, wpitch);
        else
          s0 = mesh_sad<T, W, H, false>(lds_src, reinterpret_cast<const char *>(rbase + (int64_t)srow * ref.stride + scol),
                                        (int64_t)ref.stride * ES);
        const uint32_t t0 = (s0 >> shift) + (uint32_t)cc.sad_cost(srow, scol);
        key = (unsigned long long)t0 << 32;
      }
      for (int idx = tid; idx < n_cand; idx += kMeshThreads) {
        const int ir = idx / nc, ic = idx - ir * nc;
        const int r = start_row + ir * step, c = start_col + (step > 1 ? ic * step : ic);
        uint32_t sad;
        if (fits)
          sad = mesh_sad<T, W, H, true>(lds_src, lds_win + (r - start_row) * wpitch + (c - start_col) * ES, wpitch);
        else
          sad = mesh_sad<T, W, H, false>(lds_src,
                                         reinterpret_cast<const char *>(rbase + (int64_t)(srow + r) * ref.stride + scol + c),
                                         (int64_t)ref.stride * ES);
        const uint32_t tot = (sad >> shift) + (uint32_t)cc.sad_cost(srow + r, scol + c);
        const unsigned long long k = ((unsigned long long)tot << 32) | (uint32_t)(idx + 1);
        key = k < key ? k : key;
      }
      // workgroup arg-min
#pragma unroll
      for (int msk = 1; msk < 64; msk <<= 1) {
        const unsigned long long o = __shfl_xor(key, msk, 64);
        key = o < key ? o : key;
      }
      if (lane == 0) wg_key[wave] = key;
      __syncthreads();
      key = wg_key[0];
#pragma unroll
      for (int w2 = 1; w2 < kMeshThreads / 64; ++w2) key = wg_key[w2] < key ? wg_key[w2] : key;
      const uint32_t widx = (uint32_t)key;
      bestsme = (int)(uint32_t)(key >> 32);
      if (widx == 0) {
        br = srow;
        bc = scol;
      } else {
        const int ir = (int)(widx - 1) / nc, ic = (int)(widx - 1) - ir * nc;
        br = srow + start_row + ir * step;
        bc = scol + start_col + (step > 1 ? ic * step : ic);
      }
      if (pass > 0 && interval == 1) break;
      if (pass == 0 && !progressive) break;
    }
    // get_mvpred_var_cost at the winner (wave 0; the others are done)
    if (wave == 0 && bestsme < INT_MAX) {
      uint32_t sse;
      const uint32_t v = wave_variance<T, W, H, false>(rbase + (int64_t)br * ref.stride + bc, ref.stride, 0, 0, sp, src.stride,
                                                       /*a_minus_b=*/false, bit_depth, lane, &sse);
      bestsme = (int)v + cc.var_cost(br * 8, bc * 8);
    }
  }
  if (tid == 0) {
    out_mv[2 * bi] = (int16_t)br;
    out_mv[2 * bi + 1] = (int16_t)bc;
    out_cost[bi] = bestsme;
  }
}


}  // namespace aomhip

using namespace aomhip;

extern "C" {

#ifdef AOMHIP_CELL_PROF
int aomhip_debug_cell_prof(unsigned int *out, int n_blocks) {   // out[n_blocks][16]
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cell_prof), (size_t)n_blocks * 16 * sizeof(unsigned int)) != hipSuccess) return AOMHIP_ERR_HIP;
  return AOMHIP_OK;
}
#endif

int aomhip_fullpel_diamond_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw,
                                 int bh, int clamped, int step_param, int mv_cost_type,
                                 const aomhip_search_block *d_blocks, int n_blocks, int16_t *d_best_mv,
                                 int32_t *d_best_cost) {
  int rc = check_common(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, mv_cost_type);
  if (rc != AOMHIP_OK) return rc;
  if (!d_best_mv || !d_best_cost || step_param < 0 || step_param > 10) {
    set_error("aomhip_fullpel_diamond_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  // the cell window: reach of the search's first steps around the start MVs (the whole search reaches 2 * first - 1)
  const int first_r = 1 << (10 - step_param);
  const CellPlan cp = plan_cells(ref, bw, bh, n_blocks, 2 * first_r);
#define LAUNCH(T, W, H, WAVES, CELL)                                                                                            \
  {                                                                                                                             \
    auto k = clamped ? fullpel_diamond_kernel<T, W, H, WAVES, CELL, true> : fullpel_diamond_kernel<T, W, H, WAVES, CELL, false>;  \
    if (cp.lds > 64 * 1024) AOMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, cp.lds)); \
    hipLaunchKernelGGL(k, dim3(CELL ? cp.map.n_cells : (n_blocks + WAVES - 1) / WAVES), dim3(WAVES * 64), CELL ? cp.lds : 0, ctx->stream, \
                       view_of<T>(*src), view_of<T>(*ref), frame, d_blocks, n_blocks, cp.map, clamped, step_param, mv_cost_type,  \
                       src->bit_depth, d_best_mv, d_best_cost);                                                                  \
  }
#define X(W, H)                                                                                                      \
  if (bw == W && bh == H) {                                                                                          \
    if (src->bit_depth == 8) {                                                                                       \
      if constexpr (G8<uint8_t, W, H>::KEEP) {                                                                       \
        if (cp.waves) LAUNCH(uint8_t, W, H, kCellWaves, true) else LAUNCH(uint8_t, W, H, 4, false)                    \
      } else LAUNCH(uint8_t, W, H, 4, false)                                                                         \
    } else {                                                                                                         \
      if constexpr (G8<uint16_t, W, H>::KEEP) {                                                                      \
        if (cp.waves) LAUNCH(uint16_t, W, H, kCellWaves, true) else LAUNCH(uint16_t, W, H, 4, false)                  \
      } else LAUNCH(uint16_t, W, H, 4, false)                                                                        \
    }                                                                                                                \
    AOMHIP_LAUNCH_CHECK();                                                                                           \
    return AOMHIP_OK;                                                                                                \
  }
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
#undef LAUNCH
  return AOMHIP_ERR_INVALID;
}

int aomhip_mesh_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                             int mv_cost_type, const int mesh_patterns[8], int fine_search_interval,
                             const aomhip_search_block *d_blocks, int n_blocks, int16_t *d_best_mv,
                             int32_t *d_best_cost) {
  int rc = check_common(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, mv_cost_type);
  if (rc != AOMHIP_OK) return rc;
  if (!d_best_mv || !d_best_cost || !mesh_patterns) {
    set_error("aomhip_mesh_search_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const int es = src->bit_depth == 8 ? 1 : 2;
  // LDS: the source block + the first pass's nominal window, capped at what one CU has
  const int r0 = mesh_patterns[0] < 7 ? 7 : mesh_patterns[0] > 256 ? 256 : mesh_patterns[0];
  const int64_t pitch = (((2 * r0 + bw) * es + 15) & ~15) + 16;
  int64_t want = pitch * (2 * r0 + bh) + 16;
  const int64_t src_bytes = ((int64_t)bw * bh * es + 15) & ~15;
  const int64_t cap = 150 * 1024 - src_bytes;
  if (want > cap) want = cap;
  if (want < 4096) want = 4096;
  const size_t lds = (size_t)(src_bytes + want);
  const int4 pr = make_int4(mesh_patterns[0], mesh_patterns[2], mesh_patterns[4], mesh_patterns[6]);
  const int4 pi = make_int4(mesh_patterns[1], mesh_patterns[3], mesh_patterns[5], mesh_patterns[7]);
  const dim3 grid(n_blocks), block(kMeshThreads);
#define X(W, H)                                                                                                       \
  if (bw == W && bh == H) {                                                                                           \
    if (src->bit_depth == 8) {                                                                                        \
      auto k = mesh_search_kernel<uint8_t, W, H>;                                                                     \
      if (lds > 64 * 1024) AOMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL(k, grid, block, lds, ctx->stream, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), frame, d_blocks, \
                         n_blocks, mv_cost_type, 8, pr, pi, fine_search_interval, (int)want, d_best_mv, d_best_cost);  \
    } else {                                                                                                          \
      auto k = mesh_search_kernel<uint16_t, W, H>;                                                                    \
      if (lds > 64 * 1024) AOMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      hipLaunchKernelGGL(k, grid, block, lds, ctx->stream, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), frame,    \
                         d_blocks, n_blocks, mv_cost_type, src->bit_depth, pr, pi, fine_search_interval, (int)want,   \
                         d_best_mv, d_best_cost);                                                                     \
    }                                                                                                                 \
    AOMHIP_LAUNCH_CHECK();                                                                                            \
    return AOMHIP_OK;                                                                                                 \
  }
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

}  // extern "C"
