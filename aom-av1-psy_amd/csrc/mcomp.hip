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

// LDS window for the fine diamond steps: bit-exact, but measured 5 % SLOWER than the L1/L2 path on 4K 10-bit 16x16
// (0.59 vs 0.56 ms per frame, tools/gpu_ab_search.sh) -- the staging costs more than the few r <= 8 steps save.  Off.
#ifndef AOMHIP_DIAMOND_LDS_WINDOW
#define AOMHIP_DIAMOND_LDS_WINDOW 0
#endif

namespace aomhip {

#ifndef AOMHIP_DIAMOND_WAVES
#define AOMHIP_DIAMOND_WAVES 5   // waves per SIMD the register allocation aims at (A/B: profiles/r01_search_variants.md)
#endif

template <typename T, int W, int H>
__global__ __launch_bounds__(kSearchThreads, AOMHIP_DIAMOND_WAVES) void fullpel_diamond_kernel(
    PlaneView<T> src, PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks, int n_blocks,
    int level, int step_param, int cost_type, int bit_depth, int16_t *__restrict__ out_mv,
    int32_t *__restrict__ out_cost) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;  // (uniform: the block record and everything derived from it -- centre, limits, step state -- then lives in SGPRs and the search loops branch on the scalar unit)
  const int bi = blockIdx.x * (kSearchThreads / 64) + wave;
  if (bi >= n_blocks) return;
  const aomhip_search_block b = blocks[bi];
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)b.by * src.stride + b.bx;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)b.by * ref.stride + b.bx;
  const CostCtx cc{ cost_type, b.ref_row, b.ref_col };
  const int shift = bit_depth == 10 ? 2 : bit_depth == 12 ? 4 : 0;  // vtable wrappers for highbd SAD
  const int g = lane >> 3, l = lane & 7;
  const int dr = (g == 0 || g == 4 || g == 6) ? -1 : (g == 1 || g == 5 || g == 7) ? 1 : 0;   // site order of
  const int dc = (g == 2 || g == 4 || g == 7) ? -1 : (g == 3 || g == 5 || g == 6) ? 1 : 0;   // mcomp.c:366-370

  typename G8<T, W, H>::L srcu[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1];
  group8_load_src<T, W, H>(sp, src.stride, l, srcu);
  // candidates as uniform base + 32-bit lane offset (search_device.h group8_sad_u); the base sits at the smallest legal MV
  [[maybe_unused]] uint32_t uoff[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1];
  group8_unit_offsets<T, W, H>(ref.stride, l, uoff);
  [[maybe_unused]] const char *ubase = reinterpret_cast<const char *>(rbase + (int64_t)b.row_min * ref.stride + b.col_min);
  auto site_sad = [&](int row, int col, bool active) -> uint32_t {
    if constexpr (G8<T, W, H>::KEEP)
      return group8_sad_u<T, W, H>(ubase, (uint32_t)(((row - b.row_min) * ref.stride + (col - b.col_min)) * (int)sizeof(T)), l, active, uoff, srcu);
    else
      return group8_sad<T, W, H>(sp, src.stride, rbase + (int64_t)row * ref.stride + col, ref.stride, l, active, srcu);
  };

  // LDS window for the fine steps of a search (radius <= 8): the (2*15 + H) x (2*15 + W) pixels around the current
  // centre are staged once (about the traffic of ONE diamond step) and the remaining steps of the run -- whose
  // sites stay within 8 + 4 + 2 + 1 = 15 pixels of that centre -- read LDS instead of the L1/L2 path.  Only for
  // blocks up to 32 x 32 (LDS per wavefront: 4.4 KB for 16x16 16-bit, 15 KB for 32x32); a window is only ever
  // placed over legal MV positions, so it never reaches outside the bordered plane.
  constexpr bool kUseLds = AOMHIP_DIAMOND_LDS_WINDOW && G8<T, W, H>::KEEP && W * H <= 1024 && W >= 8;
  constexpr int kRW = 15;
  constexpr int kWinRows = 2 * kRW + H, kWinPitch = ((2 * kRW + W) * (int)sizeof(T) + 15) & ~15;
  __shared__ uint32_t win_all[kUseLds ? (kSearchThreads / 64) * (kWinRows * kWinPitch / 4 + 8) : 1];
  uint32_t *win = win_all + (kUseLds ? wave * (kWinRows * kWinPitch / 4 + 8) : 0);
  int wr0 = INT_MIN / 2, wc0 = INT_MIN / 2;  // window origin in MV space; far away = nothing staged
  auto stage_window = [&](int row, int col) {
    if constexpr (kUseLds) {
      if (b.row_max - b.row_min < 2 * kRW || b.col_max - b.col_min < 2 * kRW) return;
      wr0 = min(max(row - kRW, (int)b.row_min), (int)b.row_max - 2 * kRW);
      wc0 = min(max(col - kRW, (int)b.col_min), (int)b.col_max - 2 * kRW);
      const char *g0 = reinterpret_cast<const char *>(rbase + (int64_t)wr0 * ref.stride + wc0);
      constexpr int kCpr = kWinPitch / 16;
      for (int q = lane; q < kWinRows * kCpr; q += 64) {
        const int r = q / kCpr, c = q - r * kCpr;
        const MU128 v = *reinterpret_cast<const MU128 *>(g0 + (int64_t)r * ref.stride * (int)sizeof(T) + c * 16);
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(win) + r * kWinPitch + c * 16) = make_uint4(v.v[0], v.v[1], v.v[2], v.v[3]);
      }
      // one wavefront owns the window: its own LDS writes are visible to its later reads once they have retired
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    }
  };
  auto covered = [&](int row, int col, int r) {
    return row - r >= wr0 && row + r <= wr0 + 2 * kRW && col - r >= wc0 && col + r <= wc0 + 2 * kRW;
  };

  // radius of stage k (av1_init_dsmotion_compensation): DIAMOND 2^k, CLAMPED_DIAMOND min(2^k, 256); 11 stages
  auto radius = [level](int k) { const int r = 1 << k; return (level > 0 && r > 256) ? 256 : r; };

  auto run_diamond = [&](int search_step, int *num00, int *orow, int *ocol) -> int {
    int row = min(max((int)b.start_row, (int)b.row_min), (int)b.row_max);  // clamp_fullmv
    int col = min(max((int)b.start_col, (int)b.col_min), (int)b.col_max);
    const int tot_steps = 11 - search_step;
    *num00 = 0;
    // (the centre is one position: group 0 evaluates it, the other seven groups would only repeat its loads)
    uint32_t s0 = site_sad(row, col, g == 0) >> shift;
    s0 = (uint32_t)__builtin_amdgcn_readlane((int)s0, 0);
    uint32_t bestsad = s0 + (uint32_t)cc.sad_cost(row, col);
    int is_off_center = 0;
    int next_step_size = tot_steps > 2 ? radius(tot_steps - 2) : 1;
    for (int step = tot_steps - 1; step >= 0; --step) {
      const int r = radius(step);
      if (step > 0) next_step_size = radius(step - 1);
      const int srow = row + dr * r, scol = col + dc * r;
      const bool inr = scol >= b.col_min && scol <= b.col_max && srow >= b.row_min && srow <= b.row_max;
      uint32_t mine;
      bool from_lds = false;
      if constexpr (kUseLds) {
        if (r <= 8) {
          if (!covered(row, col, r) && step >= 2) stage_window(row, col);
          from_lds = covered(row, col, r);
        }
      }
      if constexpr (kUseLds) {
        if (from_lds)
          mine = group8_sad_lds<T, W, H>(win, (unsigned)((srow - wr0) * kWinPitch + (scol - wc0) * (int)sizeof(T)), kWinPitch, l,
                                         inr, srcu) >> shift;
        else
          mine = site_sad(srow, scol, inr) >> shift;
      } else {
        mine = site_sad(srow, scol, inr) >> shift;
      }
      // The reference walks the 8 sites in order with `if (sad < best) { sad += cost; if (sad < best) take it }` (mcomp.c:1350-1395): since
      // the L1 costs of this kernel are never negative that is "the FIRST site that attains the smallest sad + cost, if that is below the
      // best so far".  Every group adds its own site's cost, the 8 keys (sad + cost) << 4 | site are min-reduced -- one DPP step inside
      // the 16-lane rows, four v_readlane, three s_min -- instead of 16 v_readlane and eight dependent scalar compare / branch sequences.
      int best_site = 0;
      {
        const uint32_t my_this = mine + (uint32_t)cc.sad_cost(srow, scol);
        uint32_t key = inr ? ((my_this << 4) | (uint32_t)(g + 1)) : 0xFFFFFFFFu;
        const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, 0x128, 0xf, 0xf, false);  // row_ror:8: the row's other group
        key = min(key, other);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, 0), k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, 16);
        const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)key, 32), k3 = (uint32_t)__builtin_amdgcn_readlane((int)key, 48);
        const uint32_t kb = min(min(k0, k1), min(k2, k3));
        if (kb != 0xFFFFFFFFu && (kb >> 4) < bestsad) {
          bestsad = kb >> 4;
          best_site = (int)(kb & 15u);
        }
      }
      if (best_site != 0) {
        const int ddr = (best_site == 1 || best_site == 5 || best_site == 7) ? -1
                        : (best_site == 2 || best_site == 6 || best_site == 8) ? 1 : 0;
        const int ddc = (best_site == 3 || best_site == 5 || best_site == 8) ? -1
                        : (best_site == 4 || best_site == 6 || best_site == 7) ? 1 : 0;
        row += ddr * r;
        col += ddc * r;
        is_off_center = 1;
      }
      if (is_off_center == 0) (*num00)++;
      if (best_site == 0) {
        while (next_step_size == radius(step) && step > 2) {
          ++(*num00);
          --step;
          next_step_size = radius(step - 1);
        }
      }
    }
    *orow = row;
    *ocol = col;
    return (int)bestsad;
  };

  auto var_cost_at = [&](int row, int col) -> int {  // get_mvpred_var_cost: vf(src, ref) + mv_err_cost_
    // lanes 0..15 (one DPP row) evaluate it: the sums reduce with four DPP steps instead of twelve 64-bit shuffles,
    // which made one variance as expensive as five diamond steps
    uint32_t sse;
    uint32_t v = group16_variance<T, W, H, false>(rbase + (int64_t)row * ref.stride + col, ref.stride, 0, 0, sp, src.stride,
                                                  /*a_minus_b=*/false, bit_depth, lane & 15, lane < 16, &sse);
    v = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    return (int)v + cc.var_cost(row * 8, col * 8);
  };

  // full_pixel_diamond (mcomp.c:1421-1470): the first search at step_param, then restarts at step_param + n that are
  // skipped while the previous search reported it would have stayed on the centre (num00).  One loop, one inlined
  // copy of the search body (two copies cost 30 VGPRs = one wave per SIMD of occupancy).
  int n = 0, num00 = 0, br = 0, bc = 0, bestsme = INT_MAX;
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
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)br;
    out_mv[2 * bi + 1] = (int16_t)bc;
    out_cost[bi] = bestsme;
  }
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
#pragma unroll(H <= 16 ? H : 4)
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
  const dim3 grid((n_blocks + 3) / 4), block(kSearchThreads);
#define X(W, H)                                                                                                      \
  if (bw == W && bh == H) {                                                                                          \
    if (src->bit_depth == 8)                                                                                         \
      hipLaunchKernelGGL((fullpel_diamond_kernel<uint8_t, W, H>), grid, block, 0, ctx->stream, view_of<uint8_t>(*src), \
                         view_of<uint8_t>(*ref), frame, d_blocks, n_blocks, clamped, step_param, mv_cost_type, 8,    \
                         d_best_mv, d_best_cost);                                                                    \
    else                                                                                                             \
      hipLaunchKernelGGL((fullpel_diamond_kernel<uint16_t, W, H>), grid, block, 0, ctx->stream,                       \
                         view_of<uint16_t>(*src), view_of<uint16_t>(*ref), frame, d_blocks, n_blocks, clamped,       \
                         step_param, mv_cost_type, src->bit_depth, d_best_mv, d_best_cost);                          \
    AOMHIP_LAUNCH_CHECK();                                                                                           \
    return AOMHIP_OK;                                                                                                \
  }
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
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
