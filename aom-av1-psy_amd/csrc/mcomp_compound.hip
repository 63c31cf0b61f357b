// The compound-reference and OBMC full-pel searches of the RD path (av1/encoder/mcomp.c), batched: one wavefront per block.
//
//   aomhip_refining_search_8p_batch      av1_refining_search_8p_c (:1621-1691) -- the 8-neighbour refinement of one MV of a compound against the
//       predictor of the OTHER reference (get_mvpred_compound_sad, :710-731: vfp->sdaf, or vfp->msdf with a wedge / diff-weighted mask),
//       with its 7 x 7 "already visited" grid -- then av1_get_mvpred_compound_var (:3679-3693: vfp->svaf / msvf at the full-pel MV + MV cost)
//       at the result: the full-pel half of av1_joint_motion_search / av1_compound_single_motion_search
//       (av1/encoder/motion_search_facade.c:496-870).
//   aomhip_obmc_full_pixel_search_batch  av1_obmc_full_pixel_search (:2272-2285): obmc_full_pixel_diamond (:2236-2270) over
//       obmc_diamond_search_sad (:2173-2234) with get_obmc_mvpred_var (:2110-2125), or -- fast_obmc_search -- obmc_refining_search_sad
//       (:2127-2171); vfp->osdf / ovf on the weighted source and mask of calc_target_weighted_pred.
//
// These searches are short (<= 25 evaluations; an OBMC diamond ~100) and every evaluation needs a per-pixel blend before the difference, so
// an evaluation is done by all 64 lanes pixel by pixel (no packed SAD applies) and the candidates of a round are taken in the reference's
// order; comparisons are the reference's, literally (the OBMC diamond compares as `int`, the others as `unsigned`).
#include <climits>

#include "fullpel_search.h"

namespace aomhip {
namespace {

struct CompoundArgs {
  int bw, bh, bit_depth, cost_type, sad_per_bit, error_per_bit, invert_mask;
  const int32_t *mvjcost, *mvcost0, *mvcost1;
};

__device__ __forceinline__ int64_t wsum(int64_t v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor((long long)v, m, 64);
  return v;
}
__device__ __forceinline__ int mv_bits(const CompoundArgs &a, int dr, int dc) {
  return a.mvjcost[(dc != 0) | ((dr != 0) << 1)] + a.mvcost0[dr] + a.mvcost1[dc];
}
__device__ __forceinline__ int sad_cost(const CompoundArgs &a, int frr, int frc, int row, int col) {   // mvsad_err_cost_ (:310-339)
  const int dr = (row - frr) * 8, dc = (col - frc) * 8;
  if (a.cost_type == kCostEntropy) return (int)(((unsigned)mv_bits(a, dr, dc) * (unsigned)a.sad_per_bit + 256u) >> 9);
  const int lambda = a.cost_type == kCostL1Low ? 32 : a.cost_type == kCostL1Mid ? 15 : a.cost_type == kCostL1Hd ? 8 : 0;
  return (lambda * (iabsm(dr) + iabsm(dc))) >> 3;
}
// the same with what does not depend on the candidate worked out once per block (left inline the compiler re-derived lambda from cost_type
// through a chain of scalar branches at every call)
struct SadCost {
  const CompoundArgs &a;
  int frr, frc, lambda;
  bool entropy;
  __device__ __forceinline__ SadCost(const CompoundArgs &a_, int frr_, int frc_) : a(a_), frr(frr_), frc(frc_) {
    entropy = a.cost_type == kCostEntropy;
    lambda = a.cost_type == kCostL1Low ? 32 : a.cost_type == kCostL1Mid ? 15 : a.cost_type == kCostL1Hd ? 8 : 0;
  }
  __device__ __forceinline__ int operator()(int row, int col) const {
    const int dr = (row - frr) * 8, dc = (col - frc) * 8;
    if (entropy) return (int)(((unsigned)mv_bits(a, dr, dc) * (unsigned)a.sad_per_bit + 256u) >> 9);
    return (lambda * (iabsm(dr) + iabsm(dc))) >> 3;
  }
};
__device__ __forceinline__ int var_cost(const CompoundArgs &a, int ref_row, int ref_col, int mrow, int mcol) {   // mv_err_cost_ (:271-308)
  const int dr = mrow - ref_row, dc = mcol - ref_col;
  if (a.cost_type == kCostEntropy) return (int)(((int64_t)mv_bits(a, dr, dc) * a.error_per_bit + (1 << 13)) >> 14);
  const int lambda = a.cost_type == kCostL1Low ? 2 : a.cost_type == kCostL1Hd ? 1 : 0;
  return (lambda * (iabsm(dr) + iabsm(dc))) >> 3;
}
__device__ __forceinline__ uint32_t finish_var(int64_t s64, uint64_t q64, int n_px, int bit_depth) {   // variance.c:141-148 / :383-420
  int32_t s;
  uint32_t q;
  if (bit_depth == 10) { q = (uint32_t)((q64 + 8) >> 4); s = (int32_t)((s64 + 2) >> 2); }
  else if (bit_depth == 12) { q = (uint32_t)((q64 + 128) >> 8); s = (int32_t)((s64 + 8) >> 4); }
  else { q = (uint32_t)q64; s = (int32_t)s64; }
  const int64_t sq = ((int64_t)s * s) >> __builtin_ctz((unsigned)n_px);   // (/ (w * h): block areas are powers of two -- a 64-bit division is ~100 instructions)
  if (bit_depth == 8) return q - (uint32_t)sq;
  const int64_t v = (int64_t)q - sq;
  return v >= 0 ? (uint32_t)v : 0u;
}

// 32-bit sum over the wavefront, wave-uniform result: four DPP steps to the sums of the four 16-lane rows, four v_readlane, scalar adds
__device__ __forceinline__ uint32_t row_sum32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);    // quad_perm [1, 0, 3, 2]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);    // quad_perm [2, 3, 0, 1]
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);   // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false);   // row_mirror
  return v;
}
// the four 16-lane sums of row_sum32 added up: row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3, the total in lane 63 -- two DPP
// adds and one v_readlane instead of four v_readlane and three s_add (these kernels are bound by the CU's one scalar unit, PMC r05e)
__device__ __forceinline__ uint32_t rows_total32(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast31
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wsum32(uint32_t v) { return rows_total32(row_sum32(v)); }
__device__ __forceinline__ uint64_t wsum32_wide(uint32_t v) {   // the same when only the row sums fit 32 bits
  v = row_sum32(v);
  return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) + (uint32_t)__builtin_amdgcn_readlane((int)v, 32) +
         (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}
// The sums of the variance's two accumulators over the wavefront without the six dependent 64-bit shuffle steps of wsum(): the sum of
// differences fits 32 bits (|d| <= 8191, <= 128 x 128 pixels), the sum of squares is summed as its low 16 bits and the rest (a lane's
// share is < 2^42: <= 256 pixels of d^2 < 2^26) -- DPP adds only.
__device__ __forceinline__ int64_t wsum_s(int64_t s) { return (int64_t)(int32_t)wsum32((uint32_t)(int32_t)s); }
__device__ __forceinline__ uint64_t wsum_q(uint64_t q) { return ((uint64_t)wsum32((uint32_t)(q >> 16)) << 16) + wsum32((uint32_t)q & 0xffffu); }
__device__ __forceinline__ uint64_t wsum32_split(uint32_t v) {   // exact for any per-lane value: the two 16-bit halves summed separately
  return ((uint64_t)wsum32(v >> 16) << 16) + wsum32(v & 0xffffu);
}

// four adjacent pixels at p (any alignment: the planes' rows start anywhere) as ints
template <typename T> __device__ __forceinline__ void load_px4(const T *p, int out[4]) {
  if constexpr (sizeof(T) == 1) {
    const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
    out[0] = (int)(w & 0xffu); out[1] = (int)((w >> 8) & 0xffu); out[2] = (int)((w >> 16) & 0xffu); out[3] = (int)(w >> 24);
  } else {
    const uint2 w = *reinterpret_cast<const uint2 *>(p);
    out[0] = (int)(w.x & 0xffffu); out[1] = (int)(w.x >> 16); out[2] = (int)(w.y & 0xffffu); out[3] = (int)(w.y >> 16);
  }
}
// The candidates of one search stage, judged in site order.  The stage's sites live one per lane: `valid` has a bit per site that is to be
// evaluated (in range, not yet visited), `off_l` is the lane's site's candidate offset.  The scalar unit walks the bits G at a time (G groups
// of lanes evaluate G candidates at once, `idle_off` is what a group without a candidate reads).  fetch(off, px) requests the pixels,
// rows_of(off, px) -> the 16-lane sums; judge(site, rows, g).
// (Requesting the NEXT batch's pixels before the current one is reduced and judged -- within a stage which sites are read does not depend
// on the outcomes -- was measured and is not taken: the rotation of the fetched registers and the second copy of the control flow cost
// more issue slots than the overlap hides, compound diamond 475 -> 520 us, OBMC 0.67 -> 0.75 ms per 4K 10-bit frame; PMC: 4038 -> 5326
// scalar and 3621 -> 4550 vector instructions per block.  These kernels are bound by instruction issue, not by latency.)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f) {   // f(0) .. f(N - 1), unrolled by construction
  if constexpr (I < N) {
    f(I);
    static_for<I + 1, N>(f);
  }
}
template <int G, typename PX, typename FetchFn, typename RowsFn, typename JudgeFn>
__device__ __forceinline__ void walk_sites(uint32_t valid, unsigned off_l, unsigned idle_off, int grp, FetchFn fetch, RowsFn rows_of, JudgeFn judge) {
  auto pop = [&](int (&idx)[G], int &cnt) -> unsigned {
    cnt = 0;
    static_for<0, G>([&](int g) {
      idx[g] = valid ? __builtin_ctz(valid) : -1;
      cnt += valid != 0;
      valid &= valid - 1;
    });
    if constexpr (G == 1) {
      return (unsigned)__builtin_amdgcn_readlane((int)off_l, idx[0]);
    } else {
      int sel = idx[0];
      static_for<1, G>([&](int g) { sel = grp == g ? idx[g] : sel; });
      return sel < 0 ? idle_off : (unsigned)__builtin_amdgcn_ds_bpermute(sel << 2, (int)off_l);
    }
  };
  while (valid) {
    int idx[G], cnt;
    PX px;
    const unsigned off = pop(idx, cnt);
    fetch(off, px);
    const uint32_t rows = rows_of(off, px);
    static_for<0, G>([&](int g) {
      if (g < cnt) judge(idx[g], rows, g);
    });
  }
}

// One block's compound error functions evaluated by the 64 lanes (a lane owns units of four adjacent pixels): get_mvpred_compound_sad
// (vfp->sdaf / msdf) and the variance of get_mvpred_compound_var[_cost] (svaf / msvf at offset 0).  What does not depend on the candidate --
// the source block, the other reference's predictor, the blend weights -- stays in registers, as packed 16-bit pairs, for blocks of up to
// 256 * UNITS pixels (a search evaluates ~25 .. ~180 candidates): UNITS = 2 for blocks of up to 512 pixels, 4 for those of up to 1024
// (32 x 32, 16 x 64, 64 x 16).  Larger blocks stream the same operands unit by unit from memory (they sit in L1 / L2 after the first candidate).
// The candidate's reference pixels are one 4-pixel load per unit, issued together.  Block widths are powers of two: row / column of pixel t
// by shift and mask.  SADs are summed in 32 bits (<= 128 x 128 x 4095).
// Small blocks leave most of the wavefront idle (an 8 x 8 block is 16 units): with G = 2 / 4 the wavefront is G groups of 64 / G lanes, every
// group holds the whole block (<= 128 / <= 64 pixels) and evaluates a DIFFERENT candidate -- sad_partial() takes the candidate's offset per
// lane, group_total() reads one group's sum -- so a search stage's sites are evaluated G at a time and then judged one after the other in the
// reference's order (the comparisons are the same: which sites are read does not depend on the running best).
template <typename T, int UNITS, int G = 1, bool PACKED = true> struct CompoundEval {
  static constexpr int kUnits = UNITS;   // 4-pixel units per lane kept in registers
  static constexpr int kGroups = G, kLanes = 64 / G;   // lanes per group
  static_assert(G == 1 || (UNITS == 1 && (G == 2 || G == 4)), "groups hold one unit per lane");
  int u, grp;   // the lane's unit within its group, its group
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const T *sp, *pred;
  const uint8_t *mask;
  int sstride, rstride, lw, wm, n_px, shift, invert, bit_depth, lane, sh;
  bool keep;
  static constexpr bool packed = PACKED;   // (the launcher's choice: !mask || bit_depth <= 10)
  // G == 1: the launcher gives blocks of exactly 256 * UNITS pixels to this instantiation (256, 512, 1024; larger ones stream), so a kept block
  // fills every unit of every lane -- no per-lane or per-unit tests inside a candidate; G > 1: blocks of 16 .. 128 pixels, tested
  static constexpr bool kFull = G == 1;
  // A candidate's pixels are addressed as base0 + 32-bit byte offset (scalar base, one vector add per unit): base0 is the reference block at
  // the search window's top-left MV (row_min, col_min), so every offset a search can ask for is non-negative; lo_ = the lane's own offset
  const char *base0;
  int rmin, cmin;
  unsigned lo_[kUnits];
  // The blend of aom_comp_avg_pred (variance.c:306-319) / aom_comp_mask_pred (:773-791) is (A * f + C) >> sh per pixel with A, C fixed for
  // the search: comp_avg A = 1, C = p + 1, sh = 1; comp_mask A = m (inverted: 64 - m), C = (64 - A) * p + 32, sh = 6.  One branch-free form
  // for the three cases: with `if (!mask) .. else if (invert) ..` inside the per-pixel blend the compiler emitted two or three scalar
  // branches PER PIXEL (7 700 scalar instructions per block, PMC r05).
  // The operands are packed 16-bit pairs (two dwords per 4-pixel unit): A * f + C fits 16 bits for every depth without a mask
  // (<= 2 * 4095 + 1) and up to 10 bits with one (<= 64 * 1023 + 32): v_pk_mad_u16, v_pk_lshrrev_b16, v_sad_u16 -- 3 instructions per pixel
  // pair (`packed`).  12-bit masked blocks hold p in place of C and blend in 32 bits.  (These searches are issue bound, not latency bound:
  // evaluating a stage's 8 sites together was 17-32 % SLOWER, profiles/r05_compound_batch8.patch.)
  uint32_t sA_[kUnits][2], sC_[kUnits][2], sS_[kUnits][2], sh2;
  template <typename U> static __device__ __forceinline__ void load_pairs(const U *p, uint32_t o[2]) {   // four adjacent pixels (any alignment) as two 16-bit pairs
    if constexpr (sizeof(U) == 1) {
      const uint32_t w = *reinterpret_cast<const uint32_t *>(p);
      o[0] = __builtin_amdgcn_perm(0, w, 0x0c010c00);   // (px0, px1) as 16-bit halves
      o[1] = __builtin_amdgcn_perm(0, w, 0x0c030c02);   // (px2, px3)
    } else {
      const uint2 w = *reinterpret_cast<const uint2 *>(p);
      o[0] = w.x; o[1] = w.y;
    }
  }
  // the candidate-independent operands of the unit at pixel t (t < n_px)
  __device__ __forceinline__ void operands(int t, uint32_t A2[2], uint32_t C2[2], uint32_t S2[2]) const {
    uint32_t p2[2], m2[2] = { 0, 0 };
    load_pairs<T>(sp + (t >> lw) * sstride + (t & wm), S2);   // (widths are multiples of 4: a unit lies in one row)
    load_pairs<T>(pred + t, p2);
    if (mask) load_pairs<uint8_t>(mask + t, m2);
    const u16x2 one = { 1, 1 }, c64 = { 64, 64 }, c32 = { 32, 32 };
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u16x2 p = __builtin_bit_cast(u16x2, p2[h]), m = __builtin_bit_cast(u16x2, m2[h]);
      const u16x2 A = !mask ? one : (invert ? c64 - m : m);
      const u16x2 Cc = !mask ? p + one : (packed ? (c64 - A) * p + c32 : p);
      A2[h] = __builtin_bit_cast(uint32_t, A);
      C2[h] = __builtin_bit_cast(uint32_t, Cc);
    }
  }
  __device__ __forceinline__ void init(const T *sp_, int sstride_, const T *rbase_, int rstride_, const T *pred_, const uint8_t *mask_, int W, int H, int invert_,
                                       int bit_depth_, int lane_, int row_min, int col_min) {
    sp = sp_; pred = pred_; mask = mask_; sstride = sstride_; rstride = rstride_;
    rmin = row_min; cmin = col_min;
    base0 = reinterpret_cast<const char *>(rbase_ + (int64_t)row_min * rstride_ + col_min);
    lw = __builtin_ctz((unsigned)W); wm = W - 1; n_px = W * H; invert = invert_; bit_depth = bit_depth_; lane = lane_;
    u = lane & (kLanes - 1); grp = lane / kLanes;
    shift = bit_depth == 10 ? 2 : bit_depth == 12 ? 4 : 0;   // the _bits10 / _bits12 vtable wrappers (encoder_utils.h)
    sh = mask ? 6 : 1;
    sh2 = (uint32_t)sh | ((uint32_t)sh << 16);
    keep = kFull ? n_px == 4 * kLanes * kUnits : n_px <= 4 * kLanes * kUnits;
    if (keep) {
#pragma unroll
      for (int k = 0; k < kUnits; ++k) {
        const int t = 4 * (k * kLanes + u);
        lo_[k] = kFull || t < n_px ? ref_off(t) : 0u;
        sA_[k][0] = sA_[k][1] = sC_[k][0] = sC_[k][1] = sS_[k][0] = sS_[k][1] = 0;   // lanes beyond the block: blend 0 against source 0
        if (kFull || t < n_px) operands(t, sA_[k], sC_[k], sS_[k]);
      }
    }
  }
  __device__ __forceinline__ unsigned ref_off(int t) const { return (unsigned)(((t >> lw) * rstride + (t & wm)) * (int)sizeof(T)); }
  // the blend of one pixel in 32 bits (12-bit masked blocks: Cc holds p)
  __device__ __forceinline__ int blend32(int f, int A, int p) const { return (A * f + (64 - A) * p + 32) >> 6; }
  // SAD of one pixel pair: blend = (A * f + C) >> sh on both halves, then |blend - s| summed into acc
  __device__ __forceinline__ uint32_t pair_sad(uint32_t f, uint32_t A, uint32_t Cc, uint32_t S, uint32_t acc) const {
    const u16x2 b = (u16x2)(__builtin_bit_cast(u16x2, A) * __builtin_bit_cast(u16x2, f) + __builtin_bit_cast(u16x2, Cc)) >> __builtin_bit_cast(u16x2, sh2);
    return __builtin_amdgcn_sad_u16(__builtin_bit_cast(uint32_t, b), S, acc);
  }
  __device__ __forceinline__ uint32_t unit_sad(const uint32_t f2[2], const uint32_t A2[2], const uint32_t C2[2], const uint32_t S2[2], uint32_t acc) const {
    if (packed) {
      acc = pair_sad(f2[0], A2[0], C2[0], S2[0], acc);   // (lanes beyond the block: (0 * f + 0) >> sh == 0 == s)
      acc = pair_sad(f2[1], A2[1], C2[1], S2[1], acc);
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        acc += (uint32_t)iabsm(blend32((int)(f2[h] & 0xffffu), (int)(A2[h] & 0xffffu), (int)(C2[h] & 0xffffu)) - (int)(S2[h] & 0xffffu));
        acc += (uint32_t)iabsm(blend32((int)(f2[h] >> 16), (int)(A2[h] >> 16), (int)(C2[h] >> 16)) - (int)(S2[h] >> 16));
      }
    }
    return acc;
  }
  // sum and sum of squares of blend - source over one unit
  __device__ __forceinline__ void unit_var(const uint32_t f2[2], const uint32_t A2[2], const uint32_t C2[2], const uint32_t S2[2], int32_t &s, uint32_t &q) const {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int b0, b1;
      if (packed) {
        const u16x2 b = (u16x2)(__builtin_bit_cast(u16x2, A2[h]) * __builtin_bit_cast(u16x2, f2[h]) + __builtin_bit_cast(u16x2, C2[h])) >> __builtin_bit_cast(u16x2, sh2);
        b0 = (int)b.x; b1 = (int)b.y;
      } else {
        b0 = blend32((int)(f2[h] & 0xffffu), (int)(A2[h] & 0xffffu), (int)(C2[h] & 0xffffu));
        b1 = blend32((int)(f2[h] >> 16), (int)(A2[h] >> 16), (int)(C2[h] >> 16));
      }
      const int d0 = b0 - (int)(S2[h] & 0xffffu), d1 = b1 - (int)(S2[h] >> 16);
      s += d0 + d1;
      q += (uint32_t)(d0 * d0) + (uint32_t)(d1 * d1);
    }
  }
  __device__ __forceinline__ unsigned cand_off(int row, int col) const {   // (row, col) inside the search window: >= 0
    return (unsigned)(((row - rmin) * rstride + (col - cmin)) * (int)sizeof(T));
  }
  __device__ __forceinline__ void load_ref(unsigned off, uint32_t f2[kUnits][2]) const {
#pragma unroll
    for (int k = 0; k < kUnits; ++k) {
      const int t = 4 * (k * kLanes + u);
      f2[k][0] = f2[k][1] = 0;
      if (kFull || (k * 4 * kLanes < n_px && t < n_px)) load_pairs<T>(reinterpret_cast<const T *>(base0 + (lo_[k] + off)), f2[k]);
    }
  }
  // the sum of one group's lanes out of the 16-lane sums row_sum32 leaves in every lane (g is uniform)
  static __device__ __forceinline__ uint32_t group_total(uint32_t rows, int g) {
    if constexpr (G == 4) return (uint32_t)__builtin_amdgcn_readlane((int)rows, 16 * g);
    if constexpr (G == 2) return (uint32_t)__builtin_amdgcn_readlane((int)rows, 32 * g) + (uint32_t)__builtin_amdgcn_readlane((int)rows, 32 * g + 16);
    return rows_total32(rows);
  }
  // the lane's share of the SAD of the candidate at byte offset `off` (cand_off; with G > 1 a value per group): fetch() requests the
  // candidate's pixels (blocks kept in registers; a streamed block reads them as it goes), sad_partial() does the arithmetic
  struct Px { uint32_t f2[kUnits][2]; };
  __device__ __forceinline__ void fetch(unsigned off, Px &px) const {
    if (keep) load_ref(off, px.f2);
  }
  __device__ __forceinline__ uint32_t sad_partial(unsigned off, const Px &px) const {
    uint32_t acc = 0;
    if (keep) {
#pragma unroll
      for (int k = 0; k < kUnits; ++k)
        if (kFull || k * 4 * kLanes < n_px) acc = unit_sad(px.f2[k], sA_[k], sC_[k], sS_[k], acc);
    } else if constexpr (G == 1) {
#pragma unroll 1
      for (int t = 4 * lane; t < n_px; t += 256) {
        uint32_t A2[2], C2[2], S2[2], f2[2];
        load_pairs<T>(reinterpret_cast<const T *>(base0 + (ref_off(t) + off)), f2);
        operands(t, A2, C2, S2);
        acc = unit_sad(f2, A2, C2, S2, acc);
      }
    }
    return acc;
  }
  // G candidates at once: rows = sad_rows(offset per group), then sad_of(rows, g) for each
  __device__ __forceinline__ uint32_t sad_rows(unsigned off, const Px &px) const { return row_sum32(sad_partial(off, px)); }
  __device__ __forceinline__ uint32_t sad_rows(unsigned off) const {
    Px px;
    fetch(off, px);
    return sad_rows(off, px);
  }
  __device__ __forceinline__ uint32_t sad_of(uint32_t rows, int g) const { return group_total(rows, g) >> shift; }
  __device__ __forceinline__ uint32_t sad(int row, int col) const { return sad_of(sad_rows(cand_off(row, col)), 0); }
  __device__ __forceinline__ uint32_t var(int row, int col) const {   // (without the MV cost)
    const unsigned off = cand_off(row, col);
    int32_t s = 0;
    uint64_t q64;
    if (keep) {
      uint32_t q = 0;   // <= 4 x kUnits x 4095^2 per lane, 16 lanes of that per row: 256 x 4095^2 < 2^32
      uint32_t f2[kUnits][2];
      load_ref(off, f2);
#pragma unroll
      for (int k = 0; k < kUnits; ++k)
        if (kFull || k * 4 * kLanes < n_px) unit_var(f2[k], sA_[k], sC_[k], sS_[k], s, q);
      if constexpr (G == 1) q64 = wsum32_wide(q);
      else q64 = group_total(row_sum32(q), 0);   // (every group evaluated the same candidate; <= 128 pixels: 32 bits)
    } else if constexpr (G > 1) {
      q64 = 0;
    } else {
      uint64_t q = 0;
      for (int t = 4 * lane; t < n_px; t += 256) {
        uint32_t A2[2], C2[2], S2[2], f2[2], qu = 0;
        load_pairs<T>(reinterpret_cast<const T *>(base0 + (ref_off(t) + off)), f2);
        operands(t, A2, C2, S2);
        unit_var(f2, A2, C2, S2, s, qu);
        q += qu;
      }
      q64 = wsum_q(q);
    }
    return finish_var((int64_t)(int32_t)group_total(row_sum32((uint32_t)s), 0), q64, n_px, bit_depth);
  }
};

template <typename T, int UNITS, int G, bool PACKED>
__global__ __launch_bounds__(256) void refining_search_8p_kernel(PlaneView<T> src, PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks,
                                                                 int n_blocks, CompoundArgs a, const T *__restrict__ second_pred,
                                                                 const uint8_t *__restrict__ masks, int16_t *__restrict__ out_mv,
                                                                 int32_t *__restrict__ out_sad, int32_t *__restrict__ out_var) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + wave;
  if (bi >= n_blocks) return;
  const BlockScalars bs = BlockScalars::of(blocks[bi]);
  if (bs.row_min > bs.row_max) return;   // an EMPTY window marks a block the caller wants skipped (fullpel_search.inc): outputs stay as they are
  const int bx = __builtin_amdgcn_readfirstlane((int)blocks[bi].bx), by = __builtin_amdgcn_readfirstlane((int)blocks[bi].by);
  const int W = a.bw, H = a.bh, n_px = W * H;
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)by * src.stride + bx;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)by * ref.stride + bx;
  const T *pred = second_pred + (size_t)bi * n_px;
  const uint8_t *mask = masks ? masks + (size_t)bi * n_px : nullptr;
  const int frr = (bs.ref_row + 3 + (bs.ref_row >= 0)) >> 3, frc = (bs.ref_col + 3 + (bs.ref_col >= 0)) >> 3;
  const SadCost cost_of(a, frr, frc);   // mvsad_err_cost_
  CompoundEval<T, UNITS, G, PACKED> ce;
  ce.init(sp, src.stride, rbase, ref.stride, pred, mask, W, H, a.invert_mask, a.bit_depth, lane, bs.row_min, bs.col_min);
  auto sad_at = [&](int row, int col) -> uint32_t { return ce.sad(row, col); };   // get_mvpred_compound_sad
  constexpr int kRange = 3, kStride = 2 * kRange + 1;   // SEARCH_RANGE_8P, SEARCH_GRID_STRIDE_8P (mcomp_structs.h:26-29)
  unsigned long long visited = 0;                       // the 49 cells of do_refine_search_grid
  int grid_center = kRange * kStride + kRange;
  int row = min(max(bs.start_row, bs.row_min), bs.row_max), col = min(max(bs.start_col, bs.col_min), bs.col_max);   // clamp_fullmv
  uint32_t best_sad = sad_at(row, col) + (uint32_t)cost_of(row, col);
  visited |= 1ull << grid_center;
  // neighbors[] (:1623-1632): (-1,0) (0,-1) (0,1) (1,0) (-1,-1) (1,-1) (-1,1) (1,1)
  auto drow_of = [](int j) { return j == 0 || j == 4 || j == 6 ? -1 : (j == 3 || j == 5 || j == 7 ? 1 : 0); };
  auto dcol_of = [](int j) { return j == 1 || j == 4 || j == 5 ? -1 : (j == 2 || j == 6 || j == 7 ? 1 : 0); };
  // (the eight neighbours by lane, the scalar unit walks the bits of `valid` in neighbour order: see compound_full_pixel_diamond_kernel)
  const int dr_l = drow_of(lane & 7), dc_l = dcol_of(lane & 7);
  constexpr unsigned long long kNeighbours = 0x1C287ull;   // the cells (-1..1, -1..1) \ (0, 0) around cell 8 of the 7-wide grid
  for (int i = 0; i < kRange; ++i) {
    int best_site = -1;
    const int r_l = row + dr_l, c_l = col + dc_l, gc_l = grid_center + dr_l * kStride + dc_l;
    const bool ok_l = lane < 8 && !((visited >> gc_l) & 1) && (unsigned)(r_l - bs.row_min) <= (unsigned)(bs.row_max - bs.row_min) &&
                      (unsigned)(c_l - bs.col_min) <= (unsigned)(bs.col_max - bs.col_min);
    visited |= kNeighbours << (grid_center - 8);   // every neighbour is marked before its range test; the centre stays >= 8: two moves at most so far
    const unsigned off_l = ok_l ? ce.cand_off(r_l, c_l) : 0u;
    uint32_t valid = (uint32_t)__ballot(ok_l);
    walk_sites<G, typename CompoundEval<T, UNITS, G, PACKED>::Px>(
        valid, off_l, ce.cand_off(row, col), ce.grp, [&](unsigned off, auto &px) { ce.fetch(off, px); },
        [&](unsigned off, const auto &px) { return ce.sad_rows(off, px); },
        [&](int j, uint32_t rows, int g) {
          uint32_t sad = ce.sad_of(rows, g);
          if (sad < best_sad) {
            sad += (uint32_t)cost_of(row + drow_of(j), col + dcol_of(j));
            if (sad < best_sad) {
              best_sad = sad;
              best_site = j;
            }
          }
        });
    if (best_site == -1) break;
    const int j = best_site;
    const int drow = j == 0 || j == 4 || j == 6 ? -1 : (j == 3 || j == 5 || j == 7 ? 1 : 0);
    const int dcol = j == 1 || j == 4 || j == 5 ? -1 : (j == 2 || j == 6 || j == 7 ? 1 : 0);
    row += drow; col += dcol;
    grid_center += drow * kStride + dcol;
  }
  // av1_get_mvpred_compound_var: svaf / msvf at sub-pel offset (0, 0) -- the bilinear passes with offset 0 are the identity -- + mv_err_cost_
  const int var = (int)ce.var(row, col) + var_cost(a, bs.ref_row, bs.ref_col, row * 8, col * 8);
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)row; out_mv[2 * bi + 1] = (int16_t)col;
    out_sad[bi] = (int32_t)best_sad;
    out_var[bi] = var;
  }
}

// full_pixel_diamond (mcomp.c:1421-1470) on a COMPOUND prediction: diamond_search_sad (:1299-1416) takes its per-site branch with
// get_mvpred_compound_sad whenever ms_buffers.second_pred is set (:1347), every run ends on get_mvpred_compound_var_cost (:676-708) and
// *second_best_mv follows every move of every run.  What av1_full_pixel_search does after it (:1756-1830) -- on the PLAIN sdf / vf even on a
// compound -- is the general kernel's (fullpel_search.inc, SearchArgs::resume).
template <typename T, int UNITS, int G, bool PACKED>
__global__ __launch_bounds__(256) void compound_full_pixel_diamond_kernel(PlaneView<T> src, PlaneView<T> ref, int frame,
                                                                          const aomhip_search_block *__restrict__ blocks, int n_blocks, CompoundArgs a,
                                                                          const SiteTable *__restrict__ sites, int step_param,
                                                                          const T *__restrict__ second_pred, const uint8_t *__restrict__ masks,
                                                                          int16_t *__restrict__ out_mv, int32_t *__restrict__ out_cost,
                                                                          int16_t *__restrict__ out_second) {
  __shared__ SiteTable S;
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(sites);
    uint32_t *d = reinterpret_cast<uint32_t *>(&S);
    for (int i = threadIdx.x; i < (int)(sizeof(SiteTable) / 4); i += 256) d[i] = g[i];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + wave;
  if (bi >= n_blocks) return;
  const BlockScalars bs = BlockScalars::of(blocks[bi]);
  if (bs.row_min > bs.row_max) return;   // an EMPTY window marks a block the caller wants skipped (fullpel_search.inc): outputs stay as they are
  const int bx = __builtin_amdgcn_readfirstlane((int)blocks[bi].bx), by = __builtin_amdgcn_readfirstlane((int)blocks[bi].by);
  const int W = a.bw, H = a.bh, n_px = W * H;
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)by * src.stride + bx;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)by * ref.stride + bx;
  const T *pred = second_pred + (size_t)bi * n_px;
  const uint8_t *mask = masks ? masks + (size_t)bi * n_px : nullptr;
  const int frr = (bs.ref_row + 3 + (bs.ref_row >= 0)) >> 3, frc = (bs.ref_col + 3 + (bs.ref_col >= 0)) >> 3;
  const SadCost cost_of(a, frr, frc);   // mvsad_err_cost_
  CompoundEval<T, UNITS, G, PACKED> ce;
  ce.init(sp, src.stride, rbase, ref.stride, pred, mask, W, H, a.invert_mask, a.bit_depth, lane, bs.row_min, bs.col_min);
  auto sad_at = [&](int row, int col) -> uint32_t { return ce.sad(row, col); };   // get_mvpred_compound_sad: sdaf / msdf
  auto var_at = [&](int row, int col) -> int {   // get_mvpred_compound_var_cost: svaf / msvf at offset (0, 0) + mv_err_cost_
    return (int)ce.var(row, col) + var_cost(a, bs.ref_row, bs.ref_col, row * 8, col * 8);
  };
  const int start_row = min(max(bs.start_row, bs.row_min), bs.row_max), start_col = min(max(bs.start_col, bs.col_min), bs.col_max);   // clamp_fullmv
  const uint32_t start_sad = sad_at(start_row, start_col) + (uint32_t)cost_of(start_row, start_col);   // (the same in every run)
  int second_row = -32768, second_col = -32768;   // MARK_MV_INVALID (av1_full_pixel_search, :1704-1707)
  const int nsteps = __builtin_amdgcn_readfirstlane(S.num_search_steps);
  auto diamond = [&](int search_step, int *num00, int *orow, int *ocol) -> int {
    // (the table lives in LDS: what is read from it arrives in a VGPR, and a search state derived from a VGPR is kept in the vector unit and
    // walked under exec masks -- 7 700 scalar + 4 000 vector instructions per block, PMC.  v_readfirstlane / v_readlane keep row, col, the
    // limits tests and the loop on the scalar unit; a stage's sites are one LDS read per lane, fetched by lane index)
    const int tot_steps = nsteps - search_step;
    int row = start_row, col = start_col;
    *num00 = 0;
    uint32_t bestsad = start_sad;
    int is_off_center = 0;
    int next_step_size = tot_steps > 2 ? __builtin_amdgcn_readfirstlane(S.radius[tot_steps - 2]) : 1;
    for (int step = tot_steps - 1; step >= 0; --step) {
      int best_site = 0;
      if (step > 0) next_step_size = __builtin_amdgcn_readfirstlane(S.radius[step - 1]);
      const int this_radius = __builtin_amdgcn_readfirstlane(S.radius[step]);
      const int nper = __builtin_amdgcn_readfirstlane(S.searches_per_step[step]);
      const int my_site = *reinterpret_cast<const int *>(&S.mv[step][lane < 17 ? lane : 0][0]);   // (row, col) of site `lane` as one dword
      // The stage's sites by lane: position, av1_is_fullmv_in_range and the candidate's offset are worked out for all of them at once in the
      // vector unit; the scalar unit then only walks the bits of `valid` in site order (per site: ~10 scalar instructions instead of ~40).
      const int r_l = row + (int)(int16_t)(my_site & 0xffff), c_l = col + (my_site >> 16);
      const bool ok_l = lane >= 1 && lane <= nper && (unsigned)(r_l - bs.row_min) <= (unsigned)(bs.row_max - bs.row_min) &&
                        (unsigned)(c_l - bs.col_min) <= (unsigned)(bs.col_max - bs.col_min);
      const unsigned off_l = ok_l ? ce.cand_off(r_l, c_l) : 0u;
      walk_sites<G, typename CompoundEval<T, UNITS, G, PACKED>::Px>(
          (uint32_t)__ballot(ok_l), off_l, 0u, ce.grp, [&](unsigned off, auto &px) { ce.fetch(off, px); },
          [&](unsigned off, const auto &px) { return ce.sad_rows(off, px); },
          [&](int idx, uint32_t rows, int g) {
            uint32_t sad = ce.sad_of(rows, g);
            if (sad < bestsad) {
              const int site = __builtin_amdgcn_readlane(my_site, idx);
              // (the cost is looked up only for a site that can win: issuing it for every site beside the pixel reads was SLOWER, 7.7 -> 8.9 ms)
              sad += (uint32_t)cost_of(row + (int)(int16_t)(site & 0xffff), col + (site >> 16));
              if (sad < bestsad) {
                bestsad = sad;
                best_site = idx;
              }
            }
          });
      if (best_site != 0) {
        second_row = row; second_col = col;
        const int site = __builtin_amdgcn_readlane(my_site, best_site);
        row += (int)(int16_t)(site & 0xffff);
        col += site >> 16;
        is_off_center = 1;
      }
      if (is_off_center == 0) (*num00)++;
      if (best_site == 0) {
        int rad = this_radius;
        while (next_step_size == rad && step > 2) {
          ++(*num00);
          --step;
          rad = __builtin_amdgcn_readfirstlane(S.radius[step]);
          next_step_size = __builtin_amdgcn_readfirstlane(S.radius[step - 1]);
        }
      }
    }
    *orow = row; *ocol = col;
    return (int)bestsad;
  };
  int n, num00 = 0, tr, tc;
  int bestsme = diamond(step_param, &n, &tr, &tc);
  if (bestsme < INT_MAX) bestsme = var_at(tr, tc);
  int best_row = tr, best_col = tc;
  const int further_steps = nsteps - 1 - step_param;
  while (n < further_steps) {
    ++n;
    if (num00) {
      num00--;
    } else {
      int thissme = diamond(step_param + n, &num00, &tr, &tc);
      if (thissme < INT_MAX) thissme = var_at(tr, tc);
      if (thissme < bestsme) {
        bestsme = thissme;
        best_row = tr; best_col = tc;
      }
    }
  }
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)best_row; out_mv[2 * bi + 1] = (int16_t)best_col;
    out_cost[bi] = bestsme;
    out_second[2 * bi] = (int16_t)second_row; out_second[2 * bi + 1] = (int16_t)second_col;
  }
}

template <typename T, int UNITS, int G>
__global__ __launch_bounds__(256) void obmc_full_pixel_search_kernel(PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks, int n_blocks,
                                                                     CompoundArgs a, const SiteTable *__restrict__ sites, int step_param, int fast,
                                                                     const int32_t *__restrict__ wsrc_all, const int32_t *__restrict__ omask_all,
                                                                     int16_t *__restrict__ out_mv, int32_t *__restrict__ out_cost) {
  __shared__ SiteTable S;
  {
    const uint32_t *g = reinterpret_cast<const uint32_t *>(sites);
    uint32_t *d = reinterpret_cast<uint32_t *>(&S);
    for (int i = threadIdx.x; i < (int)(sizeof(SiteTable) / 4); i += 256) d[i] = g[i];
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + wave;
  if (bi >= n_blocks) return;
  const BlockScalars bs = BlockScalars::of(blocks[bi]);
  const int bx = __builtin_amdgcn_readfirstlane((int)blocks[bi].bx), by = __builtin_amdgcn_readfirstlane((int)blocks[bi].by);
  const int W = a.bw, H = a.bh, n_px = W * H;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)by * ref.stride + bx;
  const int32_t *wsrc = wsrc_all + (size_t)bi * n_px, *omask = omask_all + (size_t)bi * n_px;
  const int shift = a.bit_depth == 10 ? 2 : a.bit_depth == 12 ? 4 : 0;
  const int frr = (bs.ref_row + 3 + (bs.ref_row >= 0)) >> 3, frc = (bs.ref_col + 3 + (bs.ref_col >= 0)) >> 3;
  const SadCost cost_of(a, frr, frc);   // mvsad_err_cost_
  // As CompoundEval: the weighted source and the mask of the block stay in registers for blocks of up to 256 * UNITS pixels, a candidate is then
  // 2 * UNITS independent reference loads per lane; larger blocks stream them four pixels at a time.  Widths are powers of two (row / column
  // of pixel t by shift and mask).
  // G > 1: the wavefront as G groups of lanes, each holding the whole (small) block and evaluating its own candidate (see CompoundEval).
  constexpr int kUnits = UNITS, kLanes = 64 / G;   // 4-pixel units per lane, lanes per group
  static_assert(G == 1 || UNITS == 1, "groups hold one unit per lane");
  const int u = lane & (kLanes - 1), grp = lane / kLanes;
  const int lw = __builtin_ctz((unsigned)W), wm = W - 1;
  constexpr bool kFull = G == 1;   // (G == 1: a kept block has exactly 256 * UNITS pixels -- see CompoundEval)
  const bool keep = kFull ? n_px == 4 * kLanes * kUnits : n_px <= 4 * kLanes * kUnits;
  // |wsrc - pre * mask| + 2048 as ONE v_sad_u32 on operands biased by 2^31 (the unsigned difference of the biased values is the signed one's
  // magnitude), pre * mask + 2^31 as one v_mad_u32_u24 (pre < 2^12; mask <= 4096 = 64 x 64 as calc_target_weighted_pred builds it, any
  // value below 2^24 works): 4 instructions per pixel with the shift and the sum instead of 7
  constexpr int kBias = (int)0x80000000u;
  auto round_abs12 = [](int ws_biased, int f, int m) -> uint32_t {
    uint32_t pm, r;   // (no builtin for v_sad_u32, and the compiler splits the biased product into v_mul_u32_u24 + v_xor: the constants ride in SGPRs)
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(pm) : "v"((uint32_t)f), "v"((uint32_t)m), "s"(0x80000000u));
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"((uint32_t)ws_biased), "v"(pm), "s"(2048u));
    return r >> 12;   // ROUND_POWER_OF_TWO(abs(..), 12)
  };
  int ws_[kUnits][4], om_[kUnits][4];
  // candidate pixels by 32-bit byte offset from the block at the window's top-left MV (see CompoundEval)
  const char *base0 = reinterpret_cast<const char *>(rbase + (int64_t)bs.row_min * ref.stride + bs.col_min);
  auto ref_off = [&](int t) -> unsigned { return (unsigned)(((t >> lw) * ref.stride + (t & wm)) * (int)sizeof(T)); };
  unsigned lo_[kUnits];
  if (keep) {
#pragma unroll
    for (int k = 0; k < kUnits; ++k) {
      const int t = 4 * (k * kLanes + u);
      lo_[k] = kFull || t < n_px ? ref_off(t) : 0u;
      int4 a = make_int4(0, 0, 0, 0), b = a;
      if (kFull || t < n_px) { a = *reinterpret_cast<const int4 *>(wsrc + t); b = *reinterpret_cast<const int4 *>(omask + t); }
      ws_[k][0] = a.x ^ kBias; ws_[k][1] = a.y ^ kBias; ws_[k][2] = a.z ^ kBias; ws_[k][3] = a.w ^ kBias;
      om_[k][0] = b.x; om_[k][1] = b.y; om_[k][2] = b.z; om_[k][3] = b.w;
    }
  }
  auto cand_off = [&](int row, int col) -> unsigned { return (unsigned)(((row - bs.row_min) * ref.stride + (col - bs.col_min)) * (int)sizeof(T)); };
  // vfp->osdf: obmc_sad (sad_av1.c:163-180), the lane's share for the candidate at byte offset `off` (with G > 1 a value per group); obmc_sad
  // sums in an unsigned int, so do the lanes and the reductions (modulo 2^32 like the reference)
  struct Px { int f[kUnits][4]; };
  auto ofetch = [&](unsigned off, Px &px) {   // the candidate's pixels requested (blocks kept in registers; a streamed block reads them as it goes)
    if (keep) {
#pragma unroll
      for (int k = 0; k < kUnits; ++k) {
        const int t = 4 * (k * kLanes + u);
#pragma unroll
        for (int i = 0; i < 4; ++i) px.f[k][i] = 0;
        if (kFull || (k * 4 * kLanes < n_px && t < n_px)) load_px4<T>(reinterpret_cast<const T *>(base0 + (lo_[k] + off)), px.f[k]);
      }
    }
  };
  auto osad_partial = [&](unsigned off, const Px &px) -> uint32_t {
    uint32_t acc = 0;
    if (keep) {
#pragma unroll
      for (int k = 0; k < kUnits; ++k)
        if (kFull || k * 4 * kLanes < n_px) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc += round_abs12(ws_[k][i], px.f[k][i], om_[k][i]);   // (beyond the block: |0 - 0| + 2048 >> 12 = 0)
        }
    } else if constexpr (G == 1) {
#pragma unroll 1   // (unrolled by two the int4 loads took the kernel to ~300 VGPRs)
      for (int t = 4 * lane; t < n_px; t += 256) {
        int f[4];
        load_px4<T>(reinterpret_cast<const T *>(base0 + (ref_off(t) + off)), f);
        const int4 w = *reinterpret_cast<const int4 *>(wsrc + t), m = *reinterpret_cast<const int4 *>(omask + t);
        acc += round_abs12(w.x ^ kBias, f[0], m.x) + round_abs12(w.y ^ kBias, f[1], m.y) + round_abs12(w.z ^ kBias, f[2], m.z) +
               round_abs12(w.w ^ kBias, f[3], m.w);
      }
    }
    return acc;
  };
  auto osad_rows = [&](unsigned off, const Px &px) -> uint32_t { return row_sum32(osad_partial(off, px)); };
  auto osad_of = [&](uint32_t rows, int g) -> uint32_t { return CompoundEval<T, UNITS, G>::group_total(rows, g) >> shift; };   // + the bit-depth wrapper
  auto osad_at = [&](int row, int col) -> uint32_t {
    Px px;
    ofetch(cand_off(row, col), px);
    return osad_of(osad_rows(cand_off(row, col), px), 0);
  };
  auto ovar_at = [&](int row, int col) -> int {   // get_obmc_mvpred_var: vfp->ovf (variance.c:957-1000 / :1064-1192) + mv_err_cost_
    const unsigned off = cand_off(row, col);
    int64_t s = 0, q = 0;
#pragma unroll 1
    for (int t = 4 * lane; t < n_px; t += 256) {   // (once per run: every lane of the wavefront, whatever G)
      int f[4];
      load_px4<T>(reinterpret_cast<const T *>(base0 + (ref_off(t) + off)), f);
      const int4 w4 = *reinterpret_cast<const int4 *>(wsrc + t), m4 = *reinterpret_cast<const int4 *>(omask + t);
      const int w[4] = { w4.x, w4.y, w4.z, w4.w }, m[4] = { m4.x, m4.y, m4.z, m4.w };
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int v = w[i] - f[i] * m[i];
        const int d = v < 0 ? -((-v + 2048) >> 12) : (v + 2048) >> 12;   // ROUND_POWER_OF_TWO_SIGNED(v, 12)
        s += d;
        q += (uint32_t)(d * d);
      }
    }
    return (int)finish_var(wsum_s(s), wsum_q((uint64_t)q), n_px, a.bit_depth) + var_cost(a, bs.ref_row, bs.ref_col, row * 8, col * 8);
  };
  auto in_range = [&](int r, int c) { return c >= bs.col_min && c <= bs.col_max && r >= bs.row_min && r <= bs.row_max; };
  const int start_row = min(max(bs.start_row, bs.row_min), bs.row_max), start_col = min(max(bs.start_col, bs.col_min), bs.col_max);
  int best_row, best_col, result;
  const int nsteps = __builtin_amdgcn_readfirstlane(S.num_search_steps);
  if (!fast) {
    auto diamond = [&](int search_step, int *num00, int *orow, int *ocol) -> int {   // obmc_diamond_search_sad
      const int tot_steps = nsteps - search_step;   // (table reads through v_readfirstlane / v_readlane: see compound_full_pixel_diamond_kernel)
      int row = start_row, col = start_col;
      *num00 = 0;
      int best_sad = (int)(osad_at(row, col) + (uint32_t)cost_of(row, col));
      for (int step = tot_steps - 1; step >= 0; --step) {
        int best_site = 0;
        const int nper = __builtin_amdgcn_readfirstlane(S.searches_per_step[step]);
        const int my_site = *reinterpret_cast<const int *>(&S.mv[step][lane < 17 ? lane : 0][0]);
        // (the stage's sites by lane, the scalar unit walks the bits of `valid`: see compound_full_pixel_diamond_kernel)
        const int r_l = row + (int)(int16_t)(my_site & 0xffff), c_l = col + (my_site >> 16);
        const bool ok_l = lane >= 1 && lane <= nper && (unsigned)(r_l - bs.row_min) <= (unsigned)(bs.row_max - bs.row_min) &&
                          (unsigned)(c_l - bs.col_min) <= (unsigned)(bs.col_max - bs.col_min);
        const unsigned off_l = ok_l ? cand_off(r_l, c_l) : 0u;
        walk_sites<G, Px>((uint32_t)__ballot(ok_l), off_l, 0u, grp, ofetch, osad_rows, [&](int idx, uint32_t rows, int g) {
          int sad = (int)osad_of(rows, g);   // (`int sad < int best_sad`: this function compares signed, mcomp.c:2206-2215)
          if (sad < best_sad) {
            const int site = __builtin_amdgcn_readlane(my_site, idx);
            sad += cost_of(row + (int)(int16_t)(site & 0xffff), col + (site >> 16));
            if (sad < best_sad) {
              best_sad = sad;
              best_site = idx;
            }
          }
        });
        if (best_site != 0) {
          const int site = __builtin_amdgcn_readlane(my_site, best_site);
          row += (int)(int16_t)(site & 0xffff);
          col += site >> 16;
        } else if (row == start_row && col == start_col) {   // best_address == init_ref
          (*num00)++;
        }
      }
      *orow = row; *ocol = col;
      return best_sad;
    };
    int n, num00 = 0, tr, tc;   // obmc_full_pixel_diamond
    int bestsme = diamond(step_param, &n, &tr, &tc);
    if (bestsme < INT_MAX) bestsme = ovar_at(tr, tc);
    best_row = tr; best_col = tc;
    const int further_steps = nsteps - 1 - step_param;
    while (n < further_steps) {
      ++n;
      if (num00) {
        num00--;
      } else {
        int thissme = diamond(step_param + n, &num00, &tr, &tc);
        if (thissme < INT_MAX) thissme = ovar_at(tr, tc);
        if (thissme < bestsme) {
          bestsme = thissme;
          best_row = tr; best_col = tc;
        }
      }
    }
    result = bestsme;
  } else {   // obmc_refining_search_sad from the clamped start MV
    int row = start_row, col = start_col;
    uint32_t best_sad = osad_at(row, col) + (uint32_t)cost_of(row, col);
    for (int i = 0; i < 8; ++i) {
      int best_site = -1;
      // neighbors[4] = (-1,0) (0,-1) (0,1) (1,0), one per lane
      const int jl = lane & 3, r_l = row + (jl == 0 ? -1 : jl == 3 ? 1 : 0), c_l = col + (jl == 1 ? -1 : jl == 2 ? 1 : 0);
      const bool ok_l = lane < 4 && in_range(r_l, c_l);
      walk_sites<G, Px>((uint32_t)__ballot(ok_l), ok_l ? cand_off(r_l, c_l) : 0u, 0u, grp, ofetch, osad_rows, [&](int j, uint32_t rows, int g) {
        uint32_t sad = osad_of(rows, g);
        if (sad < best_sad) {
          sad += (uint32_t)cost_of(row + (j == 0 ? -1 : j == 3 ? 1 : 0), col + (j == 1 ? -1 : j == 2 ? 1 : 0));
          if (sad < best_sad) {
            best_sad = sad;
            best_site = j;
          }
        }
      });
      if (best_site == -1) break;
      row += best_site == 0 ? -1 : best_site == 3 ? 1 : 0;
      col += best_site == 1 ? -1 : best_site == 2 ? 1 : 0;
    }
    best_row = row; best_col = col;
    result = (int)best_sad;
    if (result < INT_MAX) result = ovar_at(row, col);
  }
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)best_row; out_mv[2 * bi + 1] = (int16_t)best_col;
    out_cost[bi] = result;
  }
}

// ---- av1_find_best_obmc_sub_pixel_tree_up (mcomp.c:3588-3633): the sub-pel half of the OBMC motion search.  One wavefront per block, the
// candidates of obmc_first_level_check / obmc_second_level_check_v2 one after the other in the reference's order (they are few -- at most 8
// per precision -- and each needs a filtered prediction before the weighted difference), every evaluation by all 64 lanes pixel by pixel:
//   USE_2_TAPS_ORIG  vfp->osvf = aom_[highbd_]obmc_sub_pixel_variance (variance.c:1002-1062 / :1194-1330: the two bilinear passes, each rounded
//                    by FILTER_BITS, then obmc_variance) + estimate_obmc_mvcost (:3390-3412); the centre by setup_obmc_center_error (:3359-3374),
//                    which measures ms_buffers->ref->buf -- MV 0 -- whatever the start MV is (the reference's own TODO; reproduced)
//   USE_8_TAPS       upsampled_obmc_pref_error (:3314-3357): aom_[highbd_]upsampled_pred = the 8-tap regular kernel of phase 2 * (mv & 7),
//                    horizontal then vertical pass, each rounded and clipped (taps 0 and 7 are zero in every phase: six taps), then vfp->ovf,
//                    + mv_err_cost_
__device__ constexpr int16_t kObmcSubPel8[16][6] = {   // av1_sub_pel_filters_8 (av1/common/filter.h:124-141), taps 1 .. 6
  { 0, 0, 128, 0, 0, 0 },      { 2, -6, 126, 8, -2, 0 },    { 2, -10, 122, 18, -4, 0 },  { 2, -12, 116, 28, -8, 2 },
  { 2, -14, 110, 38, -10, 2 }, { 2, -14, 102, 48, -12, 2 }, { 2, -16, 94, 58, -12, 2 },  { 2, -14, 84, 66, -12, 2 },
  { 2, -14, 76, 76, -14, 2 },  { 2, -12, 66, 84, -14, 2 },  { 2, -12, 58, 94, -16, 2 },  { 2, -12, 48, 102, -14, 2 },
  { 2, -10, 38, 110, -14, 2 }, { 2, -8, 28, 116, -12, 2 },  { 0, -4, 18, 122, -10, 2 },  { 0, -2, 8, 126, -6, 2 }
};
__device__ constexpr int16_t kObmcSubPel4[16][6] = {   // av1_sub_pel_filters_4 (filter.h:205-215; USE_4_TAPS, av1_get_filter :270-279), taps 1 .. 6
  { 0, 0, 128, 0, 0, 0 },     { 0, -4, 126, 8, -2, 0 },   { 0, -8, 122, 18, -4, 0 },  { 0, -10, 116, 28, -6, 0 },
  { 0, -12, 110, 38, -8, 0 }, { 0, -12, 102, 48, -10, 0 }, { 0, -14, 94, 58, -10, 0 }, { 0, -12, 84, 66, -10, 0 },
  { 0, -12, 76, 76, -12, 0 }, { 0, -10, 66, 84, -12, 0 }, { 0, -10, 58, 94, -14, 0 }, { 0, -10, 48, 102, -12, 0 },
  { 0, -8, 38, 110, -12, 0 }, { 0, -6, 28, 116, -10, 0 }, { 0, -4, 18, 122, -8, 0 },  { 0, -2, 8, 126, -4, 0 }
};
// tap k + 1 of the kernel of phase ph (0 .. 15) under SUBPEL_SEARCH_TYPE type: 1 USE_2_TAPS = av1_bilinear_filters, 2 USE_4_TAPS, 3 USE_8_TAPS
__device__ __forceinline__ int obmc_up_tap(int type, int ph, int k) {
  if (type == 1) return k == 2 ? 128 - 8 * ph : (k == 3 ? 8 * ph : 0);
  return type == 2 ? kObmcSubPel4[ph][k] : kObmcSubPel8[ph][k];
}
struct ObmcSubpelArgs { int iters_per_step, allow_hp, forced_stop, upsampled; };   // upsampled: 0, or the SUBPEL_SEARCH_TYPE (1 / 2 / 3)

template <typename T>
__global__ __launch_bounds__(256) void obmc_subpel_tree_kernel(PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks, int n_blocks,
                                                               CompoundArgs a, ObmcSubpelArgs sa, const int32_t *__restrict__ wsrc_all,
                                                               const int32_t *__restrict__ omask_all, int16_t *__restrict__ out_mv,
                                                               uint32_t *__restrict__ out_err, int32_t *__restrict__ out_dist, uint32_t *__restrict__ out_sse) {
  constexpr int kObmcTileDw = 1024;   // 4 KB of LDS per wavefront: the horizontally filtered rows of a strip (form 2 below)
  __shared__ uint32_t tiles[4][kObmcTileDw];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  uint32_t *tile = tiles[wave];
  const int bi = blockIdx.x * 4 + wave;
  if (bi >= n_blocks) return;
  const BlockScalars bs = BlockScalars::of(blocks[bi]);   // start_* in 1/8 pel, limits = SubpelMvLimits
  const int bx = __builtin_amdgcn_readfirstlane((int)blocks[bi].bx), by = __builtin_amdgcn_readfirstlane((int)blocks[bi].by);
  const int W = a.bw, H = a.bh, n_px = W * H;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)by * ref.stride + bx;
  const int32_t *wsrc = wsrc_all + (size_t)bi * n_px, *omask = omask_all + (size_t)bi * n_px;
  const int pmax = sizeof(T) == 1 ? 255 : (1 << a.bit_depth) - 1;
  const int lw_ = __builtin_ctz((unsigned)W);
  constexpr uint8_t kBil[8][2] = { { 128, 0 }, { 112, 16 }, { 96, 32 }, { 80, 48 }, { 64, 64 }, { 48, 80 }, { 32, 96 }, { 16, 112 } };  // aom_filter.h:43-50
  // the up-sampling kernel's tap pairs (taps 2 k, 2 k + 1 of the six) of the eight 1/8-pel phases, one per lane: lane = 3 * phase + k.  (Looked
  // up per candidate through obmc_up_tap they were twelve table reads on the scalar unit, which bounds this kernel: PMC r05e.)
  int tap_l = 0;
  if (sa.upsampled && lane < 24) {
    const int ph = 2 * (lane / 3), k = lane % 3;
    tap_l = (int)(((uint32_t)(uint16_t)(int16_t)obmc_up_tap(sa.upsampled, ph, 2 * k)) | ((uint32_t)(uint16_t)(int16_t)obmc_up_tap(sa.upsampled, ph, 2 * k + 1) << 16));
  }
  // obmc_variance of the prediction at (mrow, mcol): form 0 = the plain block at the full-pel part (ovf), 1 = bilinear (osvf), 2 = up-sampled
  auto obmc_err = [&](int mrow, int mcol, int form, uint32_t *sse_out) -> uint32_t {
    const T *rp = rbase + (int64_t)(mrow >> 3) * ref.stride + (mcol >> 3);
    const int sx = mcol & 7, sy = mrow & 7;
    static_assert(kBil[3][0] == 128 - 16 * 3 && kBil[3][1] == 16 * 3 && kBil[7][0] == 16, "bilinear taps are 128 - 16 i, 16 i");
    const int fx1 = (sx & 7) << 4, fx0 = 128 - fx1, fy1 = (sy & 7) << 4, fy0 = 128 - fy1;   // (by arithmetic: a table index is a memory load)
    int64_t s = 0, q = 0;
    if (form == 2) {
      // The up-sampled form in two passes through the wavefront's LDS tile, a lane taking units of four adjacent pixels: the horizontal pass
      // filters every row of a strip ONCE (rows -2 .. +3 around it: three 4-pixel loads, 9 of the 12 pixels are taps) and stores it as
      // uint16 pairs; the vertical pass reads six rows of a pixel pair as dwords -- three v_perm + three v_dot2 per pixel, as in the
      // compound search's kernel (subpel_search.inc) -- and takes the weighted difference.  (Row by row per lane, the form before, every lane
      // filtered six rows for one row of output: 9 846 vector instructions per 16 x 16 block, PMC r05e.)
      // (The right-most load reaches 6 pixels beyond the block, 3 more than the taps: inside the 8 pixels the MV limits keep clear.)
      typedef short s16x2_t __attribute__((ext_vector_type(2)));
      s16x2_t kxp[3], kyp[3];   // (the tap pairs of the two phases out of the per-lane table: six v_readlane)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        kxp[k] = __builtin_bit_cast(s16x2_t, __builtin_amdgcn_readlane(tap_l, sx * 3 + k));
        kyp[k] = __builtin_bit_cast(s16x2_t, __builtin_amdgcn_readlane(tap_l, sy * 3 + k));
      }
      const int lupr = lw_ - 2, upr = 1 << lupr;                   // 4-pixel units per row
      const int S = min(H, (kObmcTileDw * 2 >> lw_) - 5);          // output rows per strip: (S + 5) rows of W uint16 fill the tile
      for (int r0 = 0; r0 < H; r0 += S) {
        const int sh = min(S, H - r0);
        for (int uu = lane; uu < ((sh + 5) << lupr); uu += 64) {   // horizontal pass: rows r0 - 2 .. r0 + sh + 2 -> tile rows 0 .. sh + 4
          const int tr = uu >> lupr, c = (uu & (upr - 1)) << 2;
          const T *r = rp + (int64_t)(r0 + tr - 2) * ref.stride + c - 2;
          uint32_t wv[6];   // pixels c - 2 .. c + 9 as pairs
          if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const uint2 w = *reinterpret_cast<const uint2 *>(r + 4 * k);
              wv[2 * k] = w.x; wv[2 * k + 1] = w.y;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const uint32_t w = *reinterpret_cast<const uint32_t *>(r + 4 * k);
              wv[2 * k] = __builtin_amdgcn_perm(0, w, 0x0c010c00);
              wv[2 * k + 1] = __builtin_amdgcn_perm(0, w, 0x0c030c02);
            }
          }
          uint32_t o2[2];
          if (!sx) {
            o2[0] = wv[1]; o2[1] = wv[2];
          } else {
            uint32_t ov[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) ov[k] = __builtin_amdgcn_alignbit(wv[k + 1], wv[k], 16);
            int o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint32_t *sw = (i & 1) ? ov : wv;
              int acc = 64;
#pragma unroll
              for (int k = 0; k < 3; ++k) acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, sw[(i >> 1) + k]), kxp[k], acc, false);
              o[i] = min(max(acc >> 7, 0), pmax);
            }
            o2[0] = (uint32_t)o[0] | ((uint32_t)o[1] << 16); o2[1] = (uint32_t)o[2] | ((uint32_t)o[3] << 16);
          }
          uint32_t *dst = tile + ((tr << lw_) + c) / 2;
          dst[0] = o2[0]; dst[1] = o2[1];
        }
        // the wavefront's own LDS writes must be visible to its reads below: one wavefront, so a wave-level fence suffices
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int uu = lane; uu < (sh << lupr); uu += 64) {   // vertical pass + the weighted difference
          const int orow = uu >> lupr, c = (uu & (upr - 1)) << 2;
          uint32_t pp[2];
          if (!sy) {
            const uint32_t *src = tile + (((orow + 2) << lw_) + c) / 2;
            pp[0] = src[0]; pp[1] = src[1];
          } else {
            uint32_t rows6[6][2];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              const uint32_t *src = tile + (((orow + k) << lw_) + c) / 2;
              rows6[k][0] = src[0]; rows6[k][1] = src[1];
            }
#pragma unroll
            for (int qd = 0; qd < 2; ++qd) {
              int a0 = 64, a1 = 64;
#pragma unroll
              for (int k = 0; k < 3; ++k) {
                const uint32_t lo = __builtin_amdgcn_perm(rows6[2 * k + 1][qd], rows6[2 * k][qd], 0x05040100u);
                const uint32_t hi = __builtin_amdgcn_perm(rows6[2 * k + 1][qd], rows6[2 * k][qd], 0x07060302u);
                a0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, lo), kyp[k], a0, false);
                a1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2_t, hi), kyp[k], a1, false);
              }
              pp[qd] = (uint32_t)min(max(a0 >> 7, 0), pmax) | ((uint32_t)min(max(a1 >> 7, 0), pmax) << 16);
            }
          }
          const int t = ((r0 + orow) << lw_) + c;
          const int4 wv4 = *reinterpret_cast<const int4 *>(wsrc + t), mv4 = *reinterpret_cast<const int4 *>(omask + t);
          const int w4[4] = { wv4.x, wv4.y, wv4.z, wv4.w }, m4[4] = { mv4.x, mv4.y, mv4.z, mv4.w };
          const int pv[4] = { (int)(pp[0] & 0xffffu), (int)(pp[0] >> 16), (int)(pp[1] & 0xffffu), (int)(pp[1] >> 16) };
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int v = w4[i] - pv[i] * m4[i];
            const int d = v < 0 ? -((-v + 2048) >> 12) : (v + 2048) >> 12;   // ROUND_POWER_OF_TWO_SIGNED(v, 12)
            s += d;
            q += (uint32_t)(d * d);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    } else
    for (int t = lane; t < n_px; t += 64) {
      const int y = t >> lw_, x = t & (W - 1);   // (block widths are powers of two)
      const T *p = rp + (int64_t)y * ref.stride + x;
      int pv;
      if (form == 0) {
        pv = (int)p[0];
      } else if (form == 1) {   // aom_var_filter_block2d_bil_first_pass / _second_pass
        const int h0 = ((int)p[0] * fx0 + (int)p[1] * fx1 + 64) >> 7;
        const int h1 = ((int)p[ref.stride] * fx0 + (int)p[ref.stride + 1] * fx1 + 64) >> 7;
        pv = (h0 * fy0 + h1 * fy1 + 64) >> 7;
      } else {
        auto hrow = [&](int dy) -> int {   // the horizontal pass at row y + dy
          const T *r = p + (int64_t)dy * ref.stride;
          if (!sx) return (int)r[0];
          int sum = 0;
#pragma unroll
          for (int k = 0; k < 6; ++k) sum += (int)r[k - 2] * obmc_up_tap(sa.upsampled, 2 * sx, k);
          return min(max((sum + 64) >> 7, 0), pmax);
        };
        if (!sy) {
          pv = hrow(0);
        } else {
          int sum = 0;
#pragma unroll
          for (int k = 0; k < 6; ++k) sum += hrow(k - 2) * obmc_up_tap(sa.upsampled, 2 * sy, k);
          pv = min(max((sum + 64) >> 7, 0), pmax);
        }
      }
      const int v = wsrc[t] - pv * omask[t];
      const int d = v < 0 ? -((-v + 2048) >> 12) : (v + 2048) >> 12;   // ROUND_POWER_OF_TWO_SIGNED(v, 12)
      s += d;
      q += (uint32_t)(d * d);
    }
    const int64_t s64 = wsum_s(s);
    const uint64_t q64 = wsum_q((uint64_t)q);
    *sse_out = a.bit_depth == 10 ? (uint32_t)((q64 + 8) >> 4) : a.bit_depth == 12 ? (uint32_t)((q64 + 128) >> 8) : (uint32_t)q64;
    return finish_var(s64, q64, n_px, a.bit_depth);
  };
  auto est_cost = [&](int mrow, int mcol) -> uint32_t {   // estimate_obmc_mvcost: the difference x 8 (GET_MV_SUBPEL), 13-bit rounding; 0 but for ENTROPY
    if (a.cost_type != kCostEntropy) return 0u;
    const int dr = (mrow - bs.ref_row) * 8, dc = (mcol - bs.ref_col) * 8;
    return ((unsigned)mv_bits(a, dr, dc) * (unsigned)a.error_per_bit + 4096u) >> 13;
  };
  uint32_t besterr, sse1;
  int distortion, best_row = bs.start_row, best_col = bs.start_col;
  {
    uint32_t q;
    const uint32_t v = sa.upsampled ? obmc_err(best_row, best_col, 2, &q) : obmc_err(0, 0, 0, &q);
    distortion = (int)v; sse1 = q;
    besterr = v + (uint32_t)var_cost(a, bs.ref_row, bs.ref_col, best_row, best_col);
  }
  auto check = [&](int mrow, int mcol, int *has_better) -> uint32_t {   // obmc_check_better / obmc_check_better_fast
    if (mcol < bs.col_min || mcol > bs.col_max || mrow < bs.row_min || mrow > bs.row_max) return (uint32_t)INT_MAX;
    uint32_t q;
    const int thismse = (int)obmc_err(mrow, mcol, sa.upsampled ? 2 : 1, &q);
    uint32_t cost = sa.upsampled ? (uint32_t)var_cost(a, bs.ref_row, bs.ref_col, mrow, mcol) : est_cost(mrow, mcol);
    cost += (uint32_t)thismse;
    if (cost < besterr) {
      besterr = cost; best_row = mrow; best_col = mcol; distortion = thismse; sse1 = q;
      *has_better |= 1;
    }
    return cost;
  };
  int hstep = 4;   // INIT_SUBPEL_STEP_SIZE
  const int round = min(3 - sa.forced_stop, 3 - (sa.allow_hp ? 0 : 1));
  for (int iter = 0; iter < round; ++iter) {
    // The up to eight candidates of an iteration through ONE call site of check() (inlined eight times the kernel was 44 KB of code):
    // steps 0 .. 3 left, right, up, down; 4 the diagonal get_best_diag_step picks (:2490-2498); 5 .. 7 obmc_second_level_check_v2
    // (:3535-3586): row, column, and both if either improved.
    const int tr = best_row, tc = best_col;
    uint32_t left = 0, right = 0, up = 0, down = 0;
    int drow = 0, dcol = 0, br = 0, bc = 0, has_better = 0, dummy = 0;
#pragma unroll 1
    for (int step = 0; step < 8; ++step) {
      int mrow, mcol;
      if (step < 4) {
        mrow = tr + (step == 2 ? -hstep : step == 3 ? hstep : 0);
        mcol = tc + (step == 0 ? -hstep : step == 1 ? hstep : 0);
      } else if (step == 4) {
        drow = up <= down ? -hstep : hstep; dcol = left <= right ? -hstep : hstep;
        mrow = tr + drow; mcol = tc + dcol;
      } else {
        if (step == 5) {
          if (!((tr != best_row || tc != best_col) && sa.iters_per_step > 1)) break;
          if (tr == best_row) drow = -drow;
          else if (tc == best_col) dcol = -dcol;
          br = best_row; bc = best_col;
        }
        if (step == 7 && !has_better) break;
        mrow = step == 6 ? br : br + drow;
        mcol = step == 5 ? bc : bc + dcol;
      }
      const uint32_t cost = check(mrow, mcol, step >= 5 ? &has_better : &dummy);
      left = step == 0 ? cost : left; right = step == 1 ? cost : right;
      up = step == 2 ? cost : up; down = step == 3 ? cost : down;
    }
    hstep >>= 1;
  }
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)best_row; out_mv[2 * bi + 1] = (int16_t)best_col;
    out_err[bi] = besterr;
    if (out_dist) out_dist[bi] = distortion;
    if (out_sse) out_sse[bi] = sse1;
  }
}

int check_compound(aomhip_ctx *ctx, const aomhip_planes *ref, int frame, int bw, int bh, const void *blocks, int n, int cost_type, const int32_t *j,
                   const int32_t *c0, const int32_t *c1, const char *who) {
  if (!ctx || !ref || !ref->base || n < 0 || (n > 0 && !blocks) || frame < 0 || frame >= ref->n_frames || !valid_block(bw, bh) || cost_type < 0 ||
      cost_type > kCostNone) {
    set_error("%s: invalid argument", who);
    return AOMHIP_ERR_INVALID;
  }
  if (cost_type == kCostEntropy && (!j || !c0 || !c1)) {
    set_error("%s: MV_COST_ENTROPY needs the three cost tables", who);
    return AOMHIP_ERR_INVALID;
  }
  return AOMHIP_OK;
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

// the full-pel kernels of this file by block size (CompoundEval's UNITS and G)
// (EXTRA: further template arguments, with their leading comma, or nothing)
#define AOMHIP_COMMA ,
#define LAUNCH_BY_UNITS(kernel, T, EXTRA, n_px, ...)                                                    \
  do {                                                                                                  \
    if ((n_px) <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(kernel<T, 1, 4 EXTRA>), __VA_ARGS__);          \
    else if ((n_px) <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(kernel<T, 1, 2 EXTRA>), __VA_ARGS__);    \
    else if ((n_px) <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(kernel<T, 1, 1 EXTRA>), __VA_ARGS__);    \
    else if ((n_px) <= 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(kernel<T, 2, 1 EXTRA>), __VA_ARGS__);    \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(kernel<T, 4, 1 EXTRA>), __VA_ARGS__);                       \
  } while (0)

extern "C" {

int aomhip_refining_search_8p_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh, int mv_cost_type,
                                    int sad_per_bit, int error_per_bit, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                    const aomhip_search_block *d_blocks, int n_blocks, const void *d_second_pred, const uint8_t *d_mask, int invert_mask,
                                    int16_t *d_best_mv, int32_t *d_best_sad, int32_t *d_best_var) {
  int rc = check_compound(ctx, ref, frame, bw, bh, d_blocks, n_blocks, mv_cost_type, d_mvjcost, d_mvcost_row, d_mvcost_col, "aomhip_refining_search_8p_batch");
  if (rc != AOMHIP_OK) return rc;
  if (!src || !src->base || frame >= src->n_frames || (src->bit_depth == 8) != (ref->bit_depth == 8) || !d_second_pred || !d_best_mv || !d_best_sad ||
      !d_best_var) {
    set_error("aomhip_refining_search_8p_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const CompoundArgs a{ bw, bh, src->bit_depth, mv_cost_type, sad_per_bit, error_per_bit, invert_mask != 0, d_mvjcost, d_mvcost_row, d_mvcost_col };
  const dim3 grid((n_blocks + 3) / 4), block(256);
  // by block size: G = 4 / 2 candidates per wavefront for blocks of up to 64 / 128 pixels; 1 / 2 / 4 units per lane in registers for blocks of
  // 256 / 512 / 1024 pixels; beyond that the operands stream
  const int wide = bw * bh;
  if (src->bit_depth == 8) {
    LAUNCH_BY_UNITS(refining_search_8p_kernel, uint8_t, AOMHIP_COMMA true, wide, grid, block, 0, ctx->stream, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), frame,
                    d_blocks, n_blocks, a, static_cast<const uint8_t *>(d_second_pred), d_mask, d_best_mv, d_best_sad, d_best_var);
  } else {
    if (!d_mask || src->bit_depth <= 10) {   // the 16-bit blend (CompoundEval::packed)
      LAUNCH_BY_UNITS(refining_search_8p_kernel, uint16_t, AOMHIP_COMMA true, wide, grid, block, 0, ctx->stream, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), frame,
                    d_blocks, n_blocks, a, static_cast<const uint16_t *>(d_second_pred), d_mask, d_best_mv, d_best_sad, d_best_var);
    } else {
      LAUNCH_BY_UNITS(refining_search_8p_kernel, uint16_t, AOMHIP_COMMA false, wide, grid, block, 0, ctx->stream, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), frame,
                    d_blocks, n_blocks, a, static_cast<const uint16_t *>(d_second_pred), d_mask, d_best_mv, d_best_sad, d_best_var);
    }
  }
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_compound_full_pixel_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                            const aomhip_search_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                                            const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, int n_blocks, const void *d_second_pred,
                                            const uint8_t *d_mask, int invert_mask, int16_t *d_best_mv, int32_t *d_best_cost, int16_t *d_second_best_mv) {
  if (!p) {
    set_error("aomhip_compound_full_pixel_search_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  int rc = check_compound(ctx, ref, frame, bw, bh, d_blocks, n_blocks, p->mv_cost_type, d_mvjcost, d_mvcost_row, d_mvcost_col,
                          "aomhip_compound_full_pixel_search_batch");
  if (rc != AOMHIP_OK) return rc;
  if (!src || !src->base || frame >= src->n_frames || (src->bit_depth == 8) != (ref->bit_depth == 8) || !d_second_pred || !d_best_mv || !d_best_cost ||
      !d_second_best_mv || p->step_param < 0) {
    set_error("aomhip_compound_full_pixel_search_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  // the compound operand enters diamond_search_sad / full_pixel_diamond only (mcomp.c:1347, :1431): the methods that run them
  if (p->search_method != kDiamond && p->search_method != kNstep && p->search_method != kNstep8 && p->search_method != kClamped) {
    set_error("aomhip_compound_full_pixel_search_batch: search method %d is a pattern search (it ignores second_pred in the reference: use "
              "aomhip_full_pixel_search_batch)", p->search_method);
    return AOMHIP_ERR_INVALID;
  }
  if (p->use_downsampled_sad) {
    set_error("aomhip_compound_full_pixel_search_batch: use_downsampled_sad does not apply to a compound search (sdaf / msdf have no row-skipping form)");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const SiteTable *d_sites = fps_device_sites(ctx->device, p->search_method);
  if (!d_sites) {
    set_error("aomhip_compound_full_pixel_search_batch: could not place the site table on device %d", ctx->device);
    return AOMHIP_ERR_HIP;
  }
  {
    int ns, per[22], rad[22];
    int16_t mv[22][17][2];
    rc = aomhip_search_sites(p->search_method, &ns, per, rad, mv);
    if (rc != AOMHIP_OK) return rc;
    if (p->step_param > ns) {   // (== ns: the start position only, as av1_single_motion_search's search_range < 1 asks for)
      set_error("aomhip_compound_full_pixel_search_batch: step_param %d > %d search steps", p->step_param, ns);
      return AOMHIP_ERR_INVALID;
    }
  }
  const CompoundArgs a{ bw, bh, src->bit_depth, p->mv_cost_type, p->sad_per_bit, p->error_per_bit, invert_mask != 0, d_mvjcost, d_mvcost_row, d_mvcost_col };
  const dim3 grid((n_blocks + 3) / 4), block(256);
  // by block size: G = 4 / 2 candidates per wavefront for blocks of up to 64 / 128 pixels; 1 / 2 / 4 units per lane in registers for blocks of
  // 256 / 512 / 1024 pixels; beyond that the operands stream
  const int wide = bw * bh;
  if (src->bit_depth == 8) {
    LAUNCH_BY_UNITS(compound_full_pixel_diamond_kernel, uint8_t, AOMHIP_COMMA true, wide, grid, block, 0, ctx->stream, view_of<uint8_t>(*src), view_of<uint8_t>(*ref),
                    frame, d_blocks, n_blocks, a, d_sites, p->step_param, static_cast<const uint8_t *>(d_second_pred), d_mask, d_best_mv, d_best_cost,
                    d_second_best_mv);
  } else {
    if (!d_mask || src->bit_depth <= 10) {   // the 16-bit blend (CompoundEval::packed)
      LAUNCH_BY_UNITS(compound_full_pixel_diamond_kernel, uint16_t, AOMHIP_COMMA true, wide, grid, block, 0, ctx->stream, view_of<uint16_t>(*src), view_of<uint16_t>(*ref),
                    frame, d_blocks, n_blocks, a, d_sites, p->step_param, static_cast<const uint16_t *>(d_second_pred), d_mask, d_best_mv,
                    d_best_cost, d_second_best_mv);
    } else {
      LAUNCH_BY_UNITS(compound_full_pixel_diamond_kernel, uint16_t, AOMHIP_COMMA false, wide, grid, block, 0, ctx->stream, view_of<uint16_t>(*src), view_of<uint16_t>(*ref),
                    frame, d_blocks, n_blocks, a, d_sites, p->step_param, static_cast<const uint16_t *>(d_second_pred), d_mask, d_best_mv,
                    d_best_cost, d_second_best_mv);
    }
  }
  AOMHIP_LAUNCH_CHECK();
  // the follow-up of av1_full_pixel_search: needed only where a mesh search can follow (NSTEP's variance threshold, or run_mesh_search)
  const bool nstep = p->search_method == kNstep || p->search_method == kNstep8;
  if (!p->run_mesh_search && !nstep) return AOMHIP_OK;
  SearchArgs q = fps_search_args(p, d_mvjcost, d_mvcost_row, d_mvcost_col, src->bit_depth, false);
  q.resume = 1;
  return (src->bit_depth == 8 ? launch_fps_u8 : launch_fps_u16)(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, d_sites, q, /*reach=*/-1, d_best_mv,
                                                               d_best_cost, nullptr, d_second_best_mv);
}

int aomhip_obmc_full_pixel_search_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int frame, int bw, int bh, int search_method, int step_param,
                                        int fast_obmc_search, int mv_cost_type, int sad_per_bit, int error_per_bit, const int32_t *d_mvjcost,
                                        const int32_t *d_mvcost_row, const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, int n_blocks,
                                        const int32_t *d_wsrc, const int32_t *d_obmc_mask, int16_t *d_best_mv, int32_t *d_best_cost) {
  int rc = check_compound(ctx, ref, frame, bw, bh, d_blocks, n_blocks, mv_cost_type, d_mvjcost, d_mvcost_row, d_mvcost_col, "aomhip_obmc_full_pixel_search_batch");
  if (rc != AOMHIP_OK) return rc;
  if (search_method < 0 || search_method >= kMethods || step_param < 0 || !d_wsrc || !d_obmc_mask || !d_best_mv || !d_best_cost) {
    set_error("aomhip_obmc_full_pixel_search_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const SiteTable *d_sites = fps_device_sites(ctx->device, search_method);
  if (!d_sites) {
    set_error("aomhip_obmc_full_pixel_search_batch: could not place the site table on device %d", ctx->device);
    return AOMHIP_ERR_HIP;
  }
  {
    int ns = 0, per[22], rad[22];
    int16_t mv[22][17][2];
    (void)aomhip_search_sites(search_method, &ns, per, rad, mv);
    if (!fast_obmc_search && step_param > ns) {
      set_error("aomhip_obmc_full_pixel_search_batch: step_param %d > %d search steps", step_param, ns);
      return AOMHIP_ERR_INVALID;
    }
  }
  const CompoundArgs a{ bw, bh, ref->bit_depth, mv_cost_type, sad_per_bit, error_per_bit, 0, d_mvjcost, d_mvcost_row, d_mvcost_col };
  const dim3 grid((n_blocks + 3) / 4), block(256);
  // by block size: G = 4 / 2 candidates per wavefront for blocks of up to 64 / 128 pixels; 1 / 2 / 4 units per lane in registers for blocks of
  // 256 / 512 / 1024 pixels; beyond that the operands stream
  const int wide = bw * bh;
  if (ref->bit_depth == 8) {
    LAUNCH_BY_UNITS(obmc_full_pixel_search_kernel, uint8_t, , wide, grid, block, 0, ctx->stream, view_of<uint8_t>(*ref), frame, d_blocks, n_blocks, a,
                    d_sites, step_param, fast_obmc_search != 0, d_wsrc, d_obmc_mask, d_best_mv, d_best_cost);
  } else {
    LAUNCH_BY_UNITS(obmc_full_pixel_search_kernel, uint16_t, , wide, grid, block, 0, ctx->stream, view_of<uint16_t>(*ref), frame, d_blocks, n_blocks, a,
                    d_sites, step_param, fast_obmc_search != 0, d_wsrc, d_obmc_mask, d_best_mv, d_best_cost);
  }
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_obmc_subpel_tree_batch(aomhip_ctx *ctx, const aomhip_planes *ref, int frame, int bw, int bh, const aomhip_subpel_params *params,
                                  const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks,
                                  int n_blocks, const int32_t *d_wsrc, const int32_t *d_obmc_mask, int16_t *d_best_mv, uint32_t *d_best_err,
                                  int32_t *d_distortion, uint32_t *d_sse) {
  if (!params) {
    set_error("aomhip_obmc_subpel_tree_batch: null params");
    return AOMHIP_ERR_INVALID;
  }
  int rc = check_compound(ctx, ref, frame, bw, bh, d_blocks, n_blocks, params->mv_cost_type, d_mvjcost, d_mvcost_row, d_mvcost_col, "aomhip_obmc_subpel_tree_batch");
  if (rc != AOMHIP_OK) return rc;
  if ((params->subpel_search_type < 0 || params->subpel_search_type > 3) || params->forced_stop < 0 || params->forced_stop > 3 || !d_wsrc || !d_obmc_mask ||
      !d_best_mv || !d_best_err) {
    set_error("aomhip_obmc_subpel_tree_batch: invalid argument (subpel_search_type 0 .. 3)");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const CompoundArgs a{ bw, bh, ref->bit_depth, params->mv_cost_type, 0, params->error_per_bit, 0, d_mvjcost, d_mvcost_row, d_mvcost_col };
  const ObmcSubpelArgs sa{ params->iters_per_step, params->allow_hp, params->forced_stop, params->subpel_search_type };
  const dim3 grid((n_blocks + 3) / 4), block(256);
  if (ref->bit_depth == 8)
    hipLaunchKernelGGL(obmc_subpel_tree_kernel<uint8_t>, grid, block, 0, ctx->stream, view_of<uint8_t>(*ref), frame, d_blocks, n_blocks, a, sa, d_wsrc,
                       d_obmc_mask, d_best_mv, d_best_err, d_distortion, d_sse);
  else
    hipLaunchKernelGGL(obmc_subpel_tree_kernel<uint16_t>, grid, block, 0, ctx->stream, view_of<uint16_t>(*ref), frame, d_blocks, n_blocks, a, sa, d_wsrc,
                       d_obmc_mask, d_best_mv, d_best_err, d_distortion, d_sse);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

}  // extern "C"
