// libaomhip -- bilinear and up-sampled sub-pel motion search on the device (gfx950): the three trees of
// av1/encoder/mcomp.c:2844-3133 (av1_find_best_sub_pixel_tree_pruned_more / _pruned / av1_find_best_sub_pixel_tree),
// one wavefront per block, up to four candidates (16 lanes each) per step.  Split from mcomp.hip so that the two
// translation units compile in parallel.
#include <climits>

#include "common.h"
#include "search_device.h"

#ifndef AOMHIP_SUBPEL_WAVES
#define AOMHIP_SUBPEL_WAVES 5   // waves per SIMD the register allocation aims at (profiles/r01_search_variants.md)
#endif

namespace aomhip {

// ---- up-sampled prediction error: upsampled_pref_error (mcomp.c:2339-2428) with subpel_search_type USE_8_TAPS ----
// aom_[highbd_]upsampled_pred_c (av1/encoder/reconinter_enc.c:424-505,562-640) on an unscaled reference is
// aom_convolve8_horiz then aom_convolve8_vert (aom_dsp/aom_convolve.c:36-108,181-253) with the EIGHTTAP_REGULAR kernel of
// phase 2 * (mv & 7): each pass rounds by FILTER_BITS = 7 and clips to the pixel range; then vfp->vf(pred, w, src).
// Phase 0 is {0,0,0,128,0,0,0,0}, the identity, so the reference's "skip the pass when the offset is 0" needs no
// special case; taps 0 and 7 are zero in every phase, so six taps (offsets -2 .. +3) give the same sums.
__device__ constexpr int16_t kSubPel8[16][8] = {  // av1_sub_pel_filters_8 (AV1 spec; av1/common/filter.h:124-141)
  { 0, 0, 0, 128, 0, 0, 0, 0 },      { 0, 2, -6, 126, 8, -2, 0, 0 },    { 0, 2, -10, 122, 18, -4, 0, 0 },
  { 0, 2, -12, 116, 28, -8, 2, 0 },  { 0, 2, -14, 110, 38, -10, 2, 0 }, { 0, 2, -14, 102, 48, -12, 2, 0 },
  { 0, 2, -16, 94, 58, -12, 2, 0 },  { 0, 2, -14, 84, 66, -12, 2, 0 },  { 0, 2, -14, 76, 76, -14, 2, 0 },
  { 0, 2, -12, 66, 84, -14, 2, 0 },  { 0, 2, -12, 58, 94, -16, 2, 0 },  { 0, 2, -12, 48, 102, -14, 2, 0 },
  { 0, 2, -10, 38, 110, -14, 2, 0 }, { 0, 2, -8, 28, 116, -12, 2, 0 },  { 0, 0, -4, 18, 122, -10, 2, 0 },
  { 0, 0, -2, 8, 126, -6, 2, 0 }
};

template <int W, int H> struct UpTile {
  static constexpr int SH = H < 8 ? H : 8;          // output rows per strip
  static constexpr int ROWS = SH + 5;               // intermediate rows of a strip: -2 .. SH + 2
  static constexpr int ELEMS = ROWS * W;            // uint16 elements per candidate
};

// Variance of (up-sampled prediction at `ap` + (xoff, yoff)/8) against the source block, by the 16 lanes of a group
// (j = lane & 15).  `tile`: this group's UpTile<W,H>::ELEMS uint16 of LDS.  The block is processed in strips of SH rows:
// horizontal pass into LDS, then vertical pass + difference accumulation.  diff = pred - src (vf(pred, w, src, stride)).
// A lane computes 8 (4 for W = 4) adjacent pixels from one 16-pixel window load; the window starts 2 pixels left of
// the unit and is read in full, i.e. up to 5 pixels beyond the filter's right-most tap -- inside the replicated border
// for any MV the limits admit (they keep 8 pixels of margin) and inside the plane allocation in any case.
template <typename T, int W, int H>
__device__ __forceinline__ uint32_t group16_upsampled_variance(const T *ap, int astride, int xoff, int yoff, const T *bp,
                                                               int bstride, int bit_depth, int j, bool active,
                                                               uint16_t *tile, uint32_t *sse_out) {
  using U = UpTile<W, H>;
  const int pmax = sizeof(T) == 1 ? 255 : (1 << bit_depth) - 1;
  int kx[6], ky[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    kx[k] = kSubPel8[2 * (xoff & 7)][k + 1];
    ky[k] = kSubPel8[2 * (yoff & 7)][k + 1];
  }
  int32_t sum = 0;
  uint64_t sse = 0;
  constexpr int UW = W < 8 ? W : 8;                   // pixels per unit: one lane computes UW adjacent outputs
  constexpr int UPR = W / UW;                          // units per row
  constexpr int LW = sizeof(T) == 2 ? 8 : 4;           // dwords of the horizontal window load (16 pixels from column c - 2)
  for (int r0 = 0; r0 < H; r0 += U::SH) {
    if (active) {
      for (int u = j; u < U::ROWS * UPR; u += 16) {    // horizontal pass: rows r0 - 2 .. r0 + SH + 2
        const int tr = u / UPR, c = (u - tr * UPR) * UW;
        const T *p = ap + (int64_t)(r0 + tr - 2) * astride + c - 2;
        uint32_t wv[LW];
        {
          const MU128 lo = *reinterpret_cast<const MU128 *>(p);
#pragma unroll
          for (int k = 0; k < 4; ++k) wv[k] = lo.v[k];
          if constexpr (sizeof(T) == 2) {
            const MU128 hi = *reinterpret_cast<const MU128 *>(p + 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[4 + k] = hi.v[k];
          }
        }
        uint32_t outw[UW / 2];
#pragma unroll
        for (int i = 0; i < UW; ++i) {
          int acc = 64;
#pragma unroll
          for (int k = 0; k < 6; ++k) acc += px_of<T>(wv, i + k) * kx[k];
          acc >>= 7;
          acc = acc < 0 ? 0 : (acc > pmax ? pmax : acc);
          if (i & 1) outw[i >> 1] |= (uint32_t)acc << 16;
          else outw[i >> 1] = (uint32_t)acc;
        }
        uint32_t *dstw = reinterpret_cast<uint32_t *>(tile + tr * W + c);
#pragma unroll
        for (int k = 0; k < UW / 2; ++k) dstw[k] = outw[k];
      }
    }
    // the group's own LDS writes must be visible to its reads below: one wavefront, so a wave-level fence suffices
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active) {
      for (int u = j; u < U::SH * UPR; u += 16) {     // vertical pass + difference
        const int orow = u / UPR, c = (u - orow * UPR) * UW;
        uint32_t rows6[6][UW / 2];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const uint32_t *srcw = reinterpret_cast<const uint32_t *>(tile + (orow + k) * W + c);
#pragma unroll
          for (int q = 0; q < UW / 2; ++q) rows6[k][q] = srcw[q];
        }
        using BL = typename MLoad<UW * (int)sizeof(T)>::type;
        const BL bv = *reinterpret_cast<const BL *>(bp + (int64_t)(r0 + orow) * bstride + c);
        uint32_t uq = 0;
        int32_t us = 0;
#pragma unroll
        for (int i = 0; i < UW; ++i) {
          int acc = 64;
#pragma unroll
          for (int k = 0; k < 6; ++k) acc += (int)((rows6[k][i >> 1] >> (16 * (i & 1))) & 0xffffu) * ky[k];
          acc >>= 7;
          const int pv = acc < 0 ? 0 : (acc > pmax ? pmax : acc);
          const int d = pv - px_of<T>(bv.v, i);
          us += d;
          uq += (uint32_t)(d * d);
        }
        sum += us;
        sse += uq;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const int64_t tsum = (int64_t)(int32_t)row16_sum_u32((uint32_t)sum);
  const uint64_t tsse = row16_sum_u64(sse);
  int32_t sfin;
  uint32_t q;
  if (bit_depth == 10) {
    q = (uint32_t)((tsse + 8) >> 4);
    sfin = (int32_t)((tsum + 2) >> 2);
  } else if (bit_depth == 12) {
    q = (uint32_t)((tsse + 128) >> 8);
    sfin = (int32_t)((tsum + 8) >> 4);
  } else {
    q = (uint32_t)tsse;
    sfin = (int32_t)tsum;
  }
  *sse_out = q;
  constexpr int LOG2N = __builtin_ctz(W * H);
  const int64_t sq = ((int64_t)sfin * sfin) >> LOG2N;
  if (bit_depth == 8) return q - (uint32_t)sq;
  const int64_t v = (int64_t)q - sq;
  return v >= 0 ? (uint32_t)v : 0;
}

// MV_COST_ENTROPY inputs of the sub-pel search (mv_err_cost, mcomp.c:271-295): joint[4] and the two component tables
// addressed from their centres, error_per_bit
struct SubpelCostTables {
  const int *mvjcost, *mvcost0, *mvcost1;
  int error_per_bit;
  int upsampled;  // tree 2 with subpel_search_type USE_8_TAPS: errors from the up-sampled prediction
};

// GENERAL = false is the lean instantiation behind aomhip_subpel_bilinear_batch (pruned_more, no cost list, L1 / no MV
// cost): the extra arguments and branches of the general form cost it 2 % on the 4K search benchmark.
template <typename T, int W, int H, bool GENERAL>
__global__ __launch_bounds__(kSearchThreads, AOMHIP_SUBPEL_WAVES) void subpel_bilinear_kernel(
    PlaneView<T> src, PlaneView<T> ref, int frame, const aomhip_search_block *__restrict__ blocks, int n_blocks,
    int cost_type, int iters_per_step, int allow_hp, int forced_stop, int bit_depth, int tree_arg,
    const int32_t *__restrict__ cost_lists_arg, SubpelCostTables ct, int16_t *__restrict__ out_mv,
    uint32_t *__restrict__ out_err, int32_t *__restrict__ out_dist, uint32_t *__restrict__ out_sse) {
  const int tree = GENERAL ? tree_arg : 0;
  const int32_t *cost_lists = GENERAL ? cost_lists_arg : nullptr;
  const bool upsampled = GENERAL && tree == 2 && ct.upsampled;
  __shared__ uint16_t up_tiles[GENERAL ? (kSearchThreads / 16) * UpTile<W, H>::ELEMS : 1];
  uint16_t *my_tile = up_tiles + (GENERAL ? (threadIdx.x >> 4) * UpTile<W, H>::ELEMS : 0);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int bi = blockIdx.x * (kSearchThreads / 64) + wave;
  if (bi >= n_blocks) return;
  const aomhip_search_block b = blocks[bi];  // start_* in 1/8 pel, limits = SubpelMvLimits
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)b.by * src.stride + b.bx;
  const T *rbase = ref.origin + (int64_t)frame * ref.frame_stride + (int64_t)b.by * ref.stride + b.bx;
  const CostCtx cc{ cost_type, b.ref_row, b.ref_col };
  auto var_cost = [&](int mrow, int mcol) -> int {  // mv_err_cost_ (mcomp.c:271-308)
    if (GENERAL && cost_type == kCostEntropy) {
      const int dr = mrow - b.ref_row, dc = mcol - b.ref_col;
      const int64_t bits = ct.mvjcost[(dc != 0) | ((dr != 0) << 1)] + ct.mvcost0[dr] + ct.mvcost1[dc];
      return (int)((bits * ct.error_per_bit + (1 << 13)) >> 14);
    }
    return cc.var_cost(mrow, mcol);
  };

  const int grp = lane >> 4, j = lane & 15;
  uint32_t besterr, sse1;
  int distortion, best_row = b.start_row, best_col = b.start_col;
  {  // setup_center_error: vf(ref at the full-pel part, src): diff = ref - src
    const int fr = b.start_row >> 3, fc = b.start_col >> 3;
    uint32_t q;
    uint32_t v = group16_variance<T, W, H, false>(rbase + (int64_t)fr * ref.stride + fc, ref.stride, 0, 0, sp, src.stride,
                                                  /*a_minus_b=*/true, bit_depth, j, grp == 0, &q);
    v = __shfl(v, 0, 64);
    sse1 = __shfl(q, 0, 64);
    distortion = (int)v;
    besterr = v + (uint32_t)var_cost(b.start_row, b.start_col);
  }
  // check_better_fast (mcomp.c:2433-2461) for up to four candidates whose POSITIONS do not depend on each other:
  // group g evaluates candidate g (one aom_sub_pixel_varianceWxH each), then every lane replays the reference's
  // sequential `if (cost < besterr)` updates in candidate order, so the outcome is that of the scalar sequence.
  int is_better = 0;  // check_better_fast's *is_better (only second_level_check_v2 looks at it)
  auto check_n = [&](int n, const int (&mrow)[4], const int (&mcol)[4], uint32_t (&cost)[4]) {
    int my_row = mrow[0], my_col = mcol[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      my_row = grp == k ? mrow[k] : my_row;
      my_col = grp == k ? mcol[k] : my_col;
    }
    const bool inb = my_col >= b.col_min && my_col <= b.col_max && my_row >= b.row_min && my_row <= b.row_max;
    uint32_t q;
    uint32_t v;
    if (GENERAL && upsampled)
      v = group16_upsampled_variance<T, W, H>(rbase + (int64_t)(my_row >> 3) * ref.stride + (my_col >> 3), ref.stride,
                                              my_col & 7, my_row & 7, sp, src.stride, bit_depth, j, inb && grp < n, my_tile, &q);
    else
      v = group16_variance<T, W, H, true>(rbase + (int64_t)(my_row >> 3) * ref.stride + (my_col >> 3), ref.stride, my_col & 7,
                                          my_row & 7, sp, src.stride, true, bit_depth, j, inb && grp < n, &q);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      cost[k] = (uint32_t)INT_MAX;
      if (k < n) {
        const uint32_t vk = __shfl(v, 16 * k, 64), qk = __shfl(q, 16 * k, 64);
        const bool in_k = mcol[k] >= b.col_min && mcol[k] <= b.col_max && mrow[k] >= b.row_min && mrow[k] <= b.row_max;
        if (in_k) {
          const int thismse = (int)vk;
          cost[k] = (uint32_t)var_cost(mrow[k], mcol[k]) + (uint32_t)thismse;
          if (cost[k] < besterr) {
            besterr = cost[k];
            best_row = mrow[k];
            best_col = mcol[k];
            distortion = thismse;
            sse1 = qk;
            is_better = 1;
          }
        }
      }
    }
  };
  auto first_level = [&](int trow, int tcol, int hstep, int *odrow, int *odcol) {  // first_level_check_fast (:2503-2543)
    uint32_t c[4];
    {
      const int r4[4] = { trow, trow, trow - hstep, trow + hstep }, c4[4] = { tcol - hstep, tcol + hstep, tcol, tcol };
      check_n(4, r4, c4, c);  // left, right, up, down
    }
    const uint32_t left = c[0], right = c[1], up = c[2], down = c[3];
    const int drow = up <= down ? -hstep : hstep, dcol = left <= right ? -hstep : hstep;
    {
      const int r1[4] = { trow + drow, 0, 0, 0 }, c1[4] = { tcol + dcol, 0, 0, 0 };
      check_n(1, r1, c1, c);
    }
    *odrow = drow;
    *odcol = dcol;
  };
  auto second_level_v2 = [&](int trow, int tcol, int drow, int dcol) {  // second_level_check_v2 (:2665-2716), bilinear branch
    if (trow == best_row && tcol == best_col) return;
    if (trow == best_row) drow = -drow;
    else if (tcol == best_col) dcol = -dcol;
    const int br = best_row, bc = best_col;
    uint32_t c[4];
    is_better = 0;
    // row_bias then col_bias: the second position does not depend on the first outcome
    const int r2[4] = { br + drow, br, 0, 0 }, c2[4] = { bc, bc + dcol, 0, 0 };
    check_n(2, r2, c2, c);
    if (is_better) {
      const int r1[4] = { br + drow, 0, 0, 0 }, c1[4] = { bc + dcol, 0, 0, 0 };
      check_n(1, r1, c1, c);
    }
  };
  auto two_level = [&](int trow, int tcol, int hstep) {  // two_level_checks_fast (mcomp.c:2503-2624)
    uint32_t c[4];
    int drow, dcol;
    first_level(trow, tcol, hstep, &drow, &dcol);
    if (iters_per_step <= 1) return;
    const int br = best_row, bc = best_col;
    if (trow != br && tcol != bc) {
      const int r2[4] = { br, br + drow, 0, 0 }, c2[4] = { bc + dcol, bc, 0, 0 };
      check_n(2, r2, c2, c);
    } else if (trow == br && tcol != bc) {
      const int r3[4] = { br + hstep, br - hstep, br - drow, 0 }, c3[4] = { bc + dcol, bc + dcol, bc, 0 };
      check_n(3, r3, c3, c);
    } else if (trow != br && tcol == bc) {
      const int r3[4] = { br + drow, br + drow, br, 0 }, c3[4] = { bc + hstep, bc - hstep, bc - dcol, 0 };
      check_n(3, r3, c3, c);
    }
  };
  int hstep = 4;          // INIT_SUBPEL_STEP_SIZE
  if (tree == 2) {         // av1_find_best_sub_pixel_tree (:3069-3133)
    const int round = min(3 - forced_stop, 3 - (allow_hp ? 0 : 1));
    for (int iter = 0; iter < round; ++iter) {
      const int cr = best_row, ccol = best_col;
      int drow, dcol;
      first_level(cr, ccol, hstep, &drow, &dcol);
      if (!(cr == best_row && ccol == best_col) && iters_per_step > 1) second_level_v2(cr, ccol, drow, dcol);
      hstep >>= 1;
    }
  } else if (forced_stop != 3) {  // FULL_PEL
    // first iteration: a usable cost list replaces the two-level check (pruned_more: the minimum of the fitted cost
    // surface, :2879-2893; pruned: the quadrant the cheaper neighbours point at, :2968-3043)
    int c0 = INT_MAX, c1 = INT_MAX, c2 = INT_MAX, c3 = INT_MAX, c4 = INT_MAX;
    if (cost_lists) {
      c0 = cost_lists[5 * bi]; c1 = cost_lists[5 * bi + 1]; c2 = cost_lists[5 * bi + 2];
      c3 = cost_lists[5 * bi + 3]; c4 = cost_lists[5 * bi + 4];
    }
    const bool usable = c0 != INT_MAX && c1 != INT_MAX && c2 != INT_MAX && c3 != INT_MAX && c4 != INT_MAX;
    uint32_t cst[4];
    if (tree == 0 && usable && c0 < c1 && c0 < c2 && c0 < c3 && c0 < c4) {
      auto div_round = [](int n, int d) { return ((n < 0) ^ (d < 0)) ? ((n - d / 2) / d) : ((n + d / 2) / d); };
      const int ic = div_round(c1 - c3, c1 - 2 * c0 + c3), ir = div_round(c4 - c2, c4 - 2 * c0 + c2);  // get_cost_surf_min, bits = 1
      if (ir != 0 || ic != 0) {
        const int r1[4] = { b.start_row + ir * hstep, 0, 0, 0 }, q1[4] = { b.start_col + ic * hstep, 0, 0, 0 };
        check_n(1, r1, q1, cst);
      }
    } else if (tree == 1 && usable) {
      const int dc = (c1 < c3) ? -hstep : hstep, dr = (c2 < c4) ? hstep : -hstep;  // left : right, bottom : top
      const int r3[4] = { b.start_row, b.start_row + dr, b.start_row + dr, 0 };
      const int q3[4] = { b.start_col + dc, b.start_col, b.start_col + dc, 0 };
      check_n(3, r3, q3, cst);
    } else {
      two_level(b.start_row, b.start_col, hstep);
    }
    if (forced_stop < 2) {  // < HALF_PEL
      hstep >>= 1;
      two_level(best_row, best_col, hstep);
    }
    if (allow_hp && forced_stop == 0) {  // EIGHTH_PEL
      hstep >>= 1;
      two_level(best_row, best_col, hstep);
    }
  }
  if (lane == 0) {
    out_mv[2 * bi] = (int16_t)best_row;
    out_mv[2 * bi + 1] = (int16_t)best_col;
    out_err[bi] = besterr;
    out_dist[bi] = distortion;
    out_sse[bi] = sse1;
  }
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

static int launch_subpel(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                         int mv_cost_type, int iters_per_step, int allow_hp, int forced_stop, int tree,
                         const int32_t *d_cost_lists, SubpelCostTables ct, const aomhip_search_block *d_blocks, int n_blocks,
                         int16_t *d_best_mv, uint32_t *d_best_err, int32_t *d_distortion, uint32_t *d_sse) {
  if (n_blocks == 0) return AOMHIP_OK;
  const dim3 grid((n_blocks + 3) / 4), block(kSearchThreads);
  const bool general = tree != 0 || d_cost_lists != nullptr || mv_cost_type == kCostEntropy || ct.upsampled;
#define X(W, H)                                                                                                      \
  if (bw == W && bh == H) {                                                                                          \
    if (src->bit_depth == 8)                                                                                         \
      hipLaunchKernelGGL((general ? subpel_bilinear_kernel<uint8_t, W, H, true> : subpel_bilinear_kernel<uint8_t, W, H, false>), grid, block, 0, ctx->stream, view_of<uint8_t>(*src), \
                         view_of<uint8_t>(*ref), frame, d_blocks, n_blocks, mv_cost_type, iters_per_step, allow_hp,  \
                         forced_stop, 8, tree, d_cost_lists, ct, d_best_mv, d_best_err, d_distortion, d_sse);        \
    else                                                                                                             \
      hipLaunchKernelGGL((general ? subpel_bilinear_kernel<uint16_t, W, H, true> : subpel_bilinear_kernel<uint16_t, W, H, false>), grid, block, 0, ctx->stream, \
                         view_of<uint16_t>(*src), view_of<uint16_t>(*ref), frame, d_blocks, n_blocks, mv_cost_type,  \
                         iters_per_step, allow_hp, forced_stop, src->bit_depth, tree, d_cost_lists, ct, d_best_mv,   \
                         d_best_err, d_distortion, d_sse);                                                           \
    AOMHIP_LAUNCH_CHECK();                                                                                           \
    return AOMHIP_OK;                                                                                                \
  }
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  return AOMHIP_ERR_INVALID;
}

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
                       SubpelCostTables{ nullptr, nullptr, nullptr, 0, 0 }, d_blocks, n_blocks, d_best_mv, d_best_err,
                       d_distortion, d_sse);
}

int aomhip_subpel_tree_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                             const aomhip_subpel_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                             const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, const int32_t *d_cost_list,
                             int n_blocks, int16_t *d_best_mv, uint32_t *d_best_err, int32_t *d_distortion, uint32_t *d_sse) {
  if (!p) {
    set_error("aomhip_subpel_tree_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const bool entropy = p->mv_cost_type == kCostEntropy;
  int rc = check_common(ctx, src, ref, frame, bw, bh, d_blocks, n_blocks, entropy ? kCostNone : p->mv_cost_type);
  if (rc != AOMHIP_OK) return rc;
  if (!d_best_mv || !d_best_err || !d_distortion || !d_sse || p->forced_stop < 0 || p->forced_stop > 3 || p->tree < 0 ||
      p->tree > 2 || (p->subpel_search_type != 0 && p->subpel_search_type != 3) ||
      (entropy && (!d_mvjcost || !d_mvcost_row || !d_mvcost_col))) {
    set_error("aomhip_subpel_tree_batch: invalid argument (tree 0..2, forced_stop 0..3, MV_COST_ENTROPY needs its tables)");
    return AOMHIP_ERR_INVALID;
  }
  return launch_subpel(ctx, src, ref, frame, bw, bh, p->mv_cost_type, p->iters_per_step, p->allow_hp, p->forced_stop, p->tree,
                       d_cost_list, SubpelCostTables{ d_mvjcost, d_mvcost_row, d_mvcost_col, p->error_per_bit, p->tree == 2 && p->subpel_search_type == 3 }, d_blocks,
                       n_blocks, d_best_mv, d_best_err, d_distortion, d_sse);
}

}  // extern "C"
