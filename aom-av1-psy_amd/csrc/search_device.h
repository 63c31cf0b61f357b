// Device helpers shared by the motion-search kernels (mcomp.hip, fullpel_search.hip): 8-lanes-per-candidate SAD,
// 16-lanes-per-candidate (sub-pel) variance, the L1 / NONE MV cost forms.  Not part of the ABI.
#ifndef AOMHIP_CSRC_SEARCH_DEVICE_H_
#define AOMHIP_CSRC_SEARCH_DEVICE_H_

#include <climits>

#include "common.h"

namespace aomhip {

struct __attribute__((packed, aligned(1))) MU128 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) MU64 { uint32_t v[2]; };
struct __attribute__((packed, aligned(1))) MU32 { uint32_t v[1]; };
template <int BYTES> struct MLoad;
template <> struct MLoad<16> { using type = MU128; };
template <> struct MLoad<8> { using type = MU64; };
template <> struct MLoad<4> { using type = MU32; };

enum { kCostEntropy = 0, kCostL1Low = 1, kCostL1Mid = 2, kCostL1Hd = 3, kCostNone = 4 };  // MV_COST_TYPE (mcomp.h:40-50)

__device__ __forceinline__ int iabsm(int v) { return v < 0 ? -v : v; }

struct CostCtx {
  int cost_type, ref_row, ref_col;  // ref_mv in 1/8 pel
  __device__ __forceinline__ int sad_cost(int row, int col) const {  // mvsad_err_cost_ (mcomp.c:310-339)
    const int frr = (ref_row + 3 + (ref_row >= 0)) >> 3, frc = (ref_col + 3 + (ref_col >= 0)) >> 3;  // GET_MV_RAWPEL
    const int d = iabsm((row - frr) * 8) + iabsm((col - frc) * 8);
    const int lambda = cost_type == kCostL1Low ? 32 : cost_type == kCostL1Mid ? 15 : cost_type == kCostL1Hd ? 8 : 0;
    return (lambda * d) >> 3;
  }
  __device__ __forceinline__ int var_cost(int mrow, int mcol) const {  // mv_err_cost_ (mcomp.c:271-308)
    const int d = iabsm(mrow - ref_row) + iabsm(mcol - ref_col);
    const int lambda = cost_type == kCostL1Low ? 2 : cost_type == kCostL1Mid ? 0 : cost_type == kCostL1Hd ? 1 : 0;
    return (lambda * d) >> 3;
  }
};

#ifndef AOMHIP_SEARCH_ALIGNED_LOADS
#define AOMHIP_SEARCH_ALIGNED_LOADS 0
#endif
struct __attribute__((aligned(4))) AU128 { uint32_t v[4]; };
struct __attribute__((aligned(4))) AU64 { uint32_t v[2]; };
// One row unit from an arbitrarily aligned address through DWORD-ALIGNED loads of the enclosing dwords + v_alignbit
// (a 16-byte global load from an address that is not dword-aligned runs at half rate, profiles/r01_inter_pred.md).
template <typename L> __device__ __forceinline__ L load_unit(const void *p) {
#if AOMHIP_SEARCH_ALIGNED_LOADS
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uintptr_t a4 = a & ~(uintptr_t)3;
  const uint32_t sh = (uint32_t)(a & 3) * 8;
  L out;
  if constexpr (sizeof(L) == 16) {
    const AU128 lo = *reinterpret_cast<const AU128 *>(a4);
    const uint32_t hi = *reinterpret_cast<const uint32_t *>(a4 + 16);
#pragma unroll
    for (int i = 0; i < 3; ++i) out.v[i] = __builtin_amdgcn_alignbit(lo.v[i + 1], lo.v[i], sh);
    out.v[3] = __builtin_amdgcn_alignbit(hi, lo.v[3], sh);
  } else if constexpr (sizeof(L) == 8) {
    const AU64 lo = *reinterpret_cast<const AU64 *>(a4);
    const uint32_t hi = *reinterpret_cast<const uint32_t *>(a4 + 8);
    out.v[0] = __builtin_amdgcn_alignbit(lo.v[1], lo.v[0], sh);
    out.v[1] = __builtin_amdgcn_alignbit(hi, lo.v[1], sh);
  } else {
    const AU64 lo = *reinterpret_cast<const AU64 *>(a4);
    out.v[0] = __builtin_amdgcn_alignbit(lo.v[1], lo.v[0], sh);
  }
  return out;
#else
  return *reinterpret_cast<const L *>(p);
#endif
}

template <typename T> __device__ __forceinline__ uint32_t sadw(uint32_t a, uint32_t b, uint32_t acc) {
  if constexpr (sizeof(T) == 1) return __builtin_amdgcn_sad_u8(a, b, acc);
  else return __builtin_amdgcn_sad_u16(a, b, acc);
}

// Geometry of the 8-lanes-per-site SAD: row units of <= 16 bytes, lane l of a group owns units l, l + 8, ...
#ifndef AOMHIP_G8_KEEP_MAX
#define AOMHIP_G8_KEEP_MAX 4
#endif
template <typename T, int W, int H> struct G8 {
  static constexpr int RB = W * (int)sizeof(T);
  static constexpr int UB = RB < 16 ? RB : 16;
  static constexpr int UE = UB / (int)sizeof(T);
  static constexpr int UPR = RB / UB;
  static constexpr int U = UPR * H;
  static constexpr int PER_LANE = (U + 7) / 8;
  // The lane's source units stay in registers across all the sites of a search when they are few: up to 4 (16 VGPRs).  With up to 8
  // (32 VGPRs: 32x32 / 64x16 / 16x64 on 8-bit planes, 32x16 / 16x32 on 16-bit ones) full_pixel_search_kernel went to 157 VGPRs + 560
  // bytes of scratch and the temporal filter's 8-bit search ran 11.1 instead of 8.4 ms per 4K frame.
  static constexpr bool KEEP = PER_LANE <= AOMHIP_G8_KEEP_MAX;
  using L = typename MLoad<UB>::type;
};

template <typename T, int W, int H>
__device__ __forceinline__ void group8_load_src(const T *sp, int sstride, int l, typename G8<T, W, H>::L (&s)[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1]) {
  using G = G8<T, W, H>;
  if constexpr (G::KEEP) {
#pragma unroll
    for (int k = 0; k < G::PER_LANE; ++k) {
      const int u = min(l + 8 * k, G::U - 1);
      const int row = u / G::UPR, col = (u % G::UPR) * G::UE;
      s[k] = *reinterpret_cast<const typename G::L *>(sp + (int64_t)row * sstride + col);
    }
  }
}

// SAD of the W x H block at sp vs rp, computed by the 8 lanes of a group (l = lane & 7); all 8 get the sum.
template <typename T, int W, int H>
__device__ __forceinline__ uint32_t group8_sad(const T *sp, int sstride, const T *rp, int rstride, int l, bool active,
                                               const typename G8<T, W, H>::L (&s)[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1]) {
  using G = G8<T, W, H>;
  using L = typename G::L;
  uint32_t acc = 0;
  if (active) {
    if constexpr (G::KEEP) {
#pragma unroll
      for (int k = 0; k < G::PER_LANE; ++k) {
        const int u = l + 8 * k;
        if (u < G::U) {
          const int row = u / G::UPR, col = (u % G::UPR) * G::UE;
          const L b = load_unit<L>(rp + (int64_t)row * rstride + col);
#pragma unroll
          for (int i = 0; i < G::UB / 4; ++i) acc = sadw<T>(s[k].v[i], b.v[i], acc);
        }
      }
    } else {
      // four units at a time, their eight reads requested before the first SAD (one read per iteration made a candidate of a large block a chain of
      // memory round trips: profiles/r06_fps_nstep.md, the 32x32 search)
      constexpr int kN = (G::U + 7) / 8, kBatch = kN >= 4 ? 4 : kN;
#pragma unroll 1
      for (int k0 = 0; k0 < kN; k0 += kBatch) {
        L a[kBatch], b[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          const int u = min(l + 8 * (k0 + j), G::U - 1);
          const int row = u / G::UPR, col = (u % G::UPR) * G::UE;
          a[j] = *reinterpret_cast<const L *>(sp + (int64_t)row * sstride + col);
          b[j] = *reinterpret_cast<const L *>(rp + (int64_t)row * rstride + col);
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
          if (k0 + j < kN && l + 8 * (k0 + j) < G::U) {
#pragma unroll
            for (int i = 0; i < G::UB / 4; ++i) acc = sadw<T>(a[j].v[i], b[j].v[i], acc);
          }
        }
      }
    }
  }
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0xB1, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x4E, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x141, 0xf, 0xf, false);
  return acc;
}

// The same SAD with the candidate addressed as a UNIFORM base (SGPR pair: the wavefront's block) + a 32-bit byte offset per lane: the
// loads take the `global_load v, v_off, s[base]` form and the per-unit address is one v_add_u32 -- the pointer form above costs three
// 64-bit VALU operations per load (v_mad_i64_i32 + v_lshl_add_u64), a fifth of a search step's vector instructions.
// site_off = ((row - row0) * rstride + (col - col0)) * sizeof(T) >= 0 relative to `base` (the caller anchors base at the block's
// smallest legal MV); unit_off[k] = this lane's k-th unit inside a block (group8_unit_offsets).
template <typename T, int W, int H>
__device__ __forceinline__ void group8_unit_offsets(int rstride, int l, uint32_t (&uo)[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1]) {
  using G = G8<T, W, H>;
  if constexpr (G::KEEP) {
#pragma unroll
    for (int k = 0; k < G::PER_LANE; ++k) {
      const int u = min(l + 8 * k, G::U - 1);
      uo[k] = (uint32_t)(((u / G::UPR) * rstride + (u % G::UPR) * G::UE) * (int)sizeof(T));
    }
  }
}
template <typename T, int W, int H>
__device__ __forceinline__ uint32_t group8_sad_u(const char *base, uint32_t site_off, int l, bool active,
                                                 const uint32_t (&uo)[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1],
                                                 const typename G8<T, W, H>::L (&s)[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1]) {
  using G = G8<T, W, H>;
  using L = typename G::L;
  static_assert(G::KEEP, "uniform-base path: blocks whose source units stay in registers");
  uint32_t acc = 0;
  if (active) {
#pragma unroll
    for (int k = 0; k < G::PER_LANE; ++k) {
      if (l + 8 * k < G::U) {
        const L b = *reinterpret_cast<const L *>(base + (size_t)(uint32_t)(site_off + uo[k]));
#pragma unroll
        for (int i = 0; i < G::UB / 4; ++i) acc = sadw<T>(s[k].v[i], b.v[i], acc);
      }
    }
  }
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0xB1, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x4E, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x141, 0xf, 0xf, false);
  return acc;
}

// The same SAD with the reference block taken from an LDS window (byte offset `off` of the block's top-left pixel,
// row pitch `pitch` bytes, both multiples of sizeof(T)): aligned dword reads + v_alignbyte, because a misaligned
// ds_read_b128 runs at 1/12 of the aligned rate (tools/lds_unaligned_probe.hip).  Source units from registers.
template <typename T, int W, int H>
__device__ __forceinline__ uint32_t group8_sad_lds(const uint32_t *win, unsigned off, int pitch, int l, bool active,
                                                   const typename G8<T, W, H>::L (&s)[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1]) {
  using G = G8<T, W, H>;
  static_assert(G::KEEP, "LDS path is only instantiated for blocks whose source units stay in registers");
  uint32_t acc = 0;
  if (active) {
    // every read of the site first, then the arithmetic: one LDS round trip per site instead of one per unit
    uint32_t d[G::PER_LANE][G::UB / 4 + 1];
    unsigned sh[G::PER_LANE];
#pragma unroll
    for (int k = 0; k < G::PER_LANE; ++k) {
      const int u = min(l + 8 * k, G::U - 1);
      const int row = u / G::UPR, colb = (u % G::UPR) * G::UB;
      const unsigned o = off + (unsigned)(row * pitch + colb);
      const uint32_t *p = win + (o >> 2);
      sh[k] = o & 3;
#pragma unroll
      for (int i = 0; i <= G::UB / 4; ++i) d[k][i] = p[i];
    }
#pragma unroll
    for (int k = 0; k < G::PER_LANE; ++k) {
      if (l + 8 * k < G::U) {
#pragma unroll
        for (int i = 0; i < G::UB / 4; ++i) acc = sadw<T>(s[k].v[i], __builtin_amdgcn_alignbyte(d[k][i + 1], d[k][i], sh[k]), acc);
      }
    }
  }
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0xB1, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x4E, 0xf, 0xf, false);
  acc += __builtin_amdgcn_update_dpp(0u, acc, 0x141, 0xf, 0xf, false);
  return acc;
}

__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor((long long)v, m, 64);
  return v;
}

// Variance (SUBPEL = false) or bilinear sub-pixel variance of a W x H block by all 64 lanes.  a = "ref" operand
// (interpolated when SUBPEL), b = "src" operand; diff = A_MINUS_B ? a - b : b - a.  Returns var, *sse.
template <typename T, int W, int H, bool SUBPEL>
__device__ __forceinline__ uint32_t wave_variance(const T *ap, int astride, int xoff, int yoff, const T *bp, int bstride,
                                                  bool a_minus_b, int bit_depth, int lane, uint32_t *sse_out) {
  constexpr int UE = 4;  // pixels per unit
  constexpr int UPR = W / UE;
  constexpr int U = UPR * H;
  constexpr uint8_t kBil[8][2] = { { 128, 0 }, { 112, 16 }, { 96, 32 }, { 80, 48 },
                                   { 64, 64 }, { 48, 80 },  { 32, 96 }, { 16, 112 } };  // aom_filter.h:43-50
  static_assert(kBil[3][0] == 128 - 16 * 3 && kBil[3][1] == 16 * 3 && kBil[7][0] == 16, "bilinear taps are 128 - 16 i, 16 i");
  const int fx1 = (xoff & 7) << 4, fx0 = 128 - fx1, fy1 = (yoff & 7) << 4, fy0 = 128 - fy1;   // = kBil[off & 7]: {128 - 16 i, 16 i} (aom_filter.h:43-50) by arithmetic -- indexed as a table the compiler put it in memory: two loads on the chain of every candidate
  int64_t sum = 0, sse = 0;
  for (int u = lane; u < U; u += 64) {
    const int row = u / UPR, col = (u % UPR) * UE;
    const T *a0 = ap + (int64_t)row * astride + col;
    const T *b0 = bp + (int64_t)row * bstride + col;
    int us = 0;
    uint32_t uq = 0;
#pragma unroll
    for (int i = 0; i < UE; ++i) {
      int av;
      if constexpr (SUBPEL) {
        const int h0 = ((int)a0[i] * fx0 + (int)a0[i + 1] * fx1 + 64) >> 7;
        const int h1 = ((int)a0[astride + i] * fx0 + (int)a0[astride + i + 1] * fx1 + 64) >> 7;
        av = (__mul24(h0, fy0) + __mul24(h1, fy1) + 64) >> 7;   // h <= 4095: 24-bit multiplies (v_mul_lo_u32 is quarter rate)
        av &= (sizeof(T) == 1) ? 0xFF : 0xFFFF;
      } else {
        av = a0[i];
      }
      const int d = a_minus_b ? av - (int)b0[i] : (int)b0[i] - av;
      us += d;
      uq += (uint32_t)__mul24(d, d);   // |d| < 2^12: the 24-bit multiplier is exact and full rate (v_mul_lo_u32 is quarter rate)
    }
    sum += us;
    sse += uq;
  }
  sum = wave_sum64(sum);
  sse = wave_sum64(sse);
  // variance.c:141-148 / :383-420
  int32_t s;
  uint32_t q;
  if (bit_depth == 10) {
    q = (uint32_t)(((uint64_t)sse + 8) >> 4);
    s = (int32_t)((sum + 2) >> 2);
  } else if (bit_depth == 12) {
    q = (uint32_t)(((uint64_t)sse + 128) >> 8);
    s = (int32_t)((sum + 8) >> 4);
  } else {
    q = (uint32_t)sse;
    s = (int32_t)sum;
  }
  *sse_out = q;
  constexpr int LOG2N = __builtin_ctz(W * H);
  const int64_t sq = ((int64_t)s * s) >> LOG2N;
  if (bit_depth == 8) return q - (uint32_t)sq;
  const int64_t v = (int64_t)q - sq;
  return v >= 0 ? (uint32_t)v : 0;
}

// ---- 16 lanes per candidate: up to four independent candidates of one block are evaluated side by side -------------
// Variance (SUBPEL = false) or bilinear sub-pixel variance of the W x H block at ap (per-GROUP pointer and phase) vs
// the source block bp, by the 16 lanes of a group (j = lane & 15); every lane of the group returns the result.
// Pixel units of 8 (4 for W = 4) per lane; the two Σ are reduced inside the DPP row (= the group) -- no LDS, no
// cross-row traffic.  Semantics identical to wave_variance (aom_dsp/variance.c:56-73,91-163,342-561).
template <typename T> __device__ __forceinline__ int px_of(const uint32_t *v, int i) {
  if constexpr (sizeof(T) == 2) return (int)((v[i >> 1] >> (16 * (i & 1))) & 0xffffu);
  else return (int)((v[i >> 2] >> (8 * (i & 3))) & 0xffu);
}
__device__ __forceinline__ uint32_t row16_sum_u32(uint32_t v) {
  v += __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xf, 0xf, false);
  return v;
}
// The minimum of a key over the wavefront when the 8 lanes of each group already hold their group's key: the row's other group (row_ror:8),
// then row_bcast15 into rows 1 and 3 and row_bcast31 into rows 2 and 3, the result read from lane 63 -- three v_min_u32 with the DPP operand
// and one v_readlane instead of one DPP step, four v_readlane and three s_min (through update_dpp the compiler emits v_mov + v_mov_dpp +
// v_min per step).  s_nop 1: the two wait states between a VALU write and a DPP read of the same register.
__device__ __forceinline__ uint32_t groups8_min_u32(uint32_t key) {
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
               : "+v"(key));
  return (uint32_t)__builtin_amdgcn_readlane((int)key, 63);
}
__device__ __forceinline__ uint64_t row16_sum_u64(uint64_t v) {
#define AOMHIP_STEP64(CTRL)                                                                      \
  {                                                                                              \
    const uint32_t lo = __builtin_amdgcn_update_dpp(0u, (uint32_t)v, CTRL, 0xf, 0xf, false);     \
    const uint32_t hi = __builtin_amdgcn_update_dpp(0u, (uint32_t)(v >> 32), CTRL, 0xf, 0xf, false); \
    v += ((uint64_t)hi << 32) | lo;                                                              \
  }
  AOMHIP_STEP64(0xB1) AOMHIP_STEP64(0x4E) AOMHIP_STEP64(0x141) AOMHIP_STEP64(0x140)
#undef AOMHIP_STEP64
  return v;
}

// The other reference's predictor of a compound search (ms_buffers.second_pred / mask / inv_mask): `second` W x H pixels, `mask` W x H blend
// weights 0..64 or null.  blend(): aom_comp_avg_pred (variance.c:306-319) / aom_comp_mask_pred (:773-791, AOM_BLEND_A64) on one pixel, `f` the
// (filtered) pixel of the searched reference.
template <typename T> struct CompoundRef {
  const T *second;
  const uint8_t *mask;
  int invert;
  __device__ __forceinline__ int blend(int f, int idx) const {
    const int p = (int)second[idx];
    if (!mask) return (p + f + 1) >> 1;
    const int m = mask[idx];
    return invert ? (m * p + (64 - m) * f + 32) >> 6 : (m * f + (64 - m) * p + 32) >> 6;
  }
  // N (4 or 8) adjacent pixels of the predictor starting at idx (a multiple of N): one vector load of the predictor and one of the weights
  // instead of 2 N element loads
  template <int N> __device__ __forceinline__ void blend_run(int *f, int idx) const {
    int p[N], m[N];
    if constexpr (sizeof(T) == 2) {
      if constexpr (N == 8) {
        const uint4 w = *reinterpret_cast<const uint4 *>(second + idx);
        const uint32_t w4[4] = { w.x, w.y, w.z, w.w };
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[2 * i] = (int)(w4[i] & 0xffffu); p[2 * i + 1] = (int)(w4[i] >> 16); }
      } else {
        const uint2 w = *reinterpret_cast<const uint2 *>(second + idx);
        p[0] = (int)(w.x & 0xffffu); p[1] = (int)(w.x >> 16); p[2] = (int)(w.y & 0xffffu); p[3] = (int)(w.y >> 16);
      }
    } else {
#pragma unroll
      for (int h = 0; h < N / 4; ++h) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(second + idx + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) p[4 * h + i] = (int)((w >> (8 * i)) & 0xffu);
      }
    }
    if (!mask) {
#pragma unroll
      for (int i = 0; i < N; ++i) f[i] = (p[i] + f[i] + 1) >> 1;
      return;
    }
#pragma unroll
    for (int h = 0; h < N / 4; ++h) {
      const uint32_t w = *reinterpret_cast<const uint32_t *>(mask + idx + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) m[4 * h + i] = (int)((w >> (8 * i)) & 0xffu);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) f[i] = invert ? (m[i] * p[i] + (64 - m[i]) * f[i] + 32) >> 6 : (m[i] * f[i] + (64 - m[i]) * p[i] + 32) >> 6;
  }
  // The same on N pixels held as N / 2 packed 16-bit pairs: (A * f + C) >> sh with A = 1, C = p + 1, sh = 1 without a mask and A = m (inverted:
  // 64 - m), C = (64 - A) * p + 32, sh = 6 with one -- v_pk_* on pixel pairs, 3 / 6 instructions per pair instead of 6 / 14.  A * f + C fits 16
  // bits for every depth without a mask and up to 10 bits with one; 12-bit masked pixels go through blend_run.
  template <int N> __device__ __forceinline__ void blend_pairs(uint32_t *f2, int idx, int bit_depth) const {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    if (mask && bit_depth > 10) {
      int f[N];
#pragma unroll
      for (int i = 0; i < N / 2; ++i) { f[2 * i] = (int)(f2[i] & 0xffffu); f[2 * i + 1] = (int)(f2[i] >> 16); }
      blend_run<N>(f, idx);
#pragma unroll
      for (int i = 0; i < N / 2; ++i) f2[i] = (uint32_t)f[2 * i] | ((uint32_t)f[2 * i + 1] << 16);
      return;
    }
    uint32_t p2[N / 2], m2[N / 2];
    if constexpr (sizeof(T) == 2) {
      if constexpr (N == 8) {
        const uint4 w = *reinterpret_cast<const uint4 *>(second + idx);
        p2[0] = w.x; p2[1] = w.y; p2[2] = w.z; p2[3] = w.w;
      } else {
        const uint2 w = *reinterpret_cast<const uint2 *>(second + idx);
        p2[0] = w.x; p2[1] = w.y;
      }
    } else {
#pragma unroll
      for (int h = 0; h < N / 4; ++h) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(second + idx + 4 * h);
        p2[2 * h] = __builtin_amdgcn_perm(0, w, 0x0c010c00);
        p2[2 * h + 1] = __builtin_amdgcn_perm(0, w, 0x0c030c02);
      }
    }
    const u16x2 one = { 1, 1 }, c64 = { 64, 64 }, c32 = { 32, 32 }, s6 = { 6, 6 };
    if (!mask) {
#pragma unroll
      for (int i = 0; i < N / 2; ++i)
        f2[i] = __builtin_bit_cast(uint32_t, (u16x2)((__builtin_bit_cast(u16x2, f2[i]) + __builtin_bit_cast(u16x2, p2[i]) + one) >> one));
      return;
    }
#pragma unroll
    for (int h = 0; h < N / 4; ++h) {
      const uint32_t w = *reinterpret_cast<const uint32_t *>(mask + idx + 4 * h);
      m2[2 * h] = __builtin_amdgcn_perm(0, w, 0x0c010c00);
      m2[2 * h + 1] = __builtin_amdgcn_perm(0, w, 0x0c030c02);
    }
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const u16x2 m = __builtin_bit_cast(u16x2, m2[i]), A = invert ? c64 - m : m;
      const u16x2 Cc = (c64 - A) * __builtin_bit_cast(u16x2, p2[i]) + c32;
      f2[i] = __builtin_bit_cast(uint32_t, (u16x2)((A * __builtin_bit_cast(u16x2, f2[i]) + Cc) >> s6));
    }
  }
};

// How many 16-lane groups share one candidate in a round of n candidates (general sub-pel instantiation): the block's rows are cut into
// that many parts (>= 4 rows each), the parts' sums added at the end (parts_sum).
// (a lone candidate over FOUR groups: at 16 x 16 the horizontal pass of a 4-row part is 9 rows x 2 units = 18 units, two passes of the 16 lanes
// like the 26 units of an 8-row part -- the same latency for twice the instructions; two groups is the default)
#ifndef AOMHIP_SUBPEL_SPLIT1
#define AOMHIP_SUBPEL_SPLIT1 2
#endif
template <int H> __device__ __forceinline__ constexpr int parts_of(int n) {
  constexpr int kMax = H >= 16 ? 4 : (H >= 8 ? 2 : 1);
  return n == 1 ? (kMax < AOMHIP_SUBPEL_SPLIT1 ? kMax : AOMHIP_SUBPEL_SPLIT1) : (n == 2 ? (kMax < 2 ? kMax : 2) : 1);
}
// the sums of the nparts (2 or 4) adjacent groups that shared a candidate, in every lane of them
__device__ __forceinline__ void parts_sum(int64_t &tsum, uint64_t &tsse, int nparts) {
  int32_t s32 = (int32_t)tsum;   // |sum| <= 4095 * 128 * 128 < 2^31
  uint32_t lo = (uint32_t)tsse, hi = (uint32_t)(tsse >> 32);
  s32 += __shfl_xor(s32, 16, 64);
  uint64_t q = tsse + (((uint64_t)(uint32_t)__shfl_xor((int)hi, 16, 64) << 32) | (uint32_t)__shfl_xor((int)lo, 16, 64));
  if (nparts == 4) {
    s32 += __shfl_xor(s32, 32, 64);
    lo = (uint32_t)q; hi = (uint32_t)(q >> 32);
    q += ((uint64_t)(uint32_t)__shfl_xor((int)hi, 32, 64) << 32) | (uint32_t)__shfl_xor((int)lo, 32, 64);
  }
  tsum = s32;
  tsse = q;
}

template <typename T, int W, int H, bool SUBPEL>
__device__ __forceinline__ uint32_t group16_variance(const T *ap, int astride, int xoff, int yoff, const T *bp, int bstride,
                                                     bool a_minus_b, int bit_depth, int j, bool active,
                                                     uint32_t *sse_out, const CompoundRef<T> comp = CompoundRef<T>{ nullptr, nullptr, 0 }, int part = 0, int nparts = 1) {
  constexpr int UE = W >= 8 ? 8 : 4;
  constexpr int UPR = W / UE;
  constexpr int U = UPR * H;
  constexpr int UB = UE * (int)sizeof(T);
  using L = typename MLoad<UB>::type;
  constexpr uint8_t kBil[8][2] = { { 128, 0 }, { 112, 16 }, { 96, 32 }, { 80, 48 },
                                   { 64, 64 }, { 48, 80 },  { 32, 96 }, { 16, 112 } };  // aom_filter.h:43-50
  static_assert(kBil[3][0] == 128 - 16 * 3 && kBil[3][1] == 16 * 3 && kBil[7][0] == 16, "bilinear taps are 128 - 16 i, 16 i");
  const int fx1 = (xoff & 7) << 4, fx0 = 128 - fx1, fy1 = (yoff & 7) << 4, fy0 = 128 - fy1;   // = kBil[off & 7]: {128 - 16 i, 16 i} (aom_filter.h:43-50) by arithmetic -- indexed as a table the compiler put it in memory: two loads on the chain of every candidate
  int32_t sum = 0;   // |Σd| <= 4095 * W * H < 2^31 for every block size
  uint64_t sse = 0;
  if (active) {
    for (int u = part * (U / nparts) + j; u < (part + 1) * (U / nparts); u += 16) {   // (nparts groups share the candidate: rows part * H / nparts ..)
      const int row = u / UPR, col = (u % UPR) * UE;
      const T *a0 = ap + (int64_t)row * astride + col;
      const L bv = *reinterpret_cast<const L *>(bp + (int64_t)row * bstride + col);
      const L r0 = *reinterpret_cast<const L *>(a0);
      int us = 0;
      uint32_t uq = 0;  // 8 * 4095^2 < 2^32
      int avv[UE];
      if constexpr (SUBPEL) {
        const L r1 = *reinterpret_cast<const L *>(a0 + astride);
        const int e0 = a0[UE], e1 = a0[astride + UE];
#pragma unroll
        for (int i = 0; i < UE; ++i) {
          const int p00 = px_of<T>(r0.v, i), p01 = i + 1 < UE ? px_of<T>(r0.v, i + 1 < UE ? i + 1 : i) : e0;
          const int p10 = px_of<T>(r1.v, i), p11 = i + 1 < UE ? px_of<T>(r1.v, i + 1 < UE ? i + 1 : i) : e1;
          const int h0 = (__mul24(p00, fx0) + __mul24(p01, fx1) + 64) >> 7;
          const int h1 = (__mul24(p10, fx0) + __mul24(p11, fx1) + 64) >> 7;
          int av = (__mul24(h0, fy0) + __mul24(h1, fy1) + 64) >> 7;   // h <= 4095: 24-bit multiplies (v_mul_lo_u32 is quarter rate)
          av &= (sizeof(T) == 1) ? 0xFF : 0xFFFF;
          avv[i] = av;
        }
      } else {
#pragma unroll
        for (int i = 0; i < UE; ++i) avv[i] = px_of<T>(r0.v, i);
      }
      if (comp.second) comp.template blend_run<UE>(avv, row * W + col);   // (the unit's predictor pixels and weights: one vector load each)
#pragma unroll
      for (int i = 0; i < UE; ++i) {
        const int bvp = px_of<T>(bv.v, i);
        const int d = a_minus_b ? avv[i] - bvp : bvp - avv[i];
        us += d;
        uq += (uint32_t)__mul24(d, d);   // |d| < 2^12: the 24-bit multiplier is exact and full rate (v_mul_lo_u32 is quarter rate)
      }
      sum += us;
      sse += uq;
    }
  }
  int64_t tsum = (int64_t)(int32_t)row16_sum_u32((uint32_t)sum);
  uint64_t tsse = row16_sum_u64(sse);
  if (nparts > 1) parts_sum(tsum, tsse, nparts);
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

#define AOMHIP_FOR_BLOCK_SIZES(X)                                                                                \
  X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) \
  X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)

inline int check_common(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                        const void *blocks, int n, int cost_type) {
  if (!ctx || !src || !ref || !src->base || !ref->base || (n > 0 && !blocks) || n < 0 || frame < 0 ||
      frame >= src->n_frames || frame >= ref->n_frames || !valid_block(bw, bh) ||
      (src->bit_depth == 8) != (ref->bit_depth == 8)) {
    set_error("motion search: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (cost_type == kCostEntropy || cost_type < 0 || cost_type > kCostNone) {
    set_error("motion search: MV_COST_ENTROPY (cost tables) is not supported on the device path yet");
    return AOMHIP_ERR_INVALID;
  }
  return AOMHIP_OK;
}

constexpr int kSearchThreads = 256;  // 4 blocks (wavefronts) per workgroup

}  // namespace aomhip

#endif  // AOMHIP_CSRC_SEARCH_DEVICE_H_
