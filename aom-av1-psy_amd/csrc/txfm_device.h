// Device-side AV1 integer transforms (forward and inverse 1-D DCT / ADST / identity),
// register-resident and fully unrolled: every index below is a compile-time constant after
// unrolling, so an N-point vector lives in N VGPRs and the cos/sin weights become literals.
//
// Reference: av1/encoder/av1_fwd_txfm1d.c, av1/common/av1_inv_txfm1d.c (straight-line
// butterfly listings), av1/common/av1_txfm.h:75-102 (round_shift, half_btf).  The networks are
// generated from their recursive structure instead of being listed:
//   DCT-N  : fold (a+b, a-b), DCT-N/2 on the sums, "odd part" on the differences, bit reversal.
//            odd part of m = N/2 points = log2(m)-1 levels of {partial rotation, grouped
//            butterflies} + m/2 final rotations by the odd angles.
//   ADST-N : signed input permutation, log2(N)-1 levels of {rotate upper half of each
//            2^(s+1) group, stride-2^s butterflies}, N/2 final rotations, output permutation.
//   inverse: the transposed graph walked backwards, clamp_value() on every add/sub.
#ifndef AOMHIP_CSRC_TXFM_DEVICE_H_
#define AOMHIP_CSRC_TXFM_DEVICE_H_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace aomhip {

// tx_type value (in aomhip_txb / uniform_tx_type) that selects the lossless 4x4 Walsh-Hadamard pair (AOMHIP_TX_WHT)
constexpr int kTxWht = 16;

namespace txfm {

__device__ constexpr int32_t kCospi[7][64] = {
#include "cospi_table.inc"
};
// av1/common/av1_txfm.c:62-69 (adjusted in the reference so that [1] + [2] == [4])
__device__ constexpr int32_t kSinpi[7][5] = { { 0, 330, 621, 836, 951 },         { 0, 660, 1241, 1672, 1901 },
                                              { 0, 1321, 2482, 3344, 3803 },     { 0, 2642, 4964, 6689, 7606 },
                                              { 0, 5283, 9929, 13377, 15212 },   { 0, 10566, 19858, 26755, 30424 },
                                              { 0, 21133, 39716, 53510, 60849 } };

constexpr int kSqrt2 = 5793, kInvSqrt2 = 2896, kSqrt2Bits = 12;  // av1_txfm.h:41-45

constexpr int ilog2c(int n) { return n <= 1 ? 0 : 1 + ilog2c(n >> 1); }
constexpr int bitrevc(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

// round_shift (av1_txfm.h:75-78) without the 64-bit add: floor((x + 2^(bit-1)) / 2^bit)
__device__ __forceinline__ int32_t rshift(int32_t x, int bit) { return (x >> bit) + ((x >> (bit - 1)) & 1); }
// round_shift of a 64-bit product
__device__ __forceinline__ int32_t rshift64(int64_t v, int bit) { return (int32_t)((v + ((int64_t)1 << (bit - 1))) >> bit); }

// half_btf (av1_txfm.h:80-102): 32-bit wrapping products, 64-bit sum + rounding, shift.
// The template argument of every network below carries one extra flag on top of the cos bit: BIT = cos_bit + kFastBtf selects the FAST
// butterfly -- two 24-bit multiply-adds into a 32-bit wrapping sum, one arithmetic shift: 3 instructions instead of 9 -- which equals
// half_btf whenever both operands lie inside (-2^23, 2^23) (the 32-bit result of v_mad_i32_i24 is then the low half of the exact product,
// i.e. the reference's wrapped product) and the exact sum w0 a + w1 b + 2^(bit-1) fits int32.  Callers may only set the flag for blocks
// whose residual magnitude is at most kSafeMax[tx_size][tx_type] (txfm_safe_max.inc): an offline interval analysis of every stage proves
// both conditions for every butterfly of both passes under that bound (the generator is named in the table's header; tests/test_oracle_txfm_bounds.py re-derives it).
// Every 8-bit video residual (|r| <= 255) qualifies at every size and type; a 10-bit one (|r| <= 1023) qualifies everywhere except where
// txfm_safe_max.inc is below 1023 -- the ADST / FLIPADST rows of TX_8X4 (915 / 781) and TX_16X4 (390 / 329): blocks of those sizes and types
// whose residual exceeds the bound silently take the exact form (same result, 3 x the butterfly instructions), as do 12-bit residuals above
// the bound of their size.  tests/test_gpu_xform_quant.py runs blocks AT the bound and one above it.  The forward transform + quantise
// kernels were VALU-issue bound on exactly these butterflies (profiles/r02_txq.md, r03_txq.md).
constexpr int kFastBtf = 32;
template <int BIT> __device__ __forceinline__ int32_t hbtf(int32_t w0, int32_t a, int32_t w1, int32_t b) {
  constexpr int B = BIT & 31;
  if constexpr (BIT >= kFastBtf) {
    // (unsigned adds: the partial sums may wrap, only the total is known to fit; the compiler fuses them into v_mad_i32_i24)
    return (int32_t)((uint32_t)__mul24(w0, a) + (uint32_t)__mul24(w1, b) + (1u << (B - 1))) >> B;
  } else {
    const int32_t p0 = (int32_t)((uint32_t)w0 * (uint32_t)a);
    const int32_t p1 = (int32_t)((uint32_t)w1 * (uint32_t)b);
    const int64_t s = (int64_t)p0 + (int64_t)p1 + ((int64_t)1 << (B - 1));
    return (int32_t)(s >> B);
  }
}
__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
// clamp_value (av1_inv_txfm1d.h:21-26); CB <= 0 disables
template <int CB> __device__ __forceinline__ int32_t clampv(int32_t v) {
  if constexpr (CB <= 0 || CB >= 32) {
    return v;
  } else {
    constexpr int32_t hi = (int32_t)(((int64_t)1 << (CB - 1)) - 1), lo = (int32_t)(-((int64_t)1 << (CB - 1)));
    return v < lo ? lo : (v > hi ? hi : v);
  }
}

// ------------------------------------------------------------------ DCT

template <int M, int J, int BIT, int OFF, int NN> __device__ __forceinline__ void dct_odd_rot(int32_t (&x)[NN]) {
  constexpr int g = M >> J, base = 64 >> J;
#pragma unroll
  for (int l = 0; l < M / 2; ++l) {
    const int pos = (l + 2 * g - g / 2) % (2 * g);
    if (pos >= g) continue;
    const int run = (l - g / 2) / (2 * g);
    const int a = (J == 1) ? 32 : base * (1 + 4 * bitrevc(run, J - 2));
    const int h = M - 1 - l;
    const int32_t ca = kCospi[(BIT & 31) - 10][a], cb = kCospi[(BIT & 31) - 10][64 - a];
    const int32_t lo = x[OFF + l], hi = x[OFF + h];
    if (pos < g / 2 || J == 1) {
      x[OFF + l] = hbtf<BIT>(-ca, lo, cb, hi);
      x[OFF + h] = hbtf<BIT>(ca, hi, cb, lo);
    } else {
      x[OFF + l] = hbtf<BIT>(-cb, lo, -ca, hi);
      x[OFF + h] = hbtf<BIT>(cb, hi, -ca, lo);
    }
  }
}
template <int M, int J, int CB, int OFF, int NN> __device__ __forceinline__ void dct_odd_bfly(int32_t (&x)[NN]) {
  constexpr int G = M >> J;
#pragma unroll
  for (int t = 0; t < M / G; ++t) {
#pragma unroll
    for (int i = 0; i < G / 2; ++i) {
      const int lo = OFF + t * G + i, hi = OFF + t * G + G - 1 - i;
      const int32_t a = x[lo], b = x[hi];
      if ((t & 1) == 0) {
        x[lo] = clampv<CB>(wadd(a, b));
        x[hi] = clampv<CB>(wsub(a, b));
      } else {
        x[lo] = clampv<CB>(wsub(b, a));
        x[hi] = clampv<CB>(wadd(b, a));
      }
    }
  }
}
template <int M, bool INV, int BIT, int OFF, int NN> __device__ __forceinline__ void dct_odd_final(int32_t (&x)[NN]) {
  constexpr int N = 2 * M, LN = ilog2c(N);
#pragma unroll
  for (int l = 0; l < M / 2; ++l) {
    const int h = M - 1 - l;
    const int th = bitrevc(M + l, LN) * 64 / N;
    const int32_t c = kCospi[(BIT & 31) - 10][64 - th], s = kCospi[(BIT & 31) - 10][th];
    const int32_t lo = x[OFF + l], hi = x[OFF + h];
    if (!INV) {
      x[OFF + l] = hbtf<BIT>(c, lo, s, hi);
      x[OFF + h] = hbtf<BIT>(c, hi, -s, lo);
    } else {
      x[OFF + l] = hbtf<BIT>(c, lo, -s, hi);
      x[OFF + h] = hbtf<BIT>(s, lo, c, hi);
    }
  }
}
template <int M, int J, bool INV, int BIT, int CB, int OFF, int NN> struct DctOddLevels {
  static __device__ __forceinline__ void run(int32_t (&x)[NN]) {
    if constexpr (J >= 1 && J < ilog2c(M)) {
      if constexpr (!INV) {
        dct_odd_rot<M, J, BIT, OFF, NN>(x);
        dct_odd_bfly<M, J, 0, OFF, NN>(x);
        DctOddLevels<M, J + 1, INV, BIT, CB, OFF, NN>::run(x);
      } else {
        dct_odd_bfly<M, J, CB, OFF, NN>(x);
        dct_odd_rot<M, J, BIT, OFF, NN>(x);
        DctOddLevels<M, J - 1, INV, BIT, CB, OFF, NN>::run(x);
      }
    }
  }
};
// in-place on x[0..N), natural (pre bit-reversal) order
template <int N, int BIT, int NN> __device__ __forceinline__ void fdct_rec(int32_t (&x)[NN]) {
  if constexpr (N == 2) {
    const int32_t c = kCospi[(BIT & 31) - 10][32];
    const int32_t a = x[0], b = x[1];
    x[0] = hbtf<BIT>(c, a, c, b);
    x[1] = hbtf<BIT>(-c, b, c, a);
  } else {
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const int32_t a = x[i], b = x[N - 1 - i];
      x[i] = wadd(a, b);
      x[N - 1 - i] = wsub(a, b);
    }
    fdct_rec<N / 2, BIT, NN>(x);
    DctOddLevels<N / 2, 1, false, BIT, 0, N / 2, NN>::run(x);
    dct_odd_final<N / 2, false, BIT, N / 2, NN>(x);
  }
}
template <int N, int BIT, int CB, int NN> __device__ __forceinline__ void idct_rec(int32_t (&x)[NN]) {
  if constexpr (N == 2) {
    const int32_t c = kCospi[(BIT & 31) - 10][32];
    const int32_t a = x[0], b = x[1];
    x[0] = hbtf<BIT>(c, a, c, b);
    x[1] = hbtf<BIT>(c, a, -c, b);
  } else {
    idct_rec<N / 2, BIT, CB, NN>(x);
    dct_odd_final<N / 2, true, BIT, N / 2, NN>(x);
    DctOddLevels<N / 2, ilog2c(N / 2) - 1, true, BIT, CB, N / 2, NN>::run(x);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      const int32_t a = x[i], b = x[N - 1 - i];
      x[i] = clampv<CB>(wadd(a, b));
      x[N - 1 - i] = clampv<CB>(wsub(a, b));
    }
  }
}

// ------------------------------------------------------------------ ADST

__device__ constexpr uint8_t kAdstSigma8[8] = { 0, 4, 6, 2, 3, 7, 5, 1 };
__device__ constexpr uint8_t kAdstSigma16[16] = { 0, 8, 12, 4, 6, 14, 10, 2, 3, 11, 15, 7, 5, 13, 9, 1 };
constexpr int adst_sigma(int n, int i) { return n == 8 ? kAdstSigma8[i] : kAdstSigma16[i]; }

template <int N, int S, int BIT> __device__ __forceinline__ void adst_level_rot(int32_t (&v)[N]) {
  constexpr int half = 1 << S, group = 2 << S, base = 64 >> S;
  constexpr int pairs = half / 2, nP = pairs >= 2 ? pairs / 2 : 1;
#pragma unroll
  for (int gb = 0; gb < N; gb += group) {
#pragma unroll
    for (int q = 0; q < pairs; ++q) {
      const int p = gb + half + 2 * q;
      const int a = base * (1 + 4 * bitrevc(q % nP, ilog2c(nP)));
      const int32_t ca = kCospi[(BIT & 31) - 10][a], cb = kCospi[(BIT & 31) - 10][64 - a];
      const int32_t x = v[p], y = v[p + 1];
      if (q < nP) {
        v[p] = hbtf<BIT>(ca, x, cb, y);
        v[p + 1] = hbtf<BIT>(cb, x, -ca, y);
      } else {
        v[p] = hbtf<BIT>(-cb, x, ca, y);
        v[p + 1] = hbtf<BIT>(ca, x, cb, y);
      }
    }
  }
}
template <int N, int S, int CB> __device__ __forceinline__ void adst_level_bfly(int32_t (&v)[N]) {
  constexpr int half = 1 << S, group = 2 << S;
#pragma unroll
  for (int gb = 0; gb < N; gb += group) {
#pragma unroll
    for (int i = 0; i < half; ++i) {
      const int32_t a = v[gb + i], b = v[gb + half + i];
      v[gb + i] = clampv<CB>(wadd(a, b));
      v[gb + half + i] = clampv<CB>(wsub(a, b));
    }
  }
}
template <int N, int BIT> __device__ __forceinline__ void adst_final_rot(int32_t (&v)[N]) {
#pragma unroll
  for (int q = 0; q < N / 2; ++q) {
    const int a = (1 + 4 * q) * 32 / N;
    const int32_t ca = kCospi[(BIT & 31) - 10][a], cb = kCospi[(BIT & 31) - 10][64 - a];
    const int32_t x = v[2 * q], y = v[2 * q + 1];
    v[2 * q] = hbtf<BIT>(ca, x, cb, y);
    v[2 * q + 1] = hbtf<BIT>(cb, x, -ca, y);
  }
}
template <int N, int S, bool INV, int BIT, int CB> struct AdstLevels {
  static __device__ __forceinline__ void run(int32_t (&v)[N]) {
    if constexpr (S >= 1 && S < ilog2c(N)) {
      if constexpr (!INV) {
        adst_level_rot<N, S, BIT>(v);
        adst_level_bfly<N, S, 0>(v);
        AdstLevels<N, S + 1, INV, BIT, CB>::run(v);
      } else {
        adst_level_bfly<N, S, CB>(v);
        adst_level_rot<N, S, BIT>(v);
        AdstLevels<N, S - 1, INV, BIT, CB>::run(v);
      }
    }
  }
};

// av1_fadst4 (av1_fwd_txfm1d.c:676-733): all products / sums wrap at 32 bits
template <int BIT> __device__ __forceinline__ void fadst4(int32_t (&x)[4]) {
  const uint32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  constexpr uint32_t s1 = kSinpi[(BIT & 31) - 10][1], s2 = kSinpi[(BIT & 31) - 10][2], s3 = kSinpi[(BIT & 31) - 10][3],
                     s4 = kSinpi[(BIT & 31) - 10][4];
  const uint32_t a = s1 * x0 + s2 * x1 + s4 * x3;
  const uint32_t b = s3 * (x0 + x1 - x3);
  const uint32_t c = s4 * x0 - s1 * x1 + s2 * x3;
  const uint32_t d = s3 * x2;
  // an all-zero input gives all-zero output through the same formula (the reference's early-out)
  x[0] = rshift((int32_t)(a + d), BIT & 31);
  x[1] = rshift((int32_t)b, BIT & 31);
  x[2] = rshift((int32_t)(c - d), BIT & 31);
  x[3] = rshift((int32_t)(c - a + d), BIT & 31);
}
// av1_iadst4 (av1_inv_txfm1d.c:656-711)
template <int BIT> __device__ __forceinline__ void iadst4(int32_t (&x)[4]) {
  const uint32_t x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  constexpr uint32_t s1 = kSinpi[(BIT & 31) - 10][1], s2 = kSinpi[(BIT & 31) - 10][2], s3 = kSinpi[(BIT & 31) - 10][3],
                     s4 = kSinpi[(BIT & 31) - 10][4];
  const uint32_t a = s1 * x0 + s4 * x2 + s2 * x3;
  const uint32_t b = s2 * x0 - s1 * x2 - s4 * x3;
  const uint32_t c = s3 * x1;
  const uint32_t d = s3 * (x0 - x2 + x3);
  x[0] = rshift((int32_t)(a + c), BIT & 31);
  x[1] = rshift((int32_t)(b + c), BIT & 31);
  x[2] = rshift((int32_t)d, BIT & 31);
  x[3] = rshift((int32_t)(a + b - c), BIT & 31);
}

// ------------------------------------------------------------------ 1-D entry points (in place)

enum Kind1D { kDct = 0, kAdst = 1, kIdtx = 3 };  // 2 (FLIPADST) is ADST on flipped data

template <int N> __device__ __forceinline__ void identity(int32_t (&x)[N]) {
  // av1_fwd_txfm1d.c:1064-1094 == av1_inv_txfm1d.c:1029-1060
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (N == 4)
      x[i] = rshift64((int64_t)x[i] * kSqrt2, kSqrt2Bits);
    else if constexpr (N == 8)
      x[i] = (int32_t)((uint32_t)x[i] * 2u);
    else if constexpr (N == 16)
      x[i] = rshift64((int64_t)x[i] * (2 * kSqrt2), kSqrt2Bits);
    else
      x[i] = (int32_t)((uint32_t)x[i] * 4u);
  }
}

template <int N, int BIT> __device__ __forceinline__ void fwd_dct(int32_t (&x)[N]) {
  fdct_rec<N, BIT, N>(x);
  constexpr int L = ilog2c(N);
  int32_t t[N];
#pragma unroll
  for (int k = 0; k < N; ++k) t[k] = x[bitrevc(k, L)];
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = t[k];
}
template <int N, int BIT, int CB> __device__ __forceinline__ void inv_dct(int32_t (&x)[N]) {
  constexpr int L = ilog2c(N);
  int32_t t[N];
#pragma unroll
  for (int p = 0; p < N; ++p) t[p] = x[bitrevc(p, L)];
  idct_rec<N, BIT, CB, N>(t);
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = t[k];
}
template <int N, int BIT> __device__ __forceinline__ void fwd_adst(int32_t (&x)[N]) {
  if constexpr (N == 4) {
    fadst4<BIT>(x);
  } else if constexpr (N == 8 || N == 16) {
    int32_t v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) v[adst_sigma(N, i)] = (i & 1) ? (int32_t)(0u - (uint32_t)x[i]) : x[i];
    AdstLevels<N, 1, false, BIT, 0>::run(v);
    adst_final_rot<N, BIT>(v);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      x[2 * i] = v[2 * i + 1];
      x[2 * i + 1] = v[N - 2 - 2 * i];
    }
  }
}
template <int N, int BIT, int CB> __device__ __forceinline__ void inv_adst(int32_t (&x)[N]) {
  if constexpr (N == 4) {
    iadst4<BIT>(x);
  } else if constexpr (N == 8 || N == 16) {
    int32_t v[N];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
      v[2 * i + 1] = x[2 * i];
      v[N - 2 - 2 * i] = x[2 * i + 1];
    }
    adst_final_rot<N, BIT>(v);
    AdstLevels<N, ilog2c(N) - 1, true, BIT, CB>::run(v);
#pragma unroll
    for (int i = 0; i < N; ++i) x[i] = (i & 1) ? (int32_t)(0u - (uint32_t)v[adst_sigma(N, i)]) : v[adst_sigma(N, i)];
  }
}

// kind: 0 DCT, 1 ADST (also used for FLIPADST after the caller's flip), 3 identity.
// Invalid combinations (ADST above 16 points, identity above 32) are never dispatched.
template <int N, int BIT> __device__ __forceinline__ void fwd_1d(int32_t (&x)[N], int kind) {
  if (kind == kDct) {
    fwd_dct<N, BIT>(x);
  } else if (kind == kIdtx) {
    if constexpr (N <= 32) identity<N>(x);
  } else {
    if constexpr (N <= 16) fwd_adst<N, BIT>(x);
  }
}
template <int N, int BIT, int CB> __device__ __forceinline__ void inv_1d(int32_t (&x)[N], int kind) {
  if (kind == kDct) {
    inv_dct<N, BIT, CB>(x);
  } else if (kind == kIdtx) {
    if constexpr (N <= 32) identity<N>(x);
  } else {
    if constexpr (N <= 16) inv_adst<N, BIT, CB>(x);
  }
}

// ------------------------------------------------------------------ 2-D configuration (compile time)

// av1_fwd_txfm2d.c:314-358 forward shifts and cos bits, av1_inv_txfm2d.c:131-157 inverse shifts
template <int W, int H> struct Cfg2D {
  static constexpr int lw = ilog2c(W) - 2, lh = ilog2c(H) - 2;
  static constexpr int kColBitTab[5][5] = {
    { 13, 13, 13, 0, 0 }, { 13, 13, 13, 12, 0 }, { 13, 13, 13, 12, 13 }, { 0, 13, 13, 12, 13 }, { 0, 0, 13, 12, 13 }
  };
  static constexpr int kRowBitTab[5][5] = {
    { 13, 13, 12, 0, 0 }, { 13, 13, 13, 12, 0 }, { 13, 13, 12, 13, 12 }, { 0, 12, 13, 12, 11 }, { 0, 0, 12, 11, 10 }
  };
  static constexpr int cos_bit_col = kColBitTab[lw][lh];
  static constexpr int cos_bit_row = kRowBitTab[lw][lh];
  static constexpr int M = W > H ? W : H, m = W > H ? H : W;
  static constexpr bool rect2 = (M == 2 * m);
  // forward {pre, mid, post}: every size is {2, -k, 0} except the 64-point ones
  static constexpr int fs0 = (W == 64 && H == 64) || (W == 32 && H == 64) || (W == 16 && H == 64) ? 0 : 2;
  static constexpr int fs1 = (M == 64) ? ((W == 64 && H < 64) ? -4 : -2)
                                       : (M == 32 ? (m == 8 ? -2 : -4) : (M == 16 ? (m == 4 ? -1 : -2) : (M == 8 ? -1 : 0)));
  static constexpr int fs2 = (W == 64 && H == 64) || (W == 32 && H == 64) || (W == 64 && H == 32) ? -2 : 0;
  // inverse {after rows, after cols}
  static constexpr int is0 = (M == 4) ? 0
                             : (M == 8 ? (m == 4 ? 0 : -1)
                                       : (M == 16 ? (m == 16 ? -2 : -1)
                                                  : (M == 32 ? ((m == 32 || m == 8) ? -2 : -1)
                                                             : ((m == 64 || m == 16) ? -2 : -1))));
  static constexpr int is1 = -4;
};

}  // namespace txfm
}  // namespace aomhip
#endif  // AOMHIP_CSRC_TXFM_DEVICE_H_
