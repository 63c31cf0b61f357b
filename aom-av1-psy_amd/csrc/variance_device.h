// Shared device pieces of the variance-family kernels (variance.hip, compound.hip): unaligned wide loads, the
// row-unit geometry of a W x H block over TPC lanes, group reductions and the reference's final formulas
// (aom_dsp/variance.c:141-148 VAR, :383-420 HIGHBD_VAR).
#pragma once
#include <type_traits>

#include "common.h"

namespace aomhip {

struct __attribute__((packed, aligned(1))) VU128 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) VU64 { uint32_t v[2]; };
struct __attribute__((packed, aligned(1))) VU32 { uint32_t v[1]; };
template <int BYTES> struct VLoad;
template <> struct VLoad<16> { using type = VU128; };
template <> struct VLoad<8> { using type = VU64; };
template <> struct VLoad<4> { using type = VU32; };

__device__ constexpr uint8_t kBilinear[8][2] = { { 128, 0 }, { 112, 16 }, { 96, 32 }, { 80, 48 },
                                                 { 64, 64 }, { 48, 80 },  { 32, 96 }, { 16, 112 } };

template <int TPC> __device__ __forceinline__ int32_t gsum32(int32_t v) {
  if constexpr (TPC >= 2) v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
  if constexpr (TPC >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
  if constexpr (TPC >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
  if constexpr (TPC >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
  if constexpr (TPC >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (TPC >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
template <int TPC> __device__ __forceinline__ uint64_t gsum64(uint64_t v) {
#pragma unroll
  for (int m = 1; m < TPC; m <<= 1) v += __shfl_xor((unsigned long long)v, m, 64);
  return v;
}

template <typename T, int W, int H> struct VarGeom {
  static constexpr int kRowBytes = W * (int)sizeof(T);
  static constexpr int kUnitBytes = kRowBytes < 16 ? kRowBytes : 16;
  static constexpr int kUnitElems = kUnitBytes / (int)sizeof(T);
  static constexpr int kUnitsPerRow = kRowBytes / kUnitBytes;
  static constexpr int kUnits = kUnitsPerRow * H;
  static constexpr int kTpcRaw = kUnits >= 2 ? kUnits / 2 : 1;
  static constexpr int kTpc = kTpcRaw > 64 ? 64 : kTpcRaw;
  static constexpr int kUnitsPerLane = kUnits / kTpc;
};

template <typename T, int N> __device__ __forceinline__ void load_elems(const T *p, int (&out)[N]) {
  // N elements from an arbitrarily aligned address with one wide load
  using L = typename VLoad<N * (int)sizeof(T)>::type;
  const L raw = *reinterpret_cast<const L *>(p);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if constexpr (sizeof(T) == 1)
      out[i] = (raw.v[i / 4] >> (8 * (i % 4))) & 0xFF;
    else
      out[i] = (raw.v[i / 2] >> (16 * (i % 2))) & 0xFFFF;
  }
}

// variance.c:141-148 VAR / :383-420 HIGHBD_VAR final formulas
template <int BD_CLASS /*0: 8-bit planes*/, int LOG2N>
__device__ __forceinline__ void finish(int64_t sum64, uint64_t sse64, int bit_depth, uint32_t *var, uint32_t *sse) {
  int32_t s;
  uint32_t q;
  if (bit_depth == 10) {
    q = (uint32_t)((sse64 + 8) >> 4);
    s = (int32_t)((sum64 + 2) >> 2);  // arithmetic shift of a possibly negative sum (aom_ports/mem.h:45)
  } else if (bit_depth == 12) {
    q = (uint32_t)((sse64 + 128) >> 8);
    s = (int32_t)((sum64 + 8) >> 4);
  } else {
    q = (uint32_t)sse64;
    s = (int32_t)sum64;
  }
  *sse = q;
  const int64_t sq = ((int64_t)s * s) >> LOG2N;  // sum^2 >= 0: shift == division by W*H
  if (bit_depth == 8) {
    *var = q - (uint32_t)sq;
  } else {
    const int64_t v = (int64_t)q - sq;
    *var = v >= 0 ? (uint32_t)v : 0;
  }
}

constexpr int ilog2v(int n) { return n <= 1 ? 0 : 1 + ilog2v(n >> 1); }
constexpr int kVarThreads = 256;

#define AOMHIP_FOR_BLOCK_SIZES(X)                                                                                \
  X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) \
  X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)

}  // namespace aomhip
