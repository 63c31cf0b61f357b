// Device helpers of the 8-tap inter predictor shared by pred.hip and the fused block kernel (encode_block.hip): the interpolation kernels,
// the packed 16-bit dot product and the dword-aligned 8-pixel load.
#pragma once
#include "common.h"

namespace aomhip {

// AV1 Subpel_Filters [set][phase][tap] (av1/common/filter.h:110-236): regular, smooth, sharp, bilinear, 4-tap regular, 4-tap smooth
__device__ const int16_t kInterp[6][16][8] __attribute__((aligned(16))) = {
#include "interp_table.inc"
};

struct __attribute__((packed, aligned(1))) PU128 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) PU64 { uint32_t v[2]; };

typedef short s16x2 __attribute__((ext_vector_type(2)));
// acc + a.lo * b.lo + a.hi * b.hi on packed signed 16-bit pairs (v_dot2c_i32_i16)
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc) {
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), acc, false);
}

struct __attribute__((aligned(4))) PA128 { uint32_t v[4]; };
struct __attribute__((aligned(4))) PA64 { uint32_t v[2]; };

// 8 consecutive pixels from an arbitrarily aligned address as four (even, odd) 16-bit pairs.  The loads themselves are
// DWORD-ALIGNED (the enclosing dwords, one more than the pixels need) and the sub-dword offset is taken out with
// v_alignbit: a 16-byte load from a 2-byte-aligned address runs at half rate on gfx950 (measured: 26.7 us -> 13.6 us per 4K
// frame for this kernel with the address rounded down, profiles/r01_inter_pred.md), dword alignment is enough for full rate.
template <typename T> __device__ __forceinline__ void load8_pairs(const T *p, uint32_t (&out)[4]) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint32_t sh = (uint32_t)(a & 3) * 8;
  if constexpr (sizeof(T) == 1) {
    const PA64 lo = *reinterpret_cast<const PA64 *>(a & ~(uintptr_t)3);
    const uint32_t hi = *reinterpret_cast<const uint32_t *>((a & ~(uintptr_t)3) + 8);
    const uint32_t r0 = __builtin_amdgcn_alignbit(lo.v[1], lo.v[0], sh), r1 = __builtin_amdgcn_alignbit(hi, lo.v[1], sh);
    out[0] = __builtin_amdgcn_perm(0, r0, 0x0c010c00);
    out[1] = __builtin_amdgcn_perm(0, r0, 0x0c030c02);
    out[2] = __builtin_amdgcn_perm(0, r1, 0x0c010c00);
    out[3] = __builtin_amdgcn_perm(0, r1, 0x0c030c02);
  } else {
    const PA128 lo = *reinterpret_cast<const PA128 *>(a & ~(uintptr_t)3);
    const uint32_t hi = *reinterpret_cast<const uint32_t *>((a & ~(uintptr_t)3) + 16);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = __builtin_amdgcn_alignbit(lo.v[i + 1], lo.v[i], sh);
    out[3] = __builtin_amdgcn_alignbit(hi, lo.v[3], sh);
  }
}

}  // namespace aomhip
