// Device helpers shared by the transform + quantise kernels (xform_quant.hip) and the fused block kernel (encode_block.hip): the
// coefficient stores, QuantArgs + quantize_one (aom_quantize_b_helper_c / aom_highbd_quantize_b_helper_c / the fp family for one
// coefficient), the arithmetic inverse scan position, the per-TX_TYPE 1-D kinds, the fast-butterfly bound and the lane-group maximum.
#pragma once
#include "common.h"
#include "txfm_device.h"

namespace aomhip {
#ifndef AOMHIP_XQ_NT_STORES
#define AOMHIP_XQ_NT_STORES 1
#endif
typedef uint32_t XqV4 __attribute__((ext_vector_type(4)));
// the coefficient outputs are written once and read by a later kernel: streaming (non-temporal) stores
__device__ __forceinline__ void xq_store4(int32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  XqV4 v = { a, b, c, d };
#if AOMHIP_XQ_NT_STORES
  __builtin_nontemporal_store(v, reinterpret_cast<XqV4 *>(p));
#else
  *reinterpret_cast<XqV4 *>(p) = v;
#endif
}
__device__ __forceinline__ void xq_store1(int32_t *p, int32_t v) {
#if AOMHIP_XQ_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

using namespace txfm;

#ifndef AOMHIP_XQ_EXTRAS
#define AOMHIP_XQ_EXTRAS 1   // 0 compiles the fp quantiser and the fused block error out (A/B of their cost on the plain path)
#endif
constexpr int kQuantFp = -2;  // QuantArgs::qs_log2 value that selects the av1_quantize_fp family

struct QuantArgs {
  int16_t zbin[2], round[2], quant[2], quant_shift[2], dequant[2];
  int8_t qs_log2[2];  // log2(quant_shift) when it is a power of two (it always is out of invert_quant,
                      // av1_quantize.c:580-588), else -1 -> generic 64-bit path
};

// aom_quantize_b_helper_c / aom_highbd_quantize_b_helper_c (aom_dsp/quantize.c:139-166,293-313) for one
// coefficient, qm == NULL (wt = iwt = 32), branch-free.  Exact re-associations used for the low-bd form:
//   ((32t * quant) >> 16)            == (t * quant) >> 11         (same rational, same floor; fits int32)
//   (t2 * 2^k) >> m                  == t2 >> (m - k)             (quant_shift is a power of two)
__device__ __forceinline__ int __mulhi24(int a, int b) {   // v_mul_hi_i32_i24: bits 32..47 of the signed 24 x 24-bit product, sign-extended
  int r;
  asm("v_mul_hi_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <bool HBD, int LS>
__device__ __forceinline__ void quantize_one(int32_t v, int zb, int rd, int quant, int qshift, int qs_log2, int dequant,
                                             int32_t *qout, int32_t *dqout) {
  const int sign = v >> 31;
  const int a = (v ^ sign) - sign;
  int q;
  if (AOMHIP_XQ_EXTRAS && qs_log2 == kQuantFp) {
    // av1_quantize_fp_no_qmatrix / highbd_quantize_fp_helper_c (av1/encoder/av1_quantize.c:36-69,181-207): no dead zone
    // table, threshold (|c| << (1 + log_scale)) >= dequant, level = ((|c| + round_fp) * quant_fp) >> (16 - log_scale);
    // the low-bd form saturates |c| + round at INT16_MAX first.  `rd` arrives log-scaled, `quant` is quant_fp.
    int64_t t = (int64_t)a + rd;
    if constexpr (!HBD) t = t > 32767 ? 32767 : t;
    q = (int)((t * quant) >> (16 - LS));
    q = (((int64_t)a << (1 + LS)) >= dequant) ? q : 0;
    const int dqf = (int)((uint32_t)q * (uint32_t)dequant) >> LS;
    *qout = (q ^ sign) - sign;
    *dqout = (dqf ^ sign) - sign;
    return;
  }
  if constexpr (!HBD) {
    int t = a + rd;
    t = t > 32767 ? 32767 : t;  // clamp(.., INT16_MIN, INT16_MAX); a + rd >= 0
    // 0 <= t < 2^15 and quant is an int16: the product is exact on the 24-bit multiplier (v_mul_i32_i24, full rate; the plain `*` compiles to
    // v_mul_lo_u32, a quarter-rate instruction, because the compiler cannot see t's lower bound) -- the kernels run at 0.82-0.96 of their VALU
    // floor (profiles/r04_txq_pmc.json), so issue cycles are launch time
    const int t2 = (__mul24(t, quant) >> 11) + (t << 5);
    if (qs_log2 >= 0)
      q = t2 >> (21 - LS - qs_log2);
    else
      q = (int)(((int64_t)t2 * qshift) >> (21 - LS));
  } else {
    // aom_highbd_quantize_b_helper_c computes in int64: tmp2 = ((32 T quant) >> 16) + 32 T, abs_q = (tmp2 quant_shift) >> (21 - log_scale), T = |c| +
    // round.  (32 T quant) >> 16 == (T quant) >> 11 (same rational, same floor).  A transform coefficient of AV1 has at most bd + 8 <= 20
    // bits, so T < 2^23 and the 48-bit product T x quant comes out of the 24-bit multiplier pair (v_mul_i32_i24 + v_mul_hi_i32_i24, full rate)
    // instead of a 64-bit multiply-add sequence; tmp2 < 2^25 then fits 32 bits and the power-of-two quant_shift is a shift.  Anything
    // outside those ranges takes the literal int64 form.
    const int T = a + rd;
    if (qs_log2 >= 0 && (unsigned)a < (1u << 22) && (unsigned)rd < (1u << 22)) {
      const uint32_t lo = (uint32_t)__mul24(T, quant);
      const int hi = __mulhi24(T, quant);
      const int t2 = (int)__builtin_amdgcn_alignbit((uint32_t)hi, lo, 11) + (T << 5);   // (hi:lo) >> 11: |T quant| < 2^38, so it fits 32 bits
      q = t2 >> (21 - LS - qs_log2);
    } else {
      const int64_t tw = ((int64_t)a + rd) * 32;
      const int64_t t2 = ((tw * quant) >> 16) + tw;
      q = (int)((t2 * qshift) >> (21 - LS));
    }
  }
  q = (a >= zb) ? q : 0;
  int dq;
  if constexpr (!HBD) dq = (int)__umul24((unsigned)q, (unsigned)dequant) >> LS;   // 0 <= q < 2^17, 0 < dequant < 2^15: exact in 24 x 24 bits
  else dq = (int)((uint32_t)q * (uint32_t)dequant) >> LS;
  *qout = (q ^ sign) - sign;
  *dqout = (dq ^ sign) - sign;
}

// The same two helpers with quantisation matrices (qm_ptr / iqm_ptr non-NULL: aom_dsp/quantize.c:39,61-77 and :283,299-312), for one
// coefficient: wt scales the dead-zone test and the level, iwt the dequantiser.  Literal 64-bit arithmetic (this is not the hot path:
// enable_qm defaults to 0, av1/av1_cx_iface.c:246).  zb / rd arrive log-scaled like quantize_one's.
template <bool HBD, int LS>
__device__ __forceinline__ void quantize_one_qm(int32_t v, int zb, int rd, int quant, int qshift, int dequant, int wt, int iwt,
                                                int32_t *qout, int32_t *dqout) {
  constexpr int QM = 5;   // AOM_QM_BITS
  const int sign = v >> 31;
  const int a = (v ^ sign) - sign;
  int q = 0;
  if constexpr (!HBD) {
    if (a * wt >= (zb << QM)) {
      int64_t t = (int64_t)a + rd;
      t = t > 32767 ? 32767 : t;   // clamp(.., INT16_MIN, INT16_MAX); a + rd >= 0
      t *= wt;
      q = (int)(((((t * quant) >> 16) + t) * qshift) >> (16 - LS + QM));
    }
  } else {
    const int cw = (int)((uint32_t)v * (uint32_t)wt);   // `coeff_ptr[rc] * wt` in int
    if (cw >= zb * (1 << QM) || cw <= -zb * (1 << QM)) {
      const int64_t tw = ((int64_t)a + rd) * wt;
      const int64_t t2 = ((tw * quant) >> 16) + tw;
      q = (int)((t2 * qshift) >> (16 - LS + QM));
    }
  }
  const int dqv = (dequant * iwt + (1 << (QM - 1))) >> QM;
  const int adq = (int)((uint32_t)q * (uint32_t)dqv) >> LS;
  *qout = (q ^ sign) - sign;
  *dqout = (adq ^ sign) - sign;
}

// The `fp` quantiser with matrices (quantize_fp_helper_c / highbd_quantize_fp_helper_c, av1/encoder/av1_quantize.c:92-121,141-169): dead zone
// a * wt >= dequant << (AOM_QM_BITS - (1 + log_scale)), level (a + round) * wt * quant >> (16 - log_scale + AOM_QM_BITS) with the
// low-bit-depth form's int16 clamp of a + round.  rd arrives log-scaled; quant / rd are the caller's quant_fp / round_fp.
template <bool HBD, int LS>
__device__ __forceinline__ void quantize_one_qm_fp(int32_t v, int rd, int quant, int dequant, int wt, int iwt, int32_t *qout, int32_t *dqout) {
  constexpr int QM = 5;   // AOM_QM_BITS
  const int sign = v >> 31;
  int64_t a = (int64_t)((v ^ sign) - sign);
  int q = 0;
  if (a * wt >= (dequant << (QM - (1 + LS)))) {
    a += rd;
    if (!HBD && a > 32767) a = 32767;   // clamp64(.., INT16_MIN, INT16_MAX); a >= 0
    q = (int)((a * wt * quant) >> (16 - LS + QM));
  }
  const int dqv = (dequant * iwt + (1 << (QM - 1))) >> QM;
  const int adq = (int)((uint32_t)q * (uint32_t)dqv) >> LS;
  *qout = (q ^ sign) - sign;
  *dqout = (adq ^ sign) - sign;
}

// av1_scan_orders (scan.c:1666-): class 0 = zig-zag (all 2-D types), 1 = "mrow" (V_* types),
// 2 = "mcol" (H_* types).  Position of coefficient (r, c) in a KW x KH scan.
template <int KW, int KH> __device__ __forceinline__ int iscan_pos(int r, int c, int scan_class) {
  if (scan_class == 2) return c * KH + r;
  if (scan_class == 1) return r * KW + c;
  constexpr int m = KW < KH ? KW : KH, M = KW < KH ? KH : KW;
  const int d = r + c;
  int before;  // coefficients on earlier anti-diagonals (all factors < 128: 24-bit multiplies, v_mul_lo_u32 is quarter rate)
  if (d <= m)
    before = __mul24(d, d + 1) / 2;
  else if (d <= M)
    before = m * (m + 1) / 2 + __mul24(d - m, m);
  else
    before = KW * KH - __mul24(KW + KH - 1 - d, KW + KH - d) / 2;
  const bool up = (KW > KH) || (KW == KH && (d & 1) == 0);
  const int cmin = d - (KH - 1) > 0 ? d - (KH - 1) : 0;
  const int rmin = d - (KW - 1) > 0 ? d - (KW - 1) : 0;
  return before + (up ? c - cmin : r - rmin);
}

// per-TX_TYPE vertical / horizontal 1-D kinds (common_data.h:149-159): 0 DCT 1 ADST 2 FLIPADST 3 IDTX
__device__ constexpr uint8_t kVKind[16] = { 0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3 };
__device__ constexpr uint8_t kHKind[16] = { 0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2 };
// ... and of the inverse transform (the same table: av1_inv_txfm2d.c reads vtx_tab / htx_tab too)
__device__ constexpr uint8_t kIVKind[16] = { 0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3 };
__device__ constexpr uint8_t kIHKind[16] = { 0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2 };

// The same four tables as 2-bit fields of one 32-bit constant each: `kVKind[tx_type]` with a block's own type is an index into constant MEMORY (a vector load
// behind the record's load at the head of every transform kernel); a shift and a mask of an immediate are not.
constexpr uint32_t pack_kinds(const uint8_t (&t)[16]) {
  uint32_t v = 0;
  for (int i = 0; i < 16; ++i) v |= (uint32_t)(t[i] & 3) << (2 * i);
  return v;
}
__device__ __forceinline__ int v_kind(int tx_type) { return (int)((pack_kinds(kVKind) >> (2 * (tx_type & 15))) & 3u); }
__device__ __forceinline__ int h_kind(int tx_type) { return (int)((pack_kinds(kHKind) >> (2 * (tx_type & 15))) & 3u); }
__device__ __forceinline__ int iv_kind(int tx_type) { return (int)((pack_kinds(kIVKind) >> (2 * (tx_type & 15))) & 3u); }
__device__ __forceinline__ int ih_kind(int tx_type) { return (int)((pack_kinds(kIHKind) >> (2 * (tx_type & 15))) & 3u); }

// Largest residual magnitude per [TX_SIZE][TX_TYPE] under which the fast butterfly is exact (txfm_device.h: kFastBtf)
__device__ constexpr int16_t kSafeMax[19][16] = {
#include "txfm_safe_max.inc"
};
constexpr int tx_index_of(int w, int h) {  // TX_SIZE (av1/common/enums.h:174-197) of a w x h transform
  constexpr int tw[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
  constexpr int th[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
  for (int i = 0; i < 19; ++i)
    if (tw[i] == w && th[i] == h) return i;
  return 0;
}
#ifndef AOMHIP_XQ_FAST_BTF
#define AOMHIP_XQ_FAST_BTF 1   // 0: every block takes the exact butterfly (A/B: profiles/history_r02_r04.md)
#endif
// the 1-D pass of a block whose residual magnitude allows (fast) / does not allow the fast butterfly
template <int N, int BIT> __device__ __forceinline__ void fwd_1d_sel(int32_t (&x)[N], int kind, bool fast) {
  if (AOMHIP_XQ_FAST_BTF && fast) fwd_1d<N, BIT + txfm::kFastBtf>(x, kind);
  else fwd_1d<N, BIT>(x, kind);
}

template <int LPB> __device__ __forceinline__ int group_max(int v) {
  if constexpr (LPB >= 2) v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
  if constexpr (LPB >= 4) v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
  if constexpr (LPB >= 8) v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
  if constexpr (LPB >= 16) v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false));
  if constexpr (LPB >= 32) v = max(v, __shfl_xor(v, 16, 64));
  if constexpr (LPB >= 64) v = max(v, __shfl_xor(v, 32, 64));
  return v;
}


// The lanes of a transform block (max(W, H) <= 64 adjacent lanes, a power of two) sit in ONE wavefront and a wavefront's LDS instructions
// execute in program order: what one lane wrote is there for the block's other lanes as soon as the compiler keeps the accesses in order --
// no s_barrier, the workgroup's other wavefronts (other blocks, private LDS regions) never wait for each other.
#ifndef AOMHIP_TX_WAVE_SYNC
#define AOMHIP_TX_WAVE_SYNC 1   // 0: workgroup barriers (A/B)
#endif
__device__ __forceinline__ void block_sync() {
#if AOMHIP_TX_WAVE_SYNC
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  __syncthreads();
#endif
}

// QuantArgs of a parameter block (quant_kind 1: the av1_quantize_fp family)
inline QuantArgs to_quant_args(const aomhip_quant_params *q, int quant_kind = 0) {
  QuantArgs a;
  for (int i = 0; i < 2; ++i) {
    a.zbin[i] = q->zbin[i];
    a.round[i] = q->round[i];
    a.quant[i] = q->quant[i];
    a.quant_shift[i] = q->quant_shift[i];
    a.dequant[i] = q->dequant[i];
    const int qs = q->quant_shift[i];
    a.qs_log2[i] = -1;
    if (qs > 0 && (qs & (qs - 1)) == 0) {
      int l = 0;
      while ((1 << l) < qs) ++l;
      a.qs_log2[i] = (int8_t)l;
    }
  }
  if (quant_kind == 1) a.qs_log2[0] = a.qs_log2[1] = (int8_t)kQuantFp;
  return a;
}

}  // namespace aomhip
