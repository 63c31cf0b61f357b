// Fused forward 2-D transform + aom_quantize_b on gfx950.
//
// Reference path: av1/encoder/encodemb.c:288-341 av1_xform_quant = av1_xform (hybrid_fwd_txfm.c:233
// -> av1_fwd_txfm2d_WxH_c, av1_fwd_txfm2d.c:56-312) followed by av1_quant ->
// av1_quantize_b_facade (av1_quantize.c:302-372) -> aom_quantize_b{,_32x32,_64x64}_c /
// aom_highbd_quantize_b* (aom_dsp/quantize.c:108-169,261-316,399-470).
//
// Mapping.  A W x H transform block is owned by max(W, H) adjacent lanes of one wavefront.
//   column pass: lane c holds column c (H values in VGPRs), runs the H-point network
//   transpose  : through a padded LDS tile (stride W + 1 words: conflict-free both ways)
//   row pass   : lane r holds row r (W values), runs the W-point network, then quantises its W
//                coefficients in registers and stores coeff / qcoeff / dqcoeff in the reference's
//                transposed layout (index c*H + r): for each c the block's lanes write one
//                contiguous 4*H-byte run.
// eob is a max-reduction of (inverse-scan position + 1) over non-zero levels; the inverse scan
// position is computed arithmetically (zig-zag / row / column rule of av1/common/scan.c) instead
// of being looked up.  The kernel moves 2 B in and 8 (+4 with coeff) B out per sample and does
// O(log N) butterflies per sample: HBM-bound for small sizes, VALU-bound towards 32 / 64 points.
// 64-point sizes only compute the 32 low-frequency outputs per dimension (the reference computes
// and then discards the rest, av1_fwd_txfm2d.c:241-312).
#include "common.h"
#include "txfm_device.h"
#include "quant_device.h"
#include "txb_cost_table.inc"

namespace aomhip {
template <int LPB> __device__ __forceinline__ int64_t group_sum64(int64_t v) {
#pragma unroll
  for (int m = 1; m < LPB; m <<= 1) v += __shfl_xor((long long)v, m, 64);
  return v;
}
// one term pair of av1_block_error_c / av1_highbd_block_error_c (av1/encoder/rdopt.c:635-682): err_shift < 0 selects the
// low-bd form, whose products are 32-bit (`diff * diff` on int operands wraps exactly as the compiled reference does)
__device__ __forceinline__ void block_err_acc(int32_t c, int32_t dq, int err_shift, int64_t &e, int64_t &z) {
  const int32_t diff = c - dq;
  if (err_shift < 0) {
    e += (int64_t)(int32_t)((uint32_t)diff * (uint32_t)diff);
    z += (int64_t)(int32_t)((uint32_t)c * (uint32_t)c);
  } else {
    e += (int64_t)diff * diff;
    z += (int64_t)c * c;
  }
}
__device__ __forceinline__ void block_err_store(int64_t *out, int bi, int64_t e, int64_t z, int err_shift) {
  if (err_shift > 0) {
    const int64_t r = (int64_t)1 << (err_shift - 1);
    e = (e + r) >> err_shift;
    z = (z + r) >> err_shift;
  }
  out[2 * (int64_t)bi] = e;
  out[2 * (int64_t)bi + 1] = z;
}

constexpr int kXqThreads = 256;

struct __attribute__((packed, aligned(1))) VecU128 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) VecU64 { uint32_t v[2]; };
struct __attribute__((packed, aligned(1))) VecU32 { uint32_t v[1]; };

// SRC: 0 = int16 residual plane; 1 = uint8 src - pred planes; 2 = uint16 src - pred planes
template <int W, int H, bool HBD, int SRC>
__global__ __launch_bounds__(kXqThreads) void xform_quant_kernel(
    const void *__restrict__ in0, const void *__restrict__ in1, int stride0, int stride1,
    const aomhip_txb *__restrict__ blocks, int n_blocks, int grid_cols, int uniform_type, QuantArgs qa,
    int32_t *__restrict__ coeff, int32_t *__restrict__ qcoeff, int32_t *__restrict__ dqcoeff,
    uint16_t *__restrict__ eob, int nblk8, int64_t *__restrict__ err_out, int err_shift) {
  using C = Cfg2D<W, H>;
  constexpr int LPB = W > H ? W : H;
  constexpr int BPW = kXqThreads / LPB;  // blocks per workgroup
  constexpr int KW = W < 32 ? W : 32, KH = H < 32 ? H : 32;
  constexpr int NC = KW * KH;                       // av1_get_max_eob
  constexpr int LS = (W * H > 256) + (W * H > 1024);  // av1_get_tx_scale (av1/common/idct.c:24-28)
  constexpr int LSTRIDE = W + 1;
  __shared__ int32_t tile[BPW][KH * LSTRIDE];

  const int slot = threadIdx.x / LPB, lane = threadIdx.x % LPB;
  const unsigned wg = xcd_chunked_index(blockIdx.x, nblk8);
  const int bi = wg * BPW + slot;
  const bool live = bi < n_blocks;
  int bx = 0, by = 0, tx_type = uniform_type;
  int64_t out_off = (int64_t)bi * NC;
  if (live) {
    if (blocks) {
      const aomhip_txb b = blocks[bi];
      bx = b.x;
      by = b.y;
      tx_type = b.tx_type;
      out_off = b.out_offset;
    } else {  // regular grid: block bi of a plane that is grid_cols blocks wide
      bx = (bi % grid_cols) * W;
      by = (bi / grid_cols) * H;
    }
  }
  const int vk = v_kind(tx_type), hk = h_kind(tx_type);
  int32_t(&t)[KH * LSTRIDE] = tile[slot];

  // ---- columns
  int32_t x[H];
  int amax = 0;  // largest residual magnitude of this lane's column; the block's decides between the fast and the exact butterfly
  if (live && lane < W) {
    const bool ud = (vk == 2);
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const int rr = ud ? H - 1 - r : r;
      int v;
      if constexpr (SRC == 0) {
        v = static_cast<const int16_t *>(in0)[(int64_t)(by + rr) * stride0 + bx + lane];
      } else if constexpr (SRC == 1) {
        v = (int)static_cast<const uint8_t *>(in0)[(int64_t)(by + rr) * stride0 + bx + lane] -
            (int)static_cast<const uint8_t *>(in1)[(int64_t)(by + rr) * stride1 + bx + lane];
      } else {
        v = (int)static_cast<const uint16_t *>(in0)[(int64_t)(by + rr) * stride0 + bx + lane] -
            (int)static_cast<const uint16_t *>(in1)[(int64_t)(by + rr) * stride1 + bx + lane];
      }
      amax = max(amax, max(v, -v));
      x[r] = v * (1 << C::fs0);  // round_shift_array with a negative bit = exact left shift of an int16
    }
  }
  const bool fast = group_max<LPB>(amax) <= kSafeMax[tx_index_of(W, H)][tx_type & 15];
  if (live && lane < W) {
    fwd_1d_sel<H, C::cos_bit_col>(x, vk == 2 ? 1 : vk, fast);
    const int dc = (hk == 2) ? W - 1 - lane : lane;
#pragma unroll
    for (int r = 0; r < KH; ++r) {
      int32_t v = x[r];
      if constexpr (C::fs1 < 0) v = rshift(v, -C::fs1);
      t[r * LSTRIDE + dc] = v;
    }
  }
  block_sync();

  // ---- rows + quantise
  int my_eob = 0;
  int64_t berr = 0, bssz = 0;
  if (live && lane < KH) {
    const int r = lane;
    int32_t y[W];
#pragma unroll
    for (int c = 0; c < W; ++c) y[c] = t[r * LSTRIDE + c];
    fwd_1d_sel<W, C::cos_bit_row>(y, hk == 2 ? 1 : hk, fast);
    const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
    const int zb[2] = { (qa.zbin[0] + ((1 << LS) >> 1)) >> LS, (qa.zbin[1] + ((1 << LS) >> 1)) >> LS };
    const int rd[2] = { (qa.round[0] + ((1 << LS) >> 1)) >> LS, (qa.round[1] + ((1 << LS) >> 1)) >> LS };
    int last_c = -1;  // the inverse scan position grows with c along a row: only the last non-zero c matters
#pragma unroll
    for (int c = 0; c < KW; ++c) {
      int32_t v = y[c];
      if constexpr (C::fs2 < 0) v = rshift(v, -C::fs2);
      if constexpr (C::rect2) v = rshift64((int64_t)v * kSqrt2, kSqrt2Bits);
      const int rc = c * KH + r;
      if (coeff) coeff[out_off + rc] = v;
      const int ac = (c == 0) ? (r != 0) : 1;  // DC is (r, c) == (0, 0)
      int32_t qv, dqv;
      quantize_one<HBD, LS>(v, zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.qs_log2[ac], qa.dequant[ac], &qv,
                            &dqv);
      last_c = qv ? c : last_c;
      xq_store1(qcoeff + out_off + rc, qv);
      xq_store1(dqcoeff + out_off + rc, dqv);
      if (AOMHIP_XQ_EXTRAS && err_out) block_err_acc(v, dqv, err_shift, berr, bssz);
    }
    if (last_c >= 0) my_eob = iscan_pos<KW, KH>(r, last_c, scan_class) + 1;
  }
  my_eob = group_max<LPB>(my_eob);
  if (live && lane == 0) eob[bi] = (uint16_t)my_eob;
  if (AOMHIP_XQ_EXTRAS && err_out) {
    berr = group_sum64<LPB>(berr);
    bssz = group_sum64<LPB>(bssz);
    if (live && lane == 0) block_err_store(err_out, bi, berr, bssz, err_shift);
  }
}


// ---------------------------------------------------------------------------------------------
// Staged variant: identical arithmetic, but every global access is 16 bytes wide.
//   1. the block's residual rows are fetched with one 16-byte load per 8 samples into LDS region A
//   2. column pass reads its column from A, writes the transpose tile (region B)
//   3. row pass reads its row from B; the quantised levels go back to LDS in the reference's
//      coefficient order (qcoeff -> A, dqcoeff -> B), and are copied out with dwordx4 stores.
// For 16x16 this is 2 loads + 8 stores per lane instead of 16 + 32.
template <int W, int H, bool HBD, int SRC>
__global__ __launch_bounds__(kXqThreads) void xform_quant_staged_kernel(
    const void *__restrict__ in0, const void *__restrict__ in1, int stride0, int stride1,
    const aomhip_txb *__restrict__ blocks, int n_blocks, int grid_cols, int uniform_type, QuantArgs qa,
    int32_t *__restrict__ coeff, int32_t *__restrict__ qcoeff, int32_t *__restrict__ dqcoeff,
    uint16_t *__restrict__ eob, int nblk8, int64_t *__restrict__ err_out, int err_shift) {
  using C = Cfg2D<W, H>;
  constexpr int LPB = W > H ? W : H;
  constexpr int BPW = kXqThreads / LPB;
  constexpr int KW = W < 32 ? W : 32, KH = H < 32 ? H : 32;
  constexpr int NC = KW * KH;
  constexpr int LS = (W * H > 256) + (W * H > 1024);
  constexpr int LSTRIDE = W + 1;
  constexpr int A_WORDS = (W * H / 2 > NC ? W * H / 2 : NC);  // int16 input tile, later qcoeff staging
  constexpr int B_WORDS = (KH * LSTRIDE + 3) & ~3;            // transpose tile, later dqcoeff staging
  __shared__ __attribute__((aligned(16))) int32_t lds[BPW][A_WORDS + B_WORDS];

  const int slot = threadIdx.x / LPB, lane = threadIdx.x % LPB;
  const unsigned wg = xcd_chunked_index(blockIdx.x, nblk8);
  const int bi = wg * BPW + slot;
  const bool live = bi < n_blocks;
  int bx = 0, by = 0, tx_type = uniform_type;
  int64_t out_off = (int64_t)bi * NC;
  if (live) {
    if (blocks) {
      const aomhip_txb b = blocks[bi];
      bx = b.x;
      by = b.y;
      tx_type = b.tx_type;
      out_off = b.out_offset;
    } else {
      bx = (bi % grid_cols) * W;
      by = (bi / grid_cols) * H;
    }
  }
  const int vk = v_kind(tx_type), hk = h_kind(tx_type);
  int32_t *A = lds[slot];
  int32_t *B = lds[slot] + A_WORDS;
  int16_t *A16 = reinterpret_cast<int16_t *>(A);

  // ---- 1. residual rows -> LDS (8 samples per access; 4 for 4-wide blocks)
  constexpr int CS = W < 8 ? W : 8;             // samples per chunk
  constexpr int CPR = W / CS;                   // chunks per row
  constexpr int CHUNKS = H * CPR;
  int amax = 0;  // largest residual magnitude this lane staged (see xform_quant_kernel)
  if (live) {
#pragma unroll
    for (int k = 0; k < (CHUNKS + LPB - 1) / LPB; ++k) {
      const int i = lane + k * LPB;
      if (i < CHUNKS) {
        const int row = i / CPR, col = (i % CPR) * CS;
        int16_t v[CS];
        if constexpr (SRC == 0) {
          const int16_t *p = static_cast<const int16_t *>(in0) + (int64_t)(by + row) * stride0 + bx + col;
          if constexpr (CS == 8) {
            const VecU128 raw = *reinterpret_cast<const VecU128 *>(p);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (int16_t)(raw.v[j / 2] >> (16 * (j % 2)));
          } else {
            const VecU64 raw = *reinterpret_cast<const VecU64 *>(p);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (int16_t)(raw.v[j / 2] >> (16 * (j % 2)));
          }
        } else if constexpr (SRC == 1) {
          const uint8_t *ps = static_cast<const uint8_t *>(in0) + (int64_t)(by + row) * stride0 + bx + col;
          const uint8_t *pp = static_cast<const uint8_t *>(in1) + (int64_t)(by + row) * stride1 + bx + col;
          uint32_t a[CS / 4], b[CS / 4];
          if constexpr (CS == 8) {
            const VecU64 ra = *reinterpret_cast<const VecU64 *>(ps), rb = *reinterpret_cast<const VecU64 *>(pp);
            a[0] = ra.v[0]; a[1] = ra.v[1]; b[0] = rb.v[0]; b[1] = rb.v[1];
          } else {
            a[0] = reinterpret_cast<const VecU32 *>(ps)->v[0];
            b[0] = reinterpret_cast<const VecU32 *>(pp)->v[0];
          }
#pragma unroll
          for (int j = 0; j < CS; ++j)
            v[j] = (int16_t)((int)((a[j / 4] >> (8 * (j % 4))) & 0xFF) - (int)((b[j / 4] >> (8 * (j % 4))) & 0xFF));
        } else {
          const uint16_t *ps = static_cast<const uint16_t *>(in0) + (int64_t)(by + row) * stride0 + bx + col;
          const uint16_t *pp = static_cast<const uint16_t *>(in1) + (int64_t)(by + row) * stride1 + bx + col;
          uint32_t a[CS / 2], b[CS / 2];
          if constexpr (CS == 8) {
            const VecU128 ra = *reinterpret_cast<const VecU128 *>(ps), rb = *reinterpret_cast<const VecU128 *>(pp);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = ra.v[j]; b[j] = rb.v[j]; }
          } else {
            const VecU64 ra = *reinterpret_cast<const VecU64 *>(ps), rb = *reinterpret_cast<const VecU64 *>(pp);
            a[0] = ra.v[0]; a[1] = ra.v[1]; b[0] = rb.v[0]; b[1] = rb.v[1];
          }
#pragma unroll
          for (int j = 0; j < CS; ++j)
            v[j] = (int16_t)((int)((a[j / 2] >> (16 * (j % 2))) & 0xFFFF) - (int)((b[j / 2] >> (16 * (j % 2))) & 0xFFFF));
        }
#pragma unroll
        for (int j = 0; j < CS; ++j) amax = max(amax, max((int)v[j], -(int)v[j]));
#pragma unroll
        for (int j = 0; j < CS; j += 2)
          *reinterpret_cast<uint32_t *>(&A16[row * W + col + j]) = (uint16_t)v[j] | ((uint32_t)(uint16_t)v[j + 1] << 16);
      }
    }
  }
  const bool fast = group_max<LPB>(amax) <= kSafeMax[tx_index_of(W, H)][tx_type & 15];
  block_sync();

  // ---- 2. columns
  if (live && lane < W) {
    int32_t x[H];
    const bool ud = (vk == 2);
#pragma unroll
    for (int r = 0; r < H; ++r) x[r] = (int)A16[(ud ? H - 1 - r : r) * W + lane] * (1 << C::fs0);
    fwd_1d_sel<H, C::cos_bit_col>(x, vk == 2 ? 1 : vk, fast);
    const int dc = (hk == 2) ? W - 1 - lane : lane;
#pragma unroll
    for (int r = 0; r < KH; ++r) {
      int32_t v = x[r];
      if constexpr (C::fs1 < 0) v = rshift(v, -C::fs1);
      B[r * LSTRIDE + dc] = v;
    }
  }
  block_sync();

  // ---- 3. rows + quantise
  int my_eob = 0;
  int32_t y[W];
  const int r = lane;
  int64_t berr = 0, bssz = 0;
  if (live && lane < KH) {
#pragma unroll
    for (int c = 0; c < W; ++c) y[c] = B[r * LSTRIDE + c];
  }
  block_sync();  // A (input) and B (tile) are dead from here: they become the output staging areas
  if (live && lane < KH) {
    fwd_1d_sel<W, C::cos_bit_row>(y, hk == 2 ? 1 : hk, fast);
    const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
    const int zb[2] = { (qa.zbin[0] + ((1 << LS) >> 1)) >> LS, (qa.zbin[1] + ((1 << LS) >> 1)) >> LS };
    const int rd[2] = { (qa.round[0] + ((1 << LS) >> 1)) >> LS, (qa.round[1] + ((1 << LS) >> 1)) >> LS };
    int last_c = -1;
#pragma unroll
    for (int c = 0; c < KW; ++c) {
      int32_t v = y[c];
      if constexpr (C::fs2 < 0) v = rshift(v, -C::fs2);
      if constexpr (C::rect2) v = rshift64((int64_t)v * kSqrt2, kSqrt2Bits);
      const int rc = c * KH + r;
      if (coeff) coeff[out_off + rc] = v;
      const int ac = (c == 0) ? (r != 0) : 1;
      int32_t qv, dqv;
      quantize_one<HBD, LS>(v, zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.qs_log2[ac], qa.dequant[ac], &qv,
                            &dqv);
      last_c = qv ? c : last_c;
      A[rc] = qv;
      B[rc] = dqv;
      if (AOMHIP_XQ_EXTRAS && err_out) block_err_acc(v, dqv, err_shift, berr, bssz);
    }
    if (last_c >= 0) my_eob = iscan_pos<KW, KH>(r, last_c, scan_class) + 1;
  }
  my_eob = group_max<LPB>(my_eob);
  if (live && lane == 0) eob[bi] = (uint16_t)my_eob;
  if (AOMHIP_XQ_EXTRAS && err_out) {
    berr = group_sum64<LPB>(berr);
    bssz = group_sum64<LPB>(bssz);
    if (live && lane == 0) block_err_store(err_out, bi, berr, bssz, err_shift);
  }
  block_sync();

  // ---- 4. copy out, 16 bytes per lane per store
  if (live) {
#pragma unroll
    for (int k = 0; k < (NC / 4 + LPB - 1) / LPB; ++k) {
      const int i = lane + k * LPB;
      if (i < NC / 4) {
        {
          const uint4 va = *reinterpret_cast<const uint4 *>(A + 4 * i), vb = *reinterpret_cast<const uint4 *>(B + 4 * i);
          xq_store4(qcoeff + out_off + 4 * i, va.x, va.y, va.z, va.w);
          xq_store4(dqcoeff + out_off + 4 * i, vb.x, vb.y, vb.z, vb.w);
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Whole-block-per-lane variant for the smallest transforms (W, H <= 8): a lane keeps the entire block in
// VGPRs, runs the column and row networks back to back (the transpose is register renaming), quantises and
// writes its coefficients as contiguous dwordx4 runs.  No LDS, no barriers, inverse-scan positions are
// compile-time constants.  For 4x4 this does ~3x fewer VALU instructions per block than 4 lanes per block.
// WHT = true: the lossless instantiation (av1_fwht4x4), chosen per launch by uniform_tx_type == AOMHIP_TX_WHT -- a
// per-block test costs the hot 4x4 path 7 % (it keeps the unshifted residual alive).
template <int W, int H, bool HBD, int SRC, bool WHT = false>
__global__ __launch_bounds__(kXqThreads) void xform_quant_lane_kernel(
    const void *__restrict__ in0, const void *__restrict__ in1, int stride0, int stride1,
    const aomhip_txb *__restrict__ blocks, int n_blocks, int grid_cols, int uniform_type, QuantArgs qa,
    int32_t *__restrict__ coeff, int32_t *__restrict__ qcoeff, int32_t *__restrict__ dqcoeff,
    uint16_t *__restrict__ eob, int nblk8, int64_t *__restrict__ err_out, int err_shift) {
  using C = Cfg2D<W, H>;
  constexpr int NC = W * H;
  constexpr int LS = 0;  // <= 64 samples: av1_get_tx_scale == 0
  const unsigned wg = xcd_chunked_index(blockIdx.x, nblk8);
  // 4x4: the stores below exchange data between the four lanes of a quad, so lanes past the end of the list stay alive
  // (they redo the last block and store nothing)
  constexpr bool kQuadStores = W == 4 && H == 4;
  // 4x4: lane l of a wavefront takes block (l & 3) * 16 + (l >> 2) of the wavefront's 64, so that a quad holds blocks q, q + 16,
  // q + 32, q + 48 and store k of the transposed quads (below) writes blocks 16 k .. 16 k + 15 = ONE contiguous KB per store
  // instruction.  (With lane = block, store k wrote every fourth block: isolated 64-byte runs, which the non-temporal path turned
  // into 1.4 x the write traffic -- PMC, profiles/r02_txq.md.)  The row loads then read 16 blocks' 8 bytes = one whole 128-byte line
  // per group of 16 lanes.
  const int t_in_wg = kQuadStores ? (int)((threadIdx.x & ~63u) + ((threadIdx.x & 3u) << 4) + ((threadIdx.x & 63u) >> 2)) : (int)threadIdx.x;
  const int bi_raw = wg * kXqThreads + t_in_wg;
  const bool valid = bi_raw < n_blocks;
  if (!kQuadStores && !valid) return;
  const int bi = valid ? bi_raw : n_blocks - 1;
  int bx, by, tx_type = uniform_type;
  int64_t out_off = (int64_t)bi * NC;
  if (blocks) {
    const aomhip_txb b = blocks[bi];
    bx = b.x; by = b.y; tx_type = b.tx_type; out_off = b.out_offset;
  } else {
    bx = (bi % grid_cols) * W;
    by = (bi / grid_cols) * H;
  }
  const int vk = v_kind(tx_type), hk = h_kind(tx_type);
  const bool ud = (vk == 2), lr = (hk == 2);

  // ---- load rows (one wide access per row), apply the up/down flip while loading
  int32_t x[H][W];
  int32_t x0[H][W];  // the raw residual (the lossless WHT takes it unshifted)
  int amax = 0;
#pragma unroll
  for (int r = 0; r < H; ++r) {
    const int rr = ud ? H - 1 - r : r;
    int v[W];
    if constexpr (SRC == 0) {
      const int16_t *p = static_cast<const int16_t *>(in0) + (int64_t)(by + rr) * stride0 + bx;
      if constexpr (W == 8) {
        const VecU128 raw = *reinterpret_cast<const VecU128 *>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (int16_t)(raw.v[j / 2] >> (16 * (j % 2)));
      } else {
        const VecU64 raw = *reinterpret_cast<const VecU64 *>(p);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (int16_t)(raw.v[j / 2] >> (16 * (j % 2)));
      }
    } else if constexpr (SRC == 1) {
      const uint8_t *ps = static_cast<const uint8_t *>(in0) + (int64_t)(by + rr) * stride0 + bx;
      const uint8_t *pp = static_cast<const uint8_t *>(in1) + (int64_t)(by + rr) * stride1 + bx;
#pragma unroll
      for (int j = 0; j < W; ++j) v[j] = (int)ps[j] - (int)pp[j];
    } else {
      const uint16_t *ps = static_cast<const uint16_t *>(in0) + (int64_t)(by + rr) * stride0 + bx;
      const uint16_t *pp = static_cast<const uint16_t *>(in1) + (int64_t)(by + rr) * stride1 + bx;
#pragma unroll
      for (int j = 0; j < W; ++j) v[j] = (int)ps[j] - (int)pp[j];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      x0[r][j] = v[j];
      x[r][j] = v[j] * (1 << C::fs0);
      amax = max(amax, max(v[j], -v[j]));
    }
  }
  const bool fast = amax <= kSafeMax[tx_index_of(W, H)][tx_type & 15];  // (a lane owns its whole block here)

  // ---- columns
#pragma unroll
  for (int c = 0; c < W; ++c) {
    int32_t col[H];
#pragma unroll
    for (int r = 0; r < H; ++r) col[r] = x[r][c];
    fwd_1d_sel<H, C::cos_bit_col>(col, vk == 2 ? 1 : vk, fast);
#pragma unroll
    for (int r = 0; r < H; ++r) x[r][c] = (C::fs1 < 0) ? rshift(col[r], C::fs1 < 0 ? -C::fs1 : 1) : col[r];
  }

  // ---- rows + quantise
  constexpr bool wht = WHT && W == 4 && H == 4;  // lossless: DCT_DCT scan (the flag, not the type, selects WHT)
  if (wht) tx_type = 0;
  const int scan_class = (wht || tx_type < 10) ? 0 : ((tx_type & 1) ? 2 : 1);
  const int zb[2] = { qa.zbin[0], qa.zbin[1] }, rd[2] = { qa.round[0], qa.round[1] };
  int my_eob = 0;
  int64_t berr = 0, bssz = 0;
  int32_t qv[H][W], dv[H][W];
  auto emit = [&](int r, int c, int32_t v) {  // coefficient (r, c) = index c * H + r of the reference's output
    if (coeff && valid) coeff[out_off + c * H + r] = v;
    const int ac = (r | c) != 0;
    quantize_one<HBD, LS>(v, zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.qs_log2[ac], qa.dequant[ac], &qv[r][c],
                          &dv[r][c]);
    // inverse-scan positions are constants here (r, c are unrolled)
    const int p0 = iscan_pos<W, H>(r, c, 0) + 1, p1 = iscan_pos<W, H>(r, c, 1) + 1, p2 = iscan_pos<W, H>(r, c, 2) + 1;
    const int p = scan_class == 0 ? p0 : (scan_class == 1 ? p1 : p2);
    my_eob = (qv[r][c] != 0 && p > my_eob) ? p : my_eob;
    if (AOMHIP_XQ_EXTRAS && err_out) block_err_acc(v, dv[r][c], err_shift, berr, bssz);
  };
  if constexpr (wht) {
    {  // av1_fwht4x4_c (hybrid_fwd_txfm.c:24-76) on the raw residual, UNIT_QUANT_FACTOR = 4
      auto bf = [](int &a, int &b, int &c, int &d) {
        int a1 = a + b, d1 = d - c;
        const int e1 = (a1 - d1) >> 1;
        const int b1 = e1 - b, c1 = e1 - c;
        a1 -= c1;
        d1 += b1;
        a = a1; b = b1; c = c1; d = d1;
      };
      int t[16], raw[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) raw[r][c] = x0[r][c];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int a = raw[0][i], b = raw[1][i], c = raw[2][i], d = raw[3][i];
        bf(a, b, c, d);
        t[4 * i + 0] = a; t[4 * i + 1] = c; t[4 * i + 2] = d; t[4 * i + 3] = b;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int a = t[i], b = t[4 + i], c = t[8 + i], d = t[12 + i];
        bf(a, b, c, d);
        // output index 4 * k + i = c * H + r with c = k, r = i
        emit(i, 0, a * 4); emit(i, 1, c * 4); emit(i, 2, d * 4); emit(i, 3, b * 4);
      }
    }
  }
  if constexpr (!wht) {
#pragma unroll
    for (int r = 0; r < H; ++r) {
      int32_t y[W];
#pragma unroll
      for (int c = 0; c < W; ++c) y[c] = x[r][lr ? W - 1 - c : c];  // left/right flip of the column-pass output
      fwd_1d_sel<W, C::cos_bit_row>(y, hk == 2 ? 1 : hk, fast);
#pragma unroll
      for (int c = 0; c < W; ++c) {
        int32_t v = y[c];
        if constexpr (C::fs2 < 0) v = rshift(v, C::fs2 < 0 ? -C::fs2 : 1);
        if constexpr (C::rect2) v = rshift64((int64_t)v * kSqrt2, kSqrt2Bits);
        emit(r, c, v);
      }
    }
  }
  if (valid) eob[bi] = (uint16_t)my_eob;
  if (AOMHIP_XQ_EXTRAS && err_out && valid) block_err_store(err_out, bi, berr, bssz, err_shift);
  if constexpr (kQuadStores) {
    // A lane's block is 64 contiguous output bytes per array, but written as four 16-byte stores the wavefront's store
    // instruction touches 64 different 64-byte chunks with a quarter of each (4x the L2 write requests of a full-chunk
    // store: PMC put this kernel at 3.4 TB/s against 4.5-4.7 for the sizes that store whole runs).  So the quad transposes
    // first: with M[k][c] = column c of the block of quad lane k, two exchange stages (lane ^ 1, lane ^ 2) leave lane j
    // with M[0..3][j], and store k then has the four lanes of a quad writing the four columns of ONE block = one 64-byte run.
    const int qj = (int)(threadIdx.x & 3);
    auto dpp_x1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); };  // quad_perm [1,0,3,2]
    auto dpp_x2 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); };  // quad_perm [2,3,0,1]
    auto transpose = [&](uint32_t (&m)[4][4]) {  // m[c][r]: column c (4 dwords) of this lane's block -> m[k][r]: this lane's column of lane k's block
#pragma unroll
      for (int c = 0; c < 4; c += 2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t send = (qj & 1) ? m[c][r] : m[c + 1][r];
          const uint32_t recv = dpp_x1(send);
          if (qj & 1) m[c][r] = recv; else m[c + 1][r] = recv;
        }
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t send = (qj & 2) ? m[c][r] : m[c + 2][r];
          const uint32_t recv = dpp_x2(send);
          if (qj & 2) m[c][r] = recv; else m[c + 2][r] = recv;
        }
    };
    uint32_t mq[4][4], md[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { mq[c][r] = (uint32_t)qv[r][c]; md[c][r] = (uint32_t)dv[r][c]; }
    transpose(mq);
    transpose(md);
    const uint32_t off_lo = (uint32_t)out_off, off_hi = (uint32_t)((uint64_t)out_off >> 32), vflag = valid ? 1u : 0u;
    auto quad_bcast = [](uint32_t v, int k) {  // quad_perm [k,k,k,k]
      switch (k) {
        case 0: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xf, 0xf, false);
        case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xf, 0xf, false);
        case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xf, 0xf, false);
        default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xf, 0xf, false);
      }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t lo = quad_bcast(off_lo, k), hi = quad_bcast(off_hi, k), ok = quad_bcast(vflag, k);
      const int64_t o = (int64_t)(((uint64_t)hi << 32) | lo) + qj * 4;
      if (ok) {
        xq_store4(qcoeff + o, mq[k][0], mq[k][1], mq[k][2], mq[k][3]);
        xq_store4(dqcoeff + o, md[k][0], md[k][1], md[k][2], md[k][3]);
      }
    }
    return;
  }
  // ---- store in the reference's transposed order (index c*H + r): column c is H contiguous values
#pragma unroll
  for (int c = 0; c < W; ++c) {
#pragma unroll
    for (int r4 = 0; r4 < H; r4 += 4) {
      *reinterpret_cast<uint4 *>(qcoeff + out_off + c * H + r4) =
          make_uint4((uint32_t)qv[r4][c], (uint32_t)qv[r4 + 1][c], (uint32_t)qv[r4 + 2][c], (uint32_t)qv[r4 + 3][c]);
      *reinterpret_cast<uint4 *>(dqcoeff + out_off + c * H + r4) =
          make_uint4((uint32_t)dv[r4][c], (uint32_t)dv[r4 + 1][c], (uint32_t)dv[r4 + 2][c], (uint32_t)dv[r4 + 3][c]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Adaptive quantiser: aom_quantize_b_adaptive_helper_c / aom_highbd_quantize_b_adaptive_helper_c
// (aom_dsp/quantize.c:16-105,173-258; selected by qparam->use_quant_b_adapt, av1_quantize.c:309-341,453-) on
// already materialised transform coefficients.  The three sequential passes of the reference are reductions:
//   non_zero_count = 1 + last scan position whose coefficient lies outside the dead zone widened by
//                    ROUND_POWER_OF_TWO(dequant * EOB_FACTOR(325), 7)          (backward pre-scan, :36-46)
//   eob / first    = last / first scan position below non_zero_count with a non-zero level
//   if first == eob and that level is +-1 and the coefficient lies inside the zone widened by
//                    dequant * (325 + SKIP_EOB_FACTOR_ADJUST(200)) / 128: drop it, eob = 0   (:84-102)
// One wavefront per block; lane l owns coefficients l, l + 64, ... of the (transposed) coefficient array.
// QMX: with quantisation matrices (qm / iqm per coefficient position, either may be NULL = flat): the helper's matrix branches -- the weight
// enters the pre-scan's test, the dead zone, the level and the single-coefficient rule, the inverse weight the dequantiser.
template <int KW, int KH, bool HBD, int LS, bool QMX = false>
__global__ __launch_bounds__(256) void quant_adaptive_kernel(const int32_t *__restrict__ coeff,
                                                             const aomhip_txb *__restrict__ blocks, int n_blocks,
                                                             int uniform_type, QuantArgs qa, int32_t *__restrict__ qcoeff,
                                                             int32_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob,
                                                             const uint8_t *__restrict__ qm = nullptr, const uint8_t *__restrict__ iqm = nullptr) {
  constexpr int NC = KW * KH;
  constexpr int PER = (NC + 63) / 64;
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (bi >= n_blocks) return;
  const int tx_type = blocks ? blocks[bi].tx_type : uniform_type;
  const int64_t off = blocks ? (int64_t)blocks[bi].out_offset : (int64_t)bi * NC;
  const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
  const int zb[2] = { (qa.zbin[0] + ((1 << LS) >> 1)) >> LS, (qa.zbin[1] + ((1 << LS) >> 1)) >> LS };
  const int rd[2] = { (qa.round[0] + ((1 << LS) >> 1)) >> LS, (qa.round[1] + ((1 << LS) >> 1)) >> LS };
  const int add1[2] = { (qa.dequant[0] * 325 + 64) >> 7, (qa.dequant[1] * 325 + 64) >> 7 };
  const int add2[2] = { (qa.dequant[0] * 525 + 64) >> 7, (qa.dequant[1] * 525 + 64) >> 7 };
  int32_t v[PER];
  int pos[PER];
  [[maybe_unused]] int wt[PER];
  int nzc = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int rc = lane + 64 * k;
    v[k] = 0;
    pos[k] = -1;
    if constexpr (QMX) wt[k] = 32;
    if (rc < NC) {
      v[k] = coeff[off + rc];
      const int c = rc / KH, r = rc % KH;  // transposed layout: rc = c * KH + r
      pos[k] = iscan_pos<KW, KH>(r, c, scan_class);
      const int ac = rc != 0;
      if constexpr (QMX) wt[k] = qm ? (int)qm[rc] : 32;
      const int64_t cw = (int64_t)v[k] * (QMX ? wt[k] : 32);  // coeff * wt (the reference forms it in int; |coeff| < 2^26 here)
      const bool inside = cw < (int64_t)zb[ac] * 32 + add1[ac] && cw > -(int64_t)zb[ac] * 32 - add1[ac];
      if (!inside) nzc = max(nzc, pos[k] + 1);
    }
  }
  nzc = group_max<64>(nzc);
  int32_t qv[PER], dv[PER];
  int last = 0, first = NC;  // last = eob (position + 1), first = position
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    qv[k] = dv[k] = 0;
    if (pos[k] >= 0 && pos[k] < nzc) {
      const int ac = (lane + 64 * k) != 0;
      if constexpr (QMX)
        quantize_one_qm<HBD, LS>(v[k], zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.dequant[ac], wt[k], iqm ? (int)iqm[lane + 64 * k] : 32, &qv[k], &dv[k]);
      else
        quantize_one<HBD, LS>(v[k], zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.qs_log2[ac], qa.dequant[ac], &qv[k],
                              &dv[k]);
      if (qv[k]) {
        last = max(last, pos[k] + 1);
        first = min(first, pos[k]);
      }
    }
  }
  last = group_max<64>(last);
  first = -group_max<64>(-first);
  if (last > 0 && first == last - 1) {  // exactly one non-zero level
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (pos[k] == first && (qv[k] == 1 || qv[k] == -1)) {
        const int ac = (lane + 64 * k) != 0;
        const int64_t cw = (int64_t)v[k] * (QMX ? wt[k] : 32);
        if (cw < (int64_t)zb[ac] * 32 + add2[ac] && cw > -(int64_t)zb[ac] * 32 - add2[ac]) qv[k] = dv[k] = 0;
      }
    }
    // did the owner drop it?  (one more vote: the eob must follow)
    int still = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) still |= (qv[k] != 0);
    still = group_max<64>(still);
    if (!still) last = 0;
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int rc = lane + 64 * k;
    if (rc < NC) {
      qcoeff[off + rc] = qv[k];
      dqcoeff[off + rc] = dv[k];
    }
  }
  if (lane == 0) eob[bi] = (uint16_t)last;
}

// Quantisation with matrices (qm_ptr / iqm_ptr non-NULL: aom_[highbd_]quantize_b_helper_c, aom_dsp/quantize.c:108-169,261-316) on already
// materialised transform coefficients, like the adaptive form above: one wavefront per block, lane l owns coefficients l, l + 64, ... of the
// (transposed) coefficient array, the matrices are indexed by the same position.  eob = 1 + the last scan position with a non-zero level.
// FP: the `fp` flavour (quantize_one_qm_fp; qa.round / qa.quant carry round_fp / quant_fp, zbin and quant_shift are not read).
template <int KW, int KH, bool HBD, int LS, bool FP = false>
__global__ __launch_bounds__(256) void quant_qm_kernel(const int32_t *__restrict__ coeff, const aomhip_txb *__restrict__ blocks, int n_blocks,
                                                       int uniform_type, QuantArgs qa, const uint8_t *__restrict__ qm,
                                                       const uint8_t *__restrict__ iqm, int32_t *__restrict__ qcoeff,
                                                       int32_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob) {
  constexpr int NC = KW * KH;
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (bi >= n_blocks) return;
  const int tx_type = blocks ? blocks[bi].tx_type : uniform_type;
  const int64_t off = blocks ? (int64_t)blocks[bi].out_offset : (int64_t)bi * NC;
  const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
  const int zb[2] = { (qa.zbin[0] + ((1 << LS) >> 1)) >> LS, (qa.zbin[1] + ((1 << LS) >> 1)) >> LS };
  const int rd[2] = { (qa.round[0] + ((1 << LS) >> 1)) >> LS, (qa.round[1] + ((1 << LS) >> 1)) >> LS };
  int last = 0;
  for (int rc = lane; rc < NC; rc += 64) {
    const int ac = rc != 0;
    int32_t qv, dv;
    if constexpr (FP)
      quantize_one_qm_fp<HBD, LS>(coeff[off + rc], rd[ac], qa.quant[ac], qa.dequant[ac], qm ? (int)qm[rc] : 32, iqm ? (int)iqm[rc] : 32, &qv, &dv);
    else
      quantize_one_qm<HBD, LS>(coeff[off + rc], zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.dequant[ac], qm ? (int)qm[rc] : 32,
                               iqm ? (int)iqm[rc] : 32, &qv, &dv);
    qcoeff[off + rc] = qv;
    dqcoeff[off + rc] = dv;
    if (qv) last = max(last, iscan_pos<KW, KH>(rc % KH, rc / KH, scan_class) + 1);   // transposed layout: rc = c * KH + r
  }
  last = group_max<64>(last);
  if (lane == 0) eob[bi] = (uint16_t)last;
}

// The low-precision quantiser of the non-RD mode search (av1_quantize_lp_c, av1/encoder/av1_quantize.c:212-240) with the transform-domain
// distortion the same caller takes next (av1_block_error_lp_c, av1/encoder/rdopt.c:650-660) on int16 coefficients: one wavefront per block as
// above.  |c| + round saturates at INT16_MAX, the level is (that * quant) >> 16, and dqcoeff is the product's low 16 bits (the reference
// stores it through an int16_t).  err (optional): sum (coeff - dqcoeff)^2, each square in 32-bit wrap-around arithmetic like the compiled `int`.
template <int KW, int KH>
__global__ __launch_bounds__(256) void quant_lp_kernel(const int16_t *__restrict__ coeff, const aomhip_txb *__restrict__ blocks, int n_blocks,
                                                       int uniform_type, QuantArgs qa, int16_t *__restrict__ qcoeff, int16_t *__restrict__ dqcoeff,
                                                       uint16_t *__restrict__ eob, int64_t *__restrict__ err) {
  constexpr int NC = KW * KH;
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (bi >= n_blocks) return;
  const int tx_type = blocks ? blocks[bi].tx_type : uniform_type;
  const int64_t off = blocks ? (int64_t)blocks[bi].out_offset : (int64_t)bi * NC;
  const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
  int last = 0;
  int64_t e = 0;
  for (int rc = lane; rc < NC; rc += 64) {
    const int ac = rc != 0;
    const int c = coeff[off + rc], sign = c >> 31;
    int t = min((c ^ sign) - sign + (int)(int16_t)qa.round[ac], 32767);
    t = (t * (int)(int16_t)qa.quant[ac]) >> 16;
    const int16_t qv = (int16_t)((t ^ sign) - sign);
    const int16_t dv = (int16_t)(qv * (int)(int16_t)qa.dequant[ac]);
    qcoeff[off + rc] = qv;
    dqcoeff[off + rc] = dv;
    if (t) last = max(last, iscan_pos<KW, KH>(rc % KH, rc / KH, scan_class) + 1);
    const uint32_t d = (uint32_t)(c - dv);
    e += (int32_t)(d * d);
  }
  last = group_max<64>(last);
  if (lane == 0) eob[bi] = (uint16_t)last;
  if (err) {
    e = group_sum64<64>(e);
    if (lane == 0) err[bi] = e;
  }
}

// av1_get_nz_map_contexts_c (av1/encoder/encodetxb.c:222-267): the context of every coefficient before the end of block, from the magnitudes of its
// causal neighbours in the padded level map (av1_txb_init_levels' output: aomhip_txb_init_levels_batch) -- get_nz_mag / get_nz_map_ctx_from_stats
// (av1/common/txb_common.h:150-224).  One wavefront per block, a lane per coefficient position: the scan index comes from iscan_pos, positions at
// or past the end of block are not written (the reference's loop never reaches them); the 2-D position offsets by Nz_Map's rule
// (av1_nz_map_ctx_offset; `rel` = sign(tx_w - tx_h) of the transform's own size: TX_64X32 / TX_32X64 code 32 x 32 coefficients with the
// rectangular offsets).
template <int KW, int KH>
__global__ __launch_bounds__(256) void nz_map_contexts_kernel(const uint8_t *__restrict__ levels, int64_t levels_pitch, const aomhip_txb *__restrict__ blocks,
                                                              int n_blocks, int uniform_type, int rel, const uint16_t *__restrict__ eobs,
                                                              int8_t *__restrict__ contexts, int64_t contexts_pitch) {
  constexpr int NC = KW * KH, BHL = KH == 4 ? 2 : (KH == 8 ? 3 : (KH == 16 ? 4 : 5));
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (bi >= n_blocks) return;
  const int tx_type = blocks ? blocks[bi].tx_type : uniform_type;
  const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
  const int tx_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 1 : 2);   // tx_type_to_class: H_* -> TX_CLASS_HORIZ (1), V_* -> TX_CLASS_VERT (2)
  const int eob = eobs[bi];
  const uint8_t *lv0 = levels + (int64_t)bi * levels_pitch;
  int8_t *out = contexts + (int64_t)bi * contexts_pitch;
  for (int pos = lane; pos < NC; pos += 64) {
    const int col = pos >> BHL, row = pos & (KH - 1);
    const int i = iscan_pos<KW, KH>(row, col, scan_class);
    if (i >= eob) continue;
    int ctx;
    if (i == eob - 1) {
      ctx = i == 0 ? 0 : (i <= NC / 8 ? 1 : (i <= NC / 4 ? 2 : 3));
    } else {
      const uint8_t *lv = lv0 + pos + (col << 2);   // get_padded_idx: TX_PAD_HOR = 4 entries after every column
      auto c3 = [](uint8_t v) { return min((int)v, 3); };
      int mag = c3(lv[KH + 4]) + c3(lv[1]);
      if (tx_class == 0) mag += c3(lv[KH + 4 + 1]) + c3(lv[2 * KH + 8]) + c3(lv[2]);
      else if (tx_class == 2) mag += c3(lv[2]) + c3(lv[3]) + c3(lv[4]);
      else mag += c3(lv[2 * KH + 8]) + c3(lv[3 * KH + 12]) + c3(lv[4 * KH + 16]);
      if ((tx_class | pos) == 0) {
        ctx = 0;
      } else {
        ctx = min((mag + 1) >> 1, 4);
        if (tx_class == 0) ctx += (rel < 0 && row < 2) ? 11 : ((rel > 0 && col < 2) ? 16 : (row + col < 2 ? 1 : (row + col < 4 ? 6 : 21)));
        else {
          const int k = tx_class == 1 ? col : row;
          ctx += 26 + (k == 0 ? 0 : (k == 1 ? 5 : 10));   // SIG_COEF_CONTEXTS_2D + nz_map_ctx_offset_1d
        }
      }
    }
    out[pos] = (int8_t)ctx;
  }
}

// av1_cost_coeffs_txb (av1/encoder/txb_rdopt.c:450-544,603-622): the rate of a block's quantised coefficients under the level-map coder's cost tables,
// everything but get_tx_type_cost.  One wavefront per block, a lane per coefficient position: its scan index from iscan_pos, its level and its
// neighbours' (min(|q|, 127), 0 outside the block: what the padded level map holds) straight from the coefficients, the three kinds of term of the
// reference's loop -- last coefficient (base_eob_cost, get_br_ctx_eob), middle (base_cost on the nz-map context, the sign bit, get_br_ctx), first
// (base_cost, dc_sign_cost) -- chosen by the index, a wavefront sum, and the block's two scalar terms (txb_skip_cost, get_eob_cost) on lane 0.
// `costs`: LV_MAP_COEFF_COST's 944 ints in declaration order, then LV_MAP_EOB_COST.eob_cost[2][11].
// LAPLACIAN: av1_cost_coeffs_txb_laplacian with adjust_eob == 0 (:546-601,624-668) -- the same two scalar terms, per coefficient costLUT[min(|q|, 14)]
// (the last one (|q| - 1) << 11) and const_term + loge_par per position.
__device__ const int kTxbCostLut[15] = AOMHIP_TXB_COST_LUT;

template <int KW, int KH, bool LAPLACIAN = false>
__global__ __launch_bounds__(256) void cost_coeffs_txb_kernel(const int32_t *__restrict__ qcoeff, const aomhip_txb *__restrict__ blocks, int n_blocks,
                                                              int uniform_type, int rel, const uint16_t *__restrict__ eobs, const uint8_t *__restrict__ txb_ctx,
                                                              const int32_t *__restrict__ costs, int32_t *__restrict__ out) {
  constexpr int NC = KW * KH, BHL = KH == 4 ? 2 : (KH == 8 ? 3 : (KH == 16 ? 4 : 5));
  constexpr int kSkip = 0, kBaseEob = 26, kBase = 38, kEobExtra = 374, kDcSign = 392, kLps = 398, kEob = 944;
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (bi >= n_blocks) return;
  const int tx_type = blocks ? blocks[bi].tx_type : uniform_type;
  const int64_t off = blocks ? (int64_t)blocks[bi].out_offset : (int64_t)bi * NC;
  const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
  const int tx_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 1 : 2);
  const int eob = eobs[bi], skip_ctx = txb_ctx[2 * bi], dc_ctx = txb_ctx[2 * bi + 1];
  if (eob == 0) {
    if (lane == 0) out[bi] = costs[kSkip + skip_ctx * 2 + 1];
    return;
  }
  const int32_t *q = qcoeff + off;
  auto L = [&](int r, int c) { return (r < KH && c < KW) ? min(abs(q[c * KH + r]), 127) : 0; };   // av1_txb_init_levels' entry (row r of column c)
  auto br_cost = [&](int level, int ctx) {   // get_br_cost + get_golomb_cost (txb_rdopt_utils.h:85-97)
    int c = costs[kLps + ctx * 26 + min(level - 3, 12)];
    if (level >= 15) c += (2 * (32 - __builtin_clz(level - 14)) - 1) * 512;
    return c;
  };
  int acc = 0;
  for (int pos = lane; pos < NC; pos += 64) {
    const int col = pos >> BHL, row = pos & (KH - 1);
    const int i = iscan_pos<KW, KH>(row, col, scan_class);
    if (i >= eob) continue;
    const int v = q[pos], level = abs(v);
    if constexpr (LAPLACIAN) {
      acc += i == eob - 1 ? (level - 1) * 2048 : kTxbCostLut[min(level, 14)];
      continue;
    }
    if (i == eob - 1) {
      const int ctx = i == 0 ? 0 : (i <= NC / 8 ? 1 : (i <= NC / 4 ? 2 : 3));
      acc += costs[kBaseEob + ctx * 3 + min(level, 3) - 1];
      if (v) {
        if (level > 2) {
          const bool near = (tx_class == 0 && row < 2 && col < 2) || (tx_class == 1 && col == 0) || (tx_class == 2 && row == 0);
          acc += br_cost(level, pos == 0 ? 0 : (near ? 7 : 14));   // get_br_ctx_eob
        }
        acc += i ? 512 : costs[kDcSign + dc_ctx * 2 + (v < 0)];
      }
      continue;
    }
    // the nz-map context (get_nz_mag / get_nz_map_ctx_from_stats, clipped at 3) and, for levels above NUM_BASE_LEVELS, get_br_ctx (unclipped)
    const int l10 = L(row + 1, col), l01 = L(row, col + 1);
    int mag = min(l10, 3) + min(l01, 3), bmag = l10 + l01;
    if (tx_class == 0) {
      const int l11 = L(row + 1, col + 1);
      mag += min(l11, 3) + min(L(row, col + 2), 3) + min(L(row + 2, col), 3);
      bmag += l11;
    } else if (tx_class == 2) {
      const int l20 = L(row + 2, col);
      mag += min(l20, 3) + min(L(row + 3, col), 3) + min(L(row + 4, col), 3);
      bmag += l20;
    } else {
      const int l02 = L(row, col + 2);
      mag += min(l02, 3) + min(L(row, col + 3), 3) + min(L(row, col + 4), 3);
      bmag += l02;
    }
    int ctx = 0;
    if ((tx_class | pos) != 0) {
      ctx = min((mag + 1) >> 1, 4);
      if (tx_class == 0) ctx += (rel < 0 && row < 2) ? 11 : ((rel > 0 && col < 2) ? 16 : (row + col < 2 ? 1 : (row + col < 4 ? 6 : 21)));
      else {
        const int k = tx_class == 1 ? col : row;
        ctx += 26 + (k == 0 ? 0 : (k == 1 ? 5 : 10));
      }
    }
    acc += costs[kBase + ctx * 8 + min(level, 3)];
    if (v) {
      acc += i ? 512 : costs[kDcSign + dc_ctx * 2 + (v < 0)];
      if (level > 2) {
        int b = min((bmag + 1) >> 1, 6);
        if (pos != 0) {
          const bool near = (tx_class == 0 && row < 2 && col < 2) || (tx_class == 1 && col == 0) || (tx_class == 2 && row == 0);
          b += near ? 7 : 14;
        }
        acc += br_cost(level, b);
      }
    }
  }
  for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) {
    // get_eob_cost (txb_rdopt_utils.h:66-83) with av1_get_eob_pos_token: group t = the last one whose start (1, 2, 3, 5, 9, 17, ..) is <= eob
    const int t = eob < 3 ? eob : 33 - __builtin_clz(eob - 1);
    const int bits = t < 3 ? 0 : t - 2;   // av1_eob_offset_bits
    int c = costs[kSkip + skip_ctx * 2] + costs[kEob + (tx_class == 0 ? 0 : 11) + t - 1];
    if (bits > 0) {
      const int extra = eob - ((1 << (t - 2)) + 1);   // av1_eob_group_start[t] for t >= 3
      c += costs[kEobExtra + (t - 3) * 2 + ((extra >> (bits - 1)) & 1)] + (bits - 1) * 512;
    }
    out[bi] = acc + c + (LAPLACIAN ? (512 + 739) * (eob - 1) : 0);   // const_term + loge_par
  }
}

// av1_get_txb_entropy_context (av1/encoder/encodetxb.c:451-467): what a coded block leaves in the above / left entropy contexts -- min(sum of the levels
// before the end of block, COEFF_CONTEXT_MASK) with the DC coefficient's sign above it (set_dc_sign).  The reference's early exit only bounds its sum:
// a lane per position adds min(|q|, 8).
template <int KW, int KH>
__global__ __launch_bounds__(256) void txb_entropy_context_kernel(const int32_t *__restrict__ qcoeff, const aomhip_txb *__restrict__ blocks, int n_blocks,
                                                                  int uniform_type, const uint16_t *__restrict__ eobs, uint8_t *__restrict__ out) {
  constexpr int NC = KW * KH;
  const int lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (bi >= n_blocks) return;
  const int tx_type = blocks ? blocks[bi].tx_type : uniform_type;
  const int64_t off = blocks ? (int64_t)blocks[bi].out_offset : (int64_t)bi * NC;
  const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
  const int eob = eobs[bi];
  const int32_t *q = qcoeff + off;
  int acc = 0;
  for (int pos = lane; pos < NC && eob; pos += 64)
    if (iscan_pos<KW, KH>(pos % KH, pos / KH, scan_class) < eob) acc += min(abs(q[pos]), 8);
  for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) {
    int cul = 0;
    if (eob) {
      cul = min(acc, 7);
      const int dc = q[0];
      if (dc < 0) cul |= 1 << 3;          // COEFF_CONTEXT_BITS
      else if (dc > 0) cul += 2 << 3;
    }
    out[bi] = (uint8_t)cul;
  }
}

// aom_quantize_b* / aom_highbd_quantize_b* with the caller's own scan tables: what the rtcd-signature entry points
// (aomhip_quantize_b ...) run -- those signatures carry `scan` / `iscan` pointers and a coefficient count instead of a
// transform size and type.  The same quantize_one as the fused kernels; eob = 1 + max iscan[rc] over non-zero levels
// (SURVEY 8a': the reference's backward pre-scan only skips coefficients the dead-zone test zeroes anyway).  ADAPT: the
// adaptive form (quant_adaptive_kernel's three reductions) with positions from the table.  One wavefront per call.
template <bool HBD, int LS, bool ADAPT>
__global__ __launch_bounds__(64) void quant_table_kernel(const int32_t *__restrict__ coeff, int n, QuantArgs qa,
                                                         const int16_t *__restrict__ iscan, int32_t *__restrict__ qcoeff,
                                                         int32_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob) {
  const int lane = threadIdx.x;
  const int zb[2] = { (qa.zbin[0] + ((1 << LS) >> 1)) >> LS, (qa.zbin[1] + ((1 << LS) >> 1)) >> LS };
  const int rd[2] = { (qa.round[0] + ((1 << LS) >> 1)) >> LS, (qa.round[1] + ((1 << LS) >> 1)) >> LS };
  const int add1[2] = { (qa.dequant[0] * 325 + 64) >> 7, (qa.dequant[1] * 325 + 64) >> 7 };
  const int add2[2] = { (qa.dequant[0] * 525 + 64) >> 7, (qa.dequant[1] * 525 + 64) >> 7 };
  int nzc = n;
  if constexpr (ADAPT) {
    nzc = 0;
    for (int rc = lane; rc < n; rc += 64) {
      const int ac = rc != 0;
      const int64_t cw = (int64_t)coeff[rc] * 32;
      const bool inside = cw < (int64_t)zb[ac] * 32 + add1[ac] && cw > -(int64_t)zb[ac] * 32 - add1[ac];
      if (!inside) nzc = max(nzc, (int)iscan[rc] + 1);
    }
    nzc = group_max<64>(nzc);
  }
  int last = 0, first = n;
  for (int rc = lane; rc < n; rc += 64) {
    const int ac = rc != 0, pos = iscan[rc];
    int32_t qv = 0, dv = 0;
    if (pos < nzc)
      quantize_one<HBD, LS>(coeff[rc], zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.qs_log2[ac], qa.dequant[ac], &qv, &dv);
    qcoeff[rc] = qv;
    dqcoeff[rc] = dv;
    if (qv) {
      last = max(last, pos + 1);
      first = min(first, pos);
    }
  }
  last = group_max<64>(last);
  if constexpr (ADAPT) {
    first = -group_max<64>(-first);
    if (last > 0 && first == last - 1) {  // exactly one non-zero level: drop a lone +-1 inside the wider zone
      int dropped = 0;
      for (int rc = lane; rc < n; rc += 64) {
        if (iscan[rc] == first && (qcoeff[rc] == 1 || qcoeff[rc] == -1)) {
          const int ac = rc != 0;
          const int64_t cw = (int64_t)coeff[rc] * 32;
          if (cw < (int64_t)zb[ac] * 32 + add2[ac] && cw > -(int64_t)zb[ac] * 32 - add2[ac]) {
            qcoeff[rc] = dqcoeff[rc] = 0;
            dropped = 1;
          }
        }
      }
      if (group_max<64>(dropped)) last = 0;
    }
  }
  if (lane == 0) *eob = (uint16_t)last;
}

struct XqLaunch {
  hipStream_t stream;
  const void *in0, *in1;
  int stride0, stride1;
  const aomhip_txb *blocks;
  int n_blocks, grid_cols, uniform_type;
  QuantArgs qa;
  int32_t *coeff, *qcoeff, *dqcoeff;
  uint16_t *eob;
  int64_t *err_out = nullptr;  // optional av1_block_error outputs: {error, ssz} per block
  int err_shift = 0;           // 0: av1_block_error_c (32-bit products); 2 * (bd - 8) >= 0 with err_hbd: the highbd form

};

template <int W, int H, bool HBD, int SRC> static int launch_xq(const XqLaunch &l) {
  constexpr int LPB = W > H ? W : H;
  constexpr int BPW = kXqThreads / LPB;
  // Variants (measured, profiles/r01_txq_variants.md):
  //   2  whole block per lane, registers only           -> 4x4
  //   1  16-byte accesses staged through LDS             -> up to 16x16
  //   0  lanes load / store their own column / row       -> larger blocks (LDS footprint would cost occupancy)
  // AOMHIP_XQ_VARIANT forces one (falls back to 1 where 2 is not instantiated).
  static const int forced = [] { const char *e = getenv("AOMHIP_XQ_VARIANT"); return e ? atoi(e) : -1; }();
  constexpr bool kLane = (W == 4 && H == 4);  // 8x8 per lane measured 60 % slower than the staged kernel
  int variant = forced >= 0 ? forced : (kLane ? 2 : (W * H <= 256 ? 1 : 0));
  if (variant == 2 && !kLane) variant = 1;
  if constexpr (kLane) {
    if (variant == 2) {
      const int nwg = (l.n_blocks + kXqThreads - 1) / kXqThreads;
      const int nwg8 = (nwg + 7) & ~7;
      if (l.uniform_type == kTxWht)
        hipLaunchKernelGGL((xform_quant_lane_kernel<W, H, HBD, SRC, true>), dim3(nwg8), dim3(kXqThreads), 0, l.stream, l.in0,
                           l.in1, l.stride0, l.stride1, l.blocks, l.n_blocks, l.grid_cols, l.uniform_type, l.qa, l.coeff,
                           l.qcoeff, l.dqcoeff, l.eob, nwg8, l.err_out, l.err_shift);
      else
        hipLaunchKernelGGL((xform_quant_lane_kernel<W, H, HBD, SRC>), dim3(nwg8), dim3(kXqThreads), 0, l.stream, l.in0,
                           l.in1, l.stride0, l.stride1, l.blocks, l.n_blocks, l.grid_cols, l.uniform_type, l.qa, l.coeff,
                           l.qcoeff, l.dqcoeff, l.eob, nwg8, l.err_out, l.err_shift);
      AOMHIP_LAUNCH_CHECK();
      return AOMHIP_OK;
    }
  }
  const int nwg = (l.n_blocks + BPW - 1) / BPW;
  const int nwg8 = (nwg + 7) & ~7;
  if (variant == 0)
    hipLaunchKernelGGL((xform_quant_kernel<W, H, HBD, SRC>), dim3(nwg8), dim3(kXqThreads), 0, l.stream, l.in0, l.in1,
                       l.stride0, l.stride1, l.blocks, l.n_blocks, l.grid_cols, l.uniform_type, l.qa, l.coeff,
                       l.qcoeff, l.dqcoeff, l.eob, nwg8, l.err_out, l.err_shift);
  else
    hipLaunchKernelGGL((xform_quant_staged_kernel<W, H, HBD, SRC>), dim3(nwg8), dim3(kXqThreads), 0, l.stream, l.in0,
                       l.in1, l.stride0, l.stride1, l.blocks, l.n_blocks, l.grid_cols, l.uniform_type, l.qa, l.coeff,
                       l.qcoeff, l.dqcoeff, l.eob, nwg8, l.err_out, l.err_shift);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

template <bool HBD, int SRC> static int dispatch_xq(int tx_size, const XqLaunch &l) {
  switch (tx_size) {  // TX_SIZE order of av1/common/enums.h:169-193
    case 0: return launch_xq<4, 4, HBD, SRC>(l);
    case 1: return launch_xq<8, 8, HBD, SRC>(l);
    case 2: return launch_xq<16, 16, HBD, SRC>(l);
    case 3: return launch_xq<32, 32, HBD, SRC>(l);
    case 4: return launch_xq<64, 64, HBD, SRC>(l);
    case 5: return launch_xq<4, 8, HBD, SRC>(l);
    case 6: return launch_xq<8, 4, HBD, SRC>(l);
    case 7: return launch_xq<8, 16, HBD, SRC>(l);
    case 8: return launch_xq<16, 8, HBD, SRC>(l);
    case 9: return launch_xq<16, 32, HBD, SRC>(l);
    case 10: return launch_xq<32, 16, HBD, SRC>(l);
    case 11: return launch_xq<32, 64, HBD, SRC>(l);
    case 12: return launch_xq<64, 32, HBD, SRC>(l);
    case 13: return launch_xq<4, 16, HBD, SRC>(l);
    case 14: return launch_xq<16, 4, HBD, SRC>(l);
    case 15: return launch_xq<8, 32, HBD, SRC>(l);
    case 16: return launch_xq<32, 8, HBD, SRC>(l);
    case 17: return launch_xq<16, 64, HBD, SRC>(l);
    case 18: return launch_xq<64, 16, HBD, SRC>(l);
  }
  set_error("bad tx_size %d", tx_size);
  return AOMHIP_ERR_INVALID;
}

static const int kTxW[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
static const int kTxH[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };

static bool type_ok(int tx_size, int tx_type) {
  if (tx_type == kTxWht) return tx_size == 0;  // lossless WHT: TX_4X4 only
  // what av1_get_fwd_txfm_cfg can serve (av1_txfm.c:89-96): ADST <= 16 points, identity <= 32, 64 DCT only
  static const uint8_t vk[16] = { 0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3 };
  static const uint8_t hk[16] = { 0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2 };
  if (tx_type < 0 || tx_type > 15) return false;
  const int vmax = vk[tx_type] == 0 ? 64 : vk[tx_type] == 3 ? 32 : 16;
  const int hmax = hk[tx_type] == 0 ? 64 : hk[tx_type] == 3 ? 32 : 16;
  return kTxH[tx_size] <= vmax && kTxW[tx_size] <= hmax;
}

__global__ void validate_txb_kernel(const aomhip_txb *blocks, int n, int tw, int th, int wht_ok, int any_type, int *status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = blocks[i].tx_type;
  bool ok;
  if (t == AOMHIP_TX_WHT) ok = wht_ok != 0;
  else if (t > 15) ok = false;
  else if (any_type) ok = true;
  else {  // av1_get_fwd_txfm_cfg (av1_txfm.c:89-96): ADST <= 16 points, identity <= 32, 64-point DCT only
    const uint8_t vkind[16] = { 0, 1, 0, 1, 2, 0, 2, 1, 2, 3, 0, 3, 1, 3, 2, 3 }, hkind[16] = { 0, 0, 1, 1, 0, 2, 2, 2, 1, 3, 3, 0, 3, 1, 3, 2 };
    const int vmax = vkind[t] == 0 ? 64 : vkind[t] == 3 ? 32 : 16, hmax = hkind[t] == 0 ? 64 : hkind[t] == 3 ? 32 : 16;
    ok = th <= vmax && tw <= hmax;
  }
  if (!ok) atomicOr(status, kStatusBadTxType);
}
int validate_txb_list(aomhip_ctx *ctx, const aomhip_txb *d_blocks, int n_blocks, int tx_size, bool wht_ok, bool any_type_0_15) {
  if (!d_blocks || n_blocks <= 0) return AOMHIP_OK;
  static const int tw[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
  static const int th[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
  hipLaunchKernelGGL(validate_txb_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, ctx->stream, d_blocks, n_blocks, tw[tx_size],
                     th[tx_size], (int)(wht_ok && tx_size == 0), (int)any_type_0_15, ctx->d_status);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
bool tx_type_ok(int tx_size, int tx_type) { return tx_size >= 0 && tx_size < 19 && type_ok(tx_size, tx_type); }

static QuantArgs to_args(const aomhip_quant_params *q, int quant_kind = 0) { return to_quant_args(q, quant_kind); }

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_tx_size_wide(int tx_size) { return tx_size >= 0 && tx_size < 19 ? kTxW[tx_size] : 0; }
int aomhip_tx_size_high(int tx_size) { return tx_size >= 0 && tx_size < 19 ? kTxH[tx_size] : 0; }
int aomhip_tx_max_eob(int tx_size) {
  if (tx_size < 0 || tx_size >= 19) return 0;
  const int w = kTxW[tx_size] < 32 ? kTxW[tx_size] : 32, h = kTxH[tx_size] < 32 ? kTxH[tx_size] : 32;
  return w * h;
}

int aomhip_xform_quant_batch(aomhip_ctx *ctx, const int16_t *d_residual, int residual_stride, int tx_size,
                             const aomhip_txb *d_blocks, int n_blocks, int grid_cols, int uniform_tx_type,
                             const aomhip_quant_params *qparams, int is_hbd, int32_t *d_coeff, int32_t *d_qcoeff,
                             int32_t *d_dqcoeff, uint16_t *d_eob) {
  if (!ctx || !d_residual || !qparams || !d_qcoeff || !d_dqcoeff || !d_eob || tx_size < 0 || tx_size >= 19 ||
      n_blocks < 0 || (!d_blocks && (grid_cols <= 0 || !type_ok(tx_size, uniform_tx_type))) ||
      (uniform_tx_type == kTxWht && tx_size != 0)) {
    set_error("aomhip_xform_quant_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, true, false)) return rc;
  XqLaunch l{ ctx->stream, d_residual, nullptr, residual_stride, 0, d_blocks, n_blocks, grid_cols, uniform_tx_type,
              to_args(qparams), d_coeff, d_qcoeff, d_dqcoeff, d_eob };
  return is_hbd ? dispatch_xq<true, 0>(tx_size, l) : dispatch_xq<false, 0>(tx_size, l);
}

int aomhip_xform_quant_ex_batch(aomhip_ctx *ctx, const int16_t *d_residual, int residual_stride, int tx_size,
                                const aomhip_txb *d_blocks, int n_blocks, int grid_cols, int uniform_tx_type,
                                const aomhip_quant_params *qparams, int is_hbd, int bit_depth, int quant_kind, int32_t *d_coeff,
                                int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob, int64_t *d_block_error) {
  if (!ctx || !d_residual || !qparams || !d_qcoeff || !d_dqcoeff || !d_eob || tx_size < 0 || tx_size >= 19 ||
      n_blocks < 0 || (!d_blocks && (grid_cols <= 0 || !type_ok(tx_size, uniform_tx_type))) ||
      (uniform_tx_type == kTxWht && tx_size != 0) || (bit_depth != 8 && bit_depth != 10 && bit_depth != 12) ||
      quant_kind < 0 || quant_kind > 1) {
    set_error("aomhip_xform_quant_ex_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, true, false)) return rc;
  XqLaunch l{ ctx->stream, d_residual, nullptr, residual_stride, 0, d_blocks, n_blocks, grid_cols, uniform_tx_type,
              to_args(qparams, quant_kind), d_coeff, d_qcoeff, d_dqcoeff, d_eob };
  l.err_out = d_block_error;
  l.err_shift = is_hbd ? 2 * (bit_depth - 8) : -1;   // the highbd form when the buffers are high bit depth (tx_search.c)
  return is_hbd ? dispatch_xq<true, 0>(tx_size, l) : dispatch_xq<false, 0>(tx_size, l);
}

int aomhip_subtract_xform_quant_ex_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *pred, int frame,
                                         int tx_size, const aomhip_txb *d_blocks, int n_blocks, int grid_cols,
                                         int uniform_tx_type, const aomhip_quant_params *qparams, int quant_kind,
                                         int32_t *d_coeff, int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob,
                                         int64_t *d_block_error) {
  if (!ctx || !src || !pred || !src->base || !pred->base || !qparams || !d_qcoeff || !d_dqcoeff || !d_eob ||
      tx_size < 0 || tx_size >= 19 || n_blocks < 0 || frame < 0 || frame >= src->n_frames ||
      frame >= pred->n_frames || (src->bit_depth == 8) != (pred->bit_depth == 8) ||
      (!d_blocks && (grid_cols <= 0 || !type_ok(tx_size, uniform_tx_type))) || quant_kind < 0 || quant_kind > 1 ||
      (uniform_tx_type == kTxWht && tx_size != 0)) {
    set_error("aomhip_subtract_xform_quant_ex_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, true, false)) return rc;
  const size_t esz = src->bit_depth == 8 ? 1 : 2;
  const char *s = static_cast<const char *>(src->base) +
                  ((size_t)frame * src->frame_stride + (size_t)src->border * src->stride + src->border) * esz;
  const char *p = static_cast<const char *>(pred->base) +
                  ((size_t)frame * pred->frame_stride + (size_t)pred->border * pred->stride + pred->border) * esz;
  XqLaunch l{ ctx->stream, s, p, src->stride, pred->stride, d_blocks, n_blocks, grid_cols, uniform_tx_type,
              to_args(qparams, quant_kind), d_coeff, d_qcoeff, d_dqcoeff, d_eob };
  l.err_out = d_block_error;
  l.err_shift = src->bit_depth == 8 ? -1 : 2 * (src->bit_depth - 8);
  // encodemb.c:323: the quantiser flavour follows the bit depth of the planes
  return src->bit_depth == 8 ? dispatch_xq<false, 1>(tx_size, l) : dispatch_xq<true, 2>(tx_size, l);
}

int aomhip_subtract_xform_quant_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *pred, int frame,
                                      int tx_size, const aomhip_txb *d_blocks, int n_blocks, int grid_cols,
                                      int uniform_tx_type, const aomhip_quant_params *qparams, int32_t *d_coeff,
                                      int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob) {
  return aomhip_subtract_xform_quant_ex_batch(ctx, src, pred, frame, tx_size, d_blocks, n_blocks, grid_cols, uniform_tx_type,
                                              qparams, 0, d_coeff, d_qcoeff, d_dqcoeff, d_eob, nullptr);
}

static int quantize_adaptive_launch(aomhip_ctx *ctx, const int32_t *d_coeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks, int uniform_tx_type,
                                    const aomhip_quant_params *qparams, int is_hbd, int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob, bool with_qm,
                                    const uint8_t *d_qm, const uint8_t *d_iqm, const char *who) {
  if (!ctx || !d_coeff || !qparams || !d_qcoeff || !d_dqcoeff || !d_eob || tx_size < 0 || tx_size >= 19 || n_blocks < 0 ||
      (!d_blocks && (uniform_tx_type < 0 || uniform_tx_type > 15))) {
    set_error("%s: invalid argument", who);
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, false, true)) return rc;
  const QuantArgs qa = to_args(qparams);
  const int w = kTxW[tx_size], h = kTxH[tx_size];
  const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;
  const int ls = (w * h > 256) + (w * h > 1024);  // av1_get_tx_scale (av1/common/idct.c:24-28)
  const dim3 grid((n_blocks + 3) / 4), block(256);
#define AOMHIP_QA_K(KW_, KH_, HBD_, LS_, QMX_)                                                                                            \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(quant_adaptive_kernel<KW_, KH_, HBD_, LS_, QMX_>), grid, block, 0, ctx->stream, d_coeff, d_blocks,   \
                     n_blocks, uniform_tx_type, qa, d_qcoeff, d_dqcoeff, d_eob, d_qm, d_iqm)
#define AOMHIP_QA(KW_, KH_, LS_)                                                                                                          \
  if (kw == KW_ && kh == KH_ && ls == LS_) {                                                                                              \
    if (is_hbd) { if (with_qm) AOMHIP_QA_K(KW_, KH_, true, LS_, true); else AOMHIP_QA_K(KW_, KH_, true, LS_, false); }                    \
    else { if (with_qm) AOMHIP_QA_K(KW_, KH_, false, LS_, true); else AOMHIP_QA_K(KW_, KH_, false, LS_, false); }                         \
    AOMHIP_LAUNCH_CHECK();                                                                                                                \
    return AOMHIP_OK;                                                                                                                     \
  }
  AOMHIP_QA(4, 4, 0) AOMHIP_QA(8, 8, 0) AOMHIP_QA(16, 16, 0) AOMHIP_QA(32, 32, 1) AOMHIP_QA(32, 32, 2)
  AOMHIP_QA(4, 8, 0) AOMHIP_QA(8, 4, 0) AOMHIP_QA(8, 16, 0) AOMHIP_QA(16, 8, 0) AOMHIP_QA(16, 32, 1) AOMHIP_QA(32, 16, 1)
  AOMHIP_QA(4, 16, 0) AOMHIP_QA(16, 4, 0) AOMHIP_QA(8, 32, 0) AOMHIP_QA(32, 8, 0)
#undef AOMHIP_QA
#undef AOMHIP_QA_K
  set_error("%s: no kernel for tx_size %d", who, tx_size);
  return AOMHIP_ERR_INVALID;
}

int aomhip_quantize_b_adaptive_batch(aomhip_ctx *ctx, const int32_t *d_coeff, int tx_size, const aomhip_txb *d_blocks,
                                     int n_blocks, int uniform_tx_type, const aomhip_quant_params *qparams, int is_hbd,
                                     int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob) {
  return quantize_adaptive_launch(ctx, d_coeff, tx_size, d_blocks, n_blocks, uniform_tx_type, qparams, is_hbd, d_qcoeff, d_dqcoeff, d_eob, false, nullptr,
                                  nullptr, "aomhip_quantize_b_adaptive_batch");
}

// aom_[highbd_]quantize_b_adaptive_helper_c with qm_ptr / iqm_ptr (use_quant_b_adapt under enable_qm); either matrix may be NULL (flat)
int aomhip_quantize_b_adaptive_qm_batch(aomhip_ctx *ctx, const int32_t *d_coeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks,
                                        int uniform_tx_type, const aomhip_quant_params *qparams, int is_hbd, const uint8_t *d_qm, const uint8_t *d_iqm,
                                        int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob) {
  return quantize_adaptive_launch(ctx, d_coeff, tx_size, d_blocks, n_blocks, uniform_tx_type, qparams, is_hbd, d_qcoeff, d_dqcoeff, d_eob, true, d_qm, d_iqm,
                                  "aomhip_quantize_b_adaptive_qm_batch");
}


static int quantize_qm_launch(aomhip_ctx *ctx, const int32_t *d_coeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks, int uniform_tx_type,
                              const aomhip_quant_params *qparams, int is_hbd, const uint8_t *d_qm, const uint8_t *d_iqm, int32_t *d_qcoeff,
                              int32_t *d_dqcoeff, uint16_t *d_eob, bool fp) {
  if (!ctx || !d_coeff || !qparams || !d_qcoeff || !d_dqcoeff || !d_eob || tx_size < 0 || tx_size >= 19 || n_blocks < 0 ||
      (!d_blocks && (uniform_tx_type < 0 || uniform_tx_type > 15))) {
    set_error("aomhip_quantize_%s_qm_batch: invalid argument", fp ? "fp" : "b");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, false, true)) return rc;
  const QuantArgs qa = to_args(qparams);
  const int w = kTxW[tx_size], h = kTxH[tx_size];
  const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;
  const int ls = (w * h > 256) + (w * h > 1024);  // av1_get_tx_scale (av1/common/idct.c:24-28)
  const dim3 grid((n_blocks + 3) / 4), block(256);
#define AOMHIP_QM_K(KW_, KH_, HBD_, LS_, FP_)                                                                                      \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(quant_qm_kernel<KW_, KH_, HBD_, LS_, FP_>), grid, block, 0, ctx->stream, d_coeff, d_blocks, n_blocks, \
                     uniform_tx_type, qa, d_qm, d_iqm, d_qcoeff, d_dqcoeff, d_eob)
#define AOMHIP_QM(KW_, KH_, LS_)                                                                                                   \
  if (kw == KW_ && kh == KH_ && ls == LS_) {                                                                                       \
    if (is_hbd) { if (fp) AOMHIP_QM_K(KW_, KH_, true, LS_, true); else AOMHIP_QM_K(KW_, KH_, true, LS_, false); }                  \
    else { if (fp) AOMHIP_QM_K(KW_, KH_, false, LS_, true); else AOMHIP_QM_K(KW_, KH_, false, LS_, false); }                       \
    AOMHIP_LAUNCH_CHECK();                                                                                                         \
    return AOMHIP_OK;                                                                                                              \
  }
  AOMHIP_QM(4, 4, 0) AOMHIP_QM(8, 8, 0) AOMHIP_QM(16, 16, 0) AOMHIP_QM(32, 32, 1) AOMHIP_QM(32, 32, 2)
  AOMHIP_QM(4, 8, 0) AOMHIP_QM(8, 4, 0) AOMHIP_QM(8, 16, 0) AOMHIP_QM(16, 8, 0) AOMHIP_QM(16, 32, 1) AOMHIP_QM(32, 16, 1)
  AOMHIP_QM(4, 16, 0) AOMHIP_QM(16, 4, 0) AOMHIP_QM(8, 32, 0) AOMHIP_QM(32, 8, 0)
#undef AOMHIP_QM
#undef AOMHIP_QM_K
  set_error("aomhip_quantize_%s_qm_batch: no kernel for tx_size %d", fp ? "fp" : "b", tx_size);
  return AOMHIP_ERR_INVALID;
}

int aomhip_quantize_b_qm_batch(aomhip_ctx *ctx, const int32_t *d_coeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks,
                               int uniform_tx_type, const aomhip_quant_params *qparams, int is_hbd, const uint8_t *d_qm, const uint8_t *d_iqm,
                               int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob) {
  return quantize_qm_launch(ctx, d_coeff, tx_size, d_blocks, n_blocks, uniform_tx_type, qparams, is_hbd, d_qm, d_iqm, d_qcoeff, d_dqcoeff, d_eob, false);
}

// av1_[highbd_]quantize_fp* with matrices (AV1_XFORM_QUANT_FP when enable_qm is on): qparams carries round_fp / quant_fp in its round / quant fields
int aomhip_quantize_fp_qm_batch(aomhip_ctx *ctx, const int32_t *d_coeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks,
                                int uniform_tx_type, const aomhip_quant_params *qparams, int is_hbd, const uint8_t *d_qm, const uint8_t *d_iqm,
                                int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob) {
  return quantize_qm_launch(ctx, d_coeff, tx_size, d_blocks, n_blocks, uniform_tx_type, qparams, is_hbd, d_qm, d_iqm, d_qcoeff, d_dqcoeff, d_eob, true);
}

// av1_quantize_lp (+ av1_block_error_lp when d_err is given): qparams carries round_fp / quant_fp in its round / quant fields
int aomhip_quantize_lp_batch(aomhip_ctx *ctx, const int16_t *d_coeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks, int uniform_tx_type,
                             const aomhip_quant_params *qparams, int16_t *d_qcoeff, int16_t *d_dqcoeff, uint16_t *d_eob, int64_t *d_err) {
  if (!ctx || !d_coeff || !qparams || !d_qcoeff || !d_dqcoeff || !d_eob || tx_size < 0 || tx_size >= 19 || n_blocks < 0 ||
      (!d_blocks && (uniform_tx_type < 0 || uniform_tx_type > 15))) {
    set_error("aomhip_quantize_lp_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, false, true)) return rc;
  const QuantArgs qa = to_args(qparams);
  const int w = kTxW[tx_size], h = kTxH[tx_size];
  const int kw = w, kh = h;   // (no 64-point sizes: the non-RD search's transforms stop at 32 x 32, and av1_quantize_lp has no log-scale form)
  const dim3 grid((n_blocks + 3) / 4), block(256);
#define AOMHIP_LP(KW_, KH_)                                                                                                        \
  if (kw == KW_ && kh == KH_) {                                                                                                    \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(quant_lp_kernel<KW_, KH_>), grid, block, 0, ctx->stream, d_coeff, d_blocks, n_blocks,       \
                       uniform_tx_type, qa, d_qcoeff, d_dqcoeff, d_eob, d_err);                                                    \
    AOMHIP_LAUNCH_CHECK();                                                                                                         \
    return AOMHIP_OK;                                                                                                              \
  }
  AOMHIP_LP(4, 4) AOMHIP_LP(8, 8) AOMHIP_LP(16, 16) AOMHIP_LP(32, 32) AOMHIP_LP(4, 8) AOMHIP_LP(8, 4) AOMHIP_LP(8, 16) AOMHIP_LP(16, 8)
  AOMHIP_LP(16, 32) AOMHIP_LP(32, 16) AOMHIP_LP(4, 16) AOMHIP_LP(16, 4) AOMHIP_LP(8, 32) AOMHIP_LP(32, 8)
#undef AOMHIP_LP
  set_error("aomhip_quantize_lp_batch: no kernel for tx_size %d", tx_size);
  return AOMHIP_ERR_INVALID;
}

int aomhip_get_nz_map_contexts_batch(aomhip_ctx *ctx, const uint8_t *d_levels, int64_t levels_pitch, int tx_size, const aomhip_txb *d_blocks, int n_blocks,
                                     int uniform_tx_type, const uint16_t *d_eob, int8_t *d_coeff_contexts, int64_t contexts_pitch) {
  if (!ctx || tx_size < 0 || tx_size >= 19 || n_blocks < 0 || (n_blocks > 0 && (!d_levels || !d_eob || !d_coeff_contexts)) ||
      (!d_blocks && (uniform_tx_type < 0 || uniform_tx_type > 15))) {
    set_error("aomhip_get_nz_map_contexts_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const int w = kTxW[tx_size], h = kTxH[tx_size];
  const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;   // av1_get_adjusted_tx_size
  if (levels_pitch < (int64_t)(kh + 4) * (kw + 4) + 16 || contexts_pitch < (int64_t)kw * kh) {
    set_error("aomhip_get_nz_map_contexts_batch: pitch too small for a %d x %d level map", kw, kh);
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, false, true)) return rc;
  const int rel = (w > h) - (w < h);
  const dim3 grid((n_blocks + 3) / 4), block(256);
#define AOMHIP_NZ(KW_, KH_)                                                                                                         \
  if (kw == KW_ && kh == KH_) {                                                                                                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(nz_map_contexts_kernel<KW_, KH_>), grid, block, 0, ctx->stream, d_levels, levels_pitch, d_blocks, n_blocks, \
                       uniform_tx_type, rel, d_eob, d_coeff_contexts, contexts_pitch);                                              \
    AOMHIP_LAUNCH_CHECK();                                                                                                          \
    return AOMHIP_OK;                                                                                                               \
  }
  AOMHIP_NZ(4, 4) AOMHIP_NZ(8, 8) AOMHIP_NZ(16, 16) AOMHIP_NZ(32, 32) AOMHIP_NZ(4, 8) AOMHIP_NZ(8, 4) AOMHIP_NZ(8, 16) AOMHIP_NZ(16, 8)
  AOMHIP_NZ(16, 32) AOMHIP_NZ(32, 16) AOMHIP_NZ(4, 16) AOMHIP_NZ(16, 4) AOMHIP_NZ(8, 32) AOMHIP_NZ(32, 8)
#undef AOMHIP_NZ
  set_error("aomhip_get_nz_map_contexts_batch: no kernel for tx_size %d", tx_size);
  return AOMHIP_ERR_INVALID;
}

static int cost_coeffs_launch(aomhip_ctx *ctx, const int32_t *d_qcoeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks, int uniform_tx_type,
                              const uint16_t *d_eob, const uint8_t *d_txb_ctx, const int32_t *d_costs, int32_t *d_cost, bool laplacian) {
  if (!ctx || tx_size < 0 || tx_size >= 19 || n_blocks < 0 || (n_blocks > 0 && (!d_qcoeff || !d_eob || !d_txb_ctx || !d_costs || !d_cost)) ||
      (!d_blocks && (uniform_tx_type < 0 || uniform_tx_type > 15))) {
    set_error("aomhip_cost_coeffs_txb_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, false, true)) return rc;
  const int w = kTxW[tx_size], h = kTxH[tx_size];
  const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;   // av1_get_adjusted_tx_size
  const int rel = (w > h) - (w < h);
  const dim3 grid((n_blocks + 3) / 4), block(256);
#define AOMHIP_CC(KW_, KH_)                                                                                                         \
  if (kw == KW_ && kh == KH_) {                                                                                                     \
    if (laplacian)                                                                                                                  \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(cost_coeffs_txb_kernel<KW_, KH_, true>), grid, block, 0, ctx->stream, d_qcoeff, d_blocks, n_blocks,        \
                         uniform_tx_type, rel, d_eob, d_txb_ctx, d_costs, d_cost);                                                  \
    else                                                                                                                            \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(cost_coeffs_txb_kernel<KW_, KH_, false>), grid, block, 0, ctx->stream, d_qcoeff, d_blocks, n_blocks,       \
                         uniform_tx_type, rel, d_eob, d_txb_ctx, d_costs, d_cost);                                                  \
    AOMHIP_LAUNCH_CHECK();                                                                                                          \
    return AOMHIP_OK;                                                                                                               \
  }
  AOMHIP_CC(4, 4) AOMHIP_CC(8, 8) AOMHIP_CC(16, 16) AOMHIP_CC(32, 32) AOMHIP_CC(4, 8) AOMHIP_CC(8, 4) AOMHIP_CC(8, 16) AOMHIP_CC(16, 8)
  AOMHIP_CC(16, 32) AOMHIP_CC(32, 16) AOMHIP_CC(4, 16) AOMHIP_CC(16, 4) AOMHIP_CC(8, 32) AOMHIP_CC(32, 8)
#undef AOMHIP_CC
  set_error("aomhip_cost_coeffs_txb_batch: no kernel for tx_size %d", tx_size);
  return AOMHIP_ERR_INVALID;
}

int aomhip_cost_coeffs_txb_batch(aomhip_ctx *ctx, const int32_t *d_qcoeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks, int uniform_tx_type,
                                 const uint16_t *d_eob, const uint8_t *d_txb_ctx, const int32_t *d_costs, int32_t *d_cost) {
  return cost_coeffs_launch(ctx, d_qcoeff, tx_size, d_blocks, n_blocks, uniform_tx_type, d_eob, d_txb_ctx, d_costs, d_cost, false);
}

// av1_cost_coeffs_txb_laplacian (adjust_eob == 0): the transform-type search's estimate
int aomhip_cost_coeffs_txb_laplacian_batch(aomhip_ctx *ctx, const int32_t *d_qcoeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks, int uniform_tx_type,
                                           const uint16_t *d_eob, const uint8_t *d_txb_ctx, const int32_t *d_costs, int32_t *d_cost) {
  return cost_coeffs_launch(ctx, d_qcoeff, tx_size, d_blocks, n_blocks, uniform_tx_type, d_eob, d_txb_ctx, d_costs, d_cost, true);
}

int aomhip_txb_entropy_context_batch(aomhip_ctx *ctx, const int32_t *d_qcoeff, int tx_size, const aomhip_txb *d_blocks, int n_blocks, int uniform_tx_type,
                                     const uint16_t *d_eob, uint8_t *d_entropy_ctx) {
  if (!ctx || tx_size < 0 || tx_size >= 19 || n_blocks < 0 || (n_blocks > 0 && (!d_qcoeff || !d_eob || !d_entropy_ctx)) ||
      (!d_blocks && (uniform_tx_type < 0 || uniform_tx_type > 15))) {
    set_error("aomhip_txb_entropy_context_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, false, true)) return rc;
  const int w = kTxW[tx_size], h = kTxH[tx_size];
  const int kw = w > 32 ? 32 : w, kh = h > 32 ? 32 : h;
  const dim3 grid((n_blocks + 3) / 4), block(256);
#define AOMHIP_EC(KW_, KH_)                                                                                                         \
  if (kw == KW_ && kh == KH_) {                                                                                                     \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(txb_entropy_context_kernel<KW_, KH_>), grid, block, 0, ctx->stream, d_qcoeff, d_blocks, n_blocks, uniform_tx_type, \
                       d_eob, d_entropy_ctx);                                                                                       \
    AOMHIP_LAUNCH_CHECK();                                                                                                          \
    return AOMHIP_OK;                                                                                                               \
  }
  AOMHIP_EC(4, 4) AOMHIP_EC(8, 8) AOMHIP_EC(16, 16) AOMHIP_EC(32, 32) AOMHIP_EC(4, 8) AOMHIP_EC(8, 4) AOMHIP_EC(8, 16) AOMHIP_EC(16, 8)
  AOMHIP_EC(16, 32) AOMHIP_EC(32, 16) AOMHIP_EC(4, 16) AOMHIP_EC(16, 4) AOMHIP_EC(8, 32) AOMHIP_EC(32, 8)
#undef AOMHIP_EC
  set_error("aomhip_txb_entropy_context_batch: no kernel for tx_size %d", tx_size);
  return AOMHIP_ERR_INVALID;
}

// av1_xform_quant with quantisation matrices: the forward transform by the fused kernel (its own flat-matrix levels are overwritten), then
// the matrix quantiser on the coefficients it stored
int aomhip_xform_quant_qm_batch(aomhip_ctx *ctx, const int16_t *d_residual, int residual_stride, int tx_size, const aomhip_txb *d_blocks,
                                int n_blocks, int grid_cols, int uniform_tx_type, const aomhip_quant_params *qparams, int is_hbd,
                                const uint8_t *d_qm, const uint8_t *d_iqm, int32_t *d_coeff, int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob) {
  if (!d_coeff) {
    set_error("aomhip_xform_quant_qm_batch: d_coeff is required (the matrix quantiser reads the transform coefficients from it)");
    return AOMHIP_ERR_INVALID;
  }
  if (int rc = aomhip_xform_quant_batch(ctx, d_residual, residual_stride, tx_size, d_blocks, n_blocks, grid_cols, uniform_tx_type, qparams, is_hbd, d_coeff,
                                        d_qcoeff, d_dqcoeff, d_eob))
    return rc;
  // (grid mode: block i's coefficients are at i * NC, which is what the quantiser assumes without a list)
  return aomhip_quantize_b_qm_batch(ctx, d_coeff, tx_size, d_blocks, n_blocks, uniform_tx_type, qparams, is_hbd, d_qm, d_iqm, d_qcoeff, d_dqcoeff, d_eob);
}

int aomhip_subtract_xform_quant_qm_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *pred, int frame, int tx_size,
                                         const aomhip_txb *d_blocks, int n_blocks, int grid_cols, int uniform_tx_type, const aomhip_quant_params *qparams,
                                         const uint8_t *d_qm, const uint8_t *d_iqm, int32_t *d_coeff, int32_t *d_qcoeff, int32_t *d_dqcoeff,
                                         uint16_t *d_eob) {
  if (!d_coeff || !src) {
    set_error("aomhip_subtract_xform_quant_qm_batch: d_coeff is required (the matrix quantiser reads the transform coefficients from it)");
    return AOMHIP_ERR_INVALID;
  }
  if (int rc = aomhip_subtract_xform_quant_batch(ctx, src, pred, frame, tx_size, d_blocks, n_blocks, grid_cols, uniform_tx_type, qparams, d_coeff, d_qcoeff,
                                                 d_dqcoeff, d_eob))
    return rc;
  return aomhip_quantize_b_qm_batch(ctx, d_coeff, tx_size, d_blocks, n_blocks, uniform_tx_type, qparams, src->bit_depth != 8, d_qm, d_iqm, d_qcoeff, d_dqcoeff,
                                    d_eob);
}


// The rtcd-signature quantisers (aom_dsp/aom_dsp_rtcd_defs.pl:653-693): host pointers, one launch, synchronous.
// log_scale 0 / 1 / 2 = aom_quantize_b / _32x32 / _64x64; is_hbd: the aom_highbd_ family; adaptive: the _adaptive family.
// A failed call records the sticky status (aomhip_status()), zeroes the outputs and returns.
void aomhip_quantize_b_any(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr,
                           const int16_t *quant_ptr, const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr,
                           const int16_t *dequant_ptr, uint16_t *eob_ptr, const int16_t *scan, const int16_t *iscan, int log_scale,
                           int is_hbd, int adaptive) {
  (void)scan;
  *eob_ptr = 0;
  // (range check BEFORE anything is sized by n_coeffs: an invalid count zeroes *eob_ptr only and touches no block memory)
  if (n_coeffs <= 0 || n_coeffs > 4096 || log_scale < 0 || log_scale > 2) {
    set_error("aomhip_quantize_b: n_coeffs %ld / log_scale %d unsupported", (long)n_coeffs, log_scale);
    return note_failure("aomhip_quantize_b", AOMHIP_ERR_INVALID);
  }
  const size_t n = (size_t)n_coeffs;
  memset(qcoeff_ptr, 0, n * 4);
  memset(dqcoeff_ptr, 0, n * 4);
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  const size_t isc_off = n * 4, q_off = (isc_off + n * 2 + 15) & ~(size_t)15, dq_off = q_off + n * 4, e_off = dq_off + n * 4;
  const size_t total = e_off + 16;
  char *h = static_cast<char *>(pinned(ctx, total)), *d = static_cast<char *>(scratch(ctx, total));
  if (!h || !d) return note_failure("aomhip_quantize_b scratch", AOMHIP_ERR_NOMEM);
  memcpy(h, coeff_ptr, n * 4);
  memcpy(h + isc_off, iscan, n * 2);
  if (hipMemcpyAsync(d, h, q_off, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return note_failure("aomhip_quantize_b H2D");
  aomhip_quant_params qp;
  for (int i = 0; i < 2; ++i) {
    qp.zbin[i] = zbin_ptr[i]; qp.round[i] = round_ptr[i]; qp.quant[i] = quant_ptr[i];
    qp.quant_shift[i] = quant_shift_ptr[i]; qp.dequant[i] = dequant_ptr[i];
  }
  const QuantArgs qa = to_args(&qp);
  const int32_t *dc = reinterpret_cast<const int32_t *>(d);
  const int16_t *di = reinterpret_cast<const int16_t *>(d + isc_off);
  int32_t *dq = reinterpret_cast<int32_t *>(d + q_off), *ddq = reinterpret_cast<int32_t *>(d + dq_off);
  uint16_t *de = reinterpret_cast<uint16_t *>(d + e_off);
#define AOMHIP_QT(HBD, LS, AD) \
  hipLaunchKernelGGL((quant_table_kernel<HBD, LS, AD>), dim3(1), dim3(64), 0, ctx->stream, dc, (int)n, qa, di, dq, ddq, de)
#define AOMHIP_QT_LS(HBD, AD) \
  do { if (log_scale == 0) AOMHIP_QT(HBD, 0, AD); else if (log_scale == 1) AOMHIP_QT(HBD, 1, AD); else AOMHIP_QT(HBD, 2, AD); } while (0)
  if (is_hbd) { if (adaptive) AOMHIP_QT_LS(true, true); else AOMHIP_QT_LS(true, false); }
  else { if (adaptive) AOMHIP_QT_LS(false, true); else AOMHIP_QT_LS(false, false); }
#undef AOMHIP_QT_LS
#undef AOMHIP_QT
  if (hipGetLastError() != hipSuccess) return note_failure("aomhip_quantize_b launch");
  if (hipMemcpyAsync(h + q_off, d + q_off, total - q_off, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return note_failure("aomhip_quantize_b D2H");
  memcpy(qcoeff_ptr, h + q_off, n * 4);
  memcpy(dqcoeff_ptr, h + dq_off, n * 4);
  *eob_ptr = *reinterpret_cast<const uint16_t *>(h + e_off);
}

}  // extern "C"
