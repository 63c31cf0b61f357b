// Batched inverse 2-D transform + reconstruction on gfx950.
// Reference: av1/common/av1_inv_txfm2d.c:234-316 inv_txfm2d_add_c (+ the 64-point input re-mapping of
// :381-450), reached from av1_inverse_transform_block (av1/common/idct.c:304) right after quantisation
// (av1/encoder/encodemb.c:454-459).  The low-bit-depth entry (av1_inv_txfm_add_c) runs the same arithmetic
// with bd = 8, so one kernel serves uint8 and uint16 planes.
//
// Mapping: max(W, H) adjacent lanes own a block.  Row pass: lane r reads its W dequantised
// coefficients from the transposed layout (coalesced across lanes for each c), clamps to bd+8, runs the
// W-point inverse network (cos_bit 12, add/sub clamped to the per-depth stage range), rounds, and writes
// the transposed LDS tile.  Column pass: lane c clamps to max(bd+6, 16), runs the H-point network, rounds
// by 4 and adds into the destination pixels with clipping.  64-point sizes only read their packed 32
// low-frequency coefficients; all-zero rows are not transformed (every network maps 0 -> 0).
#include "common.h"
#include "txfm_device.h"
#include "quant_device.h"

namespace aomhip {

using namespace txfm;


constexpr int kInvThreads = 256;

template <int W, int H, int BD, typename PIX>
__global__ __launch_bounds__(kInvThreads) void inv_txfm_add_kernel(const int32_t *__restrict__ dqcoeff,
                                                                   const aomhip_txb *__restrict__ blocks, int n_blocks,
                                                                   int grid_cols, int uniform_type,
                                                                   const uint16_t *__restrict__ eob, PIX *dst_origin,
                                                                   int dst_stride, int nblk8) {
  using C = Cfg2D<W, H>;
  constexpr int LPB = W > H ? W : H;
  constexpr int BPW = kInvThreads / LPB;
  constexpr int KW = W < 32 ? W : 32, KH = H < 32 ? H : 32;
  constexpr int NC = KW * KH;
  constexpr int LSTRIDE = W + 1;
  // av1_gen_inv_stage_range (av1_inv_txfm2d.c:188-232)
  constexpr int RNG_ROW = BD == 8 ? 16 : BD == 10 ? 18 : 20;
  constexpr int RNG_COL = BD == 8 ? 16 : BD == 10 ? 16 : 18;
  constexpr int COL_CLAMP = BD + 6 > 16 ? BD + 6 : 16;
  __shared__ int32_t tile[BPW][KH * LSTRIDE];

  const int slot = threadIdx.x / LPB, lane = threadIdx.x % LPB;
  const unsigned wg = xcd_chunked_index(blockIdx.x, nblk8);
  const int bi = wg * BPW + slot;
  bool live = bi < n_blocks;
  int bx = 0, by = 0, tx_type = uniform_type;
  int64_t in_off = (int64_t)bi * NC;
  if (live) {
    if (blocks) {
      const aomhip_txb b = blocks[bi];
      bx = b.x;
      by = b.y;
      tx_type = b.tx_type;
      in_off = b.out_offset;
    } else {
      bx = (bi % grid_cols) * W;
      by = (bi / grid_cols) * H;
    }
    if (eob && eob[bi] == 0) live = false;  // av1_inverse_transform_block: nothing to add when eob == 0
  }
  if constexpr (W == 4 && H == 4) {
    // lossless: av1_highbd_iwht4x4_add (idct.c:34-41) -> _16_add (eob > 1) or _1_add (av1_inv_txfm2d.c:20-114); one lane
    // does the 16 pixels (lossless blocks are rare and tiny)
    if (live && tx_type == kTxWht) {
      if (lane == 0) {
        constexpr int kMaxW = (1 << BD) - 1;
        auto put = [&](int r, int c, int v) {
          PIX *p = dst_origin + (int64_t)(by + r) * dst_stride + bx + c;
          int o = (int)*p + v;
          *p = (PIX)(o < 0 ? 0 : (o > kMaxW ? kMaxW : o));
        };
        auto bf = [](int &a, int &b, int &c, int &d) {
          int a1 = a + c, d1 = d - b;
          const int e1 = (a1 - d1) >> 1;
          const int b1 = e1 - b, c1 = e1 - c;
          a1 -= b1;
          d1 += c1;
          a = a1; b = b1; c = c1; d = d1;
        };
        const int32_t *in = dqcoeff + in_off;
        if (!eob || eob[bi] > 1) {
          int o[16];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int a = in[i] >> 2, c = in[4 + i] >> 2, d = in[8 + i] >> 2, b = in[12 + i] >> 2;
            bf(a, b, c, d);
            o[i] = a; o[4 + i] = b; o[8 + i] = c; o[12 + i] = d;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            int a = o[4 * i], c = o[4 * i + 1], d = o[4 * i + 2], b = o[4 * i + 3];
            bf(a, b, c, d);
            put(0, i, a); put(1, i, b); put(2, i, c); put(3, i, d);
          }
        } else {
          int a1 = in[0] >> 2, e1 = a1 >> 1;
          a1 -= e1;
          const int tmp[4] = { a1, e1, e1, e1 };
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = tmp[i] >> 1, a = tmp[i] - e;
            put(0, i, a); put(1, i, e); put(2, i, e); put(3, i, e);
          }
        }
      }
      live = false;
    }
  }
  const int vk = iv_kind(tx_type), hk = ih_kind(tx_type);
  int32_t(&t)[KH * LSTRIDE] = tile[slot];

  // ---- rows (only the KH rows that can be non-zero)
  if (live && lane < KH) {
    const int r = lane;
    int32_t x[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
      int32_t v = 0;
      if (c < KW) {
        v = dqcoeff[in_off + c * KH + r];
        if constexpr (C::rect2) v = rshift64((int64_t)v * kInvSqrt2, kSqrt2Bits);
        v = clampv<BD + 8>(v);
      }
      x[c] = v;
    }
    inv_1d<W, 12, RNG_ROW>(x, hk == 2 ? 1 : hk);
#pragma unroll
    for (int c = 0; c < W; ++c) {
      int32_t v = x[c];
      if constexpr (C::is0 < 0) v = rshift(v, -C::is0);
      t[r * LSTRIDE + c] = v;
    }
  }
  block_sync();

  // ---- columns + add
  if (live && lane < W) {
    const int c = lane;
    const int sc = (hk == 2) ? W - 1 - c : c;
    int32_t y[H];
#pragma unroll
    for (int r = 0; r < H; ++r) y[r] = r < KH ? clampv<COL_CLAMP>(t[r * LSTRIDE + sc]) : 0;
    inv_1d<H, 12, RNG_COL>(y, vk == 2 ? 1 : vk);
    const bool ud = (vk == 2);
    constexpr int kMax = (1 << BD) - 1;
#pragma unroll
    for (int r = 0; r < H; ++r) {
      const int32_t res = rshift(y[ud ? H - 1 - r : r], 4);  // -shift[1] == 4 for every size
      PIX *p = dst_origin + (int64_t)(by + r) * dst_stride + bx + c;
      int v = (int)*p + res;  // highbd_clip_pixel_add (av1_txfm.h:104-107)
      v = v < 0 ? 0 : (v > kMax ? kMax : v);
      *p = (PIX)v;
    }
  }
}

struct InvLaunch {
  hipStream_t stream;
  const int32_t *dq;
  const aomhip_txb *blocks;
  int n_blocks, grid_cols, uniform_type;
  const uint16_t *eob;
  void *dst_origin;
  int dst_stride;
};

template <int W, int H, int BD, typename PIX> static int launch_inv(const InvLaunch &l) {
  constexpr int LPB = W > H ? W : H;
  constexpr int BPW = kInvThreads / LPB;
  const int nwg = (l.n_blocks + BPW - 1) / BPW;
  const int nwg8 = (nwg + 7) & ~7;
  hipLaunchKernelGGL((inv_txfm_add_kernel<W, H, BD, PIX>), dim3(nwg8), dim3(kInvThreads), 0, l.stream, l.dq, l.blocks,
                     l.n_blocks, l.grid_cols, l.uniform_type, l.eob, static_cast<PIX *>(l.dst_origin), l.dst_stride,
                     nwg8);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

template <int BD, typename PIX> static int dispatch_inv(int tx_size, const InvLaunch &l) {
  switch (tx_size) {
    case 0: return launch_inv<4, 4, BD, PIX>(l);
    case 1: return launch_inv<8, 8, BD, PIX>(l);
    case 2: return launch_inv<16, 16, BD, PIX>(l);
    case 3: return launch_inv<32, 32, BD, PIX>(l);
    case 4: return launch_inv<64, 64, BD, PIX>(l);
    case 5: return launch_inv<4, 8, BD, PIX>(l);
    case 6: return launch_inv<8, 4, BD, PIX>(l);
    case 7: return launch_inv<8, 16, BD, PIX>(l);
    case 8: return launch_inv<16, 8, BD, PIX>(l);
    case 9: return launch_inv<16, 32, BD, PIX>(l);
    case 10: return launch_inv<32, 16, BD, PIX>(l);
    case 11: return launch_inv<32, 64, BD, PIX>(l);
    case 12: return launch_inv<64, 32, BD, PIX>(l);
    case 13: return launch_inv<4, 16, BD, PIX>(l);
    case 14: return launch_inv<16, 4, BD, PIX>(l);
    case 15: return launch_inv<8, 32, BD, PIX>(l);
    case 16: return launch_inv<32, 8, BD, PIX>(l);
    case 17: return launch_inv<16, 64, BD, PIX>(l);
    case 18: return launch_inv<64, 16, BD, PIX>(l);
  }
  set_error("bad tx_size %d", tx_size);
  return AOMHIP_ERR_INVALID;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_inv_txfm_add_batch(aomhip_ctx *ctx, const int32_t *d_dqcoeff, int tx_size, const aomhip_txb *d_blocks,
                              int n_blocks, int grid_cols, int uniform_tx_type, const uint16_t *d_eob,
                              const aomhip_planes *dst, int frame) {
  if (!ctx || !d_dqcoeff || !dst || !dst->base || tx_size < 0 || tx_size >= 19 || n_blocks < 0 || frame < 0 ||
      frame >= dst->n_frames || (!d_blocks && (grid_cols <= 0 || !tx_type_ok(tx_size, uniform_tx_type)))) {
    set_error("aomhip_inv_txfm_add_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  if (int rc = validate_txb_list(ctx, d_blocks, n_blocks, tx_size, true, false)) return rc;
  const size_t esz = dst->bit_depth == 8 ? 1 : 2;
  void *origin = static_cast<char *>(dst->base) +
                 ((size_t)frame * dst->frame_stride + (size_t)dst->border * dst->stride + dst->border) * esz;
  InvLaunch l{ ctx->stream, d_dqcoeff, d_blocks, n_blocks, grid_cols, uniform_tx_type, d_eob, origin, dst->stride };
  switch (dst->bit_depth) {
    case 8: return dispatch_inv<8, uint8_t>(tx_size, l);
    case 10: return dispatch_inv<10, uint16_t>(tx_size, l);
    default: return dispatch_inv<12, uint16_t>(tx_size, l);
  }
}

}  // extern "C"
