// The block-local middle of the encode loop as ONE kernel: for an inter block whose prediction block is one square transform block,
//   av1_enc_build_inter_predictor (av1/encoder/reconinter_enc.c:47-51; 8-tap 2-D convolve, av1/common/convolve.c:92-125)
//   -> av1_subtract_plane / aom_[highbd_]subtract_block (av1/encoder/encodemb.c:53-77)
//   -> av1_xform_quant (encodemb.c:288-341: av1_fwd_txfm2d_WxW + aom_[highbd_]quantize_b)
//   -> av1_inverse_transform_block + the clipped add (encodemb.c:454-459, av1/common/av1_inv_txfm2d.c:234-316)
// i.e. encode_block (encodemb.c:343-470) behind its predictor.  The three separate device calls (aomhip_build_inter_pred_batch,
// aomhip_subtract_xform_quant_batch, aomhip_inv_txfm_add_batch) write the prediction, read it back with the source, write the coefficients,
// read them back and read-modify-write the prediction: three launches and three trips through memory for data that one group of W lanes
// holds throughout.  Here the W lanes of a block keep
//   * the predicted column (lane c walks its column down the W + 7 reference rows exactly as inter_pred_kernel does) in registers,
//   * the residual column = source column - predicted column: already the forward column pass's input,
//   * after the LDS transpose and the row pass + quantiser, row r's dequantised coefficients: already the inverse row pass's input,
//   * after the second transpose, column c of the inverse: added to the predicted column still in registers and written ONCE.
// Memory traffic per block: the reference footprint and the source block in; the reconstruction, qcoeff, dqcoeff (each optional) and eob out.
// The arithmetic is the three kernels' own (pred_device.h, quant_device.h, txfm_device.h): bit-exact against the chain by construction,
// tests/test_gpu_encode_block.py compares it with the chain and with the oracle.
#include "common.h"
#include "pred_device.h"
#include "quant_device.h"
#include "txfm_device.h"

namespace aomhip {
namespace {

using namespace txfm;

constexpr int kEbThreads = 256;

struct EbArgs {
  int ref_frame, n_blocks, nwg8;
  int set_x, set_y, bit_depth, x_lo, x_hi, y_lo, y_hi;
  int src_stride, rec_stride, tx_type;
};

template <int W> constexpr int eb_threads() { return W >= 32 ? 128 : kEbThreads; }   // (a 32x32 block needs 12.4 KB of LDS: four per workgroup)

template <typename T, int W, int BD>
__global__ __launch_bounds__(eb_threads<W>()) void encode_inter_block_kernel(PlaneView<T> ref, const T *__restrict__ src_origin, T *__restrict__ rec_origin,
                                                                        const aomhip_search_block *__restrict__ blocks, const int16_t *__restrict__ mv,
                                                                        EbArgs a, QuantArgs qa, int32_t *__restrict__ qcoeff,
                                                                        int32_t *__restrict__ dqcoeff, uint16_t *__restrict__ eob) {
  constexpr int H = W;
  using C = Cfg2D<W, H>;
  constexpr bool HBD = sizeof(T) == 2;
  constexpr int BPW = eb_threads<W>() / W;   // blocks per workgroup
  constexpr int NC = W * H;
  constexpr int LS = (NC > 256) + (NC > 1024);
  constexpr int LSTRIDE = W + 1;
  constexpr int TILE = (H * LSTRIDE + 3) & ~3;
  // av1_gen_inv_stage_range (av1_inv_txfm2d.c:188-232)
  constexpr int RNG_ROW = BD == 8 ? 16 : BD == 10 ? 18 : 20;
  constexpr int RNG_COL = BD == 8 ? 16 : BD == 10 ? 16 : 18;
  constexpr int COL_CLAMP = BD + 6 > 16 ? BD + 6 : 16;
  // per block: Q (qcoeff staging, NC words), D (forward transpose tile, then dqcoeff staging); once both are copied out the inverse's
  // transpose tile I takes their place (LDS, not registers, bounds the wavefronts per SIMD here: 2.1 KB per block = 4 workgroups per CU)
  __shared__ __attribute__((aligned(16))) int32_t lds[BPW][NC + TILE];

  const int slot = threadIdx.x / W, lane = threadIdx.x % W;
  const unsigned wg = xcd_chunked_index(blockIdx.x, a.nwg8);
  const int bi = wg * BPW + slot;
  const bool live = bi < a.n_blocks;
  int32_t *Q = lds[slot], *D = lds[slot] + NC, *I = lds[slot];
  const int tx_type = a.tx_type;

  // ---- 1. prediction: lane = column; the W predicted pixels of the column stay in pr[] (inter_pred_kernel's walk, pred.hip)
  int pr[H];
  int bx = 0, by = 0;
  if (live) {
    bx = blocks[bi].bx;
    by = blocks[bi].by;
    int pos_x = (bx << 4) + mv[2 * bi + 1] * 2, pos_y = (by << 4) + mv[2 * bi] * 2;   // init_subpel_params (reconinter.h:130-165), luma, unscaled
    pos_x = min(max(pos_x, a.x_lo), a.x_hi);
    pos_y = min(max(pos_y, a.y_lo), a.y_hi);
    constexpr int tbd = BD;
    constexpr int r0 = tbd == 12 ? 5 : 3, r1 = 14 - r0;  // get_conv_params_no_round (convolve.h:72-81)
    constexpr int ob = tbd + 14 - r0;
    constexpr int hoff = (1 << (tbd + 6)) + ((1 << r0) >> 1);
    constexpr int voff = (1 << ob) + ((1 << r1) >> 1);
    constexpr int vsub = (1 << (ob - r1)) + (1 << (ob - r1 - 1));
    constexpr int pmax = (1 << tbd) - 1;
    const PU128 fx = *reinterpret_cast<const PU128 *>(&kInterp[a.set_x][pos_x & 15][0]);
    const PU128 fy = *reinterpret_cast<const PU128 *>(&kInterp[a.set_y][pos_y & 15][0]);
    const T *p = ref.origin + (int64_t)a.ref_frame * ref.frame_stride + (int64_t)((pos_y >> 4) - 3) * ref.stride + (pos_x >> 4) - 3 + lane;
    uint32_t win[4] = { 0, 0, 0, 0 };
#pragma unroll
    for (int r = 0; r < H + 7; ++r) {
      uint32_t px[4];
      load8_pairs<T>(p, px);
      p += ref.stride;
      int hs = hoff;
#pragma unroll
      for (int k = 0; k < 4; ++k) hs = dot2(px[k], fx.v[k], hs);
#pragma unroll
      for (int k = 0; k < 3; ++k) win[k] = __builtin_amdgcn_alignbit(win[k + 1], win[k], 16);
      win[3] = __builtin_amdgcn_alignbit((uint32_t)(hs >> r0), win[3], 16);
      if (r >= 7) {
        int vs = voff;
#pragma unroll
        for (int k = 0; k < 4; ++k) vs = dot2(win[k], fy.v[k], vs);
        int res = (vs >> r1) - vsub;
        if constexpr (!HBD) res = (int16_t)res;  // convolve.c:119 keeps it in an int16_t
        pr[r - 7] = min(max(res, 0), pmax);
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < H; ++r) pr[r] = 0;
  }

  // ---- 2. residual column (aom_[highbd_]subtract_block) = the forward column pass's input
  const int vk = v_kind(tx_type), hk = h_kind(tx_type);
  int32_t x[H];
  int amax = 0;
  if (live) {
    const T *s = src_origin + (int64_t)by * a.src_stride + bx + lane;
#pragma unroll
    for (int r = 0; r < H; ++r) {
      x[r] = (int)s[(int64_t)r * a.src_stride] - pr[r];
      amax = max(amax, max(x[r], -x[r]));
    }
  } else {
#pragma unroll
    for (int r = 0; r < H; ++r) x[r] = 0;
  }
  const bool fast = group_max<W>(amax) <= kSafeMax[tx_index_of(W, H)][tx_type & 15];

  // ---- 3. forward columns -> transpose tile D
  if (live) {
    if (vk == 2) {
#pragma unroll
      for (int r = 0; r < H / 2; ++r) { const int32_t t = x[r]; x[r] = x[H - 1 - r]; x[H - 1 - r] = t; }
    }
#pragma unroll
    for (int r = 0; r < H; ++r) x[r] = x[r] * (1 << C::fs0);
    fwd_1d_sel<H, C::cos_bit_col>(x, vk == 2 ? 1 : vk, fast);
    const int dc = (hk == 2) ? W - 1 - lane : lane;
#pragma unroll
    for (int r = 0; r < H; ++r) {
      int32_t v = x[r];
      if constexpr (C::fs1 < 0) v = rshift(v, -C::fs1);
      D[r * LSTRIDE + dc] = v;
    }
  }
  block_sync();

  // ---- 4. forward rows + quantise: lane = row r; the dequantised row stays in y[] for the inverse
  int32_t y[W];
  int my_eob = 0;
  const int r_ = lane;
  if (live) {
#pragma unroll
    for (int c = 0; c < W; ++c) y[c] = D[r_ * LSTRIDE + c];
  }
  block_sync();   // D (the tile) is dead from here: it becomes the dqcoeff staging area
  if (live) {
    fwd_1d_sel<W, C::cos_bit_row>(y, hk == 2 ? 1 : hk, fast);
    const int scan_class = tx_type < 10 ? 0 : ((tx_type & 1) ? 2 : 1);
    const int zb[2] = { (qa.zbin[0] + ((1 << LS) >> 1)) >> LS, (qa.zbin[1] + ((1 << LS) >> 1)) >> LS };
    const int rd[2] = { (qa.round[0] + ((1 << LS) >> 1)) >> LS, (qa.round[1] + ((1 << LS) >> 1)) >> LS };
    int last_c = -1;
#pragma unroll
    for (int c = 0; c < W; ++c) {
      int32_t v = y[c];
      if constexpr (C::fs2 < 0) v = rshift(v, -C::fs2);
      const int rc = c * H + r_;
      const int ac = (c == 0) ? (r_ != 0) : 1;
      int32_t qv, dqv;
      quantize_one<HBD, LS>(v, zb[ac], rd[ac], qa.quant[ac], qa.quant_shift[ac], qa.qs_log2[ac], qa.dequant[ac], &qv, &dqv);
      last_c = qv ? c : last_c;
      Q[rc] = qv;
      D[rc] = dqv;
      y[c] = dqv;
    }
    if (last_c >= 0) my_eob = iscan_pos<W, H>(r_, last_c, scan_class) + 1;
  }
  my_eob = group_max<W>(my_eob);
  if (live && lane == 0) eob[bi] = (uint16_t)my_eob;
  const bool has_coeffs = live && my_eob > 0;   // (uniform over the block's lanes)

  block_sync();

  // ---- 5. coefficients out, 16 bytes per lane per store (Q and D hold them in the reference's order)
  if (live) {
#pragma unroll
    for (int k = 0; k < (NC / 4 + W - 1) / W; ++k) {
      const int i = lane + k * W;
      if (i < NC / 4) {
        if (qcoeff) {
          const uint4 v = *reinterpret_cast<const uint4 *>(Q + 4 * i);
          xq_store4(qcoeff + (int64_t)bi * NC + 4 * i, v.x, v.y, v.z, v.w);
        }
        if (dqcoeff) {
          const uint4 v = *reinterpret_cast<const uint4 *>(D + 4 * i);
          xq_store4(dqcoeff + (int64_t)bi * NC + 4 * i, v.x, v.y, v.z, v.w);
        }
      }
    }
  }
  block_sync();   // Q and D are dead from here

  // ---- 6. inverse rows (av1_inverse_transform_block: nothing to add when eob == 0) -> transpose tile I (over Q / D)
  if (has_coeffs) {
    const int ihk = ih_kind(tx_type);
#pragma unroll
    for (int c = 0; c < W; ++c) y[c] = clampv<BD + 8>(y[c]);
    inv_1d<W, 12, RNG_ROW>(y, ihk == 2 ? 1 : ihk);
#pragma unroll
    for (int c = 0; c < W; ++c) {
      int32_t v = y[c];
      if constexpr (C::is0 < 0) v = rshift(v, -C::is0);
      I[r_ * LSTRIDE + c] = v;
    }
  }
  block_sync();

  // ---- 7. inverse columns + the clipped add onto the predicted column (highbd_clip_pixel_add, av1_txfm.h:104-107); lane = column
  if (live) {
    constexpr int kMax = (1 << BD) - 1;
    if (has_coeffs) {
      const int ivk = iv_kind(tx_type), ihk = ih_kind(tx_type);
      const int sc = (ihk == 2) ? W - 1 - lane : lane;
      int32_t z[H];
#pragma unroll
      for (int r = 0; r < H; ++r) z[r] = clampv<COL_CLAMP>(I[r * LSTRIDE + sc]);
      inv_1d<H, 12, RNG_COL>(z, ivk == 2 ? 1 : ivk);
      const bool ud = (ivk == 2);
#pragma unroll
      for (int r = 0; r < H; ++r) {
        const int32_t res = rshift(ud ? z[H - 1 - r] : z[r], 4);
        const int v = pr[r] + res;
        pr[r] = v < 0 ? 0 : (v > kMax ? kMax : v);
      }
    }
    T *d = rec_origin + (int64_t)by * a.rec_stride + bx + lane;
#pragma unroll
    for (int r = 0; r < H; ++r) d[(int64_t)r * a.rec_stride] = (T)pr[r];
  }
}

template <typename T, int W, int BD>
int launch_eb(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *ref, int ref_frame, const aomhip_planes *rec,
              int rec_frame, const aomhip_search_block *d_blocks, const int16_t *d_mv, int n, int fx, int fy, int tx_type, const QuantArgs &qa,
              int32_t *q, int32_t *dq, uint16_t *eob) {
  constexpr int BPW = eb_threads<W>() / W;
  EbArgs a;
  a.ref_frame = ref_frame; a.n_blocks = n;
  const int nwg = (n + BPW - 1) / BPW;
  a.nwg8 = (nwg + 7) & ~7;
  a.set_x = fx; a.set_y = fy; a.bit_depth = BD;
  // the 8-pixel row load starts 3 left of the integer position and the walk covers rows -3 .. W + 3 (pred.hip: launch_inter_pred)
  a.x_lo = (-ref->border + 3) << 4; a.x_hi = ((ref->width + ref->border - W - 5) << 4) | 15;
  a.y_lo = (-ref->border + 3) << 4; a.y_hi = ((ref->height + ref->border - W - 5) << 4) | 15;
  a.src_stride = src->stride; a.rec_stride = rec->stride; a.tx_type = tx_type;
  const T *s = reinterpret_cast<const T *>(src->base) + (size_t)src_frame * src->frame_stride + (size_t)src->border * src->stride + src->border;
  T *r = reinterpret_cast<T *>(rec->base) + (size_t)rec_frame * rec->frame_stride + (size_t)rec->border * rec->stride + rec->border;
  hipLaunchKernelGGL((encode_inter_block_kernel<T, W, BD>), dim3(a.nwg8), dim3(eb_threads<W>()), 0, ctx->stream, view_of<T>(*ref), s, r, d_blocks, d_mv,
                     a, qa, q, dq, eob);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_encode_inter_blocks_batch(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *ref, int ref_frame,
                                                const aomhip_planes *recon, int recon_frame, int bw, const aomhip_search_block *d_blocks,
                                                const int16_t *d_mv, int n_blocks, int interp_filter_x, int interp_filter_y, int tx_type,
                                                const aomhip_quant_params *qparams, int32_t *d_qcoeff, int32_t *d_dqcoeff, uint16_t *d_eob) {
  const int tx_size = bw == 8 ? 1 : bw == 16 ? 2 : bw == 32 ? 3 : -1;
  if (!ctx || !src || !ref || !recon || !src->base || !ref->base || !recon->base || !qparams || !d_eob || n_blocks < 0 ||
      (n_blocks > 0 && (!d_blocks || !d_mv)) || src_frame < 0 || src_frame >= src->n_frames || ref_frame < 0 || ref_frame >= ref->n_frames ||
      recon_frame < 0 || recon_frame >= recon->n_frames || tx_size < 0 || src->bit_depth != ref->bit_depth || src->bit_depth != recon->bit_depth ||
      interp_filter_x < 0 || interp_filter_x > 3 || interp_filter_y < 0 || interp_filter_y > 3 || tx_type < 0 || tx_type > 15 ||
      !tx_type_ok(tx_size, tx_type)) {
    set_error("aomhip_encode_inter_blocks_batch: invalid argument (square blocks of 8, 16 or 32 pixels; a transform type of that size)");
    return AOMHIP_ERR_INVALID;
  }
  if (ref->border < 8) {
    set_error("aomhip_encode_inter_blocks_batch: the reference planes need a border of at least 8 pixels for the 8-tap kernels");
    return AOMHIP_ERR_INVALID;
  }
  if (recon->base == ref->base && recon_frame == ref_frame) {
    set_error("aomhip_encode_inter_blocks_batch: the reconstruction cannot be written over the reference frame it is predicted from");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  const QuantArgs qa = to_quant_args(qparams, 0);
#define RUN(T, W, BD) \
  return launch_eb<T, W, BD>(ctx, src, src_frame, ref, ref_frame, recon, recon_frame, d_blocks, d_mv, n_blocks, interp_filter_x, interp_filter_y, tx_type, qa, d_qcoeff, d_dqcoeff, d_eob)
#define BY_W(T, BD) \
  switch (bw) { case 8: RUN(T, 8, BD); case 16: RUN(T, 16, BD); default: RUN(T, 32, BD); }
  switch (src->bit_depth) {
    case 8: BY_W(uint8_t, 8)
    case 10: BY_W(uint16_t, 10)
    default: BY_W(uint16_t, 12)
  }
#undef BY_W
#undef RUN
}
