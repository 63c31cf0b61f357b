// av1_estimate_txfm_yrd (av1/encoder/tx_search.c:3016-3139) for a batch of inter blocks: the luma rate / distortion estimate the RD-based second-MV
// choice of av1_single_motion_search takes for each candidate (av1/encoder/motion_search_facade.c:378-425, disable_second_mv == 0).
//
// A composite of batched calls the library already has, chained on the context's stream, per transform block k of the block size (ONE for every
// block up to 64x64 -- max_txsize_rect_lookup[bsize] is the block's own size --, 2 or 4 transform blocks of 64x64 for the 128-class sizes):
//   yrd_prepare_kernel       get_txb_ctx (av1/common/txb_common.h:251-460, plane 0) on the block's RUNNING above / left entropy contexts
//   aomhip_subtract_xform_quant_ex_batch   av1_subtract_plane + av1_xform + av1_quant (DCT_DCT, AV1_XFORM_QUANT_B, no matrices) + the
//                            transform-domain error of dist_block_tx_domain (tx_search.c:1077-1113)
//   aomhip_cost_coeffs_txb_batch           cost_coeffs -> av1_cost_coeffs_txb (txb_rdopt.c:604-623)
//   aomhip_txb_entropy_context_batch       what av1_quant leaves in txb_entropy_ctx (encodemb.c:333-340)
//   yrd_accumulate_kernel    av1_merge_rd_stats + av1_set_txb_context: the contexts transform block k + 1 reads
// and yrd_finish_kernel: the function's tail (header rates, the forced-skip check).  ref_best_rd is INT64_MAX as in the caller this serves, so
// the early exits never trigger.  Blocks lie wholly inside the frame (mb_to_right_edge / mb_to_bottom_edge >= 0).
#include "common.h"

namespace aomhip {
namespace {

struct YrdAcc { int64_t dist, sse; int32_t rate, skip; };

__device__ __forceinline__ int64_t rdcost_d(int rdmult, int64_t rate, int64_t dist) { return ((rate * rdmult + 256) >> 9) + dist * 128; }   // RDCOST, rd.h:31-33

// transform block k of every block: its position + coefficient slot, its TXB_CTX from the running contexts (k == 0: the contexts are copied in first)
__global__ __launch_bounds__(256) void yrd_prepare_kernel(const aomhip_txfm_yrd_block *__restrict__ blocks, int n, int k, int txw, int txh, int cols, int whole,
                                                          int n_coef, uint8_t *__restrict__ run_ctx, YrdAcc *__restrict__ acc, aomhip_txb *__restrict__ txb,
                                                          uint8_t *__restrict__ txb_ctx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint8_t *ra = run_ctx + (size_t)i * 64, *rl = ra + 32;
  if (k == 0) {
    for (int j = 0; j < 32; ++j) { ra[j] = blocks[i].above_ctx[j]; rl[j] = blocks[i].left_ctx[j]; }
    acc[i] = YrdAcc{ 0, 0, 0, 1 };   // av1_init_rd_stats: skip_txfm = 1
  }
  const int row = k / cols, col = k - row * cols;
  txb[i].x = blocks[i].bx + col * txw;
  txb[i].y = blocks[i].by + row * txh;
  txb[i].out_offset = (uint32_t)i * (uint32_t)n_coef;
  txb[i].tx_type = 0;
  const uint8_t *a = ra + col * (txw >> 2), *l = rl + row * (txh >> 2);
  int dc_sign = 0, top = 0, left = 0;
  for (int j = 0; j < (txw >> 2); ++j) { const int s = a[j] >> 3; dc_sign += s == 1 ? -1 : (s == 2 ? 1 : 0); top |= a[j]; }
  for (int j = 0; j < (txh >> 2); ++j) { const int s = l[j] >> 3; dc_sign += s == 1 ? -1 : (s == 2 ? 1 : 0); left |= l[j]; }
  top = min(top & 7, 4); left = min(left & 7, 4);
  // skip_contexts[5][5] (txb_common.h:330-334): rows / columns {0}, {1, 2, 3}, {4}
  const int tc = top == 0 ? 0 : (top == 4 ? 2 : 1), lc = left == 0 ? 0 : (left == 4 ? 2 : 1);
  const int tbl[3][3] = { { 1, 2, 3 }, { 2, 4, 5 }, { 3, 5, 6 } };
  txb_ctx[2 * i] = (uint8_t)(whole ? 0 : tbl[tc][lc]);
  txb_ctx[2 * i + 1] = (uint8_t)(dc_sign < 0 ? 1 : (dc_sign > 0 ? 2 : 0));
}

__global__ __launch_bounds__(256) void yrd_accumulate_kernel(int n, int k, int txw, int txh, int cols, int shift, int tx_type_rate, const uint16_t *__restrict__ eob,
                                                             const int64_t *__restrict__ err, const int32_t *__restrict__ cost, const uint8_t *__restrict__ ectx,
                                                             uint8_t *__restrict__ run_ctx, YrdAcc *__restrict__ acc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int e = eob[i];
  const int64_t d = err[2 * i], s = err[2 * i + 1];
  YrdAcc v = acc[i];
  v.rate += cost[i] + (e ? tx_type_rate : 0);                // get_tx_type_cost sits behind av1_cost_coeffs_txb's eob == 0 return
  v.dist += shift < 0 ? d << -shift : d >> shift;            // RIGHT_SIGNED_SHIFT
  v.sse += shift < 0 ? s << -shift : s >> shift;
  v.skip &= e == 0;
  acc[i] = v;
  const int row = k / cols, col = k - row * cols;
  uint8_t *a = run_ctx + (size_t)i * 64 + col * (txw >> 2), *l = run_ctx + (size_t)i * 64 + 32 + row * (txh >> 2);
  for (int j = 0; j < (txw >> 2); ++j) a[j] = ectx[i];       // av1_set_txb_context
  for (int j = 0; j < (txh >> 2); ++j) l[j] = ectx[i];
}

__global__ __launch_bounds__(256) void yrd_finish_kernel(const aomhip_txfm_yrd_block *__restrict__ blocks, int n, int rdmult, int lossless, const YrdAcc *__restrict__ acc,
                                                         aomhip_txfm_yrd_stats *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const YrdAcc v = acc[i];
  const aomhip_txfm_yrd_block b = blocks[i];
  int64_t rate = v.rate, dist = v.dist, rd;
  int skip = v.skip;
  if (skip) {
    rd = rdcost_d(rdmult, b.skip_txfm_rate, v.sse);
  } else {
    rd = rdcost_d(rdmult, rate + b.no_skip_txfm_rate + b.tx_size_rate, dist);
    rate += b.tx_size_rate;
  }
  if (!skip && !lossless) {   // does forcing the block to skip its transform cost less?
    const int64_t t = rdcost_d(rdmult, b.skip_txfm_rate, v.sse);
    if (t <= rd) { rd = t; rate = 0; dist = v.sse; skip = 1; }
  }
  out[i].rd = rd; out[i].dist = dist; out[i].sse = v.sse; out[i].rate = (int32_t)rate; out[i].skip_txfm = skip;
}

int tx_size_of(int w, int h) {
  static const int tw[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
  static const int th[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
  for (int i = 0; i < 19; ++i)
    if (tw[i] == w && th[i] == h) return i;
  return -1;
}

}  // namespace

size_t yrd_workspace_bytes(int n_blocks, int bw, int bh) {
  const int txw = bw > 64 ? 64 : bw, txh = bh > 64 ? 64 : bh;
  const size_t n = (size_t)n_blocks, nc = (size_t)aomhip_tx_max_eob(tx_size_of(txw, txh));
  return ((n * sizeof(aomhip_txb) + 255) & ~(size_t)255) + 3 * ((n * nc * 4 + 255) & ~(size_t)255) + ((n * 2 + 255) & ~(size_t)255) + ((n * 16 + 255) & ~(size_t)255) +
         ((n * 2 + 255) & ~(size_t)255) + ((n * 4 + 255) & ~(size_t)255) + ((n + 255) & ~(size_t)255) + ((n * 64 + 255) & ~(size_t)255) +
         ((n * sizeof(YrdAcc) + 255) & ~(size_t)255);
}

// the composite on caller-provided work memory (aomhip_single_motion_search_batch carves it out of its own)
int estimate_txfm_yrd_ws(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *pred, int frame, int bw, int bh, const aomhip_quant_params *qparams,
                         const int32_t *d_costs, int tx_type_rate, int rdmult, int lossless, const aomhip_txfm_yrd_block *d_blocks, int n_blocks,
                         aomhip_txfm_yrd_stats *d_stats, char *ws) {
  const int txw = bw > 64 ? 64 : bw, txh = bh > 64 ? 64 : bh;
  const int tx_size = tx_size_of(txw, txh);
  const int nc = aomhip_tx_max_eob(tx_size);
  const int pels = txw * txh, scale = (pels > 256) + (pels > 1024);
  const int shift = (1 - scale) * 2;   // (MAX_TX_SCALE - av1_get_tx_scale(tx_size)) * 2
  const size_t n = (size_t)n_blocks;
  auto take = [&](size_t bytes) { char *p = ws; ws += (bytes + 255) & ~(size_t)255; return p; };
  aomhip_txb *d_txb = reinterpret_cast<aomhip_txb *>(take(n * sizeof(aomhip_txb)));
  int32_t *d_coeff = reinterpret_cast<int32_t *>(take(n * nc * 4)), *d_q = reinterpret_cast<int32_t *>(take(n * nc * 4)), *d_dq = reinterpret_cast<int32_t *>(take(n * nc * 4));
  uint16_t *d_eob = reinterpret_cast<uint16_t *>(take(n * 2));
  int64_t *d_err = reinterpret_cast<int64_t *>(take(n * 16));
  uint8_t *d_tctx = reinterpret_cast<uint8_t *>(take(n * 2));
  int32_t *d_cost = reinterpret_cast<int32_t *>(take(n * 4));
  uint8_t *d_ectx = reinterpret_cast<uint8_t *>(take(n));
  uint8_t *d_run = reinterpret_cast<uint8_t *>(take(n * 64));
  YrdAcc *d_acc = reinterpret_cast<YrdAcc *>(take(n * sizeof(YrdAcc)));
  const int cols = bw / txw, rows = bh / txh;
  const dim3 grid((unsigned)((n_blocks + 255) / 256)), block(256);
  const int tx_type = (lossless && bw == 4 && bh == 4) ? 16 : 0;   // lossless 4x4: the Walsh-Hadamard transform (av1_fwd_txfm, hybrid_fwd_txfm.c)
  for (int k = 0; k < cols * rows; ++k) {
    hipLaunchKernelGGL(yrd_prepare_kernel, grid, block, 0, ctx->stream, d_blocks, n_blocks, k, txw, txh, cols, (int)(cols * rows == 1), nc, d_run, d_acc, d_txb, d_tctx);
    AOMHIP_LAUNCH_CHECK();
    int rc = aomhip_subtract_xform_quant_ex_batch(ctx, src, pred, frame, tx_size, d_txb, n_blocks, 0, tx_type, qparams, AOMHIP_QUANT_B, d_coeff, d_q, d_dq, d_eob, d_err);
    if (rc != AOMHIP_OK) return rc;
    rc = aomhip_cost_coeffs_txb_batch(ctx, d_q, tx_size, d_txb, n_blocks, tx_type, d_eob, d_tctx, d_costs, d_cost);
    if (rc != AOMHIP_OK) return rc;
    rc = aomhip_txb_entropy_context_batch(ctx, d_q, tx_size, d_txb, n_blocks, tx_type, d_eob, d_ectx);
    if (rc != AOMHIP_OK) return rc;
    hipLaunchKernelGGL(yrd_accumulate_kernel, grid, block, 0, ctx->stream, n_blocks, k, txw, txh, cols, shift, tx_type_rate, d_eob, d_err, d_cost, d_ectx, d_run, d_acc);
    AOMHIP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(yrd_finish_kernel, grid, block, 0, ctx->stream, d_blocks, n_blocks, rdmult, lossless, d_acc, d_stats);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_estimate_txfm_yrd_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *pred, int frame, int bw, int bh,
                                              const aomhip_quant_params *qparams, const int32_t *d_costs, int tx_type_rate, int rdmult, int lossless,
                                              const aomhip_txfm_yrd_block *d_blocks, int n_blocks, aomhip_txfm_yrd_stats *d_stats) {
  if (!ctx || !src || !pred || !src->base || !pred->base || !qparams || !d_costs || n_blocks < 0 || (n_blocks > 0 && (!d_blocks || !d_stats)) || frame < 0 ||
      frame >= src->n_frames || frame >= pred->n_frames || src->bit_depth != pred->bit_depth || !valid_block(bw, bh) ||
      tx_size_of(bw > 64 ? 64 : bw, bh > 64 ? 64 : bh) < 0) {
    set_error("aomhip_estimate_txfm_yrd_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  char *ws = static_cast<char *>(work(ctx, yrd_workspace_bytes(n_blocks, bw, bh)));
  if (!ws) return AOMHIP_ERR_NOMEM;
  return estimate_txfm_yrd_ws(ctx, src, pred, frame, bw, bh, qparams, d_costs, tx_type_rate, rdmult, lossless, d_blocks, n_blocks, d_stats, ws);
}
