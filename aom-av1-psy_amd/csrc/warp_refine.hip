// av1_refine_warped_mv (av1/encoder/mcomp.c:3224-3293) for a batch of WARPED_CAUSAL blocks: the MV of a block whose prediction is a local affine warp
// is refined by +-1 (or +-2 without high-precision MVs) in 1/8 pel, twice, and every candidate MV means a NEW warp model -- av1_selectSamples +
// av1_find_projection on the block's neighbour samples -- whose prediction is measured against the source (compute_motion_cost, :3197-3221: the warped
// predictor, vf(pred, src) and the MV's rate).
//
// On the device a round of the refinement is, for all blocks at once: `wr_candidates_kernel` (one lane per (block, neighbour): range test, sample
// selection, least-squares fit, shear decomposition -- csrc/warp_fit.h, the arithmetic the host entry points share -- and the candidate's
// aomhip_warp_block record; a candidate without a usable model gets an empty rectangle), four `aomhip_warp_affine_batch` launches (neighbour j of every
// block into slot j of a scratch ring), ONE `aomhip_variance_batch` launch over the four slots against the source frame (seen as a four-frame ring of
// stride 0), and `wr_pick_kernel` (the reference's in-order `thismse < bestmse`, the centre's move, the `best_idx == -1` exit).  No host round trip.
#include "common.h"
#include "search_device.h"
#include "warp_error_table.inc"
#include "warp_fit.h"

namespace aomhip {
namespace {

__constant__ uint16_t kWrDivLut[257] = AOMHIP_DIV_LUT;

struct WrState {        // per block, between the rounds
  int16_t mv_row, mv_col;
  int32_t bestmse;      // (unsigned in the reference; compared as such)
  int32_t num_proj_ref, cur_np, active;
  aomhip_warp_model model;
};
struct WrCand {         // per (block, neighbour) of a round
  int32_t valid, num_proj_ref;
  int16_t mv_row, mv_col;
  aomhip_warp_model model;
};

__device__ __forceinline__ int wr_mv_cost(int mrow, int mcol, int ref_row, int ref_col, int cost_type, int error_per_bit, const int32_t *mvjcost,
                                          const int32_t *mvcost0, const int32_t *mvcost1) {   // mv_err_cost_ (mcomp.c:271-308)
  const int dr = mrow - ref_row, dc = mcol - ref_col;
  if (cost_type == kCostEntropy) {
    const int64_t bits = (int64_t)mvjcost[(dc != 0) | ((dr != 0) << 1)] + mvcost0[dr] + mvcost1[dc];
    return (int)((bits * error_per_bit + (1 << 13)) >> 14);
  }
  const int lambda = cost_type == kCostL1Low ? 2 : cost_type == kCostL1Hd ? 1 : 0;   // (L1_MIDRES: 0, mcomp.c:300)
  return (lambda * (iabsm(dr) + iabsm(dc))) >> 3;
}

__device__ __forceinline__ aomhip_warp_block wr_record(const aomhip_warp_model &m, int bx, int by, int w, int h) {
  aomhip_warp_block r;
#pragma unroll
  for (int k = 0; k < 6; ++k) r.mat[k] = m.mat[k];
  r.alpha = m.alpha; r.beta = m.beta; r.gamma = m.gamma; r.delta = m.delta;
  r.p_col = bx; r.p_row = by; r.p_width = w; r.p_height = h;
  return r;
}

// the centre: the block's own MV with the model it came with
__global__ void wr_init_kernel(const aomhip_warp_refine_block *__restrict__ blocks, int n, int bw, int bh, WrState *__restrict__ st, aomhip_warp_block *__restrict__ wb,
                               aomhip_var_cand *__restrict__ vc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_warp_refine_block b = blocks[i];
  WrState s;
  s.mv_row = b.mv_row; s.mv_col = b.mv_col; s.bestmse = 0; s.num_proj_ref = b.num_proj_ref; s.cur_np = b.num_proj_ref; s.active = 1; s.model = b.model;
  st[i] = s;
  wb[i] = wr_record(b.model, b.bx, b.by, bw, bh);
  aomhip_var_cand c;
  c.sx = c.rx = b.bx; c.sy = c.ry = b.by; c.xoff = c.yoff = 0; c.reserved[0] = c.reserved[1] = 0;
  vc[i] = c;
}

__global__ void wr_center_kernel(const aomhip_warp_refine_block *__restrict__ blocks, int n, const uint32_t *__restrict__ var, int cost_type, int error_per_bit,
                                 const int32_t *__restrict__ mvjcost, const int32_t *__restrict__ mvcost0, const int32_t *__restrict__ mvcost1,
                                 WrState *__restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_warp_refine_block b = blocks[i];
  st[i].bestmse = (int32_t)(var[i] + (uint32_t)wr_mv_cost(b.mv_row, b.mv_col, b.ref_row, b.ref_col, cost_type, error_per_bit, mvjcost, mvcost0, mvcost1));
}

// one lane per (block, neighbour j): the candidate's model, or an empty rectangle
__global__ void wr_candidates_kernel(const aomhip_warp_refine_block *__restrict__ blocks, int n, int bw, int bh, int start, const WrState *__restrict__ st,
                                     WrCand *__restrict__ cand, aomhip_warp_block *__restrict__ wb) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 4 * n) return;
  const int i = t >> 2, j = t & 3;
  const aomhip_warp_refine_block b = blocks[i];
  const WrState s = st[i];
  // neighbors[] (mcomp.c:3230-3231): { 0, -1 }, { 1, 0 }, { 0, 1 }, { -1, 0 } and the same doubled
  const int step = start ? 2 : 1;
  const int dr = (j == 1 ? 1 : j == 3 ? -1 : 0) * step, dc = (j == 0 ? -1 : j == 2 ? 1 : 0) * step;
  WrCand c;
  c.valid = 0; c.num_proj_ref = s.cur_np; c.mv_row = (int16_t)(s.mv_row + dr); c.mv_col = (int16_t)(s.mv_col + dc); c.model = s.model;
  aomhip_warp_model none = s.model;
  int w = 0, h = 0;
  if (s.active && c.mv_col >= b.col_min && c.mv_col <= b.col_max && c.mv_row >= b.row_min && c.mv_row <= b.row_max) {   // av1_is_subpelmv_in_range
    int pts[16], pin[16];
    const int total = b.total_samples;
#pragma unroll
    for (int k = 0; k < 16; ++k) { pts[k] = b.pts[k]; pin[k] = b.pts_inref[k]; }
    int np = s.cur_np;   // (a single sample: mbmi->num_proj_ref stays what it was)
    if (total > 1) np = wf_select_samples(c.mv_row, c.mv_col, pts, pin, total, bw, bh);
    c.num_proj_ref = np;
    if (wf_find_affine(np, pts, pin, bw, bh, c.mv_row, c.mv_col, b.by >> 2, b.bx >> 2, kWrDivLut, c.model.mat) &&
        wf_shear(c.model.mat, kWrDivLut, &c.model.alpha)) {
      c.valid = 1;
      w = bw; h = bh;
      none = c.model;
    }
  }
  cand[t] = c;
  wb[(int64_t)j * n + i] = wr_record(none, b.bx, b.by, w, h);
}

// the reference's walk over the four neighbours in order (`thismse < bestmse`), the centre's move, the exit when none was better
__global__ void wr_pick_kernel(const aomhip_warp_refine_block *__restrict__ blocks, int n, const WrCand *__restrict__ cand, const uint32_t *__restrict__ var,
                               int cost_type, int error_per_bit, const int32_t *__restrict__ mvjcost, const int32_t *__restrict__ mvcost0,
                               const int32_t *__restrict__ mvcost1, WrState *__restrict__ st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  WrState s = st[i];
  if (!s.active) return;
  const aomhip_warp_refine_block b = blocks[i];
  int best = -1;
  for (int j = 0; j < 4; ++j) {
    const WrCand c = cand[4 * i + j];
    // mbmi->num_proj_ref follows every in-range candidate's selection, used or not (it only matters for a block with one sample, where it never changes)
    if (!c.valid) continue;
    const uint32_t mse = var[(int64_t)j * n + i] + (uint32_t)wr_mv_cost(c.mv_row, c.mv_col, b.ref_row, b.ref_col, cost_type, error_per_bit, mvjcost, mvcost0, mvcost1);
    if (mse < (uint32_t)s.bestmse) {
      best = j;
      s.bestmse = (int32_t)mse;
      s.model = c.model;
      s.num_proj_ref = c.num_proj_ref;
    }
  }
  if (best < 0) s.active = 0;
  else {
    const WrCand c = cand[4 * i + best];
    s.mv_row = c.mv_row; s.mv_col = c.mv_col;
  }
  st[i] = s;
}

__global__ void wr_finish_kernel(int n, const WrState *__restrict__ st, aomhip_warp_refine_result *__restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const WrState s = st[i];
  aomhip_warp_refine_result r;
  r.mv_row = s.mv_row; r.mv_col = s.mv_col; r.num_proj_ref = s.num_proj_ref; r.bestmse = (uint32_t)s.bestmse; r.model = s.model;
  out[i] = r;
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_refine_warped_mv_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, const aomhip_planes *pred, int bw,
                                             int bh, int allow_hp, int mv_cost_type, int error_per_bit, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                                             const int32_t *d_mvcost_col, const aomhip_warp_refine_block *d_blocks, int n_blocks,
                                             aomhip_warp_refine_result *d_results) {
  if (!ctx || !src || !ref || !pred || !src->base || !ref->base || !pred->base || frame < 0 || frame >= src->n_frames || frame >= ref->n_frames ||
      pred->n_frames < 4 || pred->bit_depth != src->bit_depth || ref->bit_depth != src->bit_depth || pred->width != src->width || pred->height != src->height ||
      !valid_block(bw, bh) || bw < 8 || bh < 8 || n_blocks < 0 || (n_blocks > 0 && (!d_blocks || !d_results)) ||
      (mv_cost_type == kCostEntropy && (!d_mvjcost || !d_mvcost_row || !d_mvcost_col)) || mv_cost_type < 0 || mv_cost_type > kCostNone) {
    set_error("aomhip_refine_warped_mv_batch: invalid argument (blocks of at least 8x8 -- is_motion_variation_allowed_bsize --, a predictor ring of four frames with the source's geometry)");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  const size_t n1 = (size_t)n_blocks;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_st = take(n1 * sizeof(WrState)), o_cand = take(4 * n1 * sizeof(WrCand)), o_wb = take(4 * n1 * sizeof(aomhip_warp_block)),
               o_vc = take(n1 * sizeof(aomhip_var_cand)), o_var = take(4 * n1 * 4), o_sse = take(4 * n1 * 4);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  WrState *st = reinterpret_cast<WrState *>(w + o_st);
  WrCand *cand = reinterpret_cast<WrCand *>(w + o_cand);
  aomhip_warp_block *wb = reinterpret_cast<aomhip_warp_block *>(w + o_wb);
  aomhip_var_cand *vc = reinterpret_cast<aomhip_var_cand *>(w + o_vc);
  uint32_t *var = reinterpret_cast<uint32_t *>(w + o_var), *sse = reinterpret_cast<uint32_t *>(w + o_sse);
  // the source frame as a four-frame ring of stride 0: slot j of the predictor ring is measured against the same source picture
  aomhip_planes src4 = *src;
  src4.base = static_cast<char *>(src->base) + (int64_t)frame * src->frame_stride * (src->bit_depth == 8 ? 1 : 2);
  src4.frame_stride = 0;
  src4.n_frames = 4;
  const unsigned g1 = (unsigned)((n1 + 255) / 256), g4 = (unsigned)((4 * n1 + 255) / 256);
  hipLaunchKernelGGL(wr_init_kernel, dim3(g1), dim3(256), 0, ctx->stream, d_blocks, n_blocks, bw, bh, st, wb, vc);
  AOMHIP_LAUNCH_CHECK();
  // compute_motion_cost at the centre: vf(dst, src): diff = prediction - source
  int rc = aomhip_warp_affine_batch(ctx, ref, frame, pred, 0, 0, 0, wb, n_blocks, bw, bh);
  if (rc != AOMHIP_OK) return rc;
  rc = aomhip_variance_batch(ctx, pred, &src4, 0, 1, bw, bh, vc, n_blocks, 0, var, sse);
  if (rc != AOMHIP_OK) return rc;
  hipLaunchKernelGGL(wr_center_kernel, dim3(g1), dim3(256), 0, ctx->stream, d_blocks, n_blocks, var, mv_cost_type, error_per_bit, d_mvjcost, d_mvcost_row,
                     d_mvcost_col, st);
  AOMHIP_LAUNCH_CHECK();
  const int start = allow_hp ? 0 : 4;
  for (int ite = 0; ite < 2; ++ite) {
    hipLaunchKernelGGL(wr_candidates_kernel, dim3(g4), dim3(256), 0, ctx->stream, d_blocks, n_blocks, bw, bh, start, st, cand, wb);
    AOMHIP_LAUNCH_CHECK();
    for (int j = 0; j < 4; ++j) {
      rc = aomhip_warp_affine_batch(ctx, ref, frame, pred, j, 0, 0, wb + (size_t)j * n1, n_blocks, bw, bh);
      if (rc != AOMHIP_OK) return rc;
    }
    rc = aomhip_variance_batch(ctx, pred, &src4, 0, 4, bw, bh, vc, n_blocks, 0, var, sse);
    if (rc != AOMHIP_OK) return rc;
    hipLaunchKernelGGL(wr_pick_kernel, dim3(g1), dim3(256), 0, ctx->stream, d_blocks, n_blocks, cand, var, mv_cost_type, error_per_bit, d_mvjcost, d_mvcost_row,
                       d_mvcost_col, st);
    AOMHIP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(wr_finish_kernel, dim3(g1), dim3(256), 0, ctx->stream, n_blocks, st, d_results);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
