// tf_motion_search (av1/encoder/temporal_filter.c:87-253) for every 32x32 block of a frame, one filter window per call:
// aomhip_tf_motion_search_frames.  The searches themselves are the library's batched av1_full_pixel_search and sub-pel
// tree kernels; this file is the frame loop of av1_tf_do_filtering_row (:849-867) turned inside out (per frame, all blocks)
// and the small element-wise kernels between the searches -- start MVs, limits, the block / sub-block bookkeeping, the partition
// decision and the ref_mv hand-over -- so that the whole chain stays in device memory with no host round trip.
#include <climits>

#include "common.h"
#include "fullpel_search.h"
#include "search_device.h"

namespace aomhip {
namespace {

constexpr int kTfBlock = 32, kTfSub = 16;
constexpr int kMaxFullPel = 1023;          // MAX_FULL_PEL_VAL (mcomp_structs.h:22)
constexpr int kMvLow = -(1 << 14), kMvUpp = 1 << 14;  // MV_LOW / MV_UPP (entropymv.h:75-76)

__device__ __forceinline__ int rawpel(int x) { return (x + 3 + (x >= 0)) >> 3; }  // GET_MV_RAWPEL (mv.h:28)
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// FullMvLimits of a search around the zero baseline MV: av1_set_mv_search_range(&mv_limits, &kZeroMv) (mcomp.c:196-215)
__device__ __forceinline__ void full_limits(const aomhip_search_block &in, aomhip_search_block *out) {
  const int lo = -kMaxFullPel > rawpel(kMvLow) + 1 ? -kMaxFullPel : rawpel(kMvLow) + 1;
  const int hi = kMaxFullPel < rawpel(kMvUpp) - 1 ? kMaxFullPel : rawpel(kMvUpp) - 1;
  out->row_min = (int16_t)(in.row_min < lo ? lo : in.row_min);
  out->row_max = (int16_t)(in.row_max > hi ? hi : in.row_max);
  out->col_min = (int16_t)(in.col_min < lo ? lo : in.col_min);
  out->col_max = (int16_t)(in.col_max > hi ? hi : in.col_max);
}
// SubpelMvLimits: av1_set_subpel_mv_search_range(.., &x->mv_limits, &kZeroMv) (mcomp.h:344-361)
__device__ __forceinline__ void subpel_limits(const aomhip_search_block &in, aomhip_search_block *out) {
  const int max_mv = kMaxFullPel * 8;
  auto lo = [&](int v) { int m = v * 8 > -max_mv ? v * 8 : -max_mv; return m > kMvLow + 1 ? m : kMvLow + 1; };
  auto hi = [&](int v) { int m = v * 8 < max_mv ? v * 8 : max_mv; return m < kMvUpp - 1 ? m : kMvUpp - 1; };
  out->row_min = (int16_t)lo(in.row_min); out->row_max = (int16_t)hi(in.row_max);
  out->col_min = (int16_t)lo(in.col_min); out->col_max = (int16_t)hi(in.col_max);
}

// full-pel list of the 32x32 blocks: start = get_fullmv_from_mv(ref_mv) (:131), baseline MV 0
__global__ void tf_full32_list_kernel(const aomhip_search_block *blocks, const int16_t *ref_mv, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aomhip_search_block b = blocks[i], o;
  o.bx = b.bx; o.by = b.by; o.ref_row = 0; o.ref_col = 0;
  o.start_row = (int16_t)rawpel(ref_mv[2 * i]); o.start_col = (int16_t)rawpel(ref_mv[2 * i + 1]);
  full_limits(b, &o);
  out[i] = o;
}

// sub-pel list from a full-pel result: subpel_start_mv = get_mv_from_fullmv(best) (:187, :232); `per` entries of `mv` per block of
// `blocks` (1 for the block itself, 4 for its sub-blocks, whose origin is the block's + (i, j) * 16 while the limits stay the block's)
__global__ void tf_subpel_list_kernel(const aomhip_search_block *blocks, const int16_t *full_mv, int n, int per, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * per) return;
  const aomhip_search_block b = blocks[i / per];
  const int k = i % per;
  aomhip_search_block o;
  o.bx = (int16_t)(b.bx + (per == 4 ? (k & 1) * kTfSub : 0));
  o.by = (int16_t)(b.by + (per == 4 ? (k >> 1) * kTfSub : 0));
  o.ref_row = 0; o.ref_col = 0;
  o.start_row = (int16_t)(full_mv[2 * i] * 8); o.start_col = (int16_t)(full_mv[2 * i + 1] * 8);
  subpel_limits(b, &o);
  out[i] = o;
}

// after the block's sub-pel search: block_mse (:190), *ref_mv = block MV (:192), and the full-pel list of the four sub-blocks started
// at get_fullmv_from_mv(ref_mv) (:198)
__global__ void tf_after_block_kernel(const aomhip_search_block *blocks, const int16_t *block_mv, const uint32_t *block_err, int n, int mse_thresh,
                                      int16_t *ref_mv, int32_t *block_mse, int16_t *block_mv_keep, aomhip_search_block *sub_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  const int row = block_mv[2 * i], col = block_mv[2 * i + 1];
  const int bmse = (int32_t)((block_err[i] + (unsigned)(kTfBlock * kTfBlock / 2)) / (unsigned)(kTfBlock * kTfBlock));  // DIVIDE_AND_ROUND, unsigned
  block_mse[i] = bmse;
  block_mv_keep[2 * i] = (int16_t)row; block_mv_keep[2 * i + 1] = (int16_t)col;   // (this frame's own copy: the sub-block chain reads it while the next frame's block search runs)
  // *ref_mv = block MV (:192), then the caller's rule (:249-252) -- which reads block_mse only, so the NEXT frame's 32x32 search does not wait
  // for this frame's sub-block searches
  const bool zero = bmse > mse_thresh;
  ref_mv[2 * i] = zero ? (int16_t)0 : (int16_t)row; ref_mv[2 * i + 1] = zero ? (int16_t)0 : (int16_t)col;
  aomhip_search_block o;
  o.ref_row = 0; o.ref_col = 0;
  o.start_row = (int16_t)rawpel(row); o.start_col = (int16_t)rawpel(col);
  full_limits(b, &o);
  for (int k = 0; k < 4; ++k) {
    o.bx = (int16_t)(b.bx + (k & 1) * kTfSub); o.by = (int16_t)(b.by + (k >> 1) * kTfSub);
    sub_out[4 * i + k] = o;
  }
}

// tf_determine_block_partition (:270-293) + the ref_mv rule (:249-252); writes the frame's outputs
__global__ void tf_finish_kernel(const int16_t *block_mv, const int32_t *block_mse, const int16_t *sub_mv, const uint32_t *sub_err, int n,
                                 int have_sub, int mse_thresh, int16_t *ref_mv /* null: tf_after_block_kernel applied the rule */, int16_t *out_mvs,
                                 int32_t *out_mses) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int mses[4], mvs[4][2];
  for (int k = 0; k < 4; ++k) {
    if (have_sub) {
      mses[k] = (int32_t)((sub_err[4 * i + k] + (unsigned)(kTfSub * kTfSub / 2)) / (unsigned)(kTfSub * kTfSub));
      mvs[k][0] = sub_mv[8 * i + 2 * k]; mvs[k][1] = sub_mv[8 * i + 2 * k + 1];
    } else {  // force_integer_mv: the caller's initial values (:861-862)
      mses[k] = INT_MAX; mvs[k][0] = mvs[k][1] = 0;
    }
  }
  const int bmse = block_mse[i];
  int mn = INT_MAX, mx = INT_MIN;
  int64_t sum = 0;
  for (int k = 0; k < 4; ++k) {
    sum += mses[k];
    mn = mses[k] < mn ? mses[k] : mn;
    mx = mses[k] > mx ? mses[k] : mx;
  }
  const int spread = (int)((unsigned)mx - (unsigned)mn);
  if (((int64_t)(bmse * 15) < sum * 4 && spread < 48) || ((int64_t)(bmse * 14) < sum * 4 && spread < 24)) {  // no split
    for (int k = 0; k < 4; ++k) {
      mvs[k][0] = block_mv[2 * i]; mvs[k][1] = block_mv[2 * i + 1];
      mses[k] = bmse;
    }
  }
  for (int k = 0; k < 4; ++k) {
    out_mvs[8 * i + 2 * k] = (int16_t)mvs[k][0]; out_mvs[8 * i + 2 * k + 1] = (int16_t)mvs[k][1];
    out_mses[4 * i + k] = mses[k];
  }
  if (ref_mv && bmse > mse_thresh) ref_mv[2 * i] = ref_mv[2 * i + 1] = 0;
}

// force_integer_mv (:158-168): error = vf(ref + mv, src) is one aomhip_variance_batch evaluation per block ...
__global__ void tf_integer_cands_kernel(const aomhip_search_block *blocks, const int16_t *full_mv, int n, aomhip_var_cand *cands) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_var_cand c;
  c.sx = b.bx; c.sy = b.by;
  c.rx = (int16_t)(b.bx + full_mv[2 * i + 1]); c.ry = (int16_t)(b.by + full_mv[2 * i]);
  c.xoff = c.yoff = 0; c.reserved[0] = c.reserved[1] = 0;
  cands[i] = c;
}
// ... and then block_mv = the full-pel MV in 1/8 pel, block_mse = DIVIDE_AND_ROUND(error, 1024); *ref_mv is NOT updated on this path
__global__ void tf_integer_finish_kernel(const int16_t *full_mv, const uint32_t *var, int n, int16_t *block_mv, int32_t *block_mse) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  block_mv[2 * i] = (int16_t)(full_mv[2 * i] * 8); block_mv[2 * i + 1] = (int16_t)(full_mv[2 * i + 1] * 8);
  block_mse[i] = (int32_t)((var[i] + (unsigned)(kTfBlock * kTfBlock / 2)) / (unsigned)(kTfBlock * kTfBlock));
}

__global__ void tf_negate_kernel(int16_t *ref_mv, int n2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) ref_mv[i] = (int16_t)-ref_mv[i];
}
__global__ void tf_fill_kernel(int16_t *mvs, int32_t *mses, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  mvs[2 * i] = mvs[2 * i + 1] = 0;
  mses[i] = INT_MAX;
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

// Joins the context's side stream back into its main stream on EVERY way out of a composite that forked them -- an error return between the
// fork and the regular join must not leave the caller's stream unordered behind side-stream work (or a capture of it forked).
struct StreamJoinGuard {
  aomhip_ctx *ctx;
  hipStream_t *side;
  const bool *forked;
  ~StreamJoinGuard() {
    if (*forked && *side) {
      (void)hipEventRecord(ctx->ev_join, *side);
      (void)hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0);
    }
  }
};

extern "C" int aomhip_tf_motion_search_frames(aomhip_ctx *ctx, const aomhip_planes *frames, int filter_frame, const uint8_t *frame_present,
                                              const aomhip_tf_params *tp, const aomhip_search_block *d_blocks, int n, int16_t *d_subblock_mvs,
                                              int32_t *d_subblock_mses, int16_t *d_ref_mv_out) {
  if (!ctx || !frames || !frames->base || !tp || n < 0 || (n > 0 && (!d_blocks || !d_subblock_mvs || !d_subblock_mses)) || filter_frame < 0 ||
      filter_frame >= frames->n_frames) {
    set_error("aomhip_tf_motion_search_frames: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (tp->full.search_method != AOMHIP_SEARCH_NSTEP || !tp->full.run_mesh_search || tp->sub.subpel_search_type != 3 || tp->sub.forced_stop != 0 ||
      tp->sub.mv_cost_type != AOMHIP_MV_COST_NONE || tp->full.mv_cost_type < AOMHIP_MV_COST_L1_LOWRES || tp->full.mv_cost_type > AOMHIP_MV_COST_L1_HDRES ||
      tp->sub.tree < 0 || tp->sub.tree > 2) {
    set_error("aomhip_tf_motion_search_frames: parameters are not tf_motion_search's (NSTEP + mesh, L1 cost; USE_8_TAPS, EIGHTH_PEL, MV_COST_NONE)");
    return AOMHIP_ERR_INVALID;
  }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  // work memory of the chain (stream-ordered re-use from call to call; growing it synchronises)
  const size_t n1 = (size_t)n, n4 = 4 * n1;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_ref = take(n1 * 4), o_l32 = take(n1 * sizeof(aomhip_search_block)), o_fmv32 = take(n1 * 4), o_fcost32 = take(n1 * 4),
               o_cl32 = take(n1 * 20), o_s32 = take(n1 * sizeof(aomhip_search_block)), o_mv32 = take(n1 * 4), o_err32 = take(n1 * 4),
               o_dist32 = take(n1 * 4), o_sse32 = take(n1 * 4), o_mse32 = take(n1 * 4),
               o_fmv16 = take(n4 * 4), o_fcost16 = take(n4 * 4), o_cl16 = take(n4 * 20), o_s16 = take(n4 * sizeof(aomhip_search_block)),
               o_mv16 = take(n4 * 4), o_err16 = take(n4 * 4), o_dist16 = take(n4 * 4), o_sse16 = take(n4 * 4),
               o_cand = take(n1 * sizeof(aomhip_var_cand));
  // per reference frame: what the sub-block chain of frame f reads after the block chain of frame f + 1 has started
  const size_t nf = (size_t)frames->n_frames;
  const size_t o_fmv32k = take(nf * n1 * 4), o_fmse32 = take(nf * n1 * 4), o_fl16 = take(nf * n4 * sizeof(aomhip_search_block));
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  auto at = [&](size_t o) { return w + o; };
  int16_t *ref_mv = reinterpret_cast<int16_t *>(at(o_ref));
  aomhip_search_block *l32 = reinterpret_cast<aomhip_search_block *>(at(o_l32)), *s32 = reinterpret_cast<aomhip_search_block *>(at(o_s32));
  aomhip_search_block *s16 = reinterpret_cast<aomhip_search_block *>(at(o_s16));
  int16_t *fmv32 = reinterpret_cast<int16_t *>(at(o_fmv32)), *mv32 = reinterpret_cast<int16_t *>(at(o_mv32));
  int16_t *fmv16 = reinterpret_cast<int16_t *>(at(o_fmv16)), *mv16 = reinterpret_cast<int16_t *>(at(o_mv16));
  int32_t *cl32 = tp->use_cost_list ? reinterpret_cast<int32_t *>(at(o_cl32)) : nullptr;
  int32_t *cl16 = tp->use_cost_list ? reinterpret_cast<int32_t *>(at(o_cl16)) : nullptr;
  int32_t *mse32 = reinterpret_cast<int32_t *>(at(o_mse32));
  uint32_t *err32 = reinterpret_cast<uint32_t *>(at(o_err32)), *err16 = reinterpret_cast<uint32_t *>(at(o_err16));

  const size_t esz = frames->bit_depth == 8 ? 1 : 2;
  auto frame_view = [&](int f) {  // one frame of the ring as a ring of one (the batched searches pair src / ref by frame index)
    aomhip_planes v = *frames;
    v.base = static_cast<char *>(frames->base) + (size_t)f * frames->frame_stride * esz;
    v.n_frames = 1;
    return v;
  };
  const aomhip_planes src = frame_view(filter_frame);
  const unsigned g1 = (unsigned)((n1 + 255) / 256), g4 = (unsigned)((n4 + 255) / 256);
  hipStream_t st = ctx->stream;
  AOMHIP_TRY(hipMemsetAsync(ref_mv, 0, n1 * 4, st));  // MV ref_mv = kZeroMv (:855)
  // ref_mv chains the frames through their 32x32 searches only: the four 16x16 searches of frame f (60 % of a frame's work) run on the
  // context's side stream beside the 32x32 search of frame f + 1, whose 8 160 wavefronts leave the chip half empty in their last round.
  // AOMHIP_TF_SERIAL=1: one stream (A/B); also while ctx->stream is being captured (no side stream then).
  aomhip_ctx side = *ctx;
  hipStream_t ss = nullptr;
  if (!tp->force_integer_mv && !([] { const char *e = getenv("AOMHIP_TF_SERIAL"); return e && atoi(e) != 0; }())) ss = aomhip::side_stream(ctx);
  if (ss) side.stream = ss;
  aomhip_ctx *cb = ss ? &side : ctx;   // where the sub-block chain is enqueued
  bool forked = false;
  StreamJoinGuard join_on_exit{ ctx, &ss, &forked };   // (the regular end of the function included)
  for (int f = 0; f < frames->n_frames; ++f) {
    int16_t *out_mvs = d_subblock_mvs + (size_t)f * n4 * 2;
    int32_t *out_mses = d_subblock_mses + (size_t)f * n4;
    if (f == filter_frame || (frame_present && !frame_present[f])) {
      hipLaunchKernelGGL(tf_fill_kernel, dim3(g4), dim3(256), 0, st, out_mvs, out_mses, (int)n4);
      if (f == filter_frame) hipLaunchKernelGGL(tf_negate_kernel, dim3((unsigned)((2 * n1 + 255) / 256)), dim3(256), 0, st, ref_mv, (int)(2 * n1));  // :864-867
      AOMHIP_LAUNCH_CHECK();
      continue;
    }
    const aomhip_planes ref = frame_view(f);
    int rc;
    hipLaunchKernelGGL(tf_full32_list_kernel, dim3(g1), dim3(256), 0, st, d_blocks, ref_mv, n, l32);
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_full_pixel_search_batch(ctx, &src, &ref, 0, kTfBlock, kTfBlock, &tp->full, nullptr, nullptr, nullptr, l32, n, fmv32,
                                        reinterpret_cast<int32_t *>(at(o_fcost32)), cl32, nullptr);
    if (rc != AOMHIP_OK) return rc;
    if (tp->force_integer_mv) {
      aomhip_var_cand *cands = reinterpret_cast<aomhip_var_cand *>(at(o_cand));
      hipLaunchKernelGGL(tf_integer_cands_kernel, dim3(g1), dim3(256), 0, st, d_blocks, fmv32, n, cands);
      AOMHIP_LAUNCH_CHECK();
      rc = aomhip_variance_batch(ctx, &src, &ref, 0, 1, kTfBlock, kTfBlock, cands, n, 0, err32, reinterpret_cast<uint32_t *>(at(o_sse32)));
      if (rc != AOMHIP_OK) return rc;
      hipLaunchKernelGGL(tf_integer_finish_kernel, dim3(g1), dim3(256), 0, st, fmv32, err32, n, mv32, mse32);
      hipLaunchKernelGGL(tf_finish_kernel, dim3(g1), dim3(256), 0, st, mv32, mse32, (const int16_t *)nullptr, (const uint32_t *)nullptr, n, 0,
                         tp->mse_thresh, ref_mv, out_mvs, out_mses);
      AOMHIP_LAUNCH_CHECK();
      continue;
    }
    hipLaunchKernelGGL(tf_subpel_list_kernel, dim3(g1), dim3(256), 0, st, d_blocks, fmv32, n, 1, s32);
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_subpel_tree_batch(ctx, &src, &ref, 0, kTfBlock, kTfBlock, &tp->sub, nullptr, nullptr, nullptr, s32, cl32, n, mv32, err32,
                                  reinterpret_cast<int32_t *>(at(o_dist32)), reinterpret_cast<uint32_t *>(at(o_sse32)));
    if (rc != AOMHIP_OK) return rc;
    int16_t *mv32f = reinterpret_cast<int16_t *>(at(o_fmv32k)) + (size_t)f * n1 * 2;
    int32_t *mse32f = reinterpret_cast<int32_t *>(at(o_fmse32)) + (size_t)f * n1;
    aomhip_search_block *l16f = reinterpret_cast<aomhip_search_block *>(at(o_fl16)) + (size_t)f * n4;
    hipLaunchKernelGGL(tf_after_block_kernel, dim3(g1), dim3(256), 0, st, d_blocks, mv32, err32, n, tp->mse_thresh, ref_mv, mse32f, mv32f, l16f);
    AOMHIP_LAUNCH_CHECK();
    if (ss) {   // the sub-block chain of this frame starts when its block chain is done; the next frame's block chain does not wait for it
      AOMHIP_TRY(hipEventRecord(ctx->ev_fork, st));
      AOMHIP_TRY(hipStreamWaitEvent(ss, ctx->ev_fork, 0));
      forked = true;
    }
    rc = aomhip_full_pixel_search_batch(cb, &src, &ref, 0, kTfSub, kTfSub, &tp->full, nullptr, nullptr, nullptr, l16f, (int)n4, fmv16,
                                        reinterpret_cast<int32_t *>(at(o_fcost16)), cl16, nullptr);
    if (rc == AOMHIP_OK) {
      hipLaunchKernelGGL(tf_subpel_list_kernel, dim3(g4), dim3(256), 0, cb->stream, d_blocks, fmv16, n, 4, s16);
      rc = aomhip_subpel_tree_batch(cb, &src, &ref, 0, kTfSub, kTfSub, &tp->sub, nullptr, nullptr, nullptr, s16, cl16, (int)n4, mv16, err16,
                                    reinterpret_cast<int32_t *>(at(o_dist16)), reinterpret_cast<uint32_t *>(at(o_sse16)));
    }
    if (rc == AOMHIP_OK)
      hipLaunchKernelGGL(tf_finish_kernel, dim3(g1), dim3(256), 0, cb->stream, mv32f, mse32f, mv16, err16, n, 1, tp->mse_thresh, (int16_t *)nullptr, out_mvs,
                         out_mses);
    if (rc != AOMHIP_OK || hipGetLastError() != hipSuccess) {
      if (rc == AOMHIP_OK) { set_error("aomhip_tf_motion_search_frames: kernel launch failed"); rc = AOMHIP_ERR_HIP; }
      return rc;   // (joined by join_on_exit)
    }
  }
  if (forked) {   // the regular join, with its errors reported; the guard then has nothing left to do
    AOMHIP_TRY(hipEventRecord(ctx->ev_join, ss));
    AOMHIP_TRY(hipStreamWaitEvent(st, ctx->ev_join, 0));
    forked = false;
  }
  if (d_ref_mv_out) AOMHIP_TRY(hipMemcpyAsync(d_ref_mv_out, ref_mv, n1 * 4, hipMemcpyDeviceToDevice, st));
  return AOMHIP_OK;
}


// ---- first pass: first_pass_motion_search (av1/encoder/firstpass.c:261-299) for a list of blocks ------------------------------
namespace aomhip {
namespace {
__global__ void fp_cands_kernel(const aomhip_search_block *blocks, const int16_t *mv, int n, aomhip_var_cand *cands) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_var_cand c;
  c.sx = b.bx; c.sy = b.by;
  c.rx = (int16_t)(b.bx + mv[2 * i + 1]); c.ry = (int16_t)(b.by + mv[2 * i]);
  c.xoff = c.yoff = 0; c.reserved[0] = c.reserved[1] = 0;
  cands[i] = c;
}
// gf_motion_error of a frame with a golden reference (firstpass.c:777-794 under :722): the smaller of the 0,0 error and the golden search's
// for a block that is searched at all, the last frame's 0,0 error otherwise -- nothing of the best_ref_mv chain enters it
__global__ void fp_gf_kernel(const uint32_t *raw, const uint32_t *err0, const uint32_t *gf0, const int32_t *gerr, int thr, int n, int32_t *gf_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int gf = (int)err0[i];
  if ((int)raw[i] > thr) { gf = (int)gf0[i]; if (gerr[i] < gf) gf = gerr[i]; }
  gf_out[i] = gf;
}
// tmp_err = sse + mv_err_cost_(get_mv_from_fullmv(best), params) + NEW_MV_MODE_PENALTY   (mcomp.c:271-308, 3637-3649)
__global__ void fp_finish_kernel(const aomhip_search_block *blocks, const int16_t *mv, const int32_t *search_cost, const uint32_t *sse, int n,
                                 int cost_type, int error_per_bit, const int32_t *mvjcost, const int32_t *mvcost0, const int32_t *mvcost1,
                                 int32_t *err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (search_cost[i] == INT_MAX) { err[i] = INT_MAX; return; }
  const aomhip_search_block b = blocks[i];
  const int mrow = mv[2 * i] * 8, mcol = mv[2 * i + 1] * 8;
  int cost;
  if (cost_type == kCostEntropy) {
    const int dr = mrow - b.ref_row, dc = mcol - b.ref_col;
    const int64_t bits = mvjcost[(dc != 0) | ((dr != 0) << 1)] + mvcost0[dr] + mvcost1[dc];
    cost = (int)((bits * error_per_bit + (1 << 13)) >> 14);
  } else {
    const CostCtx cc{ cost_type, b.ref_row, b.ref_col };
    cost = cc.var_cost(mrow, mcol);
  }
  err[i] = (int32_t)(sse[i] + (uint32_t)cost + 32u);
}
}  // namespace
}  // namespace aomhip

extern "C" int aomhip_first_pass_motion_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                                     const aomhip_search_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                                                     const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, int n, int16_t *d_best_mv,
                                                     int32_t *d_err) {
  if (!ctx || !p || !d_best_mv || !d_err || n < 0) {
    set_error("aomhip_first_pass_motion_search_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t n1 = (size_t)n, o_cost = take(n1 * 4), o_cand = take(n1 * sizeof(aomhip_var_cand)), o_var = take(n1 * 4), o_sse = take(n1 * 4);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  int32_t *cost = reinterpret_cast<int32_t *>(w + o_cost);
  aomhip_var_cand *cands = reinterpret_cast<aomhip_var_cand *>(w + o_cand);
  uint32_t *var = reinterpret_cast<uint32_t *>(w + o_var), *sse = reinterpret_cast<uint32_t *>(w + o_sse);
  int rc = aomhip_full_pixel_search_batch(ctx, src, ref, frame, bw, bh, p, d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, n, d_best_mv, cost, nullptr,
                                          nullptr);
  if (rc != AOMHIP_OK) return rc;
  const unsigned g = (unsigned)((n1 + 255) / 256);
  hipLaunchKernelGGL(fp_cands_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_best_mv, n, cands);
  AOMHIP_LAUNCH_CHECK();
  rc = aomhip_variance_batch(ctx, src, ref, frame, 1, bw, bh, cands, n, 0, var, sse);
  if (rc != AOMHIP_OK) return rc;
  hipLaunchKernelGGL(fp_finish_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_best_mv, cost, sse, n, p->mv_cost_type, p->error_per_bit, d_mvjcost,
                     d_mvcost_row, d_mvcost_col, d_err);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}


// ---- full-pel + sub-pel search of a block list (tpl_model.c motion_estimation, :248-301) --------------------------------------
namespace aomhip {
namespace {
// av1_set_mv_search_range(&limits, &ref_mv) (mcomp.c:196-215) on raw x->mv_limits
__device__ __forceinline__ void full_limits_ref(const aomhip_search_block &in, aomhip_search_block *out) {
  const int rr = in.ref_row, rc = in.ref_col;
  int col_min = rawpel(rc) - kMaxFullPel + ((rc & 7) ? 1 : 0), row_min = rawpel(rr) - kMaxFullPel + ((rr & 7) ? 1 : 0);
  int col_max = rawpel(rc) + kMaxFullPel, row_max = rawpel(rr) + kMaxFullPel;
  const int lo = rawpel(kMvLow) + 1, hi = rawpel(kMvUpp) - 1;
  col_min = col_min > lo ? col_min : lo; row_min = row_min > lo ? row_min : lo;
  col_max = col_max < hi ? col_max : hi; row_max = row_max < hi ? row_max : hi;
  out->col_min = (int16_t)(in.col_min < col_min ? col_min : in.col_min);
  out->col_max = (int16_t)(in.col_max > col_max ? col_max : in.col_max);
  out->row_min = (int16_t)(in.row_min < row_min ? row_min : in.row_min);
  out->row_max = (int16_t)(in.row_max > row_max ? row_max : in.row_max);
}
// av1_set_subpel_mv_search_range(.., &x->mv_limits, &ref_mv) (mcomp.h:344-361)
__device__ __forceinline__ void subpel_limits_ref(const aomhip_search_block &in, aomhip_search_block *out) {
  const int max_mv = kMaxFullPel * 8;
  auto mx = [](int a, int b) { return a > b ? a : b; };
  auto mn = [](int a, int b) { return a < b ? a : b; };
  out->col_min = (int16_t)mx(kMvLow + 1, mx(in.col_min * 8, in.ref_col - max_mv));
  out->col_max = (int16_t)mn(kMvUpp - 1, mn(in.col_max * 8, in.ref_col + max_mv));
  out->row_min = (int16_t)mx(kMvLow + 1, mx(in.row_min * 8, in.ref_row - max_mv));
  out->row_max = (int16_t)mn(kMvUpp - 1, mn(in.row_max * 8, in.ref_row + max_mv));
}
__global__ void me_full_list_kernel(const aomhip_search_block *blocks, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_search_block o = b;
  o.start_row = (int16_t)rawpel(b.ref_row); o.start_col = (int16_t)rawpel(b.ref_col);  // get_fullmv_from_mv(&center_mv)
  full_limits_ref(b, &o);
  if (b.row_min > b.row_max) { o.row_min = 1; o.row_max = 0; }   // an entry the caller wants skipped stays skipped (fullpel_search.inc)
  out[i] = o;
}
__global__ void me_subpel_list_kernel(const aomhip_search_block *blocks, const int16_t *full_mv, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_search_block o = b;
  o.start_row = (int16_t)(full_mv[2 * i] * 8); o.start_col = (int16_t)(full_mv[2 * i + 1] * 8);  // get_mv_from_fullmv
  subpel_limits_ref(b, &o);
  if (b.row_min > b.row_max) { o.row_min = 1; o.row_max = 0; }
  out[i] = o;
}
}  // namespace
}  // namespace aomhip

extern "C" int aomhip_motion_estimation_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                              const aomhip_search_params *full, const aomhip_subpel_params *sub, int use_cost_list,
                                              const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                              const aomhip_search_block *d_blocks, int n, int16_t *d_best_mv, uint32_t *d_best_err,
                                              int32_t *d_distortion, uint32_t *d_sse, int16_t *d_fullpel_mv) {
  if (!ctx || !full || !sub || n < 0 || (n > 0 && (!d_blocks || !d_best_mv || !d_best_err || !d_distortion || !d_sse))) {
    set_error("aomhip_motion_estimation_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t n1 = (size_t)n;
  const size_t o_fl = take(n1 * sizeof(aomhip_search_block)), o_sl = take(n1 * sizeof(aomhip_search_block)), o_fmv = take(n1 * 4), o_cost = take(n1 * 4),
               o_cl = take(n1 * 20);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  aomhip_search_block *fl = reinterpret_cast<aomhip_search_block *>(w + o_fl), *sl = reinterpret_cast<aomhip_search_block *>(w + o_sl);
  int16_t *fmv = d_fullpel_mv ? d_fullpel_mv : reinterpret_cast<int16_t *>(w + o_fmv);
  int32_t *cl = use_cost_list ? reinterpret_cast<int32_t *>(w + o_cl) : nullptr;
  const unsigned g = (unsigned)((n1 + 255) / 256);
  hipLaunchKernelGGL(me_full_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, n, fl);
  AOMHIP_LAUNCH_CHECK();
  int rc = aomhip_full_pixel_search_batch(ctx, src, ref, frame, bw, bh, full, d_mvjcost, d_mvcost_row, d_mvcost_col, fl, n, fmv,
                                          reinterpret_cast<int32_t *>(w + o_cost), cl, nullptr);
  if (rc != AOMHIP_OK) return rc;
  hipLaunchKernelGGL(me_subpel_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, fmv, n, sl);
  AOMHIP_LAUNCH_CHECK();
  return aomhip_subpel_tree_batch(ctx, src, ref, frame, bw, bh, sub, d_mvjcost, d_mvcost_row, d_mvcost_col, sl, cl, n, d_best_mv, d_best_err, d_distortion,
                                  d_sse);
}

// ---- av1_simple_motion_search / av1_simple_motion_sse_var (av1/encoder/motion_search_facade.c:925-1060): the partition-pruning search.
// Per block: av1_full_pixel_search from the caller's start_mv around ref_mv = 0 (limits av1_set_mv_search_range(&x->mv_limits, &kZeroMv)),
// the sub-pel search from get_mv_from_fullmv(best) when use_subpixel and the full-pel search returned less than INT_MAX (:1003-1024),
// the EIGHTTAP_REGULAR luma predictor at the result (:1029-1031) and the block's vf(src, pred) -> sse, var (:1052-1057).
namespace aomhip {
namespace {
__global__ void sms_full_list_kernel(const aomhip_search_block *blocks, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_search_block o = b;
  o.ref_row = 0; o.ref_col = 0;   // const MV ref_mv = kZeroMv (:948); start_row / start_col: the caller's FULLPEL start_mv
  full_limits(b, &o);
  out[i] = o;
}
__global__ void sms_subpel_list_kernel(const aomhip_search_block *blocks, const int16_t *full_mv, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_search_block o = b;
  o.ref_row = 0; o.ref_col = 0;
  o.start_row = (int16_t)(full_mv[2 * i] * 8); o.start_col = (int16_t)(full_mv[2 * i + 1] * 8);  // get_mv_from_fullmv
  subpel_limits(b, &o);
  out[i] = o;
}
// blocks whose full-pel search returned INT_MAX (or every block when there is no sub-pel stage): convert_fullmv_to_mv (:1025-1029)
__global__ void sms_fullmv_result_kernel(const int16_t *full_mv, const int32_t *full_cost, int n, int all, int16_t *best_mv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (all || full_cost[i] == INT_MAX) {
    best_mv[2 * i] = (int16_t)(full_mv[2 * i] * 8);
    best_mv[2 * i + 1] = (int16_t)(full_mv[2 * i + 1] * 8);
  }
}
// fn_ptr[bsize].vf(src, pred) (aom_dsp/variance.c:141-148 VAR, :383-420 HIGHBD_VAR): one wavefront per block
template <typename T>
__global__ __launch_bounds__(256) void sms_var_kernel(PlaneView<T> src, int src_frame, PlaneView<T> pred, int pred_frame, int bw, int bh, int bit_depth,
                                                       const aomhip_search_block *blocks, int n, uint32_t *out_sse, uint32_t *out_var) {
  const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  const T *s = src.origin + (int64_t)src_frame * src.frame_stride + (int64_t)b.by * src.stride + b.bx;
  const T *p = pred.origin + (int64_t)pred_frame * pred.frame_stride + (int64_t)b.by * pred.stride + b.bx;
  long long sum = 0;
  unsigned long long sse = 0;
  for (int q = lane; q < bw * bh; q += 64) {
    const int y = q / bw, x = q - y * bw;
    const int d = (int)s[(int64_t)y * src.stride + x] - (int)p[(int64_t)y * pred.stride + x];
    sum += d; sse += (unsigned)__mul24(d, d);
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { sum += __shfl_xor(sum, m, 64); sse += __shfl_xor(sse, m, 64); }
  if (lane == 0) {
    int32_t sm; uint32_t q;
    if (bit_depth == 10) { q = (uint32_t)((sse + 8) >> 4); sm = (int32_t)((sum + 2) >> 2); }
    else if (bit_depth == 12) { q = (uint32_t)((sse + 128) >> 8); sm = (int32_t)((sum + 8) >> 4); }
    else { q = (uint32_t)sse; sm = (int32_t)sum; }
    const int64_t sq = ((int64_t)sm * sm) / (bw * bh);
    out_sse[i] = q;
    if (bit_depth == 8) out_var[i] = q - (uint32_t)sq;
    else { const int64_t v = (int64_t)q - sq; out_var[i] = v >= 0 ? (uint32_t)v : 0u; }
  }
}
}  // namespace
}  // namespace aomhip

extern "C" int aomhip_simple_motion_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                                 const aomhip_search_params *full, const aomhip_subpel_params *sub, int use_cost_list,
                                                 const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                                 const aomhip_search_block *d_blocks, int n, const aomhip_planes *pred, int pred_frame,
                                                 int16_t *d_best_mv, uint32_t *d_sse, uint32_t *d_var) {
  if (!ctx || !src || !ref || !full || n < 0 || (n > 0 && (!d_blocks || !d_best_mv)) || (pred && (!pred->base || pred_frame < 0 || pred_frame >= pred->n_frames)) ||
      ((d_sse || d_var) && (!pred || !d_sse || !d_var))) {
    set_error("aomhip_simple_motion_search_batch: invalid argument (sse / var need the predictor plane and each other)");
    return AOMHIP_ERR_INVALID;
  }
  if (pred && (pred->width != src->width || pred->height != src->height || pred->bit_depth != src->bit_depth)) {
    set_error("aomhip_simple_motion_search_batch: the predictor plane must have the source's geometry");
    return AOMHIP_ERR_INVALID;
  }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t n1 = (size_t)n;
  const size_t o_fl = take(n1 * sizeof(aomhip_search_block)), o_sl = take(n1 * sizeof(aomhip_search_block)), o_fmv = take(n1 * 4), o_cost = take(n1 * 4),
               o_cl = take(n1 * 20), o_err = take(n1 * 4), o_dist = take(n1 * 4), o_s2 = take(n1 * 4);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  aomhip_search_block *fl = reinterpret_cast<aomhip_search_block *>(w + o_fl), *sl = reinterpret_cast<aomhip_search_block *>(w + o_sl);
  int16_t *fmv = reinterpret_cast<int16_t *>(w + o_fmv);
  int32_t *fcost = reinterpret_cast<int32_t *>(w + o_cost);
  int32_t *cl = use_cost_list ? reinterpret_cast<int32_t *>(w + o_cl) : nullptr;
  const unsigned g = (unsigned)((n1 + 255) / 256);
  hipLaunchKernelGGL(sms_full_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, n, fl);
  AOMHIP_LAUNCH_CHECK();
  int rc = aomhip_full_pixel_search_batch(ctx, src, ref, frame, bw, bh, full, d_mvjcost, d_mvcost_row, d_mvcost_col, fl, n, fmv, fcost, cl, nullptr);
  if (rc != AOMHIP_OK) return rc;
  if (sub) {
    hipLaunchKernelGGL(sms_subpel_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, fmv, n, sl);
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_subpel_tree_batch(ctx, src, ref, frame, bw, bh, sub, d_mvjcost, d_mvcost_row, d_mvcost_col, sl, cl, n, d_best_mv,
                                  reinterpret_cast<uint32_t *>(w + o_err), reinterpret_cast<int32_t *>(w + o_dist), reinterpret_cast<uint32_t *>(w + o_s2));
    if (rc != AOMHIP_OK) return rc;
  }
  hipLaunchKernelGGL(sms_fullmv_result_kernel, dim3(g), dim3(256), 0, ctx->stream, fmv, fcost, n, sub ? 0 : 1, d_best_mv);
  AOMHIP_LAUNCH_CHECK();
  if (!pred) return AOMHIP_OK;
  // av1_enc_build_inter_predictor(.., AOM_PLANE_Y, AOM_PLANE_Y) with interp_filters = EIGHTTAP_REGULAR (:944, :1029-1031)
  rc = aomhip_build_inter_pred_batch(ctx, ref, frame, pred, pred_frame, bw, bh, d_blocks, d_best_mv, n, AOMHIP_INTERP_REGULAR, AOMHIP_INTERP_REGULAR);
  if (rc != AOMHIP_OK || !d_sse) return rc;
  const unsigned gv = (unsigned)((n1 + 3) / 4);
  if (src->bit_depth == 8)
    hipLaunchKernelGGL(sms_var_kernel<uint8_t>, dim3(gv), dim3(256), 0, ctx->stream, view_of<uint8_t>(*src), frame, view_of<uint8_t>(*pred), pred_frame, bw, bh,
                       src->bit_depth, d_blocks, n, d_sse, d_var);
  else
    hipLaunchKernelGGL(sms_var_kernel<uint16_t>, dim3(gv), dim3(256), 0, ctx->stream, view_of<uint16_t>(*src), frame, view_of<uint16_t>(*pred), pred_frame, bw,
                       bh, src->bit_depth, d_blocks, n, d_sse, d_var);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

// ---- first pass: the inter half of one frame (av1/encoder/firstpass.c firstpass_inter_prediction :690-815 under the raster loop :1148-1193)
// best_ref_mv of block (r, c) is block (r, c-1)'s *best_mv and kZeroMv at c == 0 (:1165, :1190): rows are independent, columns a chain.
// Everything independent of the chain -- the three 0,0 errors and the two zero-MV legs -- goes through once for the whole frame; the
// leg started at best_ref_mv runs one block column at a time with every row in flight, list -> search -> decision, all on the stream.
namespace aomhip {
namespace {
__global__ void fpf_zero_list_kernel(const aomhip_search_block *blocks, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_search_block o = b;
  o.ref_row = o.ref_col = 0; o.start_row = o.start_col = 0;
  full_limits(b, &o);
  out[i] = o;
}
// One block column of the chain in ONE launch behind its search (a wavefront per block row): the leg's av1_get_mvpred_sse + MV cost +
// NEW_MV_MODE_PENALTY (what fp_cands / variance / fp_finish do for a list), the decision (firstpass.c:722-752, :777-794), and the next column's list
// entry (get_fullmv_from_mv(best_ref_mv), av1_set_mv_search_range).  Six launches per column were 110 us of a 4K frame's 240 columns.
template <typename T>
__global__ __launch_bounds__(256) void fpf_column_kernel(PlaneView<T> src, PlaneView<T> last, int bw, int bh, int bit_depth, const aomhip_search_block *blocks,
                                                         const aomhip_search_block *cur_list, const int32_t *search_cost, FpfLegs L, FpfCost C,
                                                         const int32_t *intra, int col, int rows, int cols, int thr, int skip_zeromv, int16_t *chain,
                                                         aomhip_search_block *next_list, int16_t *best_mv, int16_t *full_mv, int32_t *motion_error,
                                                         int32_t *gf_motion_error, int32_t *raw_motion_error) {
  const int r = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const size_t i = (size_t)r * cols + col;
  const int ref_row = chain[2 * r], ref_col = chain[2 * r + 1];
  const bool moved = (ref_row | ref_col) != 0;
  const int raw = (int)L.raw[i];
  int e1 = INT_MAX, m1r = 0, m1c = 0;
  if (raw > thr) {
    if (moved) {   // (col > 0: the chained leg was searched from cur_list[r])
      m1r = L.cmv[2 * r]; m1c = L.cmv[2 * r + 1];
      if (search_cost[r] != INT_MAX) {
        const aomhip_search_block b = cur_list[r];
        const T *sp = src.origin + (int64_t)b.by * src.stride + b.bx;
        const T *rp = last.origin + (int64_t)(b.by + m1r) * last.stride + b.bx + m1c;
        unsigned long long sse = 0;
        for (int q = lane; q < bw * bh; q += 64) {
          const int y = q / bw, x = q - y * bw;
          const int d = (int)sp[(int64_t)y * src.stride + x] - (int)rp[(int64_t)y * last.stride + x];
          sse += (unsigned)__mul24(d, d);
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) sse += __shfl_xor(sse, m, 64);
        const uint32_t q = bit_depth == 10 ? (uint32_t)((sse + 8) >> 4) : bit_depth == 12 ? (uint32_t)((sse + 128) >> 8) : (uint32_t)sse;
        const int mrow = m1r * 8, mcol = m1c * 8;
        int cost;
        if (C.type == kCostEntropy) {
          const int dr = mrow - b.ref_row, dc = mcol - b.ref_col;
          const int64_t bits = (int64_t)C.mvjcost[(dc != 0) | ((dr != 0) << 1)] + C.mvcost0[dr] + C.mvcost1[dc];
          cost = (int)((bits * C.error_per_bit + (1 << 13)) >> 14);
        } else {
          const CostCtx cc{ C.type, b.ref_row, b.ref_col };
          cost = cc.var_cost(mrow, mcol);
        }
        e1 = (int32_t)(q + (uint32_t)cost + 32u);
      }
    } else {
      e1 = L.zerr[i]; m1r = L.zmv[2 * i]; m1c = L.zmv[2 * i + 1];
    }
  }
  if (lane != 0) return;
  int err = (int)L.err0[i], mrow = 0, mcol = 0, gf;
  gf = err;
  if (raw > thr) {
    if (e1 < err) { err = e1; mrow = m1r; mcol = m1c; }
    if (!skip_zeromv && moved) {
      const int e0 = L.zerr[i];
      if (e0 < err) { err = e0; mrow = L.zmv[2 * i]; mcol = L.zmv[2 * i + 1]; }
    }
    gf = err;
    if (L.gerr) { gf = (int)L.gf0[i]; if (L.gerr[i] < gf) gf = L.gerr[i]; }
  }
  int brow = 0, bcol = 0;
  if (err <= intra[i]) { brow = mrow * 8; bcol = mcol * 8; }
  chain[2 * r] = (int16_t)brow; chain[2 * r + 1] = (int16_t)bcol;
  best_mv[2 * i] = (int16_t)brow; best_mv[2 * i + 1] = (int16_t)bcol;
  if (full_mv) { full_mv[2 * i] = (int16_t)mrow; full_mv[2 * i + 1] = (int16_t)mcol; }
  motion_error[i] = err;
  if (gf_motion_error) gf_motion_error[i] = gf;
  if (raw_motion_error) raw_motion_error[i] = raw;
  if (col + 1 < cols) {
    aomhip_search_block b = blocks[i + 1];
    b.ref_row = (int16_t)brow; b.ref_col = (int16_t)bcol;
    aomhip_search_block o = b;
    o.start_row = (int16_t)rawpel(brow); o.start_col = (int16_t)rawpel(bcol);
    full_limits_ref(b, &o);
    next_list[r] = o;
  }
}
}  // namespace
}  // namespace aomhip

extern "C" int aomhip_first_pass_inter_frame(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *last, int last_frame,
                                             const aomhip_planes *golden, int golden_frame, const aomhip_planes *last_source, int last_source_frame,
                                             int bw, int bh, const aomhip_search_params *p, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                                             const int32_t *d_mvcost_col, const aomhip_first_pass_params *fp, const aomhip_search_block *d_blocks,
                                             const int32_t *d_intra_error, int16_t *d_best_mv, int16_t *d_full_mv, int32_t *d_motion_error,
                                             int32_t *d_gf_motion_error, int32_t *d_raw_motion_error) {
  auto ring_ok = [&](const aomhip_planes *q, int f) {
    return q && q->base && f >= 0 && f < q->n_frames && q->width == src->width && q->height == src->height && q->stride == src->stride &&
           q->border == src->border && q->bit_depth == src->bit_depth;
  };
  if (!ctx || !src || !p || !fp || fp->unit_rows < 0 || fp->unit_cols < 0 || !ring_ok(src, src_frame) || !ring_ok(last, last_frame) ||
      !ring_ok(last_source, last_source_frame) || (golden && !ring_ok(golden, golden_frame))) {
    set_error("aomhip_first_pass_inter_frame: invalid argument (the source, last, golden and last-source planes must share one geometry)");
    return AOMHIP_ERR_INVALID;
  }
  const int rows = fp->unit_rows, cols = fp->unit_cols;
  const size_t n1 = (size_t)rows * cols;
  if (n1 == 0) return AOMHIP_OK;
  if (n1 > (size_t)INT_MAX / 64 || !d_blocks || !d_intra_error || !d_best_mv || !d_motion_error) {
    set_error("aomhip_first_pass_inter_frame: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const int n = (int)n1;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t r1 = (size_t)rows;
  const size_t o_zl = take(n1 * sizeof(aomhip_search_block)), o_zmv = take(n1 * 4), o_zerr = take(n1 * 4), o_gmv = take(n1 * 4), o_gerr = take(n1 * 4),
               o_e0 = take(n1 * 4), o_raw = take(n1 * 4), o_gf0 = take(n1 * 4), o_var = take(n1 * 4), o_cost = take(n1 * 4), o_sse = take(n1 * 4),
               o_cand = take(n1 * sizeof(aomhip_var_cand)), o_cl = take(r1 * sizeof(aomhip_search_block)), o_cl2 = take(r1 * sizeof(aomhip_search_block)),
               o_cmv = take(r1 * 4), o_cerr = take(r1 * 4), o_chain = take(r1 * 4),
               // the golden leg's own intermediates: it runs beside the last-frame leg and the chain (side stream)
               o_var_g = take(n1 * 4), o_cost_g = take(n1 * 4), o_sse_g = take(n1 * 4), o_cand_g = take(n1 * sizeof(aomhip_var_cand));
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  auto i32 = [&](size_t o) { return reinterpret_cast<int32_t *>(w + o); };
  auto u32 = [&](size_t o) { return reinterpret_cast<uint32_t *>(w + o); };
  auto i16 = [&](size_t o) { return reinterpret_cast<int16_t *>(w + o); };
  aomhip_search_block *zl = reinterpret_cast<aomhip_search_block *>(w + o_zl), *cl = reinterpret_cast<aomhip_search_block *>(w + o_cl);
  const size_t esz = src->bit_depth == 8 ? 1 : 2;
  auto one = [&](const aomhip_planes *q, int f) {   // one frame of a ring as a ring of one: the batched searches pair src / ref by frame index
    aomhip_planes v = *q;
    v.base = static_cast<char *>(q->base) + (size_t)f * q->frame_stride * esz;
    v.n_frames = 1;
    return v;
  };
  const aomhip_planes s1 = one(src, src_frame), l1 = one(last, last_frame), ls1 = one(last_source, last_source_frame);
  const aomhip_planes g1 = golden ? one(golden, golden_frame) : s1;
  const unsigned g = (unsigned)((n1 + 255) / 256);
  struct LegMem { size_t var, cost, sse, cand; };
  const LegMem mem_main{ o_var, o_cost, o_sse, o_cand }, mem_side{ o_var_g, o_cost_g, o_sse_g, o_cand_g };
  auto sse0 = [&](aomhip_ctx *cx, const LegMem &m, const aomhip_planes &ref, uint32_t *out) {   // get_prediction_error_bitdepth: the mse function's sse at 0,0 (:113-160)
    const unsigned gv = (unsigned)((n1 + 3) / 4);
    if (src->bit_depth == 8)
      hipLaunchKernelGGL(sms_var_kernel<uint8_t>, dim3(gv), dim3(256), 0, cx->stream, view_of<uint8_t>(s1), 0, view_of<uint8_t>(ref), 0, bw, bh, 8, d_blocks,
                         n, out, u32(m.var));
    else
      hipLaunchKernelGGL(sms_var_kernel<uint16_t>, dim3(gv), dim3(256), 0, cx->stream, view_of<uint16_t>(s1), 0, view_of<uint16_t>(ref), 0, bw, bh,
                         src->bit_depth, d_blocks, n, out, u32(m.var));
  };
  // one first_pass_motion_search leg of `m` listed blocks (the body of aomhip_first_pass_motion_search_batch on this call's work memory)
  auto leg = [&](aomhip_ctx *cx, const LegMem &mm, const aomhip_planes &ref, const aomhip_search_block *list, int m, int16_t *mv, int32_t *err) -> int {
    aomhip_var_cand *cd = reinterpret_cast<aomhip_var_cand *>(w + mm.cand);
    int rc = aomhip_full_pixel_search_batch(cx, &s1, &ref, 0, bw, bh, p, d_mvjcost, d_mvcost_row, d_mvcost_col, list, m, mv, i32(mm.cost), nullptr, nullptr);
    if (rc != AOMHIP_OK) return rc;
    const unsigned gm = (unsigned)((m + 255) / 256);
    hipLaunchKernelGGL(fp_cands_kernel, dim3(gm), dim3(256), 0, cx->stream, list, mv, m, cd);
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_variance_batch(cx, &s1, &ref, 0, 1, bw, bh, cd, m, 0, u32(mm.var), u32(mm.sse));
    if (rc != AOMHIP_OK) return rc;
    hipLaunchKernelGGL(fp_finish_kernel, dim3(gm), dim3(256), 0, cx->stream, list, mv, i32(mm.cost), u32(mm.sse), m, p->mv_cost_type, p->error_per_bit, d_mvjcost,
                       d_mvcost_row, d_mvcost_col, err);
    AOMHIP_LAUNCH_CHECK();
    return AOMHIP_OK;
  };
  hipLaunchKernelGGL(fpf_zero_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, n, zl);
  AOMHIP_LAUNCH_CHECK();
  sse0(ctx, mem_main, l1, u32(o_e0));
  AOMHIP_LAUNCH_CHECK();
  sse0(ctx, mem_main, ls1, u32(o_raw));
  AOMHIP_LAUNCH_CHECK();
  int rc = leg(ctx, mem_main, l1, zl, n, i16(o_zmv), i32(o_zerr));
  if (rc != AOMHIP_OK) return rc;
  const char *force_cols = getenv("AOMHIP_FP_COLUMNS");   // (tests: the column-at-a-time form on the sizes the row kernel serves)
  const bool by_rows = aomhip::fp_rows_supported(bw, bh, p->search_method) && !(force_cols && atoi(force_cols));
  // The golden-frame leg depends on nothing the chain produces, and gf_motion_error (:777-794) on nothing of the chain: with the row kernel
  // -- one workgroup per block row, a chip mostly idle -- it runs on the context's side stream BESIDE the chain, forked here and joined behind
  // the chain kernel (whose wavefronts raise their priority: the chain is latency, the leg throughput).  AOMHIP_FP_SERIAL=1: one stream (A/B).
  aomhip_ctx side = *ctx;
  bool forked = false;
  hipStream_t ss = nullptr;
  StreamJoinGuard join_on_exit{ ctx, &ss, &forked };   // an error return between the fork and the regular join still joins
  auto golden_leg = [&]() -> int {
    aomhip_ctx *cx = forked ? &side : ctx;
    const LegMem &mm = forked ? mem_side : mem_main;
    sse0(cx, mm, g1, u32(o_gf0));
    AOMHIP_LAUNCH_CHECK();
    return leg(cx, mm, g1, zl, n, i16(o_gmv), i32(o_gerr));
  };
  if (golden) {
    const bool serial = [] { const char *e = getenv("AOMHIP_FP_SERIAL"); return e && atoi(e) != 0; }();
    ss = (serial || !by_rows) ? nullptr : aomhip::side_stream(ctx);
    if (ss) {
      side.stream = ss;
      AOMHIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
      AOMHIP_TRY(hipStreamWaitEvent(ss, ctx->ev_fork, 0));
      forked = true;
    } else {
      rc = golden_leg();
      if (rc != AOMHIP_OK) return rc;
    }
  }
  AOMHIP_TRY(hipMemsetAsync(w + o_chain, 0, r1 * 4, ctx->stream));   // MV best_ref_mv = kZeroMv at the start of every row (:1165)
  aomhip::FpfLegs L;
  L.zmv = i16(o_zmv); L.zerr = i32(o_zerr);
  L.gmv = golden ? i16(o_gmv) : nullptr; L.gerr = golden ? i32(o_gerr) : nullptr;
  L.cmv = i16(o_cmv); L.cerr = i32(o_cerr);
  L.err0 = u32(o_e0); L.raw = u32(o_raw); L.gf0 = u32(o_gf0);
  aomhip::FpfCost C{ p->mv_cost_type, p->error_per_bit, d_mvjcost, d_mvcost_row, d_mvcost_col };
  // the chain: one launch, a workgroup per row (fp_row.hip) -- or, for block sizes that kernel is not built for, column by column
  if (by_rows) {
    aomhip::FpfOut out{ d_best_mv, d_full_mv, d_motion_error, d_gf_motion_error, d_raw_motion_error };
    if (forked) {   // the golden leg is still running beside this: gf_motion_error does not depend on the chain, it follows the join
      L.gerr = nullptr;
      out.gf_motion_error = nullptr;
    }
    rc = aomhip::launch_fp_rows(ctx, &s1, &l1, bw, bh, p, d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, L, d_intra_error, rows, cols,
                                fp->skip_motion_search_threshold, fp->skip_zeromv_motion_search, out);
    if (forked) {
      const int rcg = golden_leg();   // (queued behind the chain kernel's launch: the chain's workgroups are placed first)
      AOMHIP_TRY(hipEventRecord(ctx->ev_join, ss));   // joined on every path: a capture of ctx->stream must not end forked
      AOMHIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
      forked = false;   // (the guard has nothing left to do)
      if (rc == AOMHIP_OK) rc = rcg;
      if (rc == AOMHIP_OK && d_gf_motion_error) {
        hipLaunchKernelGGL(fp_gf_kernel, dim3(g), dim3(256), 0, ctx->stream, u32(o_raw), u32(o_e0), u32(o_gf0), i32(o_gerr), fp->skip_motion_search_threshold, n,
                           d_gf_motion_error);
        AOMHIP_LAUNCH_CHECK();
      }
    }
    return rc;
  }
  aomhip_search_block *cl2 = reinterpret_cast<aomhip_search_block *>(w + o_cl2);
  const unsigned gw = (unsigned)((r1 + 3) / 4);
  for (int c = 0; c < cols; ++c) {
    aomhip_search_block *cur = (c & 1) ? cl2 : cl, *nxt = (c & 1) ? cl : cl2;
    if (c > 0) {   // column 0 starts from kZeroMv: its ref_mv leg IS the zero-MV leg
      rc = aomhip_full_pixel_search_batch(ctx, &s1, &l1, 0, bw, bh, p, d_mvjcost, d_mvcost_row, d_mvcost_col, cur, rows, i16(o_cmv), i32(o_cost), nullptr,
                                          nullptr);
      if (rc != AOMHIP_OK) return rc;
    }
    if (src->bit_depth == 8)
      hipLaunchKernelGGL(fpf_column_kernel<uint8_t>, dim3(gw), dim3(256), 0, ctx->stream, view_of<uint8_t>(s1), view_of<uint8_t>(l1), bw, bh, 8, d_blocks, cur,
                         i32(o_cost), L, C, d_intra_error, c, rows, cols, fp->skip_motion_search_threshold, fp->skip_zeromv_motion_search, i16(o_chain), nxt,
                         d_best_mv, d_full_mv, d_motion_error, d_gf_motion_error, d_raw_motion_error);
    else
      hipLaunchKernelGGL(fpf_column_kernel<uint16_t>, dim3(gw), dim3(256), 0, ctx->stream, view_of<uint16_t>(s1), view_of<uint16_t>(l1), bw, bh,
                         src->bit_depth, d_blocks, cur, i32(o_cost), L, C, d_intra_error, c, rows, cols, fp->skip_motion_search_threshold,
                         fp->skip_zeromv_motion_search, i16(o_chain), nxt, d_best_mv, d_full_mv, d_motion_error, d_gf_motion_error, d_raw_motion_error);
    AOMHIP_LAUNCH_CHECK();
  }
  return AOMHIP_OK;
}

// ---- av1_single_motion_search, SIMPLE_TRANSLATION core (av1/encoder/motion_search_facade.c:120-495) for independent (block, reference) pairs:
// up to two full-pel searches from the caller's candidate start MVs (:271-290), the sub-pel search from the winner, optionally the second
// sub-pel search from second_best_mv on the same last_mv_search_list, kept when its error is smaller (:367-430, disable_second_mv == 1), and
// av1_mv_bit_cost of the result (:485-493).  The decisions that need the mode loop's state (mode_info[], args->single_newmv*, DRL costs:
// :300-341, :447-483) read only this call's outputs and stay with the caller.
namespace aomhip {
namespace {
constexpr int kInvalidMv = -32768;   // INVALID_MV_ROW_COL
__global__ void single_full_list_kernel(const aomhip_search_block *blocks, const int16_t *start2, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  aomhip_search_block o = b;
  if (start2) { o.start_row = start2[2 * i]; o.start_col = start2[2 * i + 1]; }
  if (o.start_row == kInvalidMv) o.start_row = o.start_col = 0;   // searched, never looked at (single_select_kernel tests the caller's value)
  full_limits_ref(b, &o);
  out[i] = o;
}
struct SingleCand { const int16_t *mv, *second; const int32_t *cost, *cl; };
__global__ void single_select_kernel(const aomhip_search_block *blocks, const int16_t *start2, SingleCand c0, SingleCand c1, int n, int16_t *full_mv,
                                     int16_t *second, int32_t *bestsme, int32_t *cl, aomhip_search_block *sub_list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  int sme = INT_MAX, mr = kInvalidMv, mc = kInvalidMv, sr = kInvalidMv, sc = kInvalidMv;
  int l0 = INT_MAX, l1 = INT_MAX, l2 = INT_MAX, l3 = INT_MAX, l4 = INT_MAX;
  auto take = [&](const SingleCand &c) {
    if (c.cl) { l0 = c.cl[5 * i]; l1 = c.cl[5 * i + 1]; l2 = c.cl[5 * i + 2]; l3 = c.cl[5 * i + 3]; l4 = c.cl[5 * i + 4]; }   // one array for all candidates
    if (c.cost[i] < sme) { sme = c.cost[i]; mr = c.mv[2 * i]; mc = c.mv[2 * i + 1]; sr = c.second[2 * i]; sc = c.second[2 * i + 1]; }
  };
  if (b.start_row != kInvalidMv) take(c0);
  if (start2 && start2[2 * i] != kInvalidMv) take(c1);
  full_mv[2 * i] = (int16_t)mr; full_mv[2 * i + 1] = (int16_t)mc;
  second[2 * i] = (int16_t)sr; second[2 * i + 1] = (int16_t)sc;
  bestsme[i] = sme;
  if (cl) { cl[5 * i] = l0; cl[5 * i + 1] = l1; cl[5 * i + 2] = l2; cl[5 * i + 3] = l3; cl[5 * i + 4] = l4; }
  aomhip_search_block o = b;
  subpel_limits_ref(b, &o);
  const bool dead = mr == kInvalidMv;
  o.start_row = (int16_t)(dead ? max(min(0, (int)o.row_max), (int)o.row_min) : mr * 8);     // get_mv_from_fullmv(best_mv) (:358)
  o.start_col = (int16_t)(dead ? max(min(0, (int)o.col_max), (int)o.col_min) : mc * 8);
  sub_list[i] = o;
}
// the second sub-pel start (:370-389): second_best_mv when it is valid, differs from the winner and lies inside the sub-pel limits; the other
// blocks start at the winner again, which the list stops at iteration 0 with INT_MAX -- the value that can never win below
__global__ void single_second_list_kernel(const aomhip_search_block *sub_list, const int16_t *full_mv, const int16_t *second, int n, aomhip_search_block *out,
                                          uint8_t *has_second) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aomhip_search_block o = sub_list[i];
  const int sr = second[2 * i], sc = second[2 * i + 1];
  const bool differs = sr != full_mv[2 * i] || sc != full_mv[2 * i + 1];
  // try_second (:370-372) && av1_is_subpelmv_in_range(&ms_params.mv_limits, subpel_start_mv) (:395-396)
  const bool ok = full_mv[2 * i] != kInvalidMv && sr != kInvalidMv && differs && sc * 8 >= o.col_min && sc * 8 <= o.col_max && sr * 8 >= o.row_min && sr * 8 <= o.row_max;
  if (ok) {
    o.start_row = (int16_t)(sr * 8); o.start_col = (int16_t)(sc * 8);
  }
  out[i] = o;
  if (has_second) has_second[i] = ok ? 1 : 0;
}
__global__ void single_fill_invalid_kernel(int16_t *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int16_t)kInvalidMv;
}
// (the RD form of the second-MV decision, sf.mv_sf.disable_second_mv == 0, motion_search_facade.c:378-425: yrd_a / yrd_b = av1_estimate_txfm_yrd of the
// predictor at each candidate, has_second = the second search ran; NULL: the variance form)
__global__ void single_finish_kernel(const aomhip_search_block *blocks, const int16_t *full_mv, int force_integer_mv, const int16_t *mv_a, const uint32_t *err_a,
                                     const uint32_t *sse_a, const int16_t *mv_b, const uint32_t *err_b, const uint32_t *sse_b, int n, const int32_t *mvjcost,
                                     const int32_t *mvcost0, const int32_t *mvcost1, int16_t *best_mv, int32_t *rate_mv, uint32_t *pred_sse,
                                     const aomhip_txfm_yrd_stats *yrd_a = nullptr, const aomhip_txfm_yrd_stats *yrd_b = nullptr, const uint8_t *has_second = nullptr,
                                     int rdmult = 0, int16_t *candidates = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int row = kInvalidMv, col = kInvalidMv, rate = 0;
  uint32_t sse = 0;
  if (candidates) {
    const bool live = full_mv[2 * i] != kInvalidMv && !force_integer_mv, two = live && yrd_a && has_second[i];
    candidates[4 * i] = live ? mv_a[2 * i] : (int16_t)kInvalidMv; candidates[4 * i + 1] = live ? mv_a[2 * i + 1] : (int16_t)kInvalidMv;
    candidates[4 * i + 2] = two ? mv_b[2 * i] : (int16_t)kInvalidMv; candidates[4 * i + 3] = two ? mv_b[2 * i + 1] : (int16_t)kInvalidMv;
  }
  if (full_mv[2 * i] != kInvalidMv) {
    const aomhip_search_block b = blocks[i];
    auto mv_rate = [&](int r, int c) {                                                  // av1_mv_bit_cost(.., MV_COST_WEIGHT) (mcomp.c:261-266)
      const int dr = r - b.ref_row, dc = c - b.ref_col;
      const int64_t bits = (int64_t)mvjcost[(dc != 0) | ((dr != 0) << 1)] + mvcost0[dr] + mvcost1[dc];
      return (int)((bits * 108 + 64) >> 7);
    };
    if (force_integer_mv) { row = full_mv[2 * i] * 8; col = full_mv[2 * i + 1] * 8; }   // convert_fullmv_to_mv (:343-345)
    else {
      row = mv_a[2 * i]; col = mv_a[2 * i + 1]; sse = sse_a[i];
      if (yrd_a) {
        if (has_second[i]) {   // RDCOST(x->rdmult, mv_rate + stats.rate, stats.dist) of both; the second one replaces the first when SMALLER (:414-418)
          const int64_t rd = ((((int64_t)mv_rate(row, col) + yrd_a[i].rate) * rdmult + 256) >> 9) + yrd_a[i].dist * 128;
          const int64_t tmp_rd = ((((int64_t)yrd_b[i].rate + mv_rate(mv_b[2 * i], mv_b[2 * i + 1])) * rdmult + 256) >> 9) + yrd_b[i].dist * 128;
          if (tmp_rd < rd) { row = mv_b[2 * i]; col = mv_b[2 * i + 1]; sse = sse_b[i]; }
        }
      } else if (mv_b && (int)err_b[i] < (int)err_a[i]) { row = mv_b[2 * i]; col = mv_b[2 * i + 1]; sse = sse_b[i]; }   // this_var < best_mv_var (:421-425)
    }
    rate = mv_rate(row, col);
  }
  best_mv[2 * i] = (int16_t)row; best_mv[2 * i + 1] = (int16_t)col;
  rate_mv[i] = rate;
  if (pred_sse) pred_sse[i] = sse;
}
}  // namespace
}  // namespace aomhip

namespace aomhip {
size_t yrd_workspace_bytes(int n_blocks, int bw, int bh);
int estimate_txfm_yrd_ws(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *pred, int frame, int bw, int bh, const aomhip_quant_params *qparams,
                         const int32_t *d_costs, int tx_type_rate, int rdmult, int lossless, const aomhip_txfm_yrd_block *d_blocks, int n_blocks,
                         aomhip_txfm_yrd_stats *d_stats, char *ws);
}  // namespace aomhip

static int single_motion_search_impl(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                     const aomhip_search_params *full, const aomhip_subpel_params *sub, int use_cost_list, int try_second_mv,
                                     int force_integer_mv, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                     const aomhip_search_block *d_blocks, const int16_t *d_start2, int n, int16_t *d_best_mv, int32_t *d_bestsme,
                                     int32_t *d_rate_mv, uint32_t *d_pred_sse, int16_t *d_full_mv, int16_t *d_second_best_mv, const aomhip_single_rd_params *rd) {
  if (rd && (!rd->pred || !rd->pred->base || !rd->qparams || !rd->d_costs || !rd->d_yrd_blocks || frame >= rd->pred->n_frames ||
             rd->pred->bit_depth != src->bit_depth || rd->pred->width != src->width || rd->pred->height != src->height)) {
    set_error("aomhip_single_motion_search_rd_batch: the RD form needs a predictor ring of the source's geometry, the quantiser, the cost tables and the blocks' rates");
    return AOMHIP_ERR_INVALID;
  }
  if (!ctx || !src || !ref || !full || (!sub && !force_integer_mv) || n < 0 || !d_mvjcost || !d_mvcost_row || !d_mvcost_col ||
      (n > 0 && (!d_blocks || !d_best_mv || !d_bestsme || !d_rate_mv))) {
    set_error("aomhip_single_motion_search_batch: invalid argument (the rate of the result needs the MV cost tables)");
    return AOMHIP_ERR_INVALID;
  }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t n1 = (size_t)n, SB = sizeof(aomhip_search_block);
  const size_t o_fl = take(n1 * SB), o_sl = take(n1 * SB), o_sl2 = take(n1 * SB), o_mv0 = take(n1 * 4), o_mv1 = take(n1 * 4), o_sec0 = take(n1 * 4),
               o_sec1 = take(n1 * 4), o_c0 = take(n1 * 4), o_c1 = take(n1 * 4), o_cl0 = take(n1 * 20), o_cl1 = take(n1 * 20), o_cl = take(n1 * 20),
               o_fmv = take(n1 * 4), o_sec = take(n1 * 4), o_lists = take(n1 * 12), o_mva = take(n1 * 4), o_erra = take(n1 * 4), o_dist = take(n1 * 4),
               o_ssea = take(n1 * 4), o_mvb = take(n1 * 4), o_errb = take(n1 * 4), o_sseb = take(n1 * 4), o_has2 = take(n1),
               o_yrda = take(n1 * sizeof(aomhip_txfm_yrd_stats)), o_yrdb = take(n1 * sizeof(aomhip_txfm_yrd_stats)),
               o_yrdws = take(rd ? aomhip::yrd_workspace_bytes(n, bw, bh) : 0);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  auto blk = [&](size_t o) { return reinterpret_cast<aomhip_search_block *>(w + o); };
  auto i16 = [&](size_t o) { return reinterpret_cast<int16_t *>(w + o); };
  auto i32 = [&](size_t o) { return reinterpret_cast<int32_t *>(w + o); };
  auto u32 = [&](size_t o) { return reinterpret_cast<uint32_t *>(w + o); };
  int16_t *fmv = d_full_mv ? d_full_mv : i16(o_fmv), *sec = d_second_best_mv ? d_second_best_mv : i16(o_sec);
  const unsigned g = (unsigned)((n1 + 255) / 256);
  hipLaunchKernelGGL(single_full_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, (const int16_t *)nullptr, n, blk(o_fl));
  AOMHIP_LAUNCH_CHECK();
  int rc = aomhip_full_pixel_search_batch(ctx, src, ref, frame, bw, bh, full, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_fl), n, i16(o_mv0), i32(o_c0),
                                          use_cost_list ? i32(o_cl0) : nullptr, i16(o_sec0));
  if (rc != AOMHIP_OK) return rc;
  aomhip::SingleCand c0{ i16(o_mv0), i16(o_sec0), i32(o_c0), use_cost_list ? i32(o_cl0) : nullptr }, c1 = c0;
  if (d_start2) {
    hipLaunchKernelGGL(single_full_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_start2, n, blk(o_fl));
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_full_pixel_search_batch(ctx, src, ref, frame, bw, bh, full, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_fl), n, i16(o_mv1), i32(o_c1),
                                        use_cost_list ? i32(o_cl1) : nullptr, i16(o_sec1));
    if (rc != AOMHIP_OK) return rc;
    c1 = aomhip::SingleCand{ i16(o_mv1), i16(o_sec1), i32(o_c1), use_cost_list ? i32(o_cl1) : nullptr };
  }
  int32_t *cl = use_cost_list ? i32(o_cl) : nullptr;
  hipLaunchKernelGGL(single_select_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_start2, c0, c1, n, fmv, sec, d_bestsme, cl, blk(o_sl));
  AOMHIP_LAUNCH_CHECK();
  const bool second = try_second_mv && !force_integer_mv;
  if (!force_integer_mv) {
    int16_t *lists = second ? i16(o_lists) : nullptr;
    if (second) {
      hipLaunchKernelGGL(single_fill_invalid_kernel, dim3((unsigned)((6 * n1 + 255) / 256)), dim3(256), 0, ctx->stream, lists, 6 * n);   // av1_set_fractional_mv
      AOMHIP_LAUNCH_CHECK();
    }
    rc = aomhip_subpel_tree_list_batch(ctx, src, ref, frame, bw, bh, sub, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_sl), cl, n, i16(o_mva), u32(o_erra),
                                       i32(o_dist), u32(o_ssea), lists);
    if (rc != AOMHIP_OK) return rc;
    if (second) {
      hipLaunchKernelGGL(single_second_list_kernel, dim3(g), dim3(256), 0, ctx->stream, blk(o_sl), fmv, sec, n, blk(o_sl2), reinterpret_cast<uint8_t *>(w + o_has2));
      AOMHIP_LAUNCH_CHECK();
      rc = aomhip_subpel_tree_list_batch(ctx, src, ref, frame, bw, bh, sub, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_sl2), cl, n, i16(o_mvb),
                                         u32(o_errb), i32(o_dist), u32(o_sseb), lists);
      if (rc != AOMHIP_OK) return rc;
    }
  }
  const aomhip_txfm_yrd_stats *ya = nullptr, *yb = nullptr;
  if (rd && second) {
    // the actual rd cost of each candidate (:378-391, :404-413): the predictor at the MV (av1_enc_build_inter_predictor, luma), its residual through
    // av1_estimate_txfm_yrd.  Both candidates of every block are measured; blocks without a second search ignore the second figure.
    aomhip_txfm_yrd_stats *sa = reinterpret_cast<aomhip_txfm_yrd_stats *>(w + o_yrda), *sb = reinterpret_cast<aomhip_txfm_yrd_stats *>(w + o_yrdb);
    const int16_t *mvs[2] = { i16(o_mva), i16(o_mvb) };
    aomhip_txfm_yrd_stats *st[2] = { sa, sb };
    for (int c = 0; c < 2; ++c) {
      rc = aomhip_build_inter_pred_batch(ctx, ref, frame, rd->pred, frame, bw, bh, d_blocks, mvs[c], n, rd->filter_x, rd->filter_y);
      if (rc != AOMHIP_OK) return rc;
      rc = aomhip::estimate_txfm_yrd_ws(ctx, src, rd->pred, frame, bw, bh, rd->qparams, rd->d_costs, rd->tx_type_rate, rd->rdmult, rd->lossless, rd->d_yrd_blocks, n,
                                        st[c], w + o_yrdws);
      if (rc != AOMHIP_OK) return rc;
    }
    ya = sa; yb = sb;
    if (rd->d_stats_first) AOMHIP_TRY(hipMemcpyAsync(rd->d_stats_first, sa, n1 * sizeof(aomhip_txfm_yrd_stats), hipMemcpyDeviceToDevice, ctx->stream));
    if (rd->d_stats_second) AOMHIP_TRY(hipMemcpyAsync(rd->d_stats_second, sb, n1 * sizeof(aomhip_txfm_yrd_stats), hipMemcpyDeviceToDevice, ctx->stream));
  }
  hipLaunchKernelGGL(single_finish_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, fmv, force_integer_mv, i16(o_mva), u32(o_erra), u32(o_ssea),
                     second ? i16(o_mvb) : nullptr, u32(o_errb), u32(o_sseb), n, d_mvjcost, d_mvcost_row, d_mvcost_col, d_best_mv, d_rate_mv, d_pred_sse, ya, yb,
                     reinterpret_cast<const uint8_t *>(w + o_has2), rd ? rd->rdmult : 0, rd ? rd->d_candidate_mvs : nullptr);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_single_motion_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                                 const aomhip_search_params *full, const aomhip_subpel_params *sub, int use_cost_list, int try_second_mv,
                                                 int force_integer_mv, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                                 const aomhip_search_block *d_blocks, const int16_t *d_start2, int n, int16_t *d_best_mv, int32_t *d_bestsme,
                                                 int32_t *d_rate_mv, uint32_t *d_pred_sse, int16_t *d_full_mv, int16_t *d_second_best_mv) {
  return single_motion_search_impl(ctx, src, ref, frame, bw, bh, full, sub, use_cost_list, try_second_mv, force_integer_mv, d_mvjcost, d_mvcost_row, d_mvcost_col,
                                   d_blocks, d_start2, n, d_best_mv, d_bestsme, d_rate_mv, d_pred_sse, d_full_mv, d_second_best_mv, nullptr);
}

extern "C" int aomhip_single_motion_search_rd_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int frame, int bw, int bh,
                                                    const aomhip_search_params *full, const aomhip_subpel_params *sub, int use_cost_list, int force_integer_mv,
                                                    const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                                    const aomhip_search_block *d_blocks, const int16_t *d_start2, int n, const aomhip_single_rd_params *rd,
                                                    int16_t *d_best_mv, int32_t *d_bestsme, int32_t *d_rate_mv, uint32_t *d_pred_sse, int16_t *d_full_mv,
                                                    int16_t *d_second_best_mv) {
  if (!rd) {
    set_error("aomhip_single_motion_search_rd_batch: null argument");
    return AOMHIP_ERR_INVALID;
  }
  return single_motion_search_impl(ctx, src, ref, frame, bw, bh, full, sub, use_cost_list, /*try_second_mv=*/1, force_integer_mv, d_mvjcost, d_mvcost_row, d_mvcost_col,
                                   d_blocks, d_start2, n, d_best_mv, d_bestsme, d_rate_mv, d_pred_sse, d_full_mv, d_second_best_mv, rd);
}

// ---- av1_joint_motion_search (av1/encoder/motion_search_facade.c:496-702) for independent compound blocks.  The branch of speed >= 1
// (disable_extensive_joint_motion_search, or COMPOUND_WEDGE): up to four alternating iterations -- the other reference's predictor at cur_mv[!id]
// (av1_enc_build_one_inter_predictor, EIGHTTAP_REGULAR), av1_refining_search_8p_c from get_fullmv_from_mv(cur_mv[id]) against it, the compound
// sub-pel tree from the result (forced_stop EIGHTH_PEL) -- a block stops at the first iteration that does not lower its reference's error
// (:689-696) or whose MVs are back at the initial ones (:544-562); then *rate_mv and min(last_besterr).  All four iterations are launched for
// the whole batch; a block that has stopped is carried along and its later results are dropped.
namespace aomhip {
namespace {
__device__ __forceinline__ void joint_prepare_one(int i, const aomhip_search_block *blocks, const int16_t *ref_mv, const int16_t *cur_mv, const int16_t *init_mv,
                                                  int ite, uint8_t *live, aomhip_search_block *full_list, int16_t *other_mv) {
  const int id = ite & 1;
  const int16_t *cm = cur_mv + 4 * i, *im = init_mv + 4 * i;
  if (live[i] && ite >= 2 && cm[2 * !id] == im[2 * !id] && cm[2 * !id + 1] == im[2 * !id + 1]) {   // (:544-562)
    if (cm[2 * id] == im[2 * id] && cm[2 * id + 1] == im[2 * id + 1]) live[i] = 0;
    else if ((cm[2 * id] >> 3) == (im[2 * id] >> 3) && (cm[2 * id + 1] >> 3) == (im[2 * id + 1] >> 3)) live[i] = 0;
  }
  aomhip_search_block b = blocks[i];
  b.ref_row = ref_mv[4 * i + 2 * id]; b.ref_col = ref_mv[4 * i + 2 * id + 1];
  aomhip_search_block o = b;
  o.start_row = (int16_t)rawpel(cm[2 * id]); o.start_col = (int16_t)rawpel(cm[2 * id + 1]);   // get_fullmv_from_mv(&cur_mv[id])
  full_limits_ref(b, &o);   // av1_make_default_fullpel_ms_params: av1_set_mv_search_range(&mv_limits, ref_mv) on x->mv_limits
  if (!live[i]) { o.row_min = 1; o.row_max = 0; }   // the block has left the loop: an empty window, the search kernels skip it
  full_list[i] = o;
  other_mv[2 * i] = cm[2 * !id]; other_mv[2 * i + 1] = cm[2 * !id + 1];
}
__global__ void joint_prepare_kernel(const aomhip_search_block *blocks, const int16_t *ref_mv, const int16_t *cur_mv, const int16_t *init_mv, int ite,
                                     int n, uint8_t *live, aomhip_search_block *full_list, int16_t *other_mv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  joint_prepare_one(i, blocks, ref_mv, cur_mv, init_mv, ite, live, full_list, other_mv);
}
__global__ void joint_subpel_list_kernel(const aomhip_search_block *blocks, const int16_t *ref_mv, const int16_t *full_mv, int id, int n,
                                         const uint8_t *live, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aomhip_search_block b = blocks[i];
  b.ref_row = ref_mv[4 * i + 2 * id]; b.ref_col = ref_mv[4 * i + 2 * id + 1];
  aomhip_search_block o = b;
  o.start_row = (int16_t)(full_mv[2 * i] * 8); o.start_col = (int16_t)(full_mv[2 * i + 1] * 8);   // get_mv_from_fullmv
  subpel_limits_ref(b, &o);   // av1_set_subpel_mv_search_range(.., &x->mv_limits, ref_mv)
  if (live && !live[i]) { o.row_min = 1; o.row_max = 0; }   // (skipped by the sub-pel kernel)
  out[i] = o;
}
// try_second (:621-623, :664-676): the sub-pel search is repeated from second_best_mv when that is valid, differs from best_mv and lies inside the
// sub-pel limits; the other blocks are carried along from best_mv and their second result is dropped (use_second 0)
__global__ void joint_second_list_kernel(const aomhip_search_block *sub_list, const int16_t *full_mv, const int16_t *second, int n, aomhip_search_block *out,
                                         uint8_t *use_second) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aomhip_search_block o = sub_list[i];
  const int sr = second[2 * i], sc = second[2 * i + 1];
  const bool differs = sr != full_mv[2 * i] || sc != full_mv[2 * i + 1];
  const bool use = !(sr == kInvalidMv && sc == kInvalidMv) && differs && sc * 8 >= o.col_min && sc * 8 <= o.col_max && sr * 8 >= o.row_min && sr * 8 <= o.row_max;
  if (use) { o.start_row = (int16_t)(sr * 8); o.start_col = (int16_t)(sc * 8); }
  else { o.row_min = 1; o.row_max = 0; }   // no second start for this block (or it has left the loop: sub_list carries the mark): skipped
  use_second[i] = use;
  out[i] = o;
}
__device__ __forceinline__ void joint_update_one(int i, int id, int force_integer_mv, const int16_t *full_mv, const int32_t *full_sad, const int16_t *sub_mv,
                                                 const uint32_t *sub_err, const uint8_t *use_second, const int16_t *sub_mv2, const uint32_t *sub_err2, uint8_t *live,
                                                 int32_t *last_besterr, int16_t *cur_mv) {
  if (!live[i]) return;
  int bestsme = full_sad[i], row = full_mv[2 * i] * 8, col = full_mv[2 * i + 1] * 8;   // convert_fullmv_to_mv (:630-632)
  if (bestsme < INT_MAX && !force_integer_mv) {
    bestsme = (int)sub_err[i]; row = sub_mv[2 * i]; col = sub_mv[2 * i + 1];
    if (use_second && use_second[i] && (int)sub_err2[i] < bestsme) { bestsme = (int)sub_err2[i]; row = sub_mv2[2 * i]; col = sub_mv2[2 * i + 1]; }
  }
  if (bestsme < last_besterr[2 * i + id]) {
    cur_mv[4 * i + 2 * id] = (int16_t)row; cur_mv[4 * i + 2 * id + 1] = (int16_t)col;
    last_besterr[2 * i + id] = bestsme;
  } else {
    live[i] = 0;
  }
}
// the end of iteration `ite` and the head of the next one in ONE launch (both are per-block; the iteration's last launch is update alone)
__global__ void joint_update_prepare_kernel(int ite, int n, int force_integer_mv, const int16_t *full_mv, const int32_t *full_sad, const int16_t *sub_mv,
                                            const uint32_t *sub_err, const uint8_t *use_second, const int16_t *sub_mv2, const uint32_t *sub_err2, uint8_t *live,
                                            int32_t *last_besterr, int16_t *cur_mv, const aomhip_search_block *blocks, const int16_t *ref_mv, const int16_t *init_mv,
                                            aomhip_search_block *full_list, int16_t *other_mv, int prepare_next) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  joint_update_one(i, ite & 1, force_integer_mv, full_mv, full_sad, sub_mv, sub_err, use_second, sub_mv2, sub_err2, live, last_besterr, cur_mv);
  if (prepare_next) joint_prepare_one(i, blocks, ref_mv, cur_mv, init_mv, ite + 1, live, full_list, other_mv);
}
__global__ void joint_finish_kernel(int n, const int16_t *cur_mv, const int16_t *ref_mv, const int32_t *last_besterr, const int32_t *mvjcost,
                                    const int32_t *mvcost0, const int32_t *mvcost1, int32_t *rate_mv, int32_t *best_err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int rate = 0;
  for (int r = 0; r < 2; ++r) {   // av1_mv_bit_cost(.., MV_COST_WEIGHT) (mcomp.c:261-266)
    const int dr = cur_mv[4 * i + 2 * r] - ref_mv[4 * i + 2 * r], dc = cur_mv[4 * i + 2 * r + 1] - ref_mv[4 * i + 2 * r + 1];
    const int64_t bits = (int64_t)mvjcost[(dc != 0) | ((dr != 0) << 1)] + mvcost0[dr] + mvcost1[dc];
    rate += (int)((bits * 108 + 64) >> 7);
  }
  rate_mv[i] = rate;
  best_err[i] = last_besterr[2 * i] < last_besterr[2 * i + 1] ? last_besterr[2 * i] : last_besterr[2 * i + 1];
}
__global__ void joint_init_kernel(int n, const int16_t *cur_mv, int16_t *init_mv, uint8_t *live, int32_t *last_besterr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 4; ++k) init_mv[4 * i + k] = cur_mv[4 * i + k];
  live[i] = 1;
  last_besterr[2 * i] = last_besterr[2 * i + 1] = INT_MAX;
}
}  // namespace
}  // namespace aomhip

// `full` null: the 8-neighbour refinement (disable_extensive_joint_motion_search, or COMPOUND_WEDGE); non-null: av1_full_pixel_search on the
// compound prediction with these parameters (:613-617) and, with allow_second_mv, the second sub-pel start
static int joint_motion_search(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref0, const aomhip_planes *ref1, int frame, int bw, int bh,
                               const aomhip_search_params *full, int allow_second_mv, int mv_cost_type, int sad_per_bit, const aomhip_subpel_params *sub,
                               int force_integer_mv, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                               const aomhip_search_block *d_blocks, const int16_t *d_ref_mv, int16_t *d_cur_mv, const uint8_t *d_mask, int n,
                               int32_t *d_rate_mv, int32_t *d_best_err) {
  if (!ctx || !src || !ref0 || !ref1 || !sub || n < 0 || !d_mvjcost || !d_mvcost_row || !d_mvcost_col ||
      (n > 0 && (!d_blocks || !d_ref_mv || !d_cur_mv || !d_rate_mv || !d_best_err))) {
    set_error("aomhip_joint_motion_search_batch: invalid argument (the rate of the result needs the MV cost tables)");
    return AOMHIP_ERR_INVALID;
  }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t n1 = (size_t)n, SB = sizeof(aomhip_search_block), px = (size_t)bw * bh * (src->bit_depth == 8 ? 1 : 2);
  const size_t o_fl = take(n1 * SB), o_sl = take(n1 * SB), o_init = take(n1 * 8), o_live = take(n1), o_last = take(n1 * 8), o_other = take(n1 * 4),
               o_fmv = take(n1 * 4), o_fsad = take(n1 * 4), o_fvar = take(n1 * 4), o_smv = take(n1 * 4), o_serr = take(n1 * 4), o_dist = take(n1 * 4),
               o_sse = take(n1 * 4), o_sec = take(n1 * 4), o_sl2 = take(n1 * SB), o_use2 = take(n1), o_smv2 = take(n1 * 4), o_serr2 = take(n1 * 4),
               o_pred = take(n1 * px);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  auto blk = [&](size_t o) { return reinterpret_cast<aomhip_search_block *>(w + o); };
  auto i16 = [&](size_t o) { return reinterpret_cast<int16_t *>(w + o); };
  auto i32 = [&](size_t o) { return reinterpret_cast<int32_t *>(w + o); };
  auto u32 = [&](size_t o) { return reinterpret_cast<uint32_t *>(w + o); };
  uint8_t *live = reinterpret_cast<uint8_t *>(w + o_live);
  const unsigned g = (unsigned)((n1 + 255) / 256);
  hipLaunchKernelGGL(joint_init_kernel, dim3(g), dim3(256), 0, ctx->stream, n, d_cur_mv, i16(o_init), live, i32(o_last));
  AOMHIP_LAUNCH_CHECK();
  aomhip_subpel_params sp = *sub;
  sp.forced_stop = 0;   // ms_params.forced_stop = EIGHTH_PEL (:645)
  sp.mv_cost_type = mv_cost_type;
  for (int ite = 0; ite < 4; ++ite) {
    const int id = ite & 1;
    const aomhip_planes *rid = id ? ref1 : ref0, *roth = id ? ref0 : ref1;
    if (ite == 0) {   // (later iterations' lists come from the previous iteration's joint_update_prepare_kernel)
      hipLaunchKernelGGL(joint_prepare_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_ref_mv, d_cur_mv, i16(o_init), ite, n, live, blk(o_fl),
                         i16(o_other));
      AOMHIP_LAUNCH_CHECK();
    }
    int rc = aomhip_build_inter_pred_contiguous_batch(ctx, roth, frame, w + o_pred, bw, bh, d_blocks, i16(o_other), n, AOMHIP_INTERP_REGULAR,
                                                      AOMHIP_INTERP_REGULAR);
    if (rc != AOMHIP_OK) return rc;
    if (full)   // bestsme = av1_full_pixel_search(start_fullmv, &full_ms_params, 5, NULL, &best_mv, &second_best_mv)
      rc = aomhip_compound_full_pixel_search_batch(ctx, src, rid, frame, bw, bh, full, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_fl), n, w + o_pred, d_mask,
                                                   id, i16(o_fmv), i32(o_fsad), i16(o_sec));
    else
      rc = aomhip_refining_search_8p_batch(ctx, src, rid, frame, bw, bh, mv_cost_type, sad_per_bit, sub->error_per_bit, d_mvjcost, d_mvcost_row, d_mvcost_col,
                                           blk(o_fl), n, w + o_pred, d_mask, id, i16(o_fmv), i32(o_fsad), i32(o_fvar));
    if (rc != AOMHIP_OK) return rc;
    const bool second = full && allow_second_mv && !force_integer_mv;
    if (!force_integer_mv) {
      hipLaunchKernelGGL(joint_subpel_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_ref_mv, i16(o_fmv), id, n, live, blk(o_sl));
      AOMHIP_LAUNCH_CHECK();
      rc = aomhip_compound_subpel_tree_batch(ctx, src, rid, frame, bw, bh, &sp, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_sl), n, w + o_pred, d_mask, id,
                                             i16(o_smv), u32(o_serr), i32(o_dist), u32(o_sse));
      if (rc != AOMHIP_OK) return rc;
      if (second) {
        hipLaunchKernelGGL(joint_second_list_kernel, dim3(g), dim3(256), 0, ctx->stream, blk(o_sl), i16(o_fmv), i16(o_sec), n, blk(o_sl2),
                           reinterpret_cast<uint8_t *>(w + o_use2));
        AOMHIP_LAUNCH_CHECK();
        rc = aomhip_compound_subpel_tree_batch(ctx, src, rid, frame, bw, bh, &sp, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_sl2), n, w + o_pred, d_mask, id,
                                               i16(o_smv2), u32(o_serr2), i32(o_dist), u32(o_sse));
        if (rc != AOMHIP_OK) return rc;
      }
    }
    hipLaunchKernelGGL(joint_update_prepare_kernel, dim3(g), dim3(256), 0, ctx->stream, ite, n, force_integer_mv, i16(o_fmv), i32(o_fsad), i16(o_smv), u32(o_serr),
                       second ? reinterpret_cast<const uint8_t *>(w + o_use2) : nullptr, i16(o_smv2), u32(o_serr2), live, i32(o_last), d_cur_mv, d_blocks, d_ref_mv,
                       i16(o_init), blk(o_fl), i16(o_other), ite < 3 ? 1 : 0);
    AOMHIP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(joint_finish_kernel, dim3(g), dim3(256), 0, ctx->stream, n, d_cur_mv, d_ref_mv, i32(o_last), d_mvjcost, d_mvcost_row, d_mvcost_col,
                     d_rate_mv, d_best_err);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_joint_motion_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref0, const aomhip_planes *ref1, int frame,
                                                int bw, int bh, int mv_cost_type, int sad_per_bit, const aomhip_subpel_params *sub, int force_integer_mv,
                                                const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                                const aomhip_search_block *d_blocks, const int16_t *d_ref_mv, int16_t *d_cur_mv, const uint8_t *d_mask, int n,
                                                int32_t *d_rate_mv, int32_t *d_best_err) {
  return joint_motion_search(ctx, src, ref0, ref1, frame, bw, bh, nullptr, 0, mv_cost_type, sad_per_bit, sub, force_integer_mv, d_mvjcost, d_mvcost_row,
                             d_mvcost_col, d_blocks, d_ref_mv, d_cur_mv, d_mask, n, d_rate_mv, d_best_err);
}

extern "C" int aomhip_joint_motion_search_extensive_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref0, const aomhip_planes *ref1,
                                                          int frame, int bw, int bh, const aomhip_search_params *full, const aomhip_subpel_params *sub,
                                                          int allow_second_mv, int force_integer_mv, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                                                          const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, const int16_t *d_ref_mv,
                                                          int16_t *d_cur_mv, const uint8_t *d_mask, int n, int32_t *d_rate_mv, int32_t *d_best_err) {
  if (!full) {
    set_error("aomhip_joint_motion_search_extensive_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  return joint_motion_search(ctx, src, ref0, ref1, frame, bw, bh, full, allow_second_mv, full->mv_cost_type, full->sad_per_bit, sub, force_integer_mv,
                             d_mvjcost, d_mvcost_row, d_mvcost_col, d_blocks, d_ref_mv, d_cur_mv, d_mask, n, d_rate_mv, d_best_err);
}

// ---- av1_compound_single_motion_search[_interinter] (av1/encoder/motion_search_facade.c:703-853): ONE component of a compound refined against the
// fixed predictor of the other -- do_masked_motion_search_indexed / the interintra search.  Always the full search: av1_full_pixel_search(start, .., 5,
// NULL, &best, NULL) on the compound prediction (:758-764), then the compound sub-pel tree with forced_stop EIGHTH_PEL (:779-793).
namespace aomhip {
namespace {
__global__ void csingle_prepare_kernel(const aomhip_search_block *blocks, const int16_t *ref_mv, const int16_t *this_mv, int n, aomhip_search_block *full_list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aomhip_search_block b = blocks[i];
  b.ref_row = ref_mv[2 * i]; b.ref_col = ref_mv[2 * i + 1];
  aomhip_search_block o = b;
  o.start_row = (int16_t)rawpel(this_mv[2 * i]); o.start_col = (int16_t)rawpel(this_mv[2 * i + 1]);   // get_fullmv_from_mv(this_mv)
  full_limits_ref(b, &o);
  full_list[i] = o;
}
__global__ void csingle_subpel_list_kernel(const aomhip_search_block *blocks, const int16_t *ref_mv, const int16_t *full_mv, int n, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  aomhip_search_block b = blocks[i];
  b.ref_row = ref_mv[2 * i]; b.ref_col = ref_mv[2 * i + 1];
  aomhip_search_block o = b;
  o.start_row = (int16_t)(full_mv[2 * i] * 8); o.start_col = (int16_t)(full_mv[2 * i + 1] * 8);
  subpel_limits_ref(b, &o);
  out[i] = o;
}
__global__ void csingle_finish_kernel(int n, int force_integer_mv, const int16_t *full_mv, const int32_t *full_var, const int16_t *sub_mv, const uint32_t *sub_err,
                                      const int16_t *ref_mv, const int32_t *mvjcost, const int32_t *mvcost0, const int32_t *mvcost1, int16_t *this_mv,
                                      int32_t *rate_mv, int32_t *bestsme_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int bestsme = full_var[i], row = full_mv[2 * i] * 8, col = full_mv[2 * i + 1] * 8;   // convert_fullmv_to_mv (:773-775)
  if (bestsme < INT_MAX && !force_integer_mv) { bestsme = (int)sub_err[i]; row = sub_mv[2 * i]; col = sub_mv[2 * i + 1]; }
  if (bestsme < INT_MAX) { this_mv[2 * i] = (int16_t)row; this_mv[2 * i + 1] = (int16_t)col; }   // (:798)
  const int dr = this_mv[2 * i] - ref_mv[2 * i], dc = this_mv[2 * i + 1] - ref_mv[2 * i + 1];   // av1_mv_bit_cost(.., MV_COST_WEIGHT)
  const int64_t bits = (int64_t)mvjcost[(dc != 0) | ((dr != 0) << 1)] + mvcost0[dr] + mvcost1[dc];
  rate_mv[i] = (int)((bits * 108 + 64) >> 7);
  bestsme_out[i] = bestsme;
}
}  // namespace
}  // namespace aomhip

extern "C" int aomhip_compound_single_motion_search_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, const aomhip_planes *ref_other,
                                                          int frame, int bw, int bh, const aomhip_search_params *full, const aomhip_subpel_params *sub,
                                                          int force_integer_mv, const int32_t *d_mvjcost, const int32_t *d_mvcost_row,
                                                          const int32_t *d_mvcost_col, const aomhip_search_block *d_blocks, const int16_t *d_ref_mv,
                                                          int16_t *d_this_mv, const int16_t *d_other_mv, int interp_filter_x, int interp_filter_y,
                                                          const void *d_second_pred, const uint8_t *d_mask, int ref_idx, int n, int32_t *d_rate_mv,
                                                          int32_t *d_bestsme) {
  if (!ctx || !src || !ref || !full || (!sub && !force_integer_mv) || n < 0 || !d_mvjcost || !d_mvcost_row || !d_mvcost_col ||
      (n > 0 && (!d_blocks || !d_ref_mv || !d_this_mv || !d_rate_mv || !d_bestsme)) || (!d_second_pred && (!ref_other || !d_other_mv)) ||
      (ref_idx != 0 && ref_idx != 1)) {
    set_error("aomhip_compound_single_motion_search_batch: invalid argument (second_pred, or the other reference and its MVs; the MV cost tables)");
    return AOMHIP_ERR_INVALID;
  }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t n1 = (size_t)n, SB = sizeof(aomhip_search_block), px = (size_t)bw * bh * (src->bit_depth == 8 ? 1 : 2);
  const size_t o_fl = take(n1 * SB), o_sl = take(n1 * SB), o_fmv = take(n1 * 4), o_fvar = take(n1 * 4), o_sec = take(n1 * 4), o_smv = take(n1 * 4),
               o_serr = take(n1 * 4), o_dist = take(n1 * 4), o_sse = take(n1 * 4), o_pred = take(d_second_pred ? 0 : n1 * px);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  auto blk = [&](size_t o) { return reinterpret_cast<aomhip_search_block *>(w + o); };
  auto i16 = [&](size_t o) { return reinterpret_cast<int16_t *>(w + o); };
  auto i32 = [&](size_t o) { return reinterpret_cast<int32_t *>(w + o); };
  auto u32 = [&](size_t o) { return reinterpret_cast<uint32_t *>(w + o); };
  const unsigned g = (unsigned)((n1 + 255) / 256);
  int rc;
  const void *pred = d_second_pred;
  if (!pred) {   // build_second_inter_pred (:803-834): the other reference at other_mv with the block's own interpolation filters
    rc = aomhip_build_inter_pred_contiguous_batch(ctx, ref_other, frame, w + o_pred, bw, bh, d_blocks, d_other_mv, n, interp_filter_x, interp_filter_y);
    if (rc != AOMHIP_OK) return rc;
    pred = w + o_pred;
  }
  hipLaunchKernelGGL(csingle_prepare_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_ref_mv, d_this_mv, n, blk(o_fl));
  AOMHIP_LAUNCH_CHECK();
  rc = aomhip_compound_full_pixel_search_batch(ctx, src, ref, frame, bw, bh, full, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_fl), n, pred, d_mask, ref_idx,
                                               i16(o_fmv), i32(o_fvar), i16(o_sec));
  if (rc != AOMHIP_OK) return rc;
  if (!force_integer_mv) {
    aomhip_subpel_params sp = *sub;
    sp.forced_stop = 0;   // EIGHTH_PEL (:787)
    sp.mv_cost_type = full->mv_cost_type;
    hipLaunchKernelGGL(csingle_subpel_list_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_ref_mv, i16(o_fmv), n, blk(o_sl));
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_compound_subpel_tree_batch(ctx, src, ref, frame, bw, bh, &sp, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_sl), n, pred, d_mask, ref_idx,
                                           i16(o_smv), u32(o_serr), i32(o_dist), u32(o_sse));
    if (rc != AOMHIP_OK) return rc;
  }
  hipLaunchKernelGGL(csingle_finish_kernel, dim3(g), dim3(256), 0, ctx->stream, n, force_integer_mv, i16(o_fmv), i32(o_fvar), i16(o_smv), u32(o_serr), d_ref_mv,
                     d_mvjcost, d_mvcost_row, d_mvcost_col, d_this_mv, d_rate_mv, d_bestsme);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}


// ---- The inter leg of tpl_model.c's mode_estimation (av1/encoder/tpl_model.c:620-770) for blocks whose centre-MV candidates the caller has
// gathered (the candidates come from the TPL stats of the blocks above / left / above-right, :652-683: a raster dependency the host walks,
// wavefront by wavefront -- blocks (r, c) with 2 r + c equal; a batch = blocks that do not depend on each other).  Per block and reference frame:
//   prune_starting_mv (:706-731): the SAD of every candidate at its clamped full-pel position, the candidates ranked by it (qsort with
//       compare_sad, :308-315; ties keep their order: glibc's qsort is a merge sort), the count cut to 4 - prune_starting_mv and once more
//       when the last SAD is more than 20 % above the one before it,
//   motion_estimation (:248-301) from every remaining candidate, the first smallest error wins (:733-743),
//   av1_enc_build_one_inter_predictor at the winner with EIGHTTAP_REGULAR (:748-757), tpl_get_satd_cost (:199-212): residual, DCT_DCT of the
//       block's size (av1_quick_txfm with use_hadamard 0), aom_satd = the sum of the coefficients' magnitudes; pred_error = max(1, cost),
// then the reference with the smallest cost (first one on ties, :759-765).
namespace aomhip {
namespace {
constexpr int kTplCands = 4;
__global__ void tpl_center_cand_kernel(const aomhip_search_block *blocks, const int16_t *centers, const uint8_t *counts, int n, int ref, int n_refs,
                                       aomhip_sad_cand *out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * kTplCands) return;
  const int i = t / kTplCands, k = t % kTplCands;
  const aomhip_search_block b = blocks[i];
  const int cnt = counts[i * n_refs + ref];
  const int16_t *c = centers + ((size_t)(i * n_refs + ref) * kTplCands + (k < cnt ? k : 0)) * 2;
  int row = rawpel(c[0]), col = rawpel(c[1]);                       // get_fullmv_from_mv
  row = min(max(row, (int)b.row_min), (int)b.row_max);              // clamp_fullmv(&mv, &x->mv_limits)
  col = min(max(col, (int)b.col_min), (int)b.col_max);
  out[t] = aomhip_sad_cand{ b.bx, b.by, (int16_t)(b.bx + col), (int16_t)(b.by + row) };
}
// the ranking and the two cuts; writes one motion_estimation entry per (block, slot): ref_mv = the centre MV, raw limits, or the skip mark
__global__ void tpl_prune_kernel(const aomhip_search_block *blocks, const int16_t *centers, const uint8_t *counts, const uint32_t *sads, int n, int ref,
                                 int n_refs, int prune, aomhip_search_block *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const aomhip_search_block b = blocks[i];
  int cnt = counts[i * n_refs + ref];   // 0: this reference does not exist for the block (:633-637): nothing is searched
  cnt = cnt > kTplCands ? kTplCands : cnt;
  int order[kTplCands] = { 0, 1, 2, 3 };
  if (prune) {
    int sad[kTplCands];
    for (int k = 0; k < kTplCands; ++k) sad[k] = k < cnt ? (int)sads[i * kTplCands + k] : INT_MAX;
    if (cnt > 1) {   // insertion sort: stable, like the merge sort behind qsort
      for (int a = 1; a < cnt; ++a) {
        const int o = order[a], v = sad[o];
        int j = a - 1;
        while (j >= 0 && sad[order[j]] > v) { order[j + 1] = order[j]; --j; }
        order[j + 1] = o;
      }
    }
    cnt = min(4 - prune, cnt);   // (refmv_count = AOMMIN(4 - prune_starting_mv, refmv_count))
    if (cnt > 1) {
      const int last = sad[order[cnt - 1]], prev = sad[order[cnt - 2]];
      if ((last - prev) * 5 > prev) --cnt;
    }
  }
  for (int k = 0; k < kTplCands; ++k) {
    aomhip_search_block o = b;
    if (k < cnt) {
      const int16_t *c = centers + ((size_t)(i * n_refs + ref) * kTplCands + order[k]) * 2;
      o.ref_row = c[0]; o.ref_col = c[1];
    } else {
      o.row_min = 1; o.row_max = 0;   // not searched
    }
    out[i * kTplCands + k] = o;
  }
}
__global__ void tpl_best_cand_kernel(const aomhip_search_block *entries, const int16_t *mvs, const uint32_t *errs, int n, int16_t *best_mv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t best = 0xFFFFFFFFu;   // bestsme = UINT32_MAX; best_rfidx_mv = { 0 }
  int r = 0, c = 0, any = 0;
  for (int k = 0; k < kTplCands; ++k) {
    const int e = i * kTplCands + k;
    if (entries[e].row_min > entries[e].row_max) continue;
    any = 1;
    if (errs[e] < best) { best = errs[e]; r = mvs[2 * e]; c = mvs[2 * e + 1]; }
  }
  if (!any) r = c = -32768;   // INVALID_MV: the reference does not exist for this block
  best_mv[2 * i] = (int16_t)r; best_mv[2 * i + 1] = (int16_t)c;
}
template <typename T>
__global__ void tpl_residual_kernel(PlaneView<T> src, int frame, const aomhip_search_block *blocks, const T *pred, int n, int bw, int bh, int16_t *res) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int px = bw * bh;
  if (t >= (int64_t)n * px) return;
  const int i = (int)(t / px), q = (int)(t % px), y = q / bw, x = q % bw;
  const T *sp = src.origin + (int64_t)frame * src.frame_stride + (int64_t)(blocks[i].by + y) * src.stride + blocks[i].bx + x;
  res[t] = (int16_t)((int)*sp - (int)pred[t]);   // av1_subtract_block; block i = rows i * bh .. of a bw-wide residual plane
}
__global__ void tpl_satd_kernel(const int32_t *coeff, const uint8_t *counts, int n, int nc, int ref, int n_refs, int32_t *raw_cost, int32_t *pred_error) {
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (wave >= n) return;
  int acc = 0;   // aom_satd_c / aom_highbd_satd_c: int satd += abs(coeff[i])
  for (int k = lane; k < nc; k += 64) { const int v = coeff[(size_t)wave * nc + k]; acc += v < 0 ? -v : v; }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) {
    const bool have = counts[wave * n_refs + ref] != 0;
    raw_cost[wave * n_refs + ref] = have ? acc : INT_MAX;                    // inter_cost: what the references are compared by
    pred_error[wave * n_refs + ref] = have ? (acc > 1 ? acc : 1) : INT_MAX;   // tpl_stats->pred_error = AOMMAX(1, inter_cost)
  }
}
__global__ void tpl_best_ref_kernel(const int32_t *raw_cost, const uint8_t *counts, int n, int n_refs, int8_t *best_rf, int32_t *best_cost) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int best = INT_MAX, rf = -1;   // best_inter_cost = INT32_MAX, best_rf_idx = -1
  for (int r = 0; r < n_refs; ++r) {
    if (!counts[i * n_refs + r]) continue;
    const int c = raw_cost[i * n_refs + r];
    if (c < best) { best = c; rf = r; }   // (inter_cost < best_inter_cost: the first smallest)
  }
  best_rf[i] = (int8_t)rf;
  best_cost[i] = best;
}
}  // namespace
}  // namespace aomhip

extern "C" int aomhip_tpl_inter_estimation_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *const *refs, int n_refs, int frame, int bw,
                                                 const aomhip_search_params *full, const aomhip_subpel_params *sub, int use_cost_list,
                                                 int prune_starting_mv, const int32_t *d_mvjcost, const int32_t *d_mvcost_row, const int32_t *d_mvcost_col,
                                                 const aomhip_search_block *d_blocks, const int16_t *d_center_mvs, const uint8_t *d_center_counts, int n,
                                                 int16_t *d_best_mv, int32_t *d_pred_error, int8_t *d_best_rf_idx, int32_t *d_best_inter_cost) {
  const int tx_size = bw == 8 ? 1 : bw == 16 ? 2 : bw == 32 ? 3 : -1;
  if (!ctx || !src || !refs || n_refs < 1 || n_refs > 7 || !full || !sub || n < 0 || tx_size < 0 || prune_starting_mv < 0 || prune_starting_mv > 3 ||
      (n > 0 && (!d_blocks || !d_center_mvs || !d_center_counts || !d_best_mv || !d_pred_error || !d_best_rf_idx || !d_best_inter_cost))) {
    set_error("aomhip_tpl_inter_estimation_batch: invalid argument (square blocks of 8, 16 or 32; 1 .. 7 references)");
    return AOMHIP_ERR_INVALID;
  }
  for (int r = 0; r < n_refs; ++r)
    if (!refs[r] || !refs[r]->base || refs[r]->bit_depth != src->bit_depth) {
      set_error("aomhip_tpl_inter_estimation_batch: reference %d missing or of another bit depth", r);
      return AOMHIP_ERR_INVALID;
    }
  if (n == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  const int bh = bw, px = bw * bh, K = kTplCands;
  const size_t n1 = (size_t)n, SB = sizeof(aomhip_search_block), es = src->bit_depth == 8 ? 1 : 2;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t o_cand = take(n1 * K * sizeof(aomhip_sad_cand)), o_sad = take(n1 * K * 4), o_ent = take(n1 * K * SB), o_fl = take(n1 * K * SB),
               o_sl = take(n1 * K * SB), o_fmv = take(n1 * K * 4), o_fcost = take(n1 * K * 4), o_cl = take(n1 * K * 20), o_mv = take(n1 * K * 4),
               o_err = take(n1 * K * 4), o_dist = take(n1 * K * 4), o_sse = take(n1 * K * 4), o_bmv = take(n1 * 4), o_pred = take(n1 * px * es),
               o_res = take(n1 * px * 2), o_coeff = take(n1 * px * 4), o_q = take(n1 * px * 4), o_dq = take(n1 * px * 4), o_eob = take(n1 * 2),
               o_raw = take(n1 * n_refs * 4);
  char *w = static_cast<char *>(work(ctx, off));
  if (!w) return AOMHIP_ERR_NOMEM;
  auto blk = [&](size_t o) { return reinterpret_cast<aomhip_search_block *>(w + o); };
  auto i16 = [&](size_t o) { return reinterpret_cast<int16_t *>(w + o); };
  auto i32 = [&](size_t o) { return reinterpret_cast<int32_t *>(w + o); };
  auto u32 = [&](size_t o) { return reinterpret_cast<uint32_t *>(w + o); };
  const unsigned g = (unsigned)((n1 + 255) / 256), gk = (unsigned)((n1 * K + 255) / 256);
  aomhip_quant_params qp;   // (the transform kernel quantises as well: any valid parameters, its levels are not used)
  for (int k = 0; k < 2; ++k) { qp.zbin[k] = 64; qp.round[k] = 32; qp.quant[k] = 1; qp.quant_shift[k] = 1 << 14; qp.dequant[k] = 64; }
  for (int r = 0; r < n_refs; ++r) {
    const aomhip_planes *ref = refs[r];
    int rc;
    if (prune_starting_mv) {
      hipLaunchKernelGGL(tpl_center_cand_kernel, dim3(gk), dim3(256), 0, ctx->stream, d_blocks, d_center_mvs, d_center_counts, n, r, n_refs,
                         reinterpret_cast<aomhip_sad_cand *>(w + o_cand));
      AOMHIP_LAUNCH_CHECK();
      rc = aomhip_sad_batch(ctx, src, ref, frame, 1, bw, bh, 0, reinterpret_cast<const aomhip_sad_cand *>(w + o_cand), n * K, 0, u32(o_sad));
      if (rc != AOMHIP_OK) return rc;
    }
    hipLaunchKernelGGL(tpl_prune_kernel, dim3(g), dim3(256), 0, ctx->stream, d_blocks, d_center_mvs, d_center_counts, u32(o_sad), n, r, n_refs,
                       prune_starting_mv, blk(o_ent));
    AOMHIP_LAUNCH_CHECK();
    // motion_estimation for every (block, slot) entry (aomhip_motion_estimation_batch's steps on this call's own work memory)
    int32_t *cl = use_cost_list ? i32(o_cl) : nullptr;
    hipLaunchKernelGGL(me_full_list_kernel, dim3(gk), dim3(256), 0, ctx->stream, blk(o_ent), n * K, blk(o_fl));
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_full_pixel_search_batch(ctx, src, ref, frame, bw, bh, full, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_fl), n * K, i16(o_fmv), i32(o_fcost),
                                        cl, nullptr);
    if (rc != AOMHIP_OK) return rc;
    hipLaunchKernelGGL(me_subpel_list_kernel, dim3(gk), dim3(256), 0, ctx->stream, blk(o_ent), i16(o_fmv), n * K, blk(o_sl));
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_subpel_tree_batch(ctx, src, ref, frame, bw, bh, sub, d_mvjcost, d_mvcost_row, d_mvcost_col, blk(o_sl), cl, n * K, i16(o_mv), u32(o_err),
                                  i32(o_dist), u32(o_sse));
    if (rc != AOMHIP_OK) return rc;
    hipLaunchKernelGGL(tpl_best_cand_kernel, dim3(g), dim3(256), 0, ctx->stream, blk(o_ent), i16(o_mv), u32(o_err), n, i16(o_bmv));
    AOMHIP_LAUNCH_CHECK();
    AOMHIP_TRY(hipMemcpy2DAsync(d_best_mv + 2 * r, (size_t)n_refs * 4, w + o_bmv, 4, 4, n1, hipMemcpyDeviceToDevice, ctx->stream));
    // predictor at the winner, residual, DCT_DCT, satd
    rc = aomhip_build_inter_pred_contiguous_batch(ctx, ref, frame, w + o_pred, bw, bh, d_blocks, i16(o_bmv), n, AOMHIP_INTERP_REGULAR, AOMHIP_INTERP_REGULAR);
    if (rc != AOMHIP_OK) return rc;
    const unsigned gp = (unsigned)((n1 * px + 255) / 256);
    if (es == 1)
      hipLaunchKernelGGL(tpl_residual_kernel<uint8_t>, dim3(gp), dim3(256), 0, ctx->stream, view_of<uint8_t>(*src), frame, d_blocks,
                         reinterpret_cast<const uint8_t *>(w + o_pred), n, bw, bh, i16(o_res));
    else
      hipLaunchKernelGGL(tpl_residual_kernel<uint16_t>, dim3(gp), dim3(256), 0, ctx->stream, view_of<uint16_t>(*src), frame, d_blocks,
                         reinterpret_cast<const uint16_t *>(w + o_pred), n, bw, bh, i16(o_res));
    AOMHIP_LAUNCH_CHECK();
    rc = aomhip_xform_quant_batch(ctx, i16(o_res), bw, tx_size, nullptr, n, 1, 0, &qp, src->bit_depth != 8, i32(o_coeff), i32(o_q), i32(o_dq),
                                  reinterpret_cast<uint16_t *>(w + o_eob));
    if (rc != AOMHIP_OK) return rc;
    hipLaunchKernelGGL(tpl_satd_kernel, dim3((unsigned)((n1 * 64 + 255) / 256)), dim3(256), 0, ctx->stream, i32(o_coeff), d_center_counts, n, px, r, n_refs,
                       i32(o_raw), d_pred_error);
    AOMHIP_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(tpl_best_ref_kernel, dim3(g), dim3(256), 0, ctx->stream, i32(o_raw), d_center_counts, n, n_refs, d_best_rf_idx, d_best_inter_cost);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
