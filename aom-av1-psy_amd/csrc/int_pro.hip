// The projection-based motion estimation of the real-time path: av1_int_pro_motion_estimation (av1/encoder/mcomp.c:1897-2105; callers
// av1/encoder/var_based_part.c -- the superblock's vector before variance partitioning -- and av1/encoder/nonrd_pickmode.c) with
// aom_int_pro_row / aom_int_pro_col / aom_vector_var (aom_dsp/avg.c:536-581).  One wavefront per block:
//   projections    the reference window of twice the block's size onto a row of column sums (hbuf, 2 bw entries, rows of the block) and a column of
//                  row sums (vbuf, 2 bh entries, columns of the block), the block itself onto src_hbuf / src_vbuf, each sum normalised by its
//                  shift; 16-bit values in the wavefront's 1.5 KB of LDS.  Column sums: a lane per column, rows walked (coalesced); row sums: a
//                  lane per row;
//   vector_match   per direction: the offsets 0, 16, .. bw, then +- 8, 4, 2, 1 around the running best (strict <, first wins), every candidate's
//                  projection variance sse - mean^2 / width over the lanes with two wavefront sums (the reference's unsigned arithmetic);
//   the decision   the SAD of the block at that vector, at the zero vector if it differs, at its four neighbours and at one diagonal, in the
//                  reference's order with its strict comparisons; the vector times 8, clamped to the sub-pel limits of (mv_limits, ref_mv).
// Above 8 bits the reference only measures the zero vector (its vtable SAD, >> (bd - 8)); so does this.  A few hundred bytes to ~100 KB per block;
// the work per block is dominated by the projections and the seven SADs.
#include "common.h"
#include "search_device.h"

namespace aomhip {
namespace {

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// aom_vector_var_c: the variance of the difference of two projections of `width` = 4 << bwl entries
__device__ __forceinline__ int vector_var(const int16_t *ref, const int16_t *src, int width, int bwl, int lane) {
  int mean = 0, sse = 0;
  for (int i = lane; i < width; i += 64) {
    const int d = ref[i] - src[i];
    mean += d;
    sse += d * d;
  }
  mean = wave_sum(mean);
  sse = wave_sum(sse);
  const unsigned mean_abs = (unsigned)abs(mean);
  return (int)((unsigned)sse - ((mean_abs * mean_abs) >> (bwl + 2)));
}

// vector_match (mcomp.c:1897-1960): the offset of `src` in `ref` (bw + 1 positions), relative to the centre
__device__ __forceinline__ int vector_match(const int16_t *ref, const int16_t *src, int bw, int bwl, int lane) {
  int best = INT_MAX, offset = 0;
  for (int d = 0; d <= bw; d += 16) {
    const int s = vector_var(ref + d, src, bw, bwl, lane);
    if (s < best) { best = s; offset = d; }
  }
  int center = offset;
  for (int step = 8; step >= 1; step >>= 1) {
    for (int d = -step; d <= step; d += 2 * step) {
      const int pos = offset + d;
      if (pos < 0 || pos > bw) continue;
      const int s = vector_var(ref + pos, src, bw, bwl, lane);
      if (s < best) { best = s; center = pos; }
    }
    offset = center;
  }
  return center - (bw >> 1);
}

template <typename T>
__device__ __forceinline__ unsigned block_sad(const T *__restrict__ s, int s_stride, const T *__restrict__ r, int r_stride, int bw, int bwl, int n_px, int lane) {
  int acc = 0;
  for (int i = lane; i < n_px; i += 64) {
    const int y = i >> (bwl + 2), x = i & (bw - 1);
    acc += abs((int)s[y * s_stride + x] - (int)r[y * r_stride + x]);   // (offsets inside a plane fit 32 bits)
  }
  return (unsigned)wave_sum(acc);
}

template <typename T>
__global__ __launch_bounds__(256) void int_pro_kernel(PlaneView<T> src, int src_frame, PlaneView<T> ref, int ref_frame, int bw, int bh, int bd,
                                                      const aomhip_search_block *__restrict__ blocks, int n_blocks, int16_t *__restrict__ best_mv,
                                                      uint32_t *__restrict__ best_sad_out) {
  __shared__ int16_t proj[4][256 + 256 + 128 + 128];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int bi = blockIdx.x * 4 + wave;
  if (bi >= n_blocks) return;
  const aomhip_search_block b = blocks[bi];
  const int bx = __builtin_amdgcn_readfirstlane((int)b.bx), by = __builtin_amdgcn_readfirstlane((int)b.by);
  const T *s = src.origin + (int64_t)src_frame * src.frame_stride + (int64_t)by * src.stride + bx;
  const T *r = ref.origin + (int64_t)ref_frame * ref.frame_stride + (int64_t)by * ref.stride + bx;
  const int bwl = 31 - __builtin_clz(bw) - 2, bhl = 31 - __builtin_clz(bh) - 2, n_px = bw * bh;
  if constexpr (sizeof(T) == 2) {   // xd->bd != 8: the zero vector's SAD through the vtable's wrapper (encoder_utils.h:155-208)
    const unsigned sad = block_sad<T>(s, src.stride, r, ref.stride, bw, bwl, n_px, lane) >> (bd - 8);
    if (lane == 0) {
      best_mv[2 * bi] = 0;
      best_mv[2 * bi + 1] = 0;
      best_sad_out[bi] = sad;
    }
    return;
  } else {
    int16_t *hbuf = proj[wave], *vbuf = hbuf + 256, *src_h = vbuf + 256, *src_v = src_h + 128;
    const int row_norm = bhl + 1, col_norm = 3 + (bw >> 5);
    // aom_int_pro_row: column sums over the block's rows -- the reference window from bw / 2 to the left, then the block
    for (int idx = lane; idx < 2 * bw; idx += 64) {
      const T *p = r - (bw >> 1) + idx;
      int acc = 0;
      for (int i = 0; i < bh; ++i) acc += (int)p[i * ref.stride];
      hbuf[idx] = (int16_t)(acc >> row_norm);
    }
    for (int idx = lane; idx < bw; idx += 64) {
      int acc = 0;
      for (int i = 0; i < bh; ++i) acc += (int)s[i * src.stride + idx];
      src_h[idx] = (int16_t)(acc >> row_norm);
    }
    // aom_int_pro_col: row sums over the block's columns -- the reference window from bh / 2 above
    // (a lane per row reads 64 different cache lines per instruction, but the lanes' loops are independent; a row over min(bw, 64) neighbouring
    //  lanes -- one coalesced read and a butterfly sum per row -- makes every row a dependent load -> 6 cross-lane steps chain and measured
    //  163 us instead of 89 us per 4K frame of 64 x 64 blocks, 180 instead of 127 us at 32 x 32: at ~2 wavefronts per SIMD the kernel is bound
    //  by latency, not by the texture path)
    for (int ht = lane; ht < 2 * bh; ht += 64) {
      const T *p = r + (int64_t)(ht - (bh >> 1)) * ref.stride;
      int acc = 0;
      for (int idx = 0; idx < bw; ++idx) acc += (int)p[idx];
      vbuf[ht] = (int16_t)(acc >> col_norm);
    }
    for (int ht = lane; ht < bh; ht += 64) {
      const T *p = s + (int64_t)ht * src.stride;
      int acc = 0;
      for (int idx = 0; idx < bw; ++idx) acc += (int)p[idx];
      src_v[ht] = (int16_t)(acc >> col_norm);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int col = vector_match(hbuf, src_h, bw, bwl, lane), row = vector_match(vbuf, src_v, bh, bhl, lane);
    int trow = row, tcol = col;   // this_mv
    auto sad_at = [&](int rr, int cc) { return block_sad<T>(s, src.stride, r + (rr * ref.stride + cc), ref.stride, bw, bwl, n_px, lane); };
    unsigned best = sad_at(trow, tcol);
    if (row != 0 || col != 0) {   // the zero vector
      const unsigned t = sad_at(0, 0);
      if (t < best) { row = col = trow = tcol = 0; best = t; }
    }
    const unsigned s_up = sad_at(trow - 1, tcol), s_left = sad_at(trow, tcol - 1), s_right = sad_at(trow, tcol + 1), s_down = sad_at(trow + 1, tcol);
    if (s_up < best) { best = s_up; row = trow - 1; col = tcol; }
    if (s_left < best) { best = s_left; row = trow; col = tcol - 1; }
    if (s_right < best) { best = s_right; row = trow; col = tcol + 1; }
    if (s_down < best) { best = s_down; row = trow + 1; col = tcol; }
    trow += s_up < s_down ? -1 : 1;
    tcol += s_left < s_right ? -1 : 1;
    const unsigned t = sad_at(trow, tcol);
    if (best > t) { row = trow; col = tcol; best = t; }
    if (lane == 0) {
      // convert_fullmv_to_mv, clamp_mv to av1_set_subpel_mv_search_range(x->mv_limits, ref_mv) (mcomp.h:344-361)
      const int max_mv = 1023 * 8, lo = -(1 << 14) + 1, hi = (1 << 14) - 1;   // MAX_FULL_PEL_VAL, MV_LOW + 1, MV_UPP - 1
      const int minc = max(max((int)b.col_min * 8, (int)b.ref_col - max_mv), lo), maxc = min(min((int)b.col_max * 8, (int)b.ref_col + max_mv), hi);
      const int minr = max(max((int)b.row_min * 8, (int)b.ref_row - max_mv), lo), maxr = min(min((int)b.row_max * 8, (int)b.ref_row + max_mv), hi);
      const int mvc = col * 8, mvr = row * 8;
      best_mv[2 * bi] = (int16_t)(mvr < minr ? minr : (mvr > maxr ? maxr : mvr));
      best_mv[2 * bi + 1] = (int16_t)(mvc < minc ? minc : (mvc > maxc ? maxc : mvc));
      best_sad_out[bi] = best;
    }
  }
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_int_pro_motion_estimation_batch(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *ref, int ref_frame, int bw,
                                                      int bh, const aomhip_search_block *d_blocks, int n_blocks, int16_t *d_best_mv, uint32_t *d_best_sad) {
  const bool size_ok = (bw == 16 || bw == 32 || bw == 64 || bw == 128) && (bh == 16 || bh == 32 || bh == 64 || bh == 128);   // aom_vector_var: bwl 2 .. 5
  if (!ctx || !src || !ref || !src->base || !ref->base || n_blocks < 0 || (n_blocks > 0 && !d_blocks) || src_frame < 0 || src_frame >= src->n_frames ||
      ref_frame < 0 || ref_frame >= ref->n_frames || src->bit_depth != ref->bit_depth || !size_ok || !d_best_mv || !d_best_sad ||
      (src->bit_depth == 8 && ref->border < (bw > bh ? bw : bh) / 2 + 1)) {
    set_error("aomhip_int_pro_motion_estimation_batch: invalid argument (blocks of 16 .. 128 pixels a side; the reference's border must hold the window)");
    return AOMHIP_ERR_INVALID;
  }
  if (n_blocks == 0) return AOMHIP_OK;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  const dim3 grid((unsigned)((n_blocks + 3) / 4)), block(256);
  if (src->bit_depth == 8)
    hipLaunchKernelGGL(int_pro_kernel<uint8_t>, grid, block, 0, ctx->stream, view_of<uint8_t>(*src), src_frame, view_of<uint8_t>(*ref), ref_frame, bw, bh, 8, d_blocks,
                       n_blocks, d_best_mv, d_best_sad);
  else
    hipLaunchKernelGGL(int_pro_kernel<uint16_t>, grid, block, 0, ctx->stream, view_of<uint16_t>(*src), src_frame, view_of<uint16_t>(*ref), ref_frame, bw, bh,
                       src->bit_depth, d_blocks, n_blocks, d_best_mv, d_best_sad);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
