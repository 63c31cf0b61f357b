// The small members of the files the north star names that nothing else in the library covers (round 6, VERDICT r5 "missing" #4):
//   aom_get_mb_ss                         aom_dsp/variance.c:46-54            the sum of squares of a 16 x 16 int16 residual
//   aom_mse_wxh_16bit / _16xh_ / _highbd  aom_dsp/variance.c:1258-1297        CDEF search's distortion of an 8 / 16-bit plane against the 16-bit
//                                                                             filtered block (av1/encoder/pickcdef.c compute_cdef_dist)
//   aom_[highbd_]comp_mask_pred           aom_dsp/variance.c:773-791,841-862  the masked compound predictor as a stand-alone call
// Entry points with the reference's rtcd signatures (host pointers, aom_dsp/aom_dsp_rtcd_defs.pl:1344,1359,1362,1780,2032,2036): operands
// are staged through the default context's pinned / device scratch like the other exact-signature calls, one launch each.  The batched
// forms the encoder's hot loops use are elsewhere (aomhip_sum_sse_2d_i16_batch, aomhip_cdef_search..., aomhip_compound_batch): these exist so
// that every member of the rtcd table the named files define can be bound to the device.
#include "common.h"

namespace aomhip {
namespace {

// one workgroup: sum over a w x h block of (dst - src)^2, 64-bit
template <typename D>
__global__ __launch_bounds__(256) void mse_wxh_16bit_kernel(const D *__restrict__ dst, int dstride, const uint16_t *__restrict__ src, int sstride, int w,
                                                             int h, int n_blks, int blk_src_step, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long part[4];
  unsigned long long sum = 0;
  // (n_blks > 1: aom_mse_16xh_16bit -- blocks side by side in dst, one after the other in src)
  for (int t = threadIdx.x; t < n_blks * w * h; t += 256) {
    const int b = t / (w * h), r = t - b * (w * h), i = r / w, j = r - i * w;
    const int e = (int)dst[(int64_t)i * dstride + b * w + j] - (int)src[(int64_t)b * blk_src_step + (int64_t)i * sstride + j];
    sum += (unsigned long long)((long long)e * e);
  }
  for (int m = 1; m < 64; m <<= 1) sum += __shfl_xor(sum, m, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) *out = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void get_mb_ss_kernel(const int16_t *__restrict__ a, uint32_t *__restrict__ out) {
  __shared__ uint32_t part[4];
  const int v = a[threadIdx.x];
  uint32_t s = (uint32_t)(v * v);
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = part[0] + part[1] + part[2] + part[3];
}

// comp_pred[i * width + j] = AOM_BLEND_A64(mask, invert ? pred : ref, invert ? ref : pred) (aom_dsp/blend.h:24-33)
template <typename T>
__global__ __launch_bounds__(256) void comp_mask_pred_kernel(T *__restrict__ comp_pred, const T *__restrict__ pred, int width, int height,
                                                              const T *__restrict__ ref, int ref_stride, const uint8_t *__restrict__ mask,
                                                              int mask_stride, int invert_mask) {
  for (int t = blockIdx.x * 256 + threadIdx.x; t < width * height; t += gridDim.x * 256) {
    const int i = t / width, j = t - i * width;
    const int p = pred[t], r = ref[(int64_t)i * ref_stride + j], m = mask[(int64_t)i * mask_stride + j];
    const int a = invert_mask ? p : r, b = invert_mask ? r : p;
    comp_pred[t] = (T)((m * a + (64 - m) * b + 32) >> 6);
  }
}

// rows of `elems` elements gathered from a strided host block into a packed staging area
template <typename T> void pack_rows(T *dst, const T *src, int stride, int w, int h) {
  for (int r = 0; r < h; ++r) memcpy(dst + (size_t)r * w, src + (size_t)r * stride, (size_t)w * sizeof(T));
}

template <typename D>
uint64_t host_mse_16bit(const D *dst, int dstride, const uint16_t *src, int sstride, int w, int h, int n_blks, const char *who) {
  const uint64_t kFail = ~0ull;   // (a distortion: the largest value loses every comparison)
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return kFail;
  if (w < 1 || h < 1 || n_blks < 1 || w > 128 || h > 128 || !dst || !src) {
    set_error("%s: invalid argument", who);
    note_failure(who, AOMHIP_ERR_INVALID);
    return kFail;
  }
  const int dw = w * n_blks;
  const size_t d_bytes = ((size_t)dw * h * sizeof(D) + 15) & ~(size_t)15, s_bytes = ((size_t)w * h * n_blks * 2 + 15) & ~(size_t)15, total = d_bytes + s_bytes + 16;
  char *hb = static_cast<char *>(pinned(ctx, total)), *db = static_cast<char *>(scratch(ctx, total));
  if (!hb || !db) { note_failure(who, AOMHIP_ERR_NOMEM); return kFail; }
  pack_rows(reinterpret_cast<D *>(hb), dst, dstride, dw, h);
  if (n_blks == 1) pack_rows(reinterpret_cast<uint16_t *>(hb + d_bytes), src, sstride, w, h);
  else memcpy(hb + d_bytes, src, (size_t)w * h * n_blks * 2);   // (aom_mse_16xh_16bit: the blocks already lie packed, pitch w)
  if (hipMemcpyAsync(db, hb, d_bytes + s_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { note_failure(who); return kFail; }
  hipLaunchKernelGGL(mse_wxh_16bit_kernel<D>, dim3(1), dim3(256), 0, ctx->stream, reinterpret_cast<const D *>(db), dw,
                     reinterpret_cast<const uint16_t *>(db + d_bytes), w, w, h, n_blks, w * h,
                     reinterpret_cast<unsigned long long *>(db + d_bytes + s_bytes));
  if (hipGetLastError() != hipSuccess ||
      hipMemcpyAsync(hb + d_bytes + s_bytes, db + d_bytes + s_bytes, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    note_failure(who);
    return kFail;
  }
  return *reinterpret_cast<const unsigned long long *>(hb + d_bytes + s_bytes);
}

template <typename T>
void host_comp_mask_pred(T *comp_pred, const T *pred, int width, int height, const T *ref, int ref_stride, const uint8_t *mask, int mask_stride,
                         int invert_mask, const char *who) {
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  if (width < 1 || height < 1 || width > 128 || height > 128 || !comp_pred || !pred || !ref || !mask) {
    set_error("%s: invalid argument", who);
    note_failure(who, AOMHIP_ERR_INVALID);
    return;
  }
  const size_t n = (size_t)width * height, pb = (n * sizeof(T) + 15) & ~(size_t)15, mb = (n + 15) & ~(size_t)15, total = 3 * pb + mb;
  char *hb = static_cast<char *>(pinned(ctx, total)), *db = static_cast<char *>(scratch(ctx, total));
  if (!hb || !db) { note_failure(who, AOMHIP_ERR_NOMEM); return; }
  memcpy(hb, pred, n * sizeof(T));
  pack_rows(reinterpret_cast<T *>(hb + pb), ref, ref_stride, width, height);
  pack_rows(reinterpret_cast<uint8_t *>(hb + 2 * pb), mask, mask_stride, width, height);
  if (hipMemcpyAsync(db, hb, 2 * pb + mb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { note_failure(who); return; }
  hipLaunchKernelGGL(comp_mask_pred_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, reinterpret_cast<T *>(db + 2 * pb + mb),
                     reinterpret_cast<const T *>(db), width, height, reinterpret_cast<const T *>(db + pb), width,
                     reinterpret_cast<const uint8_t *>(db + 2 * pb), width, invert_mask);
  if (hipGetLastError() != hipSuccess ||
      hipMemcpyAsync(hb + 2 * pb + mb, db + 2 * pb + mb, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    note_failure(who);
    return;
  }
  memcpy(comp_pred, hb + 2 * pb + mb, n * sizeof(T));
}

}  // namespace
}  // namespace aomhip

using namespace aomhip;

extern "C" {

unsigned int aomhip_get_mb_ss(const int16_t *a) {
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return kFailedVarCost;
  if (!a) { set_error("aomhip_get_mb_ss: null argument"); note_failure("aomhip_get_mb_ss", AOMHIP_ERR_INVALID); return kFailedVarCost; }
  char *hb = static_cast<char *>(pinned(ctx, 512 + 16)), *db = static_cast<char *>(scratch(ctx, 512 + 16));
  if (!hb || !db) { note_failure("aomhip_get_mb_ss", AOMHIP_ERR_NOMEM); return kFailedVarCost; }
  memcpy(hb, a, 512);
  if (hipMemcpyAsync(db, hb, 512, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { note_failure("aomhip_get_mb_ss"); return kFailedVarCost; }
  hipLaunchKernelGGL(get_mb_ss_kernel, dim3(1), dim3(256), 0, ctx->stream, reinterpret_cast<const int16_t *>(db), reinterpret_cast<uint32_t *>(db + 512));
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hb + 512, db + 512, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    note_failure("aomhip_get_mb_ss");
    return kFailedVarCost;
  }
  return *reinterpret_cast<const uint32_t *>(hb + 512);
}

uint64_t aomhip_mse_wxh_16bit(uint8_t *dst, int dstride, uint16_t *src, int sstride, int w, int h) {
  return host_mse_16bit<uint8_t>(dst, dstride, src, sstride, w, h, 1, "aomhip_mse_wxh_16bit");
}

uint64_t aomhip_mse_16xh_16bit(uint8_t *dst, int dstride, uint16_t *src, int w, int h) {
  if (w < 1 || w > 16) { set_error("aomhip_mse_16xh_16bit: w = %d", w); note_failure("aomhip_mse_16xh_16bit", AOMHIP_ERR_INVALID); return ~0ull; }
  return host_mse_16bit<uint8_t>(dst, dstride, src, w, w, h, 16 / w, "aomhip_mse_16xh_16bit");
}

uint64_t aomhip_mse_wxh_16bit_highbd(uint16_t *dst, int dstride, uint16_t *src, int sstride, int w, int h) {
  return host_mse_16bit<uint16_t>(dst, dstride, src, sstride, w, h, 1, "aomhip_mse_wxh_16bit_highbd");
}

void aomhip_comp_mask_pred(uint8_t *comp_pred, const uint8_t *pred, int width, int height, const uint8_t *ref, int ref_stride, const uint8_t *mask,
                           int mask_stride, int invert_mask) {
  host_comp_mask_pred<uint8_t>(comp_pred, pred, width, height, ref, ref_stride, mask, mask_stride, invert_mask, "aomhip_comp_mask_pred");
}

void aomhip_highbd_comp_mask_pred(uint8_t *comp_pred8, const uint8_t *pred8, int width, int height, const uint8_t *ref8, int ref_stride,
                                  const uint8_t *mask, int mask_stride, int invert_mask) {
  uint16_t *comp_pred = reinterpret_cast<uint16_t *>(reinterpret_cast<uintptr_t>(comp_pred8) << 1);   // CONVERT_TO_SHORTPTR
  const uint16_t *pred = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(pred8) << 1);
  const uint16_t *ref = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(ref8) << 1);
  host_comp_mask_pred<uint16_t>(comp_pred, pred, width, height, ref, ref_stride, mask, mask_stride, invert_mask, "aomhip_highbd_comp_mask_pred");
}

}  // extern "C"
