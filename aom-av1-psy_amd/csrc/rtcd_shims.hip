// The rest of the rtcd-signature conformance surface: av1_fwd_txfm2d_WxH, av1_inv_txfm2d_add_WxH,
// aom_[highbd_]subtract_block, and the macro-stamped names of the quantiser / loop-filter / CDEF families, plus the
// installer aomhip_rtcd() that fills a table of those pointers the way setup_rtcd_internal assigns the reference's
// (build/cmake/rtcd.pl:189-209,262-290).  Host pointers in, results out, one launch per call, synchronous on the calling
// thread's default context: for the reference's own unit tests and plumbing, not for speed.  Every function runs the
// device code of the batched entry points (the transform / inverse go through aomhip_xform_quant_batch /
// aomhip_inv_txfm_add_batch themselves).  A failed call records the sticky status (aomhip_status()), leaves its outputs
// zeroed / untouched and returns: it never aborts and there is no CPU fallback.
#include "common.h"

namespace aomhip {

// aom_subtract_block_c / aom_highbd_subtract_block_c (aom_dsp/subtract.c:20-53): diff = src - pred
template <typename PIX>
__global__ __launch_bounds__(256) void subtract_kernel(const PIX *__restrict__ src, const PIX *__restrict__ pred, int16_t *__restrict__ diff, int n) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) diff[i] = (int16_t)((int)src[i] - (int)pred[i]);
}

static int tx_size_of(int w, int h) {
  for (int t = 0; t < 19; ++t)
    if (aomhip_tx_size_wide(t) == w && aomhip_tx_size_high(t) == h) return t;
  return -1;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

void aomhip_quantize_b_any(const int32_t *, intptr_t, const int16_t *, const int16_t *, const int16_t *, const int16_t *, int32_t *, int32_t *,
                           const int16_t *, uint16_t *, const int16_t *, const int16_t *, int, int, int);
void aomhip_lpf_any(void *, int, int, int, int, const uint8_t *, const uint8_t *, const uint8_t *, const uint8_t *, const uint8_t *,
                    const uint8_t *, int, int);
void aomhip_cdef_filter_any(void *, int, const uint16_t *, int, int, int, int, int, int, int, int, int, int);

// av1_fwd_txfm2d_WxH_c (av1/common/av1_rtcd_defs.pl:355-399; av1/encoder/av1_fwd_txfm2d.c:129-312): the coefficients in
// the reference's layout (transposed; 64-point sizes packed to their 32 low frequencies) -- aomhip_tx_max_eob(tx) values
// are written, which is what every caller reads.
void aomhip_fwd_txfm2d(const int16_t *input, int32_t *output, int stride, int tx_type, int bd, int w, int h) {
  const int tx = tx_size_of(w, h), nc = aomhip_tx_max_eob(tx);
  if (tx >= 0) memset(output, 0, (size_t)nc * 4);
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  if (tx < 0) {
    set_error("aomhip_fwd_txfm2d: no %dx%d transform", w, h);
    return note_failure("aomhip_fwd_txfm2d", AOMHIP_ERR_INVALID);
  }
  const size_t res_bytes = ((size_t)w * h * 2 + 15) & ~(size_t)15, c_off = res_bytes, q_off = c_off + (size_t)nc * 4, total = q_off + (size_t)nc * 8 + 16;
  char *hb = static_cast<char *>(pinned(ctx, total)), *d = static_cast<char *>(scratch(ctx, total));
  if (!hb || !d) return note_failure("aomhip_fwd_txfm2d scratch", AOMHIP_ERR_NOMEM);
  for (int r = 0; r < h; ++r) memcpy(hb + (size_t)r * w * 2, input + (ptrdiff_t)r * stride, (size_t)w * 2);
  if (hipMemcpyAsync(d, hb, res_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return note_failure("aomhip_fwd_txfm2d H2D");
  aomhip_quant_params qp = { { 1, 1 }, { 0, 0 }, { 1, 1 }, { 1, 1 }, { 1, 1 } };  // the quantiser's outputs are not used
  int32_t *dc = reinterpret_cast<int32_t *>(d + c_off), *dq = reinterpret_cast<int32_t *>(d + q_off);
  if (aomhip_xform_quant_batch(ctx, reinterpret_cast<const int16_t *>(d), w, tx, nullptr, 1, 1, tx_type, &qp, bd > 8, dc, dq, dq + nc,
                               reinterpret_cast<uint16_t *>(d + q_off + (size_t)nc * 8)) != AOMHIP_OK)
    return note_failure("aomhip_fwd_txfm2d launch", AOMHIP_ERR_INVALID);
  if (hipMemcpyAsync(hb + c_off, d + c_off, (size_t)nc * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return note_failure("aomhip_fwd_txfm2d D2H");
  memcpy(output, hb + c_off, (size_t)nc * 4);
}

// av1_inv_txfm2d_add_WxH_c (av1_rtcd_defs.pl:137-243; av1/common/av1_inv_txfm2d.c:311-): input = dequantised coefficients in
// the reference layout, output = uint16 pixels the residual is added to (clipped to bd).
void aomhip_inv_txfm2d_add(const int32_t *input, uint16_t *output, int stride, int tx_type, int bd, int w, int h) {
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  const int tx = tx_size_of(w, h), nc = aomhip_tx_max_eob(tx);
  if (tx < 0 || (bd != 8 && bd != 10 && bd != 12)) {
    set_error("aomhip_inv_txfm2d_add: no %dx%d transform / bit depth %d", w, h, bd);
    return note_failure("aomhip_inv_txfm2d_add", AOMHIP_ERR_INVALID);
  }
  const size_t pix_bytes = ((size_t)w * h * 2 + 255) & ~(size_t)255, c_off = pix_bytes, total = c_off + (size_t)nc * 4;
  char *hb = static_cast<char *>(pinned(ctx, total)), *d = static_cast<char *>(scratch(ctx, total));
  if (!hb || !d) return note_failure("aomhip_inv_txfm2d_add scratch", AOMHIP_ERR_NOMEM);
  // the batched entry types its planes by bit depth: bd 8 runs on uint8 pixels (the same arithmetic, av1_inv_txfm_add_c)
  const size_t esz = bd == 8 ? 1 : 2;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const uint16_t v = output[(ptrdiff_t)r * stride + c];
      if (esz == 1) reinterpret_cast<uint8_t *>(hb)[(size_t)r * w + c] = (uint8_t)v;
      else reinterpret_cast<uint16_t *>(hb)[(size_t)r * w + c] = v;
    }
  memcpy(hb + c_off, input, (size_t)nc * 4);
  if (hipMemcpyAsync(d, hb, total, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return note_failure("aomhip_inv_txfm2d_add H2D");
  aomhip_planes p;
  p.base = d; p.frame_stride = (int64_t)w * h; p.width = w; p.height = h; p.stride = w; p.border = 0; p.bit_depth = bd; p.n_frames = 1;
  if (aomhip_inv_txfm_add_batch(ctx, reinterpret_cast<const int32_t *>(d + c_off), tx, nullptr, 1, 1, tx_type, nullptr, &p, 0) != AOMHIP_OK)
    return note_failure("aomhip_inv_txfm2d_add launch", AOMHIP_ERR_INVALID);
  if (hipMemcpyAsync(hb, d, (size_t)w * h * esz, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
    return note_failure("aomhip_inv_txfm2d_add D2H");
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c)
      output[(ptrdiff_t)r * stride + c] = esz == 1 ? reinterpret_cast<const uint8_t *>(hb)[(size_t)r * w + c] : reinterpret_cast<const uint16_t *>(hb)[(size_t)r * w + c];
}

static void subtract_any(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const void *src, ptrdiff_t src_stride, const void *pred,
                         ptrdiff_t pred_stride, int is_hbd) {
  // (one failure convention for the shims, common.h: costs UINT32_MAX, coefficient-like outputs -- this residual -- zeroed, pixels untouched;
  // the size is checked before anything is written through it)
  if (rows < 1 || cols < 1 || rows > 128 || cols > 128) {
    set_error("aomhip_subtract_block: %dx%d unsupported", cols, rows);
    return note_failure("aomhip_subtract_block", AOMHIP_ERR_INVALID);
  }
  for (int r = 0; r < rows; ++r) memset(diff + r * diff_stride, 0, (size_t)cols * 2);
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  const size_t esz = is_hbd ? 2 : 1, n = (size_t)rows * cols, pb = (n * esz + 15) & ~(size_t)15, total = 2 * pb + n * 2;
  char *hb = static_cast<char *>(pinned(ctx, total)), *d = static_cast<char *>(scratch(ctx, total));
  if (!hb || !d) return note_failure("aomhip_subtract_block scratch", AOMHIP_ERR_NOMEM);
  for (int r = 0; r < rows; ++r) {
    memcpy(hb + (size_t)r * cols * esz, static_cast<const char *>(src) + r * src_stride * (ptrdiff_t)esz, (size_t)cols * esz);
    memcpy(hb + pb + (size_t)r * cols * esz, static_cast<const char *>(pred) + r * pred_stride * (ptrdiff_t)esz, (size_t)cols * esz);
  }
  if (hipMemcpyAsync(d, hb, 2 * pb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return note_failure("aomhip_subtract_block H2D");
  int16_t *dd = reinterpret_cast<int16_t *>(d + 2 * pb);
  if (is_hbd)
    hipLaunchKernelGGL(subtract_kernel<uint16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, reinterpret_cast<const uint16_t *>(d),
                       reinterpret_cast<const uint16_t *>(d + pb), dd, (int)n);
  else
    hipLaunchKernelGGL(subtract_kernel<uint8_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, reinterpret_cast<const uint8_t *>(d),
                       reinterpret_cast<const uint8_t *>(d + pb), dd, (int)n);
  if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hb + 2 * pb, dd, n * 2, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess)
    return note_failure("aomhip_subtract_block");
  for (int r = 0; r < rows; ++r) memcpy(diff + r * diff_stride, hb + 2 * pb + (size_t)r * cols * 2, (size_t)cols * 2);
}
// aom_dsp_rtcd_defs.pl:723,733.  The highbd form takes CONVERT_TO_BYTEPTR-encoded pointers like the reference (aom_ports/mem.h:79-80).
void aomhip_subtract_block(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src_ptr, ptrdiff_t src_stride,
                           const uint8_t *pred_ptr, ptrdiff_t pred_stride) {
  subtract_any(rows, cols, diff_ptr, diff_stride, src_ptr, src_stride, pred_ptr, pred_stride, 0);
}
void aomhip_highbd_subtract_block(int rows, int cols, int16_t *diff_ptr, ptrdiff_t diff_stride, const uint8_t *src8, ptrdiff_t src_stride,
                                  const uint8_t *pred8, ptrdiff_t pred_stride) {
  subtract_any(rows, cols, diff_ptr, diff_stride, reinterpret_cast<const void *>((uintptr_t)src8 << 1), src_stride,
               reinterpret_cast<const void *>((uintptr_t)pred8 << 1), pred_stride, 1);
}

// ---- the stamped names ----
#define AOMHIP_QB(NAME, LS, HBD, AD)                                                                                              \
  void NAME(const int32_t *coeff_ptr, intptr_t n_coeffs, const int16_t *zbin_ptr, const int16_t *round_ptr, const int16_t *quant_ptr, \
            const int16_t *quant_shift_ptr, int32_t *qcoeff_ptr, int32_t *dqcoeff_ptr, const int16_t *dequant_ptr, uint16_t *eob_ptr, \
            const int16_t *scan, const int16_t *iscan) {                                                                            \
    aomhip_quantize_b_any(coeff_ptr, n_coeffs, zbin_ptr, round_ptr, quant_ptr, quant_shift_ptr, qcoeff_ptr, dqcoeff_ptr, dequant_ptr, \
                          eob_ptr, scan, iscan, LS, HBD, AD);                                                                       \
  }
AOMHIP_QB(aomhip_quantize_b, 0, 0, 0) AOMHIP_QB(aomhip_quantize_b_32x32, 1, 0, 0) AOMHIP_QB(aomhip_quantize_b_64x64, 2, 0, 0)
AOMHIP_QB(aomhip_highbd_quantize_b, 0, 1, 0) AOMHIP_QB(aomhip_highbd_quantize_b_32x32, 1, 1, 0) AOMHIP_QB(aomhip_highbd_quantize_b_64x64, 2, 1, 0)
AOMHIP_QB(aomhip_quantize_b_adaptive, 0, 0, 1) AOMHIP_QB(aomhip_quantize_b_32x32_adaptive, 1, 0, 1) AOMHIP_QB(aomhip_quantize_b_64x64_adaptive, 2, 0, 1)
AOMHIP_QB(aomhip_highbd_quantize_b_adaptive, 0, 1, 1) AOMHIP_QB(aomhip_highbd_quantize_b_32x32_adaptive, 1, 1, 1)
AOMHIP_QB(aomhip_highbd_quantize_b_64x64_adaptive, 2, 1, 1)
#undef AOMHIP_QB

#define AOMHIP_TX(W, H)                                                                                                         \
  void aomhip_fwd_txfm2d_##W##x##H(const int16_t *input, int32_t *output, int stride, int tx_type, int bd) {                    \
    aomhip_fwd_txfm2d(input, output, stride, tx_type, bd, W, H);                                                                \
  }                                                                                                                             \
  void aomhip_inv_txfm2d_add_##W##x##H(const int32_t *input, uint16_t *output, int stride, int tx_type, int bd) {               \
    aomhip_inv_txfm2d_add(input, output, stride, tx_type, bd, W, H);                                                            \
  }
AOMHIP_RTCD_TX_SIZES(AOMHIP_TX)
#undef AOMHIP_TX

#define AOMHIP_LPF(DIR, HZ, LEN)                                                                                                 \
  void aomhip_lpf_##DIR##_##LEN(uint8_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh) {      \
    aomhip_lpf_any(s, pitch, HZ, LEN, 4, blimit, limit, thresh, nullptr, nullptr, nullptr, 8, 0);                                 \
  }                                                                                                                              \
  void aomhip_lpf_##DIR##_##LEN##_dual(uint8_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, \
                                       const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1) {                  \
    aomhip_lpf_any(s, pitch, HZ, LEN, 8, blimit0, limit0, thresh0, blimit1, limit1, thresh1, 8, 0);                               \
  }                                                                                                                              \
  void aomhip_lpf_##DIR##_##LEN##_quad(uint8_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0) { \
    aomhip_lpf_any(s, pitch, HZ, LEN, 16, blimit0, limit0, thresh0, nullptr, nullptr, nullptr, 8, 0);                             \
  }                                                                                                                              \
  void aomhip_highbd_lpf_##DIR##_##LEN(uint16_t *s, int pitch, const uint8_t *blimit, const uint8_t *limit, const uint8_t *thresh, int bd) { \
    aomhip_lpf_any(s, pitch, HZ, LEN, 4, blimit, limit, thresh, nullptr, nullptr, nullptr, bd, 1);                                \
  }                                                                                                                              \
  void aomhip_highbd_lpf_##DIR##_##LEN##_dual(uint16_t *s, int pitch, const uint8_t *blimit0, const uint8_t *limit0, const uint8_t *thresh0, \
                                              const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1, int bd) {   \
    aomhip_lpf_any(s, pitch, HZ, LEN, 8, blimit0, limit0, thresh0, blimit1, limit1, thresh1, bd, 1);                              \
  }
AOMHIP_LPF(horizontal, 1, 4) AOMHIP_LPF(horizontal, 1, 6) AOMHIP_LPF(horizontal, 1, 8) AOMHIP_LPF(horizontal, 1, 14)
AOMHIP_LPF(vertical, 0, 4) AOMHIP_LPF(vertical, 0, 6) AOMHIP_LPF(vertical, 0, 8) AOMHIP_LPF(vertical, 0, 14)
#undef AOMHIP_LPF

#define AOMHIP_CDEF(BITS, IS16, V)                                                                                               \
  void aomhip_cdef_filter_##BITS##_##V(void *dst, int dstride, const uint16_t *in, int pri_strength, int sec_strength, int dir,  \
                                       int pri_damping, int sec_damping, int coeff_shift, int block_width, int block_height) {  \
    aomhip_cdef_filter_any(dst, dstride, in, pri_strength, sec_strength, dir, pri_damping, sec_damping, coeff_shift, block_width, \
                           block_height, IS16, V);                                                                              \
  }
AOMHIP_CDEF(8, 0, 0) AOMHIP_CDEF(8, 0, 1) AOMHIP_CDEF(8, 0, 2) AOMHIP_CDEF(8, 0, 3)
AOMHIP_CDEF(16, 1, 0) AOMHIP_CDEF(16, 1, 1) AOMHIP_CDEF(16, 1, 2) AOMHIP_CDEF(16, 1, 3)
#undef AOMHIP_CDEF

// The installer: every pointer of the table, the reference's names minus the aom_ / av1_ prefix.
int aomhip_rtcd(aomhip_rtcd_table *t) {
  if (!t) return AOMHIP_ERR_INVALID;
  memset(t, 0, sizeof(*t));
  if (aomhip_device_count() <= 0) {
    set_error("aomhip_rtcd: no HIP device (libaomhip has no CPU fallback)");
    return AOMHIP_ERR_NO_DEVICE;  // the caller keeps its C / SIMD pointers, exactly like a missing ISA in setup_rtcd_internal
  }
  t->sad16x16 = aomhip_sad16x16; t->sad16x16x4d = aomhip_sad16x16x4d; t->variance16x16 = aomhip_variance16x16;
  // all 22 block sizes x {8-bit, highbd 10, highbd 12}: the same fixed-size functions the vtable binder installs
  for (int d = 0; d < 3; ++d)
    if (aomhip_bind_variance_vtable(t->block_fns[d], 8 + 2 * d) != AOMHIP_OK) return AOMHIP_ERR_INVALID;
  t->subtract_block = aomhip_subtract_block; t->highbd_subtract_block = aomhip_highbd_subtract_block;
  t->quantize_b = aomhip_quantize_b; t->quantize_b_32x32 = aomhip_quantize_b_32x32; t->quantize_b_64x64 = aomhip_quantize_b_64x64;
  t->highbd_quantize_b = aomhip_highbd_quantize_b; t->highbd_quantize_b_32x32 = aomhip_highbd_quantize_b_32x32;
  t->highbd_quantize_b_64x64 = aomhip_highbd_quantize_b_64x64;
  t->quantize_b_adaptive = aomhip_quantize_b_adaptive; t->quantize_b_32x32_adaptive = aomhip_quantize_b_32x32_adaptive;
  t->quantize_b_64x64_adaptive = aomhip_quantize_b_64x64_adaptive; t->highbd_quantize_b_adaptive = aomhip_highbd_quantize_b_adaptive;
  t->highbd_quantize_b_32x32_adaptive = aomhip_highbd_quantize_b_32x32_adaptive;
  t->highbd_quantize_b_64x64_adaptive = aomhip_highbd_quantize_b_64x64_adaptive;
  int k = 0;
#define AOMHIP_TX(W, H) t->fwd_txfm2d[k] = aomhip_fwd_txfm2d_##W##x##H; t->inv_txfm2d_add[k] = aomhip_inv_txfm2d_add_##W##x##H; ++k;
  AOMHIP_RTCD_TX_SIZES(AOMHIP_TX)
#undef AOMHIP_TX
#define AOMHIP_LPF(DIR, IDX, LEN, LI)                                                                             \
  t->lpf[IDX][LI] = aomhip_lpf_##DIR##_##LEN; t->lpf_dual[IDX][LI] = aomhip_lpf_##DIR##_##LEN##_dual;            \
  t->lpf_quad[IDX][LI] = aomhip_lpf_##DIR##_##LEN##_quad; t->highbd_lpf[IDX][LI] = aomhip_highbd_lpf_##DIR##_##LEN; \
  t->highbd_lpf_dual[IDX][LI] = aomhip_highbd_lpf_##DIR##_##LEN##_dual;
  AOMHIP_LPF(horizontal, 0, 4, 0) AOMHIP_LPF(horizontal, 0, 6, 1) AOMHIP_LPF(horizontal, 0, 8, 2) AOMHIP_LPF(horizontal, 0, 14, 3)
  AOMHIP_LPF(vertical, 1, 4, 0) AOMHIP_LPF(vertical, 1, 6, 1) AOMHIP_LPF(vertical, 1, 8, 2) AOMHIP_LPF(vertical, 1, 14, 3)
#undef AOMHIP_LPF
  t->cdef_find_dir = aomhip_cdef_find_dir; t->cdef_find_dir_dual = aomhip_cdef_find_dir_dual;
  t->cdef_filter_8[0] = aomhip_cdef_filter_8_0; t->cdef_filter_8[1] = aomhip_cdef_filter_8_1; t->cdef_filter_8[2] = aomhip_cdef_filter_8_2;
  t->cdef_filter_8[3] = aomhip_cdef_filter_8_3; t->cdef_filter_16[0] = aomhip_cdef_filter_16_0; t->cdef_filter_16[1] = aomhip_cdef_filter_16_1;
  t->cdef_filter_16[2] = aomhip_cdef_filter_16_2; t->cdef_filter_16[3] = aomhip_cdef_filter_16_3;
  return AOMHIP_OK;
}

}  // extern "C"
