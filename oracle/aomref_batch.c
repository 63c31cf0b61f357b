/*
 * oracle/aomref_batch.c -- whole-work-list drivers over the scalar oracle functions.
 * TEST INFRASTRUCTURE ONLY (see aomref.h): used as the checker for the batched HIP
 * entry points and as bench.py's `cpu_baseline` ("port", OpenMP over candidates,
 * mirroring the reference's static tile-thread partition, av1/encoder/ethread.c).
 */
#include <omp.h>

#include "aomref.h"

typedef struct { int16_t sx, sy, rx, ry; } orc_cand;            /* == aomhip_sad_cand */
typedef struct { int16_t sx, sy, rx[4], ry[4]; } orc_x4d_group;  /* == aomhip_sad_x4d_cand */

int orc_max_threads(void) { return omp_get_max_threads(); }

/* origin pointers address pixel (0,0) of a bordered plane; elem16 selects uint16 planes. */
void orc_sad_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16,
                   int bd, int w, int h, int skip, const orc_cand *c, int n, uint32_t *out, int threads, int reps) {
  if (threads < 1) threads = 1;
  if (reps < 1) reps = 1;
  /* static partition of the list over the threads; reps > 1 repeats each thread's own slice (no shared output lines) */
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < n; ++i)
   for (int rep = 0; rep < reps; ++rep) {
    if (!elem16) {
      const uint8_t *s = (const uint8_t *)src_origin + (ptrdiff_t)c[i].sy * src_stride + c[i].sx;
      const uint8_t *r = (const uint8_t *)ref_origin + (ptrdiff_t)c[i].ry * ref_stride + c[i].rx;
      out[i] = skip ? orc_sad_skip(s, src_stride, r, ref_stride, w, h) : orc_sad(s, src_stride, r, ref_stride, w, h);
    } else {
      const uint16_t *s = (const uint16_t *)src_origin + (ptrdiff_t)c[i].sy * src_stride + c[i].sx;
      const uint16_t *r = (const uint16_t *)ref_origin + (ptrdiff_t)c[i].ry * ref_stride + c[i].rx;
      out[i] = skip ? orc_highbd_sad_skip(s, src_stride, r, ref_stride, w, h, bd)
                    : orc_highbd_sad(s, src_stride, r, ref_stride, w, h, bd);
    }
  }
}

void orc_sad_x4d_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16,
                       int bd, int w, int h, int skip, const orc_x4d_group *g, int n, uint32_t *out, int threads, int reps) {
  if (threads < 1) threads = 1;
  if (reps < 1) reps = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < n; ++i)
   for (int rep = 0; rep < reps; ++rep) {
    for (int k = 0; k < 4; ++k) {
      if (!elem16) {
        const uint8_t *s = (const uint8_t *)src_origin + (ptrdiff_t)g[i].sy * src_stride + g[i].sx;
        const uint8_t *r = (const uint8_t *)ref_origin + (ptrdiff_t)g[i].ry[k] * ref_stride + g[i].rx[k];
        out[4 * i + k] =
            skip ? orc_sad_skip(s, src_stride, r, ref_stride, w, h) : orc_sad(s, src_stride, r, ref_stride, w, h);
      } else {
        const uint16_t *s = (const uint16_t *)src_origin + (ptrdiff_t)g[i].sy * src_stride + g[i].sx;
        const uint16_t *r = (const uint16_t *)ref_origin + (ptrdiff_t)g[i].ry[k] * ref_stride + g[i].rx[k];
        out[4 * i + k] = skip ? orc_highbd_sad_skip(s, src_stride, r, ref_stride, w, h, bd)
                              : orc_highbd_sad(s, src_stride, r, ref_stride, w, h, bd);
      }
    }
  }
}

/* ---- av1_xform_quant over a block list (aomhip_xform_quant_batch's checker / CPU baseline) ---- */
typedef struct { int32_t x, y; uint32_t out_offset; uint8_t tx_type; uint8_t reserved[3]; } orc_txb; /* == aomhip_txb */

void orc_quantize_fp(const int32_t *coeff, intptr_t n, const int16_t *round_fp, const int16_t *quant_fp, int32_t *qcoeff,
                     int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale,
                     int highbd);
static int g_quant_kind = 0; /* 0 quantize_b, 1 quantize_fp: set by orc_xform_quant_set_kind (tests only, single-threaded setup) */
void orc_xform_quant_set_kind(int kind) { g_quant_kind = kind; }

/* quantisation matrices of the following orc_xform_quant_batch calls (NULL, NULL: none): one (qm, iqm) pair for the batch's transform size */
static const uint8_t *g_qm, *g_iqm;
void orc_set_qm(const uint8_t *qm, const uint8_t *iqm) { g_qm = qm; g_iqm = iqm; }

void orc_xform_quant_batch(const int16_t *residual, int stride, int tx_size, const orc_txb *blocks, int n,
                           int grid_cols, int uniform_type, const int16_t q[5][2], int is_hbd, int32_t *coeff,
                           int32_t *qcoeff, int32_t *dqcoeff, uint16_t *eob, int threads, int reps) {
  const int w = orc_tx_wide[tx_size], h = orc_tx_high[tx_size];
  const int kw = w < 32 ? w : 32, kh = h < 32 ? h : 32, nc = kw * kh;
  const int log_scale = (w * h > 256) + (w * h > 1024); /* av1_get_tx_scale, av1/common/idct.c:24-28 */
  int16_t scans[17][1024], iscans[17][1024];
  uint8_t have[17] = { 0 };
  for (int i = 0; i < n; ++i) {
    const int tt = blocks ? blocks[i].tx_type : uniform_type;
    if (!have[tt]) { orc_get_scan(tx_size, tt == ORC_TX_WHT ? 0 : tt, scans[tt], iscans[tt]); have[tt] = 1; } /* lossless: DCT_DCT scan */
  }
  if (threads < 1) threads = 1;
  if (reps < 1) reps = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int i = 0; i < n; ++i)
   for (int rep = 0; rep < reps; ++rep) {
    int32_t full[64 * 64];
    const int bx = blocks ? blocks[i].x : (i % grid_cols) * w, by = blocks ? blocks[i].y : (i / grid_cols) * h;
    const int tt = blocks ? blocks[i].tx_type : uniform_type;
    const size_t off = blocks ? blocks[i].out_offset : (size_t)i * nc;
    if (tt == ORC_TX_WHT) orc_fwht4x4(residual + (ptrdiff_t)by * stride + bx, full, stride);
    else orc_fwd_txfm2d(residual + (ptrdiff_t)by * stride + bx, full, stride, tx_size, tt, is_hbd ? 10 : 8);
    if (coeff) for (int k = 0; k < nc; ++k) coeff[off + k] = full[k];
    if (g_qm && g_quant_kind != 1) {
      if (is_hbd) orc_highbd_quantize_b_qm(full, nc, q[0], q[1], q[2], q[3], qcoeff + off, dqcoeff + off, q[4], &eob[i], scans[tt], log_scale, g_qm, g_iqm);
      else orc_quantize_b_qm(full, nc, q[0], q[1], q[2], q[3], qcoeff + off, dqcoeff + off, q[4], &eob[i], scans[tt], log_scale, g_qm, g_iqm);
    } else if (g_quant_kind == 1)
      orc_quantize_fp(full, nc, q[1], q[2], qcoeff + off, dqcoeff + off, q[4], &eob[i], scans[tt], log_scale, is_hbd);
    else if (is_hbd)
      orc_highbd_quantize_b(full, nc, q[0], q[1], q[2], q[3], qcoeff + off, dqcoeff + off, q[4], &eob[i], scans[tt],
                            iscans[tt], log_scale);
    else
      orc_quantize_b(full, nc, q[0], q[1], q[2], q[3], qcoeff + off, dqcoeff + off, q[4], &eob[i], scans[tt],
                     iscans[tt], log_scale);
  }
}

/* ---- av1_inverse_transform_block over a block list: adds into a uint16 working copy of the plane ---- */
void orc_inv_txfm_add_batch(const int32_t *dqcoeff, int tx_size, const orc_txb *blocks, int n, int grid_cols,
                            int uniform_type, const uint16_t *eob, uint16_t *dst, int dst_stride, int bd) {
  const int w = orc_tx_wide[tx_size], h = orc_tx_high[tx_size];
  const int kw = w < 32 ? w : 32, kh = h < 32 ? h : 32, nc = kw * kh;
  for (int i = 0; i < n; ++i) {
    if (eob && eob[i] == 0) continue;
    const int bx = blocks ? blocks[i].x : (i % grid_cols) * w, by = blocks ? blocks[i].y : (i / grid_cols) * h;
    const int tt = blocks ? blocks[i].tx_type : uniform_type;
    const size_t off = blocks ? blocks[i].out_offset : (size_t)i * nc;
    if (tt == ORC_TX_WHT) orc_iwht4x4_add(dqcoeff + off, dst + (ptrdiff_t)by * dst_stride + bx, dst_stride, eob ? eob[i] : 16, bd);
    else orc_inv_txfm2d_add(dqcoeff + off, dst + (ptrdiff_t)by * dst_stride + bx, dst_stride, tx_size, tt, bd);
  }
}
