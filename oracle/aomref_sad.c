/*
 * oracle/aomref_sad.c -- SAD / variance / sub-pixel variance / subtract.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Restates aom_dsp/sad.c,
 * aom_dsp/variance.c, aom_dsp/subtract.c of the reference.
 */
#include "aomref.h"

#include <stdlib.h>
#include <string.h>

/* aom_ports/mem.h:45 ROUND_POWER_OF_TWO -- (v + half) >> n, arithmetic shift for signed v */
#define RPOT(v, n) (((v) + ((1 << (n)) >> 1)) >> (n))

/* aom_dsp/aom_filter.h:43-50 bilinear_filters_2t, FILTER_BITS = 7 */
static const uint8_t k_bilin[8][2] = { { 128, 0 }, { 112, 16 }, { 96, 32 }, { 80, 48 },
                                       { 64, 64 }, { 48, 80 },  { 32, 96 }, { 16, 112 } };

/* ------------------------------------------------------------------ SAD, 8-bit */

unsigned orc_sad(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int w, int h) {
  unsigned acc = 0;
  for (int r = 0; r < h; ++r, src += src_stride, ref += ref_stride)
    for (int c = 0; c < w; ++c) acc += (unsigned)abs((int)src[c] - (int)ref[c]);
  return acc;
}

unsigned orc_sad_skip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int w, int h) {
  return 2u * orc_sad(src, 2 * src_stride, ref, 2 * ref_stride, w, h / 2);
}

void orc_sad_x4d(const uint8_t *src, int src_stride, const uint8_t *const ref[4], int ref_stride, int w, int h,
                 uint32_t out[4]) {
  for (int k = 0; k < 4; ++k) out[k] = orc_sad(src, src_stride, ref[k], ref_stride, w, h);
}

void orc_sad_skip_x4d(const uint8_t *src, int src_stride, const uint8_t *const ref[4], int ref_stride, int w,
                      int h, uint32_t out[4]) {
  for (int k = 0; k < 4; ++k) out[k] = orc_sad_skip(src, src_stride, ref[k], ref_stride, w, h);
}

unsigned orc_sad_avg(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride,
                     const uint8_t *second_pred, int w, int h) {
  /* comp_pred = round((second_pred + ref) / 2), second_pred is w-contiguous */
  unsigned acc = 0;
  for (int r = 0; r < h; ++r) {
    for (int c = 0; c < w; ++c) {
      const int p = RPOT((int)second_pred[r * w + c] + (int)ref[r * ref_stride + c], 1);
      acc += (unsigned)abs((int)src[r * src_stride + c] - p);
    }
  }
  return acc;
}

/* Compound-average SAD, all flavours (sad.c:50-64 SADMXN avg / dist_wtd avg; highbd :282-297 with
 * aom_highbd_comp_avg_pred_c / aom_highbd_dist_wtd_comp_avg_pred_c, variance.c:731-766; 8-bit preds :306-339):
 * fwd_offset == bck_offset == 0 selects the plain rounded average, otherwise
 * comp = ROUND_POWER_OF_TWO(pred * bck_offset + ref * fwd_offset, DIST_PRECISION_BITS = 4).
 * `bd` applies the encoder's _bits10 / _bits12 wrappers (encoder_utils.h:210-262) as for orc_highbd_sad. */
unsigned orc_sad_avg_any(const void *src, int src_stride, const void *ref, int ref_stride, const void *second_pred,
                         int w, int h, int elem16, int bd, int fwd_offset, int bck_offset) {
  unsigned acc = 0;
  for (int r = 0; r < h; ++r) {
    for (int c = 0; c < w; ++c) {
      const int s = elem16 ? ((const uint16_t *)src)[r * src_stride + c] : ((const uint8_t *)src)[r * src_stride + c];
      const int f = elem16 ? ((const uint16_t *)ref)[r * ref_stride + c] : ((const uint8_t *)ref)[r * ref_stride + c];
      const int p = elem16 ? ((const uint16_t *)second_pred)[r * w + c] : ((const uint8_t *)second_pred)[r * w + c];
      int comp;
      if (fwd_offset == 0 && bck_offset == 0)
        comp = RPOT(p + f, 1);
      else
        comp = RPOT(p * bck_offset + f * fwd_offset, 4);
      if (!elem16) comp = (uint8_t)comp;
      acc += (unsigned)abs(s - comp);
    }
  }
  if (elem16) return bd == 10 ? acc >> 2 : bd == 12 ? acc >> 4 : acc;
  return acc;
}

/* ------------------------------------------------------------------ SAD, highbd */

static unsigned hbd_sad_raw(const uint16_t *src, int src_stride, const uint16_t *ref, int ref_stride, int w,
                            int h) {
  unsigned acc = 0;
  for (int r = 0; r < h; ++r, src += src_stride, ref += ref_stride)
    for (int c = 0; c < w; ++c) acc += (unsigned)abs((int)src[c] - (int)ref[c]);
  return acc;
}

static unsigned hbd_wrap(unsigned v, int bd) { /* encoder_utils.h:155-208 _bits8/_bits10/_bits12 */
  return bd == 10 ? v >> 2 : bd == 12 ? v >> 4 : v;
}

unsigned orc_highbd_sad(const uint16_t *src, int src_stride, const uint16_t *ref, int ref_stride, int w, int h,
                        int bd) {
  return hbd_wrap(hbd_sad_raw(src, src_stride, ref, ref_stride, w, h), bd);
}

unsigned orc_highbd_sad_skip(const uint16_t *src, int src_stride, const uint16_t *ref, int ref_stride, int w,
                             int h, int bd) {
  return hbd_wrap(2u * hbd_sad_raw(src, 2 * src_stride, ref, 2 * ref_stride, w, h / 2), bd);
}

void orc_highbd_sad_x4d(const uint16_t *src, int src_stride, const uint16_t *const ref[4], int ref_stride, int w,
                        int h, int bd, uint32_t out[4]) {
  for (int k = 0; k < 4; ++k) out[k] = orc_highbd_sad(src, src_stride, ref[k], ref_stride, w, h, bd);
}

/* ------------------------------------------------------------------ variance, 8-bit */

uint32_t orc_variance(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse,
                      int *sum) {
  int s = 0;
  uint32_t q = 0;
  for (int r = 0; r < h; ++r, a += a_stride, b += b_stride) {
    for (int c = 0; c < w; ++c) {
      const int d = (int)a[c] - (int)b[c];
      s += d;
      q += (uint32_t)(d * d);
    }
  }
  *sse = q;
  if (sum) *sum = s;
  return q - (uint32_t)(((int64_t)s * s) / (w * h));
}

uint32_t orc_sub_pixel_variance(const uint8_t *a, int a_stride, int xoff, int yoff, const uint8_t *b, int b_stride,
                                int w, int h, uint32_t *sse) {
  /* pass 1: horizontal 2-tap on h+1 rows -> uint16; pass 2: vertical 2-tap -> uint8 */
  uint16_t *mid = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(h + 1) * w);
  uint8_t *fin = (uint8_t *)malloc((size_t)h * w);
  const uint8_t *fx = k_bilin[xoff], *fy = k_bilin[yoff];
  for (int r = 0; r < h + 1; ++r)
    for (int c = 0; c < w; ++c)
      mid[r * w + c] = (uint16_t)RPOT((int)a[r * a_stride + c] * fx[0] + (int)a[r * a_stride + c + 1] * fx[1], 7);
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c)
      fin[r * w + c] = (uint8_t)RPOT((int)mid[r * w + c] * fy[0] + (int)mid[(r + 1) * w + c] * fy[1], 7);
  const uint32_t v = orc_variance(fin, w, b, b_stride, w, h, sse, NULL);
  free(mid);
  free(fin);
  return v;
}

/* ------------------------------------------------------------------ variance, highbd */

uint32_t orc_highbd_variance(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h,
                             int bd, uint32_t *sse, int *sum) {
  int64_t s64 = 0;
  uint64_t q64 = 0;
  for (int r = 0; r < h; ++r, a += a_stride, b += b_stride) {
    int32_t row = 0;
    for (int c = 0; c < w; ++c) {
      const int d = (int)a[c] - (int)b[c];
      row += d;
      q64 += (uint32_t)(d * d);
    }
    s64 += row;
  }
  int s;
  uint32_t q;
  if (bd == 10) {
    q = (uint32_t)((q64 + 8) >> 4);
    s = (int)((s64 + 2) >> 2); /* arithmetic shift of a possibly negative sum, mem.h:45 */
  } else if (bd == 12) {
    q = (uint32_t)((q64 + 128) >> 8);
    s = (int)((s64 + 8) >> 4);
  } else {
    q = (uint32_t)q64;
    s = (int)s64;
  }
  *sse = q;
  if (sum) *sum = s;
  if (bd == 8) return q - (uint32_t)(((int64_t)s * s) / (w * h));
  const int64_t var = (int64_t)q - (((int64_t)s * s) / (w * h));
  return var >= 0 ? (uint32_t)var : 0;
}

uint32_t orc_highbd_sub_pixel_variance(const uint16_t *a, int a_stride, int xoff, int yoff, const uint16_t *b,
                                       int b_stride, int w, int h, int bd, uint32_t *sse) {
  uint16_t *mid = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(h + 1) * w);
  uint16_t *fin = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)h * w);
  const uint8_t *fx = k_bilin[xoff], *fy = k_bilin[yoff];
  for (int r = 0; r < h + 1; ++r)
    for (int c = 0; c < w; ++c)
      mid[r * w + c] = (uint16_t)RPOT((int)a[r * a_stride + c] * fx[0] + (int)a[r * a_stride + c + 1] * fx[1], 7);
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c)
      fin[r * w + c] = (uint16_t)RPOT((int)mid[r * w + c] * fy[0] + (int)mid[(r + 1) * w + c] * fy[1], 7);
  const uint32_t v = orc_highbd_variance(fin, w, b, b_stride, w, h, bd, sse, NULL);
  free(mid);
  free(fin);
  return v;
}

/* ------------------------------------------------------------------ subtract */

void orc_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint8_t *src,
                        ptrdiff_t src_stride, const uint8_t *pred, ptrdiff_t pred_stride) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c)
      diff[r * diff_stride + c] = (int16_t)((int)src[r * src_stride + c] - (int)pred[r * pred_stride + c]);
}

void orc_highbd_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint16_t *src,
                               ptrdiff_t src_stride, const uint16_t *pred, ptrdiff_t pred_stride) {
  for (int r = 0; r < rows; ++r)
    for (int c = 0; c < cols; ++c)
      diff[r * diff_stride + c] = (int16_t)((int)src[r * src_stride + c] - (int)pred[r * pred_stride + c]);
}
