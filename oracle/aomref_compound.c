/*
 * oracle/aomref_compound.c -- the compound / masked / OBMC members of the encoder's kernel table
 * (aom_variance_fn_ptr_t: svaf, jsvaf, msdf, msvf, osdf, ovf, osvf; aom_dsp/variance.h:84-103).
 *
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Plain-C restatement with run-time (w, h); one pixel type switch
 * (`elem16`) instead of the reference's separate 8-bit / highbd symbol families.  Pinned by
 * tests/golden/ref_eval_compound.npz (the reference's own functions, interpreted where they lie).
 *
 *   svaf / jsvaf  aom_[highbd_N_][dist_wtd_]sub_pixel_avg_varianceWxH_c   aom_dsp/variance.c:165-200,563-690
 *   msvf          aom_[highbd_N_]masked_sub_pixel_varianceWxH_c           aom_dsp/variance.c:773-811,840-928
 *   msdf          aom_[highbd_]masked_sadWxH_c                            aom_dsp/sad_av1.c:20-52,92-126
 *   ovf / osvf    aom_[highbd_N_]obmc_[sub_pixel_]varianceWxH_c           aom_dsp/variance.c:957-1000,1064-1192
 *   osdf          aom_[highbd_]obmc_sadWxH_c                              aom_dsp/sad_av1.c:163-186,215-239
 * and the encoder's _bits10 / _bits12 SAD wrappers (av1/encoder/encoder_utils.h:363-387,527-542).
 */
#include <stdlib.h>
#include <string.h>

#include "aomref.h"

#define RPOT(v, n) (((v) + ((1 << (n)) >> 1)) >> (n))

static const uint8_t k_bilin2[8][2] = { { 128, 0 }, { 112, 16 }, { 96, 32 }, { 80, 48 },
                                        { 64, 64 }, { 48, 80 },  { 32, 96 }, { 16, 112 } };

static int px(const void *p, int elem16, ptrdiff_t i) { return elem16 ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i]; }

/* aom_var_filter_block2d_bil_first_pass_c + _second_pass_c (variance.c:91-139; highbd :475-520): h+1 rows of
 * horizontal 2-tap kept in uint16, then vertical 2-tap stored in the pixel type.  out: w * h, row stride w. */
static void bilinear_block(const void *a, int a_stride, int xoff, int yoff, int w, int h, int elem16, uint16_t *out) {
  uint16_t *mid = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(h + 1) * w);
  const uint8_t *fx = k_bilin2[xoff], *fy = k_bilin2[yoff];
  for (int r = 0; r < h + 1; ++r)
    for (int c = 0; c < w; ++c)
      mid[r * w + c] =
          (uint16_t)RPOT(px(a, elem16, (ptrdiff_t)r * a_stride + c) * fx[0] + px(a, elem16, (ptrdiff_t)r * a_stride + c + 1) * fx[1], 7);
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const int v = RPOT((int)mid[r * w + c] * fy[0] + (int)mid[(r + 1) * w + c] * fy[1], 7);
      out[r * w + c] = elem16 ? (uint16_t)v : (uint8_t)v;
    }
  free(mid);
}

/* variance of a w-stride uint16 block `t` against plane block b (the pixel type of the planes), with the final
 * forms of aom_varianceWxH_c (variance.c:141-148) / aom_highbd_{8,10,12}_varianceWxH_c (:383-420) */
static uint32_t block_variance(const uint16_t *t, const void *b, int b_stride, int w, int h, int elem16, int bd, uint32_t *sse) {
  int64_t s64 = 0;
  uint64_t q64 = 0;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const int d = (int)t[r * w + c] - px(b, elem16, (ptrdiff_t)r * b_stride + c);
      s64 += d;
      q64 += (uint32_t)(d * d);
    }
  int s;
  uint32_t q;
  if (elem16 && bd == 10) {
    q = (uint32_t)((q64 + 8) >> 4);
    s = (int)((s64 + 2) >> 2);
  } else if (elem16 && bd == 12) {
    q = (uint32_t)((q64 + 128) >> 8);
    s = (int)((s64 + 8) >> 4);
  } else {
    q = (uint32_t)q64;
    s = (int)s64;
  }
  *sse = q;
  if (!elem16 || bd == 8) return q - (uint32_t)(((int64_t)s * s) / (w * h));
  const int64_t v = (int64_t)q - (((int64_t)s * s) / (w * h));
  return v >= 0 ? (uint32_t)v : 0;
}

/* kind 0: aom_comp_avg_pred (variance.c:306-319); 1: aom_dist_wtd_comp_avg_pred (:321-339, weights fwd/bck);
 * 2: aom_comp_mask_pred (:773-791, AOM_BLEND_A64, aom_dsp/blend.h:24-33).  `t` (the filtered block) is blended in place. */
static void blend_block(uint16_t *t, const void *second_pred, int w, int h, int elem16, int kind, int fwd, int bck, const uint8_t *mask,
                        int mask_stride, int invert) {
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const int ref = t[r * w + c], pred = px(second_pred, elem16, (ptrdiff_t)r * w + c);
      int v;
      if (kind == 0) {
        v = RPOT(pred + ref, 1);
      } else if (kind == 1) {
        v = RPOT(pred * bck + ref * fwd, 4);
      } else {
        const int m = mask[r * mask_stride + c];
        v = invert ? RPOT(m * pred + (64 - m) * ref, 6) : RPOT(m * ref + (64 - m) * pred, 6);
      }
      t[r * w + c] = elem16 ? (uint16_t)v : (uint8_t)v;
    }
}

/* svaf / jsvaf / msvf: `a` (interpolated at xoff/8, yoff/8) blended with second_pred, variance against b. */
uint32_t orc_compound_sub_pixel_variance(const void *a, int a_stride, int xoff, int yoff, const void *b, int b_stride, int w, int h,
                                         int elem16, int bd, int kind, const void *second_pred, int fwd_offset, int bck_offset,
                                         const uint8_t *mask, int mask_stride, int invert_mask, uint32_t *sse) {
  uint16_t *t = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
  bilinear_block(a, a_stride, xoff, yoff, w, h, elem16, t);
  blend_block(t, second_pred, w, h, elem16, kind, fwd_offset, bck_offset, mask, mask_stride, invert_mask);
  const uint32_t v = block_variance(t, b, b_stride, w, h, elem16, bd, sse);
  free(t);
  return v;
}

/* msdf: masked_sad / highbd_masked_sad (sad_av1.c:20-52,92-126) + the _bits10 / _bits12 wrappers */
unsigned orc_masked_sad(const void *src, int src_stride, const void *ref, int ref_stride, const void *second_pred, const uint8_t *mask,
                        int mask_stride, int invert_mask, int w, int h, int elem16, int bd) {
  unsigned sad = 0;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) {
      const int m = mask[r * mask_stride + c];
      const int f = px(ref, elem16, (ptrdiff_t)r * ref_stride + c), p = px(second_pred, elem16, (ptrdiff_t)r * w + c);
      const int16_t pred = (int16_t)(invert_mask ? RPOT(m * p + (64 - m) * f, 6) : RPOT(m * f + (64 - m) * p, 6));
      sad += (unsigned)abs(pred - px(src, elem16, (ptrdiff_t)r * src_stride + c));
    }
  if (elem16) return bd == 10 ? sad >> 2 : bd == 12 ? sad >> 4 : sad;
  return sad;
}

/* osdf: obmc_sad / highbd_obmc_sad (sad_av1.c:163-180,215-232) + the _bits10 / _bits12 wrappers */
unsigned orc_obmc_sad(const void *pre, int pre_stride, const int32_t *wsrc, const int32_t *mask, int w, int h, int elem16, int bd) {
  unsigned sad = 0;
  for (int r = 0; r < h; ++r)
    for (int c = 0; c < w; ++c) sad += (unsigned)RPOT(abs(wsrc[r * w + c] - px(pre, elem16, (ptrdiff_t)r * pre_stride + c) * mask[r * w + c]), 12);
  if (elem16) return bd == 10 ? sad >> 2 : bd == 12 ? sad >> 4 : sad;
  return sad;
}

/* ovf / osvf: obmc_variance (variance.c:957-1000) and the highbd forms (:1064-1192); xoff = yoff = 0 with
 * `subpel` 0 is the plain form, `subpel` 1 runs the bilinear passes first (also for offset 0, as the reference does). */
uint32_t orc_obmc_variance(const void *pre, int pre_stride, int subpel, int xoff, int yoff, const int32_t *wsrc, const int32_t *mask, int w,
                           int h, int elem16, int bd, uint32_t *sse) {
  uint16_t *t = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
  if (subpel) {
    bilinear_block(pre, pre_stride, xoff, yoff, w, h, elem16, t);
  } else {
    for (int r = 0; r < h; ++r)
      for (int c = 0; c < w; ++c) t[r * w + c] = (uint16_t)px(pre, elem16, (ptrdiff_t)r * pre_stride + c);
  }
  int64_t s64 = 0;
  uint64_t q64 = 0;
  for (int i = 0; i < w * h; ++i) {
    const int v = wsrc[i] - (int)t[i] * mask[i];
    const int diff = v < 0 ? -RPOT(-v, 12) : RPOT(v, 12); /* ROUND_POWER_OF_TWO_SIGNED */
    s64 += diff;
    q64 += (uint32_t)(diff * diff);
  }
  free(t);
  int s;
  uint32_t q;
  if (elem16 && bd == 10) {
    q = (uint32_t)((q64 + 8) >> 4);
    s = (int)((s64 + 2) >> 2);
  } else if (elem16 && bd == 12) {
    q = (uint32_t)((q64 + 128) >> 8);
    s = (int)((s64 + 8) >> 4);
  } else {
    q = (uint32_t)q64;
    s = (int)s64;
  }
  *sse = q;
  if (!elem16 || bd == 8) return q - (uint32_t)(((int64_t)s * s) / (w * h));
  const int64_t var = (int64_t)q - (((int64_t)s * s) / (w * h));
  return var >= 0 ? (uint32_t)var : 0;
}
