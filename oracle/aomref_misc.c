/*
 * oracle/aomref_misc.c -- the small members of the files BASELINE.json's north_star names that no other oracle module restates.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_leftovers.npz (the reference's own functions interpreted where they
 * lie, tests/golden/gen_ref_eval_leftovers.py).
 *
 *   orc_get_mb_ss                 aom_get_mb_ss_c                   aom_dsp/variance.c:46-54
 *   orc_mse_wxh_16bit[_highbd]    aom_mse_wxh_16bit[_highbd]_c      aom_dsp/variance.c:1258-1268,1287-1297
 *   orc_mse_16xh_16bit            aom_mse_16xh_16bit_c              aom_dsp/variance.c:1270-1284
 *   orc_comp_mask_pred            aom_[highbd_]comp_mask_pred_c     aom_dsp/variance.c:773-791,841-862 (AOM_BLEND_A64, aom_dsp/blend.h:24-33)
 *   orc_return_extreme_sub_pixel_mv   av1_return_max / _min_sub_pixel_mv   av1/encoder/mcomp.c:3139-3190 (lower_mv_precision, av1/common/mvref_common.h:88-97)
 */
#include "aomref.h"

uint32_t orc_get_mb_ss(const int16_t *a) {
  uint32_t sum = 0;   /* (the reference's `unsigned int sum` takes int products: modulo 2^32) */
  for (int i = 0; i < 256; ++i) sum += (uint32_t)(a[i] * a[i]);
  return sum;
}

uint64_t orc_mse_wxh_16bit(const void *dst, int dstride, int dst16, const uint16_t *src, int sstride, int w, int h) {
  uint64_t sum = 0;
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j) {
      const int d = dst16 ? ((const uint16_t *)dst)[i * dstride + j] : ((const uint8_t *)dst)[i * dstride + j];
      const int e = d - src[i * sstride + j];
      sum += (uint64_t)((int64_t)e * e);   /* (the reference's int product `e * e` is defined for |e| < 46341 -- its operands are pixel values, 16
                                            * bits at most on one side and 8 .. 12 on the other in pickcdef.c --; the same values there) */
    }
  return sum;
}

/* 16 / w blocks of w x h side by side in `dst`, one after the other (w * h entries each) in `src` */
uint64_t orc_mse_16xh_16bit(const uint8_t *dst, int dstride, const uint16_t *src, int w, int h) {
  int64_t sum = 0;
  for (int i = 0; i < 16 / w; ++i) sum += (int64_t)orc_mse_wxh_16bit(dst + i * w, dstride, 0, src + i * (w * h), w, w, h);
  return (uint64_t)sum;
}

/* comp_pred (width x height, pitch width) = AOM_BLEND_A64(mask, invert_mask ? pred : ref, invert_mask ? ref : pred) */
void orc_comp_mask_pred(void *comp_pred, const void *pred, int width, int height, const void *ref, int ref_stride, const uint8_t *mask, int mask_stride,
                        int invert_mask, int elem16) {
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j) {
      const int p = elem16 ? ((const uint16_t *)pred)[i * width + j] : ((const uint8_t *)pred)[i * width + j];
      const int r = elem16 ? ((const uint16_t *)ref)[i * ref_stride + j] : ((const uint8_t *)ref)[i * ref_stride + j];
      const int m = mask[i * mask_stride + j];
      const int a = invert_mask ? p : r, b = invert_mask ? r : p;
      const int v = (m * a + (64 - m) * b + 32) >> 6;
      if (elem16) ((uint16_t *)comp_pred)[i * width + j] = (uint16_t)v; else ((uint8_t *)comp_pred)[i * width + j] = (uint8_t)v;
    }
}

/* limits = SubpelMvLimits {col_min, col_max, row_min, row_max}; want_max 1: av1_return_max_sub_pixel_mv, 0: _min_.  Returns besterr (0). */
int orc_return_extreme_sub_pixel_mv(const int *limits, int allow_hp, int want_max, int16_t *bestmv /* row, col */) {
  int row = want_max ? limits[3] : limits[2], col = want_max ? limits[1] : limits[0];
  if (!allow_hp) {
    if (row & 1) row += row > 0 ? -1 : 1;
    if (col & 1) col += col > 0 ? -1 : 1;
  }
  bestmv[0] = (int16_t)row;
  bestmv[1] = (int16_t)col;
  return 0;
}
