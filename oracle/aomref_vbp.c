/* TEST INFRASTRUCTURE ONLY (see aomref.h): CPU restatement of the leaves of the variance-based partitioning's tree --
 * fill_variance_8x8avg / compute_minmax_8x8 / fill_variance_4x4avg (av1/encoder/var_based_part.c:255-430) on aom_avg_8x8 / aom_avg_4x4 /
 * aom_minmax_8x8 and their high-bit-depth forms (aom_dsp/avg.c:18-100).  Pinned by tests/golden/ref_eval_vbp.npz (the three functions
 * interpreted). */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "aomref.h"

static int px(const void *p, int hbd, ptrdiff_t i) { return hbd ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i]; }
static int avg_nxn(const void *s, int hbd, int stride, int x, int y, int n) {
  int sum = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) sum += px(s, hbd, (ptrdiff_t)(y + i) * stride + x + j);
  return n == 8 ? (sum + 32) >> 6 : (sum + 8) >> 4;
}

/* src / dst: the superblock's first pixel; x16 / y16, pixels_wide / pixels_high relative to it.  sum[k], sse[k]: what fill_variance stores for the
 * 16 x 16 block's k-th 8 x 8 (0 outside) */
void orc_vbp_fill_8x8avg(const void *src, int src_stride, const void *dst, int dst_stride, int x16, int y16, int hbd, int pixels_wide, int pixels_high,
                         int32_t *sum, uint32_t *sse) {
  for (int k = 0; k < 4; ++k) {
    const int x8 = x16 + ((k & 1) << 3), y8 = y16 + ((k >> 1) << 3);
    sum[k] = 0;
    sse[k] = 0;
    if (x8 < pixels_wide && y8 < pixels_high) {
      sum[k] = avg_nxn(src, hbd, src_stride, x8, y8, 8) - avg_nxn(dst, hbd, dst_stride, x8, y8, 8);
      sse[k] = (uint32_t)(sum[k] * sum[k]);
    }
  }
}
int orc_vbp_minmax_8x8(const void *src, int src_stride, const void *dst, int dst_stride, int x16, int y16, int hbd, int pixels_wide, int pixels_high) {
  int minmax_max = 0, minmax_min = 255;
  for (int k = 0; k < 4; ++k) {
    const int x8 = x16 + ((k & 1) << 3), y8 = y16 + ((k >> 1) << 3);
    if (x8 < pixels_wide && y8 < pixels_high) {
      int mn = 255, mx = 0;
      for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
          const int d = abs(px(src, hbd, (ptrdiff_t)(y8 + i) * src_stride + x8 + j) - px(dst, hbd, (ptrdiff_t)(y8 + i) * dst_stride + x8 + j));
          mn = d < mn ? d : mn;
          mx = d > mx ? d : mx;
        }
      if (mx - mn > minmax_max) minmax_max = mx - mn;
      if (mx - mn < minmax_min) minmax_min = mx - mn;
    }
  }
  return minmax_max - minmax_min;
}
void orc_vbp_fill_4x4avg(const void *src, int src_stride, int x8, int y8, int hbd, int pixels_wide, int pixels_high, int border_offset_4x4, int32_t *sum,
                         uint32_t *sse) {
  for (int k = 0; k < 4; ++k) {
    const int x4 = x8 + ((k & 1) << 2), y4 = y8 + ((k >> 1) << 2);
    sum[k] = 0;
    sse[k] = 0;
    if (x4 < pixels_wide - border_offset_4x4 && y4 < pixels_high - border_offset_4x4) {
      sum[k] = avg_nxn(src, hbd, src_stride, x4, y4, 4) - 128;
      sse[k] = (uint32_t)(sum[k] * sum[k]);
    }
  }
}
