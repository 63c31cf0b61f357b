/*
 * oracle/aomref_lrstats.c -- the Wiener-filter statistics of the loop-restoration search: av1_compute_stats_c
 * (av1/encoder/pickrst.c:948-1025, find_average pickrst.h:32-42) and av1_compute_stats_highbd_c (:1027-1083).
 *
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_lrstats.npz (the reference's own functions,
 * interpreted where they lie).  M[k] = sum Y[k] * X and H[k][l] = sum Y[k] * Y[l] over the unit's pixels, with
 * Y = the wiener_win x wiener_win window of the degraded frame minus the unit's average (index = column offset major,
 * row offset minor) and X = the source pixel minus that average.  Exact integers, so one restatement with 64-bit sums
 * covers the reference's int32 per-row partial sums too (they cannot overflow for pixel data).
 */
#include <string.h>

#include "aomref.h"

void orc_compute_stats(int wiener_win, const void *dgd, const void *src, int h_start, int h_end, int v_start, int v_end, int dgd_stride,
                       int src_stride, int elem16, int bit_depth, int use_downsampled_wiener_stats, int64_t *M, int64_t *H) {
  const int win2 = wiener_win * wiener_win, half = wiener_win >> 1;
#define PX(p, i) (elem16 ? (int)((const uint16_t *)(p))[i] : (int)((const uint8_t *)(p))[i])
  uint64_t sum = 0;
  for (int i = v_start; i < v_end; ++i)
    for (int j = h_start; j < h_end; ++j) sum += (uint64_t)PX(dgd, (ptrdiff_t)i * dgd_stride + j);
  const int avg = (int)(sum / (uint64_t)((v_end - v_start) * (h_end - h_start))); /* find_average[_highbd] */
  memset(M, 0, sizeof(*M) * (size_t)win2);
  memset(H, 0, sizeof(*H) * (size_t)win2 * win2);
  /* the 8-bit function can visit every 4th row and weigh it by 4 (by what is left for the last one); highbd has no such mode */
  int step = (!elem16 && use_downsampled_wiener_stats) ? 4 : 1;
  for (int i = v_start; i < v_end; i += step) {
    if (step > 1 && v_end - i < 4) step = v_end - i;
    for (int j = h_start; j < h_end; ++j) {
      const int X = PX(src, (ptrdiff_t)i * src_stride + j) - avg;
      int Y[49], idx = 0;
      for (int k = -half; k <= half; ++k)
        for (int l = -half; l <= half; ++l) Y[idx++] = PX(dgd, (ptrdiff_t)(i + l) * dgd_stride + (j + k)) - avg;
      for (int k = 0; k < win2; ++k) {
        M[k] += (int64_t)Y[k] * X * step;
        for (int l = k; l < win2; ++l) H[k * win2 + l] += (int64_t)Y[k] * Y[l] * step;
      }
    }
  }
#undef PX
  const int div = !elem16 ? 1 : bit_depth == 12 ? 16 : bit_depth == 10 ? 4 : 1; /* bit_depth_divider (:1041-1045), C division */
  for (int k = 0; k < win2; ++k) {
    M[k] /= div;
    for (int l = k; l < win2; ++l) {
      H[k * win2 + l] /= div;
      H[l * win2 + k] = H[k * win2 + l];
    }
  }
}
