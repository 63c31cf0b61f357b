/*
 * oracle/aomref_lrstats.c -- the Wiener-filter statistics of the loop-restoration search: av1_compute_stats_c
 * (av1/encoder/pickrst.c:948-1025, find_average pickrst.h:32-42) and av1_compute_stats_highbd_c (:1027-1083); and the self-guided filter's
 * projection statistics av1_calc_proj_params[_high_bd] (:470-657) and av1_[lowbd|highbd]_pixel_proj_error (:226-370), pinned by
 * tests/golden/ref_eval_proj.npz.
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

#define PX(p, i) (elem16 ? (int)((const uint16_t *)(p))[i] : (int)((const uint8_t *)(p))[i])
/* av1_calc_proj_params_c / _high_bd_c (pickrst.c:470-657): with u = dat << SGRPROJ_RST_BITS, s = (src << 4) - u, f1 = flt0 - u, f2 = flt1 - u:
 * H = { sum f1 f1, sum f1 f2; .., sum f2 f2 } / size, C = { sum f1 s, sum f2 s } / size (C's integer division: towards zero); only the entries of
 * the radii in use are computed, the others stay 0 (get_proj_subspace zeroes them). */
void orc_calc_proj_params(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, const int32_t *flt0, int flt0_stride,
                          const int32_t *flt1, int flt1_stride, int elem16, int r0, int r1, int64_t H[4], int64_t C[2]) {
  int64_t h00 = 0, h01 = 0, h11 = 0, c0 = 0, c1 = 0;
  const int64_t size = (int64_t)width * height;
  H[0] = H[1] = H[2] = H[3] = 0; C[0] = C[1] = 0;
  if (r0 <= 0 && r1 <= 0) return;
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j) {
      const int32_t u = (int32_t)(PX(dat, (ptrdiff_t)i * dat_stride + j) << 4);
      const int32_t sv = (int32_t)(PX(src, (ptrdiff_t)i * src_stride + j) << 4) - u;
      const int32_t f1 = r0 > 0 ? flt0[(ptrdiff_t)i * flt0_stride + j] - u : 0, f2 = r1 > 0 ? flt1[(ptrdiff_t)i * flt1_stride + j] - u : 0;
      h00 += (int64_t)f1 * f1; h11 += (int64_t)f2 * f2; h01 += (int64_t)f1 * f2;
      c0 += (int64_t)f1 * sv; c1 += (int64_t)f2 * sv;
    }
  if (r0 > 0) { H[0] = h00 / size; C[0] = c0 / size; }
  if (r1 > 0) { H[3] = h11 / size; C[1] = c1 / size; }
  if (r0 > 0 && r1 > 0) H[1] = H[2] = h01 / size;
}

/* av1_lowbd_pixel_proj_error_c / av1_highbd_pixel_proj_error_c (pickrst.c:226-370): sum of e^2 with
 * e = ((xq0 (flt0 - u) + xq1 (flt1 - u) + half) >> (SGRPROJ_RST_BITS + SGRPROJ_PRJ_BITS)) + dat - src over the radii in use (the two functions
 * write the same value differently: (u << 7) inside the rounded sum IS dat << 11); neither radius: the plain SSE */
int64_t orc_pixel_proj_error(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, const int32_t *flt0, int flt0_stride,
                             const int32_t *flt1, int flt1_stride, int elem16, int r0, int r1, int xq0, int xq1) {
  int64_t err = 0;
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j) {
      const int32_t d = PX(dat, (ptrdiff_t)i * dat_stride + j), sv = PX(src, (ptrdiff_t)i * src_stride + j);
      const int32_t u = d << 4;
      int32_t v = 1 << 10;
      if (r0 > 0) v += xq0 * (flt0[(ptrdiff_t)i * flt0_stride + j] - u);
      if (r1 > 0) v += xq1 * (flt1[(ptrdiff_t)i * flt1_stride + j] - u);
      const int32_t e = (r0 > 0 || r1 > 0 ? (v >> 11) : 0) + d - sv;
      err += (int64_t)e * e;
    }
  return err;
}
#undef PX
