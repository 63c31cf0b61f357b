/* TEST INFRASTRUCTURE ONLY (see aomref.h): CPU restatement of the projection-based motion estimation of the real-time path,
 * av1_int_pro_motion_estimation (av1/encoder/mcomp.c:1897-2105) with aom_int_pro_row_c / aom_int_pro_col_c / aom_vector_var_c
 * (aom_dsp/avg.c:536-581): the block and a window of twice its size are projected onto a row and a column of sums, the two 1-D offsets are
 * searched coarse to fine on the projections' variance (vector_match), then the 2-D SAD decides between that vector, the zero vector, its four
 * neighbours and one diagonal.  Above 8 bits the reference only measures the zero vector.  Pinned by tests/golden/ref_eval_intpro.npz (the
 * function itself interpreted). */
#include <limits.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "aomref.h"

static void int_pro_row(int16_t *hbuf, const uint8_t *ref, int ref_stride, int width, int height, int norm_factor) {
  for (int idx = 0; idx < width; ++idx) {
    int16_t s = 0;
    for (int i = 0; i < height; ++i) s = (int16_t)(s + ref[(ptrdiff_t)i * ref_stride + idx]);
    hbuf[idx] = (int16_t)(s >> norm_factor);
  }
}
static void int_pro_col(int16_t *vbuf, const uint8_t *ref, int ref_stride, int width, int height, int norm_factor) {
  for (int ht = 0; ht < height; ++ht) {
    int16_t s = 0;
    for (int idx = 0; idx < width; ++idx) s = (int16_t)(s + ref[(ptrdiff_t)ht * ref_stride + idx]);
    vbuf[ht] = (int16_t)(s >> norm_factor);
  }
}
static int vector_var(const int16_t *ref, const int16_t *src, int bwl) {
  const int width = 4 << bwl;
  int sse = 0, mean = 0;
  for (int i = 0; i < width; ++i) {
    const int diff = ref[i] - src[i];
    mean += diff;
    sse += diff * diff;
  }
  const unsigned mean_abs = (unsigned)abs(mean);
  return (int)((unsigned)sse - ((mean_abs * mean_abs) >> (bwl + 2)));
}
static int vector_match(const int16_t *ref, const int16_t *src, int bwl) {
  int best = INT_MAX, offset = 0;
  const int bw = 4 << bwl;
  for (int d = 0; d <= bw; d += 16) {
    const int s = vector_var(&ref[d], src, bwl);
    if (s < best) { best = s; offset = d; }
  }
  int center = offset;
  for (int step = 8; step >= 1; step >>= 1) {   /* +- 8, 4, 2, 1 around the running centre */
    for (int d = -step; d <= step; d += 2 * step) {
      const int pos = offset + d;
      if (pos < 0 || pos > bw) continue;
      const int s = vector_var(&ref[pos], src, bwl);
      if (s < best) { best = s; center = pos; }
    }
    offset = center;
  }
  return center - (bw >> 1);
}

static int log2i(int v) { int n = 0; while ((1 << n) < v) ++n; return n; }

/* src / ref: the block's first pixel in its plane (uint8, or uint16 when bd > 8); limits: x->mv_limits {col_min, col_max, row_min, row_max} in full pels;
 * ref_mv in 1/8 pel; out_mv {row, col} in 1/8 pel (xd->mi[0]->mv[0] as the function leaves it); returns best_sad */
unsigned orc_int_pro_motion_estimation(const void *src, int src_stride, const void *ref, int ref_stride, int bw, int bh, int bd, const int *limits,
                                       const int16_t *ref_mv, int16_t *out_mv) {
  if (bd != 8) {
    out_mv[0] = out_mv[1] = 0;
    return orc_highbd_sad((const uint16_t *)src, src_stride, (const uint16_t *)ref, ref_stride, bw, bh, bd);
  }
  const uint8_t *s = (const uint8_t *)src, *r = (const uint8_t *)ref;
  int16_t hbuf[256], vbuf[256], src_hbuf[128], src_vbuf[128];
  const int row_norm = log2i(bh) - 2 + 1, col_norm = 3 + (bw >> 5);   /* mi_size_high_log2 + 1 */
  int_pro_row(hbuf, r - (bw >> 1), ref_stride, bw << 1, bh, row_norm);
  int_pro_col(vbuf, r - (ptrdiff_t)(bh >> 1) * ref_stride, ref_stride, bw, bh << 1, col_norm);
  int_pro_row(src_hbuf, s, src_stride, bw, bh, row_norm);
  int_pro_col(src_vbuf, s, src_stride, bw, bh, col_norm);
  int col = vector_match(hbuf, src_hbuf, log2i(bw) - 2), row = vector_match(vbuf, src_vbuf, log2i(bh) - 2);
  int trow = row, tcol = col;   /* this_mv */
  unsigned best_sad = orc_sad(s, src_stride, r + (ptrdiff_t)trow * ref_stride + tcol, ref_stride, bw, bh);
  if (row != 0 || col != 0) {
    const unsigned t = orc_sad(s, src_stride, r, ref_stride, bw, bh);
    if (t < best_sad) { row = col = trow = tcol = 0; best_sad = t; }
  }
  const uint8_t *rb = r + (ptrdiff_t)trow * ref_stride + tcol;
  static const int pos[4][2] = { { -1, 0 }, { 0, -1 }, { 0, 1 }, { 1, 0 } };
  unsigned this_sad[4];
  for (int i = 0; i < 4; ++i) this_sad[i] = orc_sad(s, src_stride, rb + (ptrdiff_t)pos[i][0] * ref_stride + pos[i][1], ref_stride, bw, bh);
  for (int i = 0; i < 4; ++i)
    if (this_sad[i] < best_sad) { best_sad = this_sad[i]; row = pos[i][0] + trow; col = pos[i][1] + tcol; }
  trow += this_sad[0] < this_sad[3] ? -1 : 1;
  tcol += this_sad[1] < this_sad[2] ? -1 : 1;
  const unsigned t = orc_sad(s, src_stride, r + (ptrdiff_t)trow * ref_stride + tcol, ref_stride, bw, bh);
  if (best_sad > t) { row = trow; col = tcol; best_sad = t; }
  /* convert_fullmv_to_mv, then clamp_mv to av1_set_subpel_mv_search_range(x->mv_limits, ref_mv) (av1/encoder/mcomp.h:344-361) */
  int mvr = row * 8, mvc = col * 8;
  const int max_mv = 1023 * 8, lo = -(1 << 14) + 1, hi = (1 << 14) - 1;
  int minc = limits[0] * 8 > ref_mv[1] - max_mv ? limits[0] * 8 : ref_mv[1] - max_mv;
  int maxc = limits[1] * 8 < ref_mv[1] + max_mv ? limits[1] * 8 : ref_mv[1] + max_mv;
  int minr = limits[2] * 8 > ref_mv[0] - max_mv ? limits[2] * 8 : ref_mv[0] - max_mv;
  int maxr = limits[3] * 8 < ref_mv[0] + max_mv ? limits[3] * 8 : ref_mv[0] + max_mv;
  minc = minc > lo ? minc : lo; maxc = maxc < hi ? maxc : hi; minr = minr > lo ? minr : lo; maxr = maxr < hi ? maxr : hi;
  mvc = mvc < minc ? minc : (mvc > maxc ? maxc : mvc);
  mvr = mvr < minr ? minr : (mvr > maxr ? maxr : mvr);
  out_mv[0] = (int16_t)mvr; out_mv[1] = (int16_t)mvc;
  return best_sad;
}
