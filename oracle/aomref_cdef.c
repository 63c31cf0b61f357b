/*
 * oracle/aomref_cdef.c -- CDEF direction search, constrained directional filter, luma plane driver.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Restates av1/common/cdef_block.c:57-426, the constrain()
 * of av1/common/cdef.h:59-67 and the per-64x64 logic of av1/common/cdef.c:138-345.
 *
 * PINNED by interpreting the reference's cdef_find_dir_c, cdef_filter_{8,16}_{0..3}_c and av1_cdef_filter_fb
 * (luma + the four chroma subsamplings, whole 64x64 filter blocks) on seeded inputs: tests/golden/ref_eval_cdef.npz,
 * ref_eval_cdef_fb.npz, checked bit for bit in tests/test_golden_ref_eval.py.  (The reference's own CDEF gtests are
 * SIMD-vs-C only, test/cdef_test.cc:408-436.)  tests/test_oracle_cdef.py adds definitional properties.
 */
#include "aomref.h"

#include <stdlib.h>
#include <string.h>

#define VERY_LARGE 0x4000 /* cdef_block.h CDEF_VERY_LARGE */

static int msb(unsigned v) { /* aom_ports/bitops.h get_msb, v != 0 */
  int n = 0;
  while (v >>= 1) ++n;
  return n;
}

int orc_cdef_find_dir(const uint16_t *img, int stride, int32_t *var, int coeff_shift) {
  /* line index of pixel (i, j) for each of the 8 directions (cdef_block.c:71-87) and the number of
   * pixels on each line -> weight 840 / n */
  static const int div_table[9] = { 0, 840, 420, 280, 210, 168, 140, 120, 105 };
  int partial[8][15];
  int32_t cost[8] = { 0 };
  memset(partial, 0, sizeof(partial));
  for (int i = 0; i < 8; ++i) {
    for (int j = 0; j < 8; ++j) {
      const int x = (img[i * stride + j] >> coeff_shift) - 128;
      const int line[8] = { i + j, i + j / 2, i, 3 + i - j / 2, 7 + i - j, 3 - i / 2 + j, j, i / 2 + j };
      for (int d = 0; d < 8; ++d) partial[d][line[d]] += x;
    }
  }
  for (int d = 0; d < 8; ++d) {
    if (d == 2 || d == 6) { /* 8 lines of 8 */
      for (int k = 0; k < 8; ++k) cost[d] += partial[d][k] * partial[d][k];
      cost[d] *= div_table[8];
    } else if (d == 0 || d == 4) { /* diagonals: lines of 1..8..1 pixels */
      for (int k = 0; k < 7; ++k)
        cost[d] += (partial[d][k] * partial[d][k] + partial[d][14 - k] * partial[d][14 - k]) * div_table[k + 1];
      cost[d] += partial[d][7] * partial[d][7] * div_table[8];
    } else { /* odd directions: 5 full lines, then lines of 2, 4, 6 pixels on both ends */
      for (int k = 0; k < 5; ++k) cost[d] += partial[d][3 + k] * partial[d][3 + k];
      cost[d] *= div_table[8];
      for (int k = 0; k < 3; ++k)
        cost[d] += (partial[d][k] * partial[d][k] + partial[d][10 - k] * partial[d][10 - k]) * div_table[2 * k + 2];
    }
  }
  int best = 0;
  int32_t best_cost = 0;
  for (int d = 0; d < 8; ++d)
    if (cost[d] > best_cost) {
      best_cost = cost[d];
      best = d;
    }
  *var = (best_cost - cost[(best + 4) & 7]) >> 10;
  return best;
}

static int constrain(int diff, int threshold, int damping) { /* cdef.h:59-67 */
  if (!threshold) return 0;
  int shift = damping - msb((unsigned)threshold);
  if (shift < 0) shift = 0;
  const int a = abs(diff);
  int m = threshold - (a >> shift);
  if (m < 0) m = 0;
  if (m > a) m = a;
  return diff < 0 ? -m : m;
}

/* Cdef_Directions (AV1 spec 7.15.3; cdef_block.c:25-48): (dy, dx) of taps k = 0, 1 for direction d */
static const int8_t k_dir[8][2][2] = { { { -1, 1 }, { -2, 2 } }, { { 0, 1 }, { -1, 2 } }, { { 0, 1 }, { 0, 2 } },
                                       { { 0, 1 }, { 1, 2 } },   { { 1, 1 }, { 2, 2 } },  { { 1, 0 }, { 2, 1 } },
                                       { { 1, 0 }, { 2, 0 } },   { { 1, 0 }, { 2, -1 } } };

void orc_cdef_filter_block(uint8_t *dst8, uint16_t *dst16, int dstride, const uint16_t *in, int pri_strength,
                           int sec_strength, int dir, int pri_damping, int sec_damping, int coeff_shift,
                           int block_w, int block_h, int enable_primary, int enable_secondary) {
  /* `in` points into a buffer of stride 144 (CDEF_BSTRIDE, cdef_block.h:26-27) */
  const int s = 144;
  static const int pri_taps[2][2] = { { 4, 2 }, { 3, 3 } };
  static const int sec_taps[2] = { 2, 1 };
  const int *pt = pri_taps[(pri_strength >> coeff_shift) & 1];
  const int clip = enable_primary && enable_secondary;
  for (int i = 0; i < block_h; ++i) {
    for (int j = 0; j < block_w; ++j) {
      const int x = in[i * s + j];
      int sum = 0, mx = x, mn = x;
      for (int k = 0; k < 2; ++k) {
        if (enable_primary) {
          const int o = k_dir[dir][k][0] * s + k_dir[dir][k][1];
          const int p[2] = { in[i * s + j + o], in[i * s + j - o] };
          for (int t = 0; t < 2; ++t) {
            sum += pt[k] * constrain(p[t] - x, pri_strength, pri_damping);
            if (clip) {
              if (p[t] != VERY_LARGE && p[t] > mx) mx = p[t];
              if (p[t] < mn) mn = p[t];
            }
          }
        }
        if (enable_secondary) {
          const int o1 = k_dir[(dir + 2) & 7][k][0] * s + k_dir[(dir + 2) & 7][k][1];
          const int o2 = k_dir[(dir + 6) & 7][k][0] * s + k_dir[(dir + 6) & 7][k][1];
          const int q[4] = { in[i * s + j + o1], in[i * s + j - o1], in[i * s + j + o2], in[i * s + j - o2] };
          for (int t = 0; t < 4; ++t) {
            if (clip) {
              if (q[t] != VERY_LARGE && q[t] > mx) mx = q[t];
              if (q[t] < mn) mn = q[t];
            }
            sum += sec_taps[k] * constrain(q[t] - x, sec_strength, sec_damping);
          }
        }
      }
      int y = (int16_t)x + ((8 + (int16_t)sum - ((int16_t)sum < 0)) >> 4);
      if (clip) y = y < mn ? mn : (y > mx ? mx : y);
      if (dst8)
        dst8[i * dstride + j] = (uint8_t)y;
      else
        dst16[i * dstride + j] = (uint16_t)y;
    }
  }
}

static int adjust_strength(int strength, int32_t var) { /* cdef_block.c:289-293 */
  const int i = (var >> 6) ? (msb((unsigned)(var >> 6)) < 12 ? msb((unsigned)(var >> 6)) : 12) : 0;
  return var ? (strength * (4 + i) + 8) >> 4 : 0;
}

/* Luma plane driver: av1_cdef_frame -> cdef_fb_col -> cdef_prepare_fb -> av1_cdef_filter_fb for pli == 0.
 * src: deblocked plane (uint8 or uint16), dst: output plane; both `stride` elements per row.
 * fb_pri / fb_sec: per 64x64 filter block the primary level and the secondary strength AFTER the
 * "3 -> 4" rule (cdef.c:309-313); a block with pri == sec == 0 and every skipped 8x8 is copied.
 * skip: one byte per 8x8 block (row-major, (width/8) per row), non-zero = all four 4x4 are skip_txfm.
 * dir_out / var_out (optional): per-8x8 direction and variance of every non-skipped block, for the chroma planes. */
void orc_cdef_plane_luma(const void *src, void *dst, int stride, int width, int height, int elem16, int bd,
                         const uint8_t *fb_pri, const uint8_t *fb_sec, int fb_stride, const uint8_t *skip, int damping,
                         uint8_t *dir_out, int32_t *var_out) {
  const int coeff_shift = bd - 8;
  const int b8w = width / 8, b8h = height / 8;
  const int pw = width + 16, ph = height + 4; /* frame with CDEF_HBORDER / CDEF_VBORDER of VERY_LARGE */
  uint16_t *pad = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)pw * ph);
  for (int i = 0; i < pw * ph; ++i) pad[i] = VERY_LARGE;
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
      pad[(y + 2) * pw + x + 8] = elem16 ? ((const uint16_t *)src)[(size_t)y * stride + x]
                                         : ((const uint8_t *)src)[(size_t)y * stride + x];
  /* start from a copy: unfiltered pixels keep their value */
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      if (elem16)
        ((uint16_t *)dst)[(size_t)y * stride + x] = pad[(y + 2) * pw + x + 8];
      else
        ((uint8_t *)dst)[(size_t)y * stride + x] = (uint8_t)pad[(y + 2) * pw + x + 8];
    }
  uint16_t in[144 * (8 + 4)];
  for (int by = 0; by < b8h; ++by) {
    for (int bx = 0; bx < b8w; ++bx) {
      const int fb = (by / 8) * fb_stride + bx / 8;
      const int level = fb_pri[fb], sec = fb_sec[fb];
      if (dir_out) dir_out[by * b8w + bx] = 0;
      if (var_out) var_out[by * b8w + bx] = 0;
      if (skip[by * b8w + bx]) continue;
      /* A filter block whose luma strengths are zero is still searched for directions when the chroma
       * strengths are not (cdef.c:334-345 "do not skip ... luma ... direction is computed based on luma"); the
       * side outputs therefore cover every non-skipped block whenever they are requested. */
      if (level == 0 && sec == 0 && !dir_out && !var_out) continue;
      /* local CDEF_BSTRIDE buffer: rows -2..9, cols -8..(8+8) around the block */
      for (int r = -2; r < 10; ++r)
        for (int c = -8; c < 16; ++c) in[(r + 2) * 144 + c + 8] = pad[(by * 8 + r + 2) * pw + bx * 8 + c + 8];
      const uint16_t *blk = in + 2 * 144 + 8;
      int32_t var;
      const int dir = orc_cdef_find_dir(blk, 144, &var, coeff_shift);
      if (dir_out) dir_out[by * b8w + bx] = (uint8_t)dir;
      if (var_out) var_out[by * b8w + bx] = var;
      const int pri_strength = level << coeff_shift, sec_strength = sec << coeff_shift;
      const int t = adjust_strength(pri_strength, var);
      const int dmp = damping + coeff_shift;
      if (elem16)
        orc_cdef_filter_block(NULL, (uint16_t *)dst + (size_t)by * 8 * stride + bx * 8, stride, blk, t, sec_strength,
                              pri_strength ? dir : 0, dmp, dmp, coeff_shift, 8, 8, t != 0, sec_strength != 0);
      else
        orc_cdef_filter_block((uint8_t *)dst + (size_t)by * 8 * stride + bx * 8, NULL, stride, blk, t, sec_strength,
                              pri_strength ? dir : 0, dmp, dmp, coeff_shift, 8, 8, t != 0, sec_strength != 0);
    }
  }
  free(pad);
}

/* Chroma plane driver: av1_cdef_filter_fb for pli > 0 (cdef_block.c:323-426): block (8 >> xdec) x (8 >> ydec) per
 * luma 8x8, direction taken from luma (converted by conv422 / conv440 when xdec != ydec, :362-371), no
 * adjust_strength, damping - 1 (:333), CDEF_VERY_LARGE outside the chroma plane (cdef.c:138-245).
 * width / height: chroma plane size; dir: luma directions [height_blocks][width_blocks]; fb_pri / fb_sec: the uv
 * level and secondary strength per 64x64 luma filter block (after the "3 -> 4" rule, cdef.c:318-322). */
void orc_cdef_plane_chroma(const void *src, void *dst, int stride, int width, int height, int elem16, int bd, int xdec,
                           int ydec, const uint8_t *dir, const uint8_t *fb_pri, const uint8_t *fb_sec, int fb_stride,
                           const uint8_t *skip, int damping) {
  static const int conv422[8] = { 7, 0, 2, 4, 5, 6, 6, 6 };
  static const int conv440[8] = { 1, 2, 2, 2, 3, 4, 6, 0 };
  const int coeff_shift = bd - 8;
  const int bw = 8 >> xdec, bh = 8 >> ydec;
  const int nbx = width / bw, nby = height / bh;
  const int pw = width + 16, ph = height + 4;
  uint16_t *pad = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)pw * ph);
  for (int i = 0; i < pw * ph; ++i) pad[i] = VERY_LARGE;
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const int v = elem16 ? ((const uint16_t *)src)[(size_t)y * stride + x] : ((const uint8_t *)src)[(size_t)y * stride + x];
      pad[(y + 2) * pw + x + 8] = (uint16_t)v;
      if (elem16)
        ((uint16_t *)dst)[(size_t)y * stride + x] = (uint16_t)v;
      else
        ((uint8_t *)dst)[(size_t)y * stride + x] = (uint8_t)v;
    }
  uint16_t in[144 * (8 + 4)];
  for (int by = 0; by < nby; ++by) {
    for (int bx = 0; bx < nbx; ++bx) {
      const int fb = (by / 8) * fb_stride + bx / 8;
      const int level = fb_pri[fb], sec = fb_sec[fb];
      if ((level == 0 && sec == 0) || skip[by * nbx + bx]) continue;
      for (int r = -2; r < bh + 2; ++r)
        for (int c = -8; c < bw + 8; ++c) in[(r + 2) * 144 + c + 8] = pad[(by * bh + r + 2) * pw + bx * bw + c + 8];
      const uint16_t *blk = in + 2 * 144 + 8;
      int d = dir[by * nbx + bx];
      if (xdec != ydec) d = (xdec ? conv422 : conv440)[d];
      const int pri_strength = level << coeff_shift, sec_strength = sec << coeff_shift;
      const int dmp = damping + coeff_shift - 1;
      if (elem16)
        orc_cdef_filter_block(NULL, (uint16_t *)dst + (size_t)by * bh * stride + bx * bw, stride, blk, pri_strength,
                              sec_strength, pri_strength ? d : 0, dmp, dmp, coeff_shift, bw, bh, pri_strength != 0,
                              sec_strength != 0);
      else
        orc_cdef_filter_block((uint8_t *)dst + (size_t)by * bh * stride + bx * bw, NULL, stride, blk, pri_strength,
                              sec_strength, pri_strength ? d : 0, dmp, dmp, coeff_shift, bw, bh, pri_strength != 0,
                              sec_strength != 0);
    }
  }
  free(pad);
}
