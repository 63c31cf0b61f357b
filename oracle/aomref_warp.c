/*
 * oracle/aomref_warp.c -- the warped-motion predictor of a single reference: av1_warp_affine / av1_highbd_warp_affine
 * (av1/common/warped_motion.c:264-393,538-675; AV1 spec 7.11.3.5 block warp process), the form av1_warp_plane runs for a WARPED_CAUSAL block or a
 * global-motion reference when the prediction is not a compound (conv_params->is_compound == 0).
 *
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_warp.npz (the reference's own two functions, interpreted where
 * they lie).  One loop nest for both pixel types; every intermediate keeps the reference's width (the horizontal sums are int32, the
 * 15 x 8 intermediate block is per 8 x 8 output tile as in the reference).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "aomref.h"

static const int16_t k_warped_filter[193][8] = {
#include "aomref_warp.inc"
};

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int rpot(int v, int n) { return (v + ((1 << n) >> 1)) >> n; }   /* ROUND_POWER_OF_TWO on a signed int as the reference applies it */

/* mat[6] = wmmat, (alpha, beta, gamma, delta) = the shear parameters (av1_get_shear_params); round_0 = conv_params->round_0 (3, or 5 at 12 bits:
 * get_conv_params_no_round); elem16: uint16 planes with bit depth bd, else uint8 (bd = 8) */
static void warp_affine(const int32_t *mat, const void *ref, int elem16, int width, int height, int stride, void *pred, int p_col, int p_row, int p_width,
                        int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd, int round_0, int alpha, int beta, int gamma, int delta,
                        int is_compound, int do_average, int use_dist_wtd, int fwd_offset, int bck_offset, uint16_t *conv, int conv_stride) {
  int32_t tmp[15 * 8];
  if (!elem16) bd = 8;
  const int extra = elem16 ? (bd + 7 - round_0 - 14 > 0 ? bd + 7 - round_0 - 14 : 0) : 0;   /* (highbd only: AOMMAX(bd + FILTER_BITS - round_0 - 14, 0)) */
  const int reduce_bits_horiz = round_0 + extra;
  const int round_1 = is_compound ? 7 : 2 * 7 - round_0;   /* COMPOUND_ROUND1_BITS (get_conv_params_no_round) */
  const int reduce_bits_vert = is_compound ? round_1 : 2 * 7 - reduce_bits_horiz;
  const int round_bits = 2 * 7 - round_0 - round_1, offset_bits = bd + 2 * 7 - round_0;
  const int offset_bits_horiz = bd + 7 - 1;
  const int offset_bits_vert = bd + 2 * 7 - reduce_bits_horiz;
  const int pmax = (1 << bd) - 1;
  for (int i = p_row; i < p_row + p_height; i += 8) {
    for (int j = p_col; j < p_col + p_width; j += 8) {
      const int32_t src_x = (j + 4) << subsampling_x, src_y = (i + 4) << subsampling_y;
      const int64_t dst_x = (int64_t)mat[2] * src_x + (int64_t)mat[3] * src_y + (int64_t)mat[0];
      const int64_t dst_y = (int64_t)mat[4] * src_x + (int64_t)mat[5] * src_y + (int64_t)mat[1];
      const int64_t x4 = dst_x >> subsampling_x, y4 = dst_y >> subsampling_y;
      const int32_t ix4 = (int32_t)(x4 >> 16), iy4 = (int32_t)(y4 >> 16);
      int32_t sx4 = (int32_t)(x4 & 0xffff), sy4 = (int32_t)(y4 & 0xffff);
      sx4 += alpha * (-4) + beta * (-4);
      sy4 += gamma * (-4) + delta * (-4);
      sx4 &= ~63;   /* WARP_PARAM_REDUCE_BITS */
      sy4 &= ~63;
      for (int k = -7; k < 8; ++k) {   /* horizontal filter */
        const int iy = clampi(iy4 + k, 0, height - 1);
        int sx = sx4 + beta * (k + 4);
        for (int l = -4; l < 4; ++l) {
          const int ix = ix4 + l - 3;
          const int offs = rpot(sx, 10) + 64;   /* WARPEDDIFF_PREC_BITS, WARPEDPIXEL_PREC_SHIFTS */
          const int16_t *c = k_warped_filter[offs];
          int32_t sum = 1 << offset_bits_horiz;
          for (int m = 0; m < 8; ++m) {
            const int sample_x = clampi(ix + m, 0, width - 1);
            const int px = elem16 ? ((const uint16_t *)ref)[(ptrdiff_t)iy * stride + sample_x] : ((const uint8_t *)ref)[(ptrdiff_t)iy * stride + sample_x];
            sum += px * c[m];
          }
          tmp[(k + 7) * 8 + (l + 4)] = rpot(sum, reduce_bits_horiz);
          sx += alpha;
        }
      }
      const int kmax = 4 < p_row + p_height - i - 4 ? 4 : p_row + p_height - i - 4, lmax = 4 < p_col + p_width - j - 4 ? 4 : p_col + p_width - j - 4;
      for (int k = -4; k < kmax; ++k) {   /* vertical filter */
        int sy = sy4 + delta * (k + 4);
        for (int l = -4; l < lmax; ++l) {
          const int offs = rpot(sy, 10) + 64;
          const int16_t *c = k_warped_filter[offs];
          int32_t sum = 1 << offset_bits_vert;
          for (int m = 0; m < 8; ++m) sum += tmp[(k + m + 4) * 8 + (l + 4)] * c[m];
          sum = rpot(sum, reduce_bits_vert);
          const ptrdiff_t o = (ptrdiff_t)(i - p_row + k + 4) * p_stride + (j - p_col + l + 4);
          int v;
          if (is_compound) {
            uint16_t *cp = &conv[(ptrdiff_t)(i - p_row + k + 4) * conv_stride + (j - p_col + l + 4)];
            if (!do_average) { *cp = (uint16_t)sum; sy += gamma; continue; }   /* CONV_BUF_TYPE is uint16_t */
            int32_t t32 = *cp;
            if (use_dist_wtd) t32 = (t32 * fwd_offset + sum * bck_offset) >> 4;   /* DIST_PRECISION_BITS */
            else t32 = (t32 + sum) >> 1;
            t32 = t32 - (1 << (offset_bits - round_1)) - (1 << (offset_bits - round_1 - 1));
            v = clampi(rpot(t32, round_bits), 0, pmax);
          } else {
            v = clampi(sum - (1 << (bd - 1)) - (1 << bd), 0, pmax);
          }
          if (elem16) ((uint16_t *)pred)[o] = (uint16_t)v;
          else ((uint8_t *)pred)[o] = (uint8_t)v;
          sy += gamma;
        }
      }
    }
  }
}

void orc_warp_affine(const int32_t *mat, const void *ref, int elem16, int width, int height, int stride, void *pred, int p_col, int p_row, int p_width,
                     int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd, int round_0, int alpha, int beta, int gamma, int delta) {
  warp_affine(mat, ref, elem16, width, height, stride, pred, p_col, p_row, p_width, p_height, p_stride, subsampling_x, subsampling_y, bd, round_0, alpha, beta,
              gamma, delta, 0, 0, 0, 0, 0, NULL, 0);
}
/* conv_params->is_compound = 1: do_average 0 writes the block's CONV_BUF (conv, conv_stride: the block's own buffer, element (0, 0) = its first pixel),
 * do_average 1 blends with it -- plain average or the distance weights -- into pred */
void orc_warp_affine_compound(const int32_t *mat, const void *ref, int elem16, int width, int height, int stride, void *pred, int p_col, int p_row, int p_width,
                              int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd, int round_0, int alpha, int beta, int gamma, int delta,
                              int do_average, int use_dist_wtd, int fwd_offset, int bck_offset, uint16_t *conv, int conv_stride) {
  warp_affine(mat, ref, elem16, width, height, stride, pred, p_col, p_row, p_width, p_height, p_stride, subsampling_x, subsampling_y, bd, round_0, alpha, beta,
              gamma, delta, 1, do_average, use_dist_wtd, fwd_offset, bck_offset, conv, conv_stride);
}

/* ---- the global-motion search's use of the warp: av1_get_shear_params (av1/common/warped_motion.c:186-247), av1_warp_error (av1/encoder/global_motion.c:
 * 128-224: 32 x 32 tiles of the model's prediction against the frame, only where the segment map holds inliers, every pixel's difference through
 * error_measure_lut -- interpolated between neighbouring entries above 8 bits, warped_motion.c:248-259) and av1_segmented_frame_error (warped_motion.c:
 * 400-460,687-760: the same metric without a warp).  Pinned by tests/golden/ref_eval_warp_error.npz. */
#include "aomref_warp_error.inc"
static const int k_error_measure_lut[512] = AOMHIP_ERROR_MEASURE_LUT;
static const uint16_t k_div_lut[257] = AOMHIP_DIV_LUT;

static int64_t rpot_s64(int64_t v, int n) { return v < 0 ? -((-v + ((int64_t)1 << n >> 1)) >> n) : (v + ((int64_t)1 << n >> 1)) >> n; }
static int rpot_s(int v, int n) { return v < 0 ? -((-v + ((1 << n) >> 1)) >> n) : (v + ((1 << n) >> 1)) >> n; }

int orc_get_shear_params(const int32_t *mat, int16_t *abgd) {
  if (mat[2] <= 0) return 0;   /* is_affine_valid */
  int alpha = clampi(mat[2] - (1 << 16), INT16_MIN, INT16_MAX), beta = clampi(mat[3], INT16_MIN, INT16_MAX);
  const uint32_t D = (uint32_t)mat[2];
  int shift = 31;
  while (!(D >> shift)) --shift;   /* get_msb */
  const int32_t e = (int32_t)(D - ((uint32_t)1 << shift));
  const int32_t f = shift > 8 ? (e + ((1 << (shift - 8)) >> 1)) >> (shift - 8) : e << (8 - shift);
  shift += 14;
  const int16_t y = (int16_t)k_div_lut[f];
  int64_t v = ((int64_t)mat[4] * (1 << 16)) * y;
  int gamma = clampi((int)rpot_s64(v, shift), INT16_MIN, INT16_MAX);
  v = ((int64_t)mat[3] * mat[4]) * y;
  int delta = clampi(mat[5] - (int)rpot_s64(v, shift) - (1 << 16), INT16_MIN, INT16_MAX);
  alpha = rpot_s(alpha, 6) * 64;   /* WARP_PARAM_REDUCE_BITS */
  beta = rpot_s(beta, 6) * 64;
  gamma = rpot_s(gamma, 6) * 64;
  delta = rpot_s(delta, 6) * 64;
  abgd[0] = (int16_t)alpha; abgd[1] = (int16_t)beta; abgd[2] = (int16_t)gamma; abgd[3] = (int16_t)delta;
  if (4 * abs(alpha) + 7 * abs(beta) >= (1 << 16) || 4 * abs(gamma) + 4 * abs(delta) >= (1 << 16)) return 0;   /* is_affine_shear_allowed */
  return 1;
}

static int64_t frame_error(const void *ref, int stride, const void *dst, int w, int h, int p_stride, int elem16, int bd) {
  int64_t sum = 0;
  const int b = bd - 8, bmask = (1 << b) - 1, v = 1 << b;
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j) {
      if (elem16) {
        const int err = abs((int)((const uint16_t *)dst)[j + (ptrdiff_t)i * p_stride] - (int)((const uint16_t *)ref)[j + (ptrdiff_t)i * stride]);
        const int e1 = err >> b, e2 = err & bmask;
        sum += k_error_measure_lut[255 + e1] * (v - e2) + k_error_measure_lut[256 + e1] * e2;
      } else {
        sum += k_error_measure_lut[255 + (int)((const uint8_t *)dst)[j + (ptrdiff_t)i * p_stride] - (int)((const uint8_t *)ref)[j + (ptrdiff_t)i * stride]];
      }
    }
  return sum;
}

/* mat + the four shear values as av1_get_shear_params left them in the model; ref / dst point at pixel (0, 0) of their planes */
int64_t orc_warp_error(const int32_t *mat, const int16_t *abgd, const void *ref, int elem16, int width, int height, int stride, const void *dst, int p_col,
                       int p_row, int p_width, int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd, int64_t best_error,
                       const uint8_t *segment_map, int segment_map_stride) {
  int64_t sumerr = 0;
  const int bw = p_width < 32 ? p_width : 32, bh = p_height < 32 ? p_height : 32;   /* WARP_ERROR_BLOCK */
  uint16_t tmp16[32 * 32];
  uint8_t tmp8[32 * 32];
  const int es = elem16 ? 2 : 1;
  for (int i = p_row; i < p_row + p_height; i += 32)
    for (int j = p_col; j < p_col + p_width; j += 32) {
      if (!segment_map[(i >> 5) * segment_map_stride + (j >> 5)]) continue;
      const int ww = bw < p_col + p_width - j ? bw : p_col + p_width - j, wh = bh < p_row + p_height - i ? bh : p_row + p_height - i;
      void *tmp = elem16 ? (void *)tmp16 : (void *)tmp8;
      orc_warp_affine(mat, ref, elem16, width, height, stride, tmp, j, i, ww, wh, 32, subsampling_x, subsampling_y, bd, bd == 12 ? 5 : 3, abgd[0], abgd[1],
                      abgd[2], abgd[3]);
      sumerr += frame_error(tmp, 32, (const char *)dst + ((ptrdiff_t)i * p_stride + j) * es, ww, wh, p_stride, elem16, bd);
      if (sumerr > best_error) return INT64_MAX;
    }
  return sumerr;
}

int64_t orc_segmented_frame_error(const void *ref, int elem16, int stride, const void *dst, int p_width, int p_height, int p_stride, int bd,
                                  const uint8_t *segment_map, int segment_map_stride) {
  int64_t sum = 0;
  const int bw = p_width < 32 ? p_width : 32, bh = p_height < 32 ? p_height : 32, es = elem16 ? 2 : 1;
  for (int i = 0; i < p_height; i += 32)
    for (int j = 0; j < p_width; j += 32) {
      if (!segment_map[(i >> 5) * segment_map_stride + (j >> 5)]) continue;
      const int pw = bw < p_width - j ? bw : p_width - j, ph = bh < p_height - i ? bh : p_height - i;
      sum += frame_error((const char *)ref + ((ptrdiff_t)i * stride + j) * es, stride, (const char *)dst + ((ptrdiff_t)i * p_stride + j) * es, pw, ph,
                         p_stride, elem16, bd);
    }
  return sum;
}

