/* The shear decomposition of an affine motion model (av1_get_shear_params, av1/common/warped_motion.c:186-247; AV1 specification 7.11.3.6): host-side scalar
 * arithmetic the global-motion search runs on every candidate before it measures it (aomhip_warp_error_batch).  Plain C, no GPU. */
#include <stdint.h>
#include <stdlib.h>

#include "aomhip.h"
#include "../csrc/warp_error_table.inc"

static const uint16_t k_div_lut[257] = AOMHIP_DIV_LUT;   /* round(2^22 / (256 + i)) */

static int clamp16(int64_t v) { return v < INT16_MIN ? INT16_MIN : (v > INT16_MAX ? INT16_MAX : (int)v); }
/* ROUND_POWER_OF_TWO_SIGNED[_64] */
static int64_t round_signed(int64_t v, int n) { return v < 0 ? -((-v + (((int64_t)1 << n) >> 1)) >> n) : (v + (((int64_t)1 << n) >> 1)) >> n; }
static int reduce(int v) { return (int)round_signed(v, 6) * 64; }   /* WARP_PARAM_REDUCE_BITS */

int aomhip_get_shear_params(aomhip_warp_model *model) {
  const int32_t *mat = model->mat;
  if (mat[2] <= 0) return 0;   /* is_affine_valid */
  int alpha = clamp16((int64_t)mat[2] - (1 << 16)), beta = clamp16(mat[3]);   /* WARPEDMODEL_PREC_BITS */
  /* resolve_divisor_32: the reciprocal of mat[2] as a 14-bit table entry and a shift */
  const uint32_t d = (uint32_t)mat[2];
  int shift = 31;
  while (!(d >> shift)) --shift;
  const int32_t e = (int32_t)(d - ((uint32_t)1 << shift));
  const int32_t f = shift > 8 ? (e + ((1 << (shift - 8)) >> 1)) >> (shift - 8) : e << (8 - shift);
  shift += 14;
  const int16_t y = (int16_t)k_div_lut[f];
  int64_t v = ((int64_t)mat[4] * (1 << 16)) * y;
  int gamma = clamp16((int)round_signed(v, shift));   /* (the reference truncates the rounded quotient to int before it clamps: degenerate models wrap) */
  v = ((int64_t)mat[3] * mat[4]) * y;
  int delta = clamp16((int64_t)mat[5] - (int)round_signed(v, shift) - (1 << 16));
  alpha = reduce(alpha);
  beta = reduce(beta);
  gamma = reduce(gamma);
  delta = reduce(delta);
  model->alpha = (int16_t)alpha;
  model->beta = (int16_t)beta;
  model->gamma = (int16_t)gamma;
  model->delta = (int16_t)delta;
  /* is_affine_shear_allowed */
  return !(4 * abs(alpha) + 7 * abs(beta) >= (1 << 16) || 4 * abs(gamma) + 4 * abs(delta) >= (1 << 16));
}

/* ---- the LOCAL warp model (WARPED_CAUSAL blocks): av1_selectSamples and av1_find_projection; the arithmetic is csrc/warp_fit.h, shared with the device
 * composite that refines such a block's MV ---- */
#include "../csrc/warp_fit.h"

int aomhip_select_samples(int mv_row, int mv_col, int32_t *pts, int32_t *pts_inref, int n_samples, int bw, int bh) {
  if (!pts || !pts_inref || n_samples < 1 || n_samples > 8) return -1;   /* LEAST_SQUARES_SAMPLES_MAX */
  return wf_select_samples(mv_row, mv_col, pts, pts_inref, n_samples, bw, bh);
}

int aomhip_find_projection(int n_samples, const int32_t *pts, const int32_t *pts_inref, int bw, int bh, int mv_row, int mv_col, aomhip_warp_model *model,
                           int mi_row, int mi_col) {
  if (!pts || !pts_inref || !model || n_samples < 1) return 0;
  if (!wf_find_affine(n_samples, pts, pts_inref, bw, bh, mv_row, mv_col, mi_row, mi_col, k_div_lut, model->mat)) return 0;
  return wf_shear(model->mat, k_div_lut, &model->alpha);
}
