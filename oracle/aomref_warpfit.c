/*
 * oracle/aomref_warpfit.c -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline): the local warp model's fit.
 *
 *   orc_select_samples    av1_selectSamples (av1/common/mvref_common.c:1083-1104): keep the samples whose motion is within
 *                         clamp(max(bw, bh), 16, 112) (L1, 1/8 pel) of the block's MV, compacted in place; at least one.
 *   orc_find_projection   av1_find_projection (av1/common/warped_motion.c:1004-1015) = find_affine_int (:894-1002: the sums of the two 2 x 2
 *                         least-squares systems through LS_SQUARE / LS_PRODUCT1 / LS_PRODUCT2 (:807-815), the determinant through resolve_divisor_64
 *                         (:170-185), get_mult_shift_diag / _ndiag (:881-892), the translation that keeps the block's centre on its MV) followed by
 *                         av1_get_shear_params' verdict (aomref_warp.c).  Returns what the reference returns: 1 = no usable model.
 *
 * PINNED by tests/golden/ref_eval_warpfit.npz (both functions interpreted, 260 neighbourhoods).
 */
#include <stdint.h>
#include <stdlib.h>

#include "aomref.h"
#include "aomref_warp_error.inc"

static const uint16_t k_div_lut_fit[257] = AOMHIP_DIV_LUT;

int orc_select_samples(int mv_row, int mv_col, int *pts, int *pts_inref, int len, int bw, int bh) {
  int thresh = bw > bh ? bw : bh;
  thresh = thresh < 16 ? 16 : thresh > 112 ? 112 : thresh;
  int ret = 0;
  for (int i = 0; i < len; ++i) {
    const int diff = abs(pts_inref[2 * i] - pts[2 * i] - mv_col) + abs(pts_inref[2 * i + 1] - pts[2 * i + 1] - mv_row);
    if (diff > thresh) continue;
    if (ret != i) {
      pts[2 * ret] = pts[2 * i]; pts[2 * ret + 1] = pts[2 * i + 1];
      pts_inref[2 * ret] = pts_inref[2 * i]; pts_inref[2 * ret + 1] = pts_inref[2 * i + 1];
    }
    ++ret;
  }
  return ret > 1 ? ret : 1;
}

/* LS_STEP 8, LS_MAT_DOWN_BITS 2 (warped_motion.c:784-815) */
static int32_t ls_square(int a) { return (a * a * 4 + a * 4 * 8 + 8 * 8 * 2) >> 4; }
static int32_t ls_product1(int a, int b) { return (a * b * 4 + (a + b) * 2 * 8 + 8 * 8) >> 4; }
static int32_t ls_product2(int a, int b) { return (a * b * 4 + (a + b) * 2 * 8 + 8 * 8 * 2) >> 4; }
static int64_t rpot_signed_64(int64_t v, int n) { return v < 0 ? -((-v + (((int64_t)1 << n) >> 1)) >> n) : (v + (((int64_t)1 << n) >> 1)) >> n; }
static int64_t clamp_64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : v > hi ? hi : v; }

int orc_find_projection(int np, const int *pts1, const int *pts2, int bw, int bh, int mvy, int mvx, int32_t *mat, int16_t *abgd, int mi_row, int mi_col) {
  int32_t A00 = 0, A01 = 0, A11 = 0, Bx0 = 0, Bx1 = 0, By0 = 0, By1 = 0;
  const int rsuy = bh / 2 - 1, rsux = bw / 2 - 1, suy = rsuy * 8, sux = rsux * 8, duy = suy + mvy, dux = sux + mvx;
  for (int i = 0; i < np; ++i) {
    const int dx = pts2[2 * i] - dux, dy = pts2[2 * i + 1] - duy, sx = pts1[2 * i] - sux, sy = pts1[2 * i + 1] - suy;
    if (abs(sx - dx) < 256 && abs(sy - dy) < 256) {   /* LS_MV_MAX */
      A00 += ls_square(sx); A01 += ls_product1(sx, sy); A11 += ls_square(sy);
      Bx0 += ls_product2(sx, dx); Bx1 += ls_product1(sy, dx);
      By0 += ls_product1(sx, dy); By1 += ls_product2(sy, dy);
    }
  }
  const int64_t det = (int64_t)A00 * A11 - (int64_t)A01 * A01;
  if (det == 0) return 1;
  /* resolve_divisor_64(|det|): 1 / D = y / 2^shift */
  const uint64_t D = (uint64_t)(det < 0 ? -det : det);
  int16_t shift = 63;
  while (!(D >> shift)) --shift;
  const int64_t e = (int64_t)(D - ((uint64_t)1 << shift));
  const int64_t f = shift > 8 ? (e + (((int64_t)1 << (shift - 8)) >> 1)) >> (shift - 8) : e << (8 - shift);
  shift += 14;
  int16_t idet = (int16_t)((int16_t)k_div_lut_fit[f] * (det < 0 ? -1 : 1));
  shift -= 16;   /* WARPEDMODEL_PREC_BITS */
  if (shift < 0) {
    idet = (int16_t)(idet << (-shift));   /* (an int16_t left shift in the reference: the int result is stored back into 16 bits) */
    shift = 0;
  }
  const int64_t px0 = (int64_t)A11 * Bx0 - (int64_t)A01 * Bx1, px1 = -(int64_t)A01 * Bx0 + (int64_t)A00 * Bx1;
  const int64_t py0 = (int64_t)A11 * By0 - (int64_t)A01 * By1, py1 = -(int64_t)A01 * By0 + (int64_t)A00 * By1;
  const int64_t nd = (1 << 13) - 1;   /* WARPEDMODEL_NONDIAGAFFINE_CLAMP - 1 */
  mat[2] = (int32_t)clamp_64(rpot_signed_64(px0 * idet, shift), (1 << 16) - nd, (1 << 16) + nd);
  mat[3] = (int32_t)clamp_64(rpot_signed_64(px1 * idet, shift), -nd, nd);
  mat[4] = (int32_t)clamp_64(rpot_signed_64(py0 * idet, shift), -nd, nd);
  mat[5] = (int32_t)clamp_64(rpot_signed_64(py1 * idet, shift), (1 << 16) - nd, (1 << 16) + nd);
  const int isuy = mi_row * 4 + rsuy, isux = mi_col * 4 + rsux;
  const int32_t vx = mvx * (1 << 13) - (isux * (mat[2] - (1 << 16)) + isuy * mat[3]);
  const int32_t vy = mvy * (1 << 13) - (isux * mat[4] + isuy * (mat[5] - (1 << 16)));
  mat[0] = vx < -(1 << 23) ? -(1 << 23) : vx > (1 << 23) - 1 ? (1 << 23) - 1 : vx;   /* WARPEDMODEL_TRANS_CLAMP */
  mat[1] = vy < -(1 << 23) ? -(1 << 23) : vy > (1 << 23) - 1 ? (1 << 23) - 1 : vy;
  return !orc_get_shear_params(mat, abgd);
}
