// The local warp model of a WARPED_CAUSAL block, fitted to the motion of its neighbours: av1_selectSamples (av1/common/mvref_common.c:1083-1104) and
// av1_find_projection (av1/common/warped_motion.c:894-1015).  Scalar integer arithmetic per candidate MV; shared by the host entry points
// (host/warp_model.c, plain C99) and the device composite that refines a warped block's MV (csrc/warp_refine.hip), hence a header of static functions
// with no dependency but <stdint.h>.  Not part of the ABI.
#ifndef AOMHIP_CSRC_WARP_FIT_H_
#define AOMHIP_CSRC_WARP_FIT_H_

#include <stdint.h>

#ifdef __HIPCC__
#define AOMHIP_WF __host__ __device__ static inline
#else
#define AOMHIP_WF static inline
#endif

AOMHIP_WF int wf_iabs(int v) { return v < 0 ? -v : v; }
AOMHIP_WF int64_t wf_round_signed(int64_t v, int n) {   // ROUND_POWER_OF_TWO_SIGNED_64
  const int64_t half = ((int64_t)1 << n) >> 1;
  return v < 0 ? -((-v + half) >> n) : (v + half) >> n;
}
AOMHIP_WF int64_t wf_clamp64(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// av1_selectSamples: the samples whose own motion (pts_inref - pts) lies within clamp(max(bw, bh), 16, 112) of the block's MV in the L1 norm
// (1/8 pel) move to the front of both arrays, in order; at least one sample counts as kept.
AOMHIP_WF int wf_select_samples(int mv_row, int mv_col, int *pts, int *pts_inref, int len, int bw, int bh) {
  const int m = bw > bh ? bw : bh;
  const int thresh = m < 16 ? 16 : (m > 112 ? 112 : m);
  int kept = 0;
  for (int i = 0; i < len; ++i) {
    if (wf_iabs(pts_inref[2 * i] - pts[2 * i] - mv_col) + wf_iabs(pts_inref[2 * i + 1] - pts[2 * i + 1] - mv_row) > thresh) continue;
    if (kept != i) {
      pts[2 * kept] = pts[2 * i]; pts[2 * kept + 1] = pts[2 * i + 1];
      pts_inref[2 * kept] = pts_inref[2 * i]; pts_inref[2 * kept + 1] = pts_inref[2 * i + 1];
    }
    ++kept;
  }
  return kept > 1 ? kept : 1;
}

// find_affine_int (warped_motion.c:894-1002).  div_lut: the 257-entry reciprocal table (warp_error_table.inc).  mat[0 .. 5] = wmmat; returns 0 and
// leaves mat alone when the normal equations are singular, 1 otherwise (the caller then asks for the shear decomposition).
// The products of the normal equations keep 1/8-pel coordinates in 32 bits: (a + LS_STEP / 2)-style roundings with LS_STEP = 8 and two bits dropped.
AOMHIP_WF int wf_find_affine(int np, const int *pts1, const int *pts2, int bw, int bh, int mvy, int mvx, int mi_row, int mi_col, const uint16_t *div_lut,
                             int32_t *mat) {
  const int rsuy = bh / 2 - 1, rsux = bw / 2 - 1;
  const int suy = rsuy * 8, sux = rsux * 8, duy = suy + mvy, dux = sux + mvx;
  int32_t a00 = 0, a01 = 0, a11 = 0, bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
  for (int i = 0; i < np; ++i) {
    const int dx = pts2[2 * i] - dux, dy = pts2[2 * i + 1] - duy;
    const int sx = pts1[2 * i] - sux, sy = pts1[2 * i + 1] - suy;
    if (wf_iabs(sx - dx) >= 256 || wf_iabs(sy - dy) >= 256) continue;   // LS_MV_MAX
    a00 += (sx * sx * 4 + sx * 32 + 128) >> 4;                          // LS_SQUARE
    a01 += (sx * sy * 4 + (sx + sy) * 16 + 64) >> 4;                    // LS_PRODUCT1
    a11 += (sy * sy * 4 + sy * 32 + 128) >> 4;
    bx0 += (sx * dx * 4 + (sx + dx) * 16 + 128) >> 4;                   // LS_PRODUCT2
    bx1 += (sy * dx * 4 + (sy + dx) * 16 + 64) >> 4;
    by0 += (sx * dy * 4 + (sx + dy) * 16 + 64) >> 4;
    by1 += (sy * dy * 4 + (sy + dy) * 16 + 128) >> 4;
  }
  const int64_t det = (int64_t)a00 * a11 - (int64_t)a01 * a01;
  if (det == 0) return 0;
  // resolve_divisor_64: 1 / |det| = y / 2^shift with y from the table at the 8 bits below |det|'s leading one
  const uint64_t d = (uint64_t)(det < 0 ? -det : det);
  int msb = 63;
  while (!((d >> msb) & 1)) --msb;
  const int64_t e = (int64_t)(d - ((uint64_t)1 << msb));
  const int64_t f = msb > 8 ? (e + (((int64_t)1 << (msb - 8)) >> 1)) >> (msb - 8) : e << (8 - msb);
  int shift = msb + 14 - 16;                                            // DIV_LUT_PREC_BITS - WARPEDMODEL_PREC_BITS
  int idet = det < 0 ? -(int)div_lut[f] : (int)div_lut[f];
  if (shift < 0) {
    idet = (int16_t)(idet << -shift);                                   // (the reference shifts an int16_t in place)
    shift = 0;
  }
  const int64_t px0 = (int64_t)a11 * bx0 - (int64_t)a01 * bx1, px1 = (int64_t)a00 * bx1 - (int64_t)a01 * bx0;
  const int64_t py0 = (int64_t)a11 * by0 - (int64_t)a01 * by1, py1 = (int64_t)a00 * by1 - (int64_t)a01 * by0;
  const int64_t lim = (1 << 13) - 1, one = 1 << 16;                     // WARPEDMODEL_NONDIAGAFFINE_CLAMP - 1, 1 << WARPEDMODEL_PREC_BITS
  mat[2] = (int32_t)wf_clamp64(wf_round_signed(px0 * idet, shift), one - lim, one + lim);
  mat[3] = (int32_t)wf_clamp64(wf_round_signed(px1 * idet, shift), -lim, lim);
  mat[4] = (int32_t)wf_clamp64(wf_round_signed(py0 * idet, shift), -lim, lim);
  mat[5] = (int32_t)wf_clamp64(wf_round_signed(py1 * idet, shift), one - lim, one + lim);
  // the translation that maps the block's centre to itself + the MV
  const int isuy = mi_row * 4 + rsuy, isux = mi_col * 4 + rsux;
  const int32_t vx = mvx * (1 << 13) - (isux * (mat[2] - (int32_t)one) + isuy * mat[3]);
  const int32_t vy = mvy * (1 << 13) - (isux * mat[4] + isuy * (mat[5] - (int32_t)one));
  const int32_t tmax = (1 << 23) - 1, tmin = -(1 << 23);                // WARPEDMODEL_TRANS_CLAMP
  mat[0] = vx < tmin ? tmin : (vx > tmax ? tmax : vx);
  mat[1] = vy < tmin ? tmin : (vy > tmax ? tmax : vy);
  return 1;
}

// av1_get_shear_params (warped_motion.c:186-247) on mat[2 .. 5]: the four shear values and is_affine_shear_allowed's verdict (1 = usable)
AOMHIP_WF int wf_shear(const int32_t *mat, const uint16_t *div_lut, int16_t *abgd) {
  if (mat[2] <= 0) return 0;
  const uint32_t d = (uint32_t)mat[2];
  int msb = 31;
  while (!((d >> msb) & 1)) --msb;
  const int32_t e = (int32_t)(d - ((uint32_t)1 << msb));
  const int32_t f = msb > 8 ? (e + ((1 << (msb - 8)) >> 1)) >> (msb - 8) : e << (8 - msb);
  const int shift = msb + 14;
  const int64_t y = (int16_t)div_lut[f];
  int v[4];
  v[0] = (int)wf_clamp64((int64_t)mat[2] - (1 << 16), INT16_MIN, INT16_MAX);
  v[1] = (int)wf_clamp64(mat[3], INT16_MIN, INT16_MAX);
  v[2] = (int)wf_clamp64((int)wf_round_signed(((int64_t)mat[4] * (1 << 16)) * y, shift), INT16_MIN, INT16_MAX);   // (rounded quotient truncated to int first)
  v[3] = (int)wf_clamp64((int64_t)mat[5] - (int)wf_round_signed(((int64_t)mat[3] * mat[4]) * y, shift) - (1 << 16), INT16_MIN, INT16_MAX);
  for (int i = 0; i < 4; ++i) {
    v[i] = (int)wf_round_signed(v[i], 6) * 64;                          // WARP_PARAM_REDUCE_BITS
    abgd[i] = (int16_t)v[i];
  }
  return !(4 * wf_iabs(v[0]) + 7 * wf_iabs(v[1]) >= (1 << 16) || 4 * wf_iabs(v[2]) + 4 * wf_iabs(v[3]) >= (1 << 16));
}

#endif  // AOMHIP_CSRC_WARP_FIT_H_
