/*
 * oracle/aomref_quant.c -- dead-zone quantisers, quantiser tables, scan orders.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Restates aom_dsp/quantize.c:108-169,261-316,
 * av1/encoder/av1_quantize.c:580-674, av1/common/quant_common.c:193-215 and the scan tables
 * of av1/common/scan.c (generated from their zig-zag / row / column rule instead of 7 900
 * literal entries; tests compare every generated table with the reference initialiser).
 *
 * PINNED by interpreting the reference's aom_[highbd_]quantize_b{,_32x32,_64x64}[_adaptive]_c (688 cases) and
 * av1_build_quantizer (8/10/12-bit, all 256 qindex, Y/U/V with delta_q): tests/golden/ref_eval_quant.npz,
 * ref_eval_tables.npz, checked bit for bit in tests/test_golden_ref_eval.py.  (The reference's own tests for the
 * quantisers are SIMD-vs-C only, test/quantize_func_test.cc.)
 */
#include "aomref.h"

#include <string.h>

#define QM_BITS 5 /* aom_dsp/quantize.h AOM_QM_BITS; qmatrix pointers NULL => wt = iwt = 32 (the _qm forms take them) */

static int rpot(int v, int n) { return (v + ((1 << n) >> 1)) >> n; } /* ROUND_POWER_OF_TWO */

/* aom_quantize_b_helper_c (quantize.c:108-169) with its qm_ptr / iqm_ptr arguments (NULL: wt = iwt = 32).  The quantisation matrices are
 * indexed by the coefficient's buffer position rc like the coefficients themselves.  PINNED with matrices by
 * tests/golden/ref_eval_qm.npz (the helpers interpreted with the level 0 / 8 / 14 matrices of av1/common/quant_common.c). */
void orc_quantize_b_qm(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                       const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                       const int16_t *dequant, uint16_t *eob, const int16_t *scan, int log_scale, const uint8_t *qm,
                       const uint8_t *iqm) {
  const int zb[2] = { rpot(zbin[0], log_scale), rpot(zbin[1], log_scale) };
  int last = -1;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  for (int i = 0; i < (int)n; ++i) {
    const int rc = scan[i], ac = (rc != 0);
    const int wt = qm ? qm[rc] : 1 << QM_BITS, iwt = iqm ? iqm[rc] : 1 << QM_BITS;
    const int c = coeff[rc];
    const int sign = c >> 31;
    const int a = (c ^ sign) - sign;
    if (a * wt < (zb[ac] << QM_BITS)) continue; /* pre-scan (:127-137) and :62 are the same test */
    int64_t t = a + rpot(round[ac], log_scale);
    if (t > INT16_MAX) t = INT16_MAX;
    if (t < INT16_MIN) t = INT16_MIN;
    t *= wt;
    const int q = (int)(((((t * quant[ac]) >> 16) + t) * quant_shift[ac]) >> (16 - log_scale + QM_BITS));
    qcoeff[rc] = (q ^ sign) - sign;
    const int dq = (dequant[ac] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
    const int adq = (q * dq) >> log_scale;
    dqcoeff[rc] = (adq ^ sign) - sign;
    if (q) last = i;
  }
  *eob = (uint16_t)(last + 1);
}

/* aom_highbd_quantize_b_helper_c (quantize.c:261-316) with qm_ptr / iqm_ptr */
void orc_highbd_quantize_b_qm(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                              const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                              const int16_t *dequant, uint16_t *eob, const int16_t *scan, int log_scale,
                              const uint8_t *qm, const uint8_t *iqm) {
  const int zb[2] = { rpot(zbin[0], log_scale), rpot(zbin[1], log_scale) };
  int last = -1;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  for (int i = 0; i < (int)n; ++i) {
    const int rc = scan[i], ac = (rc != 0);
    const int wt = qm ? qm[rc] : 1 << QM_BITS, iwt = iqm ? iqm[rc] : 1 << QM_BITS;
    const int c = coeff[rc];
    const int cw = (int)((uint32_t)c * (uint32_t)wt);
    if (!(cw >= zb[ac] * (1 << QM_BITS) || cw <= -zb[ac] * (1 << QM_BITS))) continue;
    const int sign = c >> 31;
    const int a = (c ^ sign) - sign;
    const int64_t t1 = a + rpot(round[ac], log_scale);
    const int64_t tw = t1 * wt;
    const int64_t t2 = ((tw * quant[ac]) >> 16) + tw;
    const int q = (int)((t2 * quant_shift[ac]) >> (16 - log_scale + QM_BITS));
    qcoeff[rc] = (q ^ sign) - sign;
    const int dq = (dequant[ac] * iwt + (1 << (QM_BITS - 1))) >> QM_BITS;
    const int adq = (int)((uint32_t)q * (uint32_t)dq) >> log_scale;
    dqcoeff[rc] = (adq ^ sign) - sign;
    if (q) last = i;
  }
  *eob = (uint16_t)(last + 1);
}

void orc_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                    const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                    const int16_t *dequant, uint16_t *eob, const int16_t *scan, const int16_t *iscan,
                    int log_scale) {
  (void)iscan;
  const int zb[2] = { rpot(zbin[0], log_scale), rpot(zbin[1], log_scale) };
  const int wt = 1 << QM_BITS;
  int last = -1;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  /* The reference first walks the scan backwards to drop the all-inside-dead-zone tail
   * (quantize.c:127-137); coefficients it drops fail the test below anyway, so a single
   * forward pass over every position is equivalent. */
  for (int i = 0; i < (int)n; ++i) {
    const int rc = scan[i], ac = (rc != 0);
    const int c = coeff[rc];
    const int sign = c >> 31; /* AOMSIGN */
    const int a = (c ^ sign) - sign;
    if (a * wt < (zb[ac] << QM_BITS)) continue;
    int64_t t = a + rpot(round[ac], log_scale);
    if (t > INT16_MAX) t = INT16_MAX; /* clamp(..., INT16_MIN, INT16_MAX): low-bd only */
    if (t < INT16_MIN) t = INT16_MIN;
    t *= wt;
    const int q = (int)(((((t * quant[ac]) >> 16) + t) * quant_shift[ac]) >> (16 - log_scale + QM_BITS));
    qcoeff[rc] = (q ^ sign) - sign;
    const int dq = (dequant[ac] * wt + (1 << (QM_BITS - 1))) >> QM_BITS;
    const int adq = (q * dq) >> log_scale;
    dqcoeff[rc] = (adq ^ sign) - sign;
    if (q) last = i;
  }
  *eob = (uint16_t)(last + 1);
}

void orc_highbd_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                           const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff,
                           int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob, const int16_t *scan,
                           const int16_t *iscan, int log_scale) {
  (void)iscan;
  const int zb[2] = { rpot(zbin[0], log_scale), rpot(zbin[1], log_scale) };
  const int wt = 1 << QM_BITS;
  int last = -1;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  for (int i = 0; i < (int)n; ++i) {
    const int rc = scan[i], ac = (rc != 0);
    const int c = coeff[rc];
    /* pre-scan test (quantize.c:281-291): keep when coeff*wt is outside (-zbin*32, zbin*32) */
    const int cw = (int)((uint32_t)c * (uint32_t)wt);
    if (!(cw >= zb[ac] * (1 << QM_BITS) || cw <= -zb[ac] * (1 << QM_BITS))) continue;
    const int sign = c >> 31;
    const int a = (c ^ sign) - sign;
    const int64_t t1 = a + rpot(round[ac], log_scale);
    const int64_t tw = t1 * wt;
    const int64_t t2 = ((tw * quant[ac]) >> 16) + tw;
    const int q = (int)((t2 * quant_shift[ac]) >> (16 - log_scale + QM_BITS));
    qcoeff[rc] = (q ^ sign) - sign;
    const int dq = (dequant[ac] * wt + (1 << (QM_BITS - 1))) >> QM_BITS;
    const int adq = (int)((uint32_t)q * (uint32_t)dq) >> log_scale;
    dqcoeff[rc] = (adq ^ sign) - sign;
    if (q) last = i;
  }
  *eob = (uint16_t)(last + 1);
}

/* ------------------------------------------------------------------ tables */

static const int16_t k_qlookup[6][256] = {
#include "aomref_qlookup.inc"
};

static int clampq(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
static int bd_row(int bit_depth) { return bit_depth == 10 ? 1 : bit_depth == 12 ? 2 : 0; }

int16_t orc_dc_q(int qindex, int delta, int bit_depth) { return k_qlookup[bd_row(bit_depth)][clampq(qindex + delta)]; }
int16_t orc_ac_q(int qindex, int delta, int bit_depth) {
  return k_qlookup[3 + bd_row(bit_depth)][clampq(qindex + delta)];
}

void orc_build_quantizer_y(int bit_depth, int qindex, int16_t tables[5][2]) {
  /* get_qzbin_factor (av1_quantize.c:590-602) */
  const int dcq = orc_dc_q(qindex, 0, bit_depth);
  const int thr = bit_depth == 8 ? 148 : bit_depth == 10 ? 592 : 2368;
  const int zbin_factor = qindex == 0 ? 64 : (dcq < thr ? 84 : 80);
  const int round_factor = qindex == 0 ? 64 : 48;
  for (int i = 0; i < 2; ++i) {
    const int d = i == 0 ? dcq : orc_ac_q(qindex, 0, bit_depth);
    /* invert_quant (:580-588) */
    int l = 0;
    for (uint32_t t = (uint32_t)d; t > 1; t >>= 1) ++l;
    const int m = 1 + (1 << (16 + l)) / d;
    tables[0][i] = (int16_t)rpot(zbin_factor * d, 7); /* y_zbin  */
    tables[1][i] = (int16_t)((round_factor * d) >> 7); /* y_round */
    tables[2][i] = (int16_t)(m - (1 << 16));           /* y_quant */
    tables[3][i] = (int16_t)(1 << (16 - l));           /* y_quant_shift */
    tables[4][i] = (int16_t)d;                         /* y_dequant_QTX */
  }
}

/* ------------------------------------------------------------------ scans */

int orc_get_scan(int tx_size, int tx_type, int16_t *scan, int16_t *iscan) {
  /* av1_scan_orders (scan.c:1666-): 2-D types use the zig-zag, V_* the row scan, H_* the column
   * scan; 64-point dimensions use the 32-point table.  Index = c*h + r (transposed layout). */
  int w = orc_tx_wide[tx_size], h = orc_tx_high[tx_size];
  if (w > 32) w = 32;
  if (h > 32) h = 32;
  const int n = w * h;
  int k = 0;
  if (tx_type >= ORC_V_DCT && (tx_type & 1)) { /* H_DCT, H_ADST, H_FLIPADST: "mcol" = identity */
    for (int i = 0; i < n; ++i) scan[k++] = (int16_t)i;
  } else if (tx_type >= ORC_V_DCT) { /* V_*: "mrow" */
    for (int r = 0; r < h; ++r)
      for (int c = 0; c < w; ++c) scan[k++] = (int16_t)(c * h + r);
  } else {
    for (int d = 0; d < w + h - 1; ++d) {
      /* walk the anti-diagonal r + c = d; wide blocks always upwards, tall always downwards,
       * square blocks alternate starting upwards on even d */
      const int up = (w > h) || (w == h && (d & 1) == 0);
      for (int t = 0; t < h; ++t) {
        const int r = up ? h - 1 - t : t, c = d - r;
        if (c < 0 || c >= w) continue;
        scan[k++] = (int16_t)(c * h + r);
      }
    }
  }
  if (iscan)
    for (int i = 0; i < n; ++i) iscan[scan[i]] = (int16_t)i;
  return n;
}

/* aom_quantize_b_adaptive_helper_c / aom_highbd_quantize_b_adaptive_helper_c (aom_dsp/quantize.c:16-105,173-258),
 * qm_ptr == iqm_ptr == NULL; EOB_FACTOR 325, SKIP_EOB_FACTOR_ADJUST 200 (aom_dsp/quantize.h:23-24).  Literal:
 * backward pre-scan over a dead zone widened by dequant * 325 / 128, forward quantisation of what is left, and the
 * "single +-1 coefficient" kill with the zone widened by dequant * 525 / 128.  Pinned like quantize_b (file header). */
/* (with qm / iqm non-NULL: the same functions' matrix branches -- every `wt` / `iwt` below per coefficient; pinned by ref_eval_qm_adaptive.npz) */
void orc_quantize_b_adaptive_qm(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                                const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                                const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale, int highbd, const uint8_t *qm,
                                const uint8_t *iqm) {
  const int zb[2] = { rpot(zbin[0], log_scale), rpot(zbin[1], log_scale) };
  const int nzb[2] = { -zb[0], -zb[1] };
#define wt (qm ? (int)qm[rc] : (1 << QM_BITS))
  int non_zero_count = (int)n, eob = -1, first = -1;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  int prescan_add[2];
  for (int i = 0; i < 2; ++i) prescan_add[i] = rpot(dequant[i] * 325, 7);
  for (int i = (int)n - 1; i >= 0; --i) {
    const int rc = scan[i], ac = (rc != 0);
    const int c = coeff[rc] * wt;
    if (c < (zb[ac] * (1 << QM_BITS) + prescan_add[ac]) && c > (nzb[ac] * (1 << QM_BITS) - prescan_add[ac]))
      non_zero_count--;
    else
      break;
  }
  for (int i = 0; i < non_zero_count; ++i) {
    const int rc = scan[i], ac = (rc != 0);
    const int c = coeff[rc];
    const int sign = c >> 31;
    const int a = (c ^ sign) - sign;
    if (a * wt < (zb[ac] << QM_BITS)) continue;
    int q;
    if (!highbd) {
      int64_t t = a + rpot(round[ac], log_scale);
      if (t > INT16_MAX) t = INT16_MAX;
      if (t < INT16_MIN) t = INT16_MIN;
      t *= wt;
      q = (int)(((((t * quant[ac]) >> 16) + t) * quant_shift[ac]) >> (16 - log_scale + QM_BITS));
    } else {
      const int64_t t1 = a + rpot(round[ac], log_scale);
      const int64_t tw = t1 * wt;
      const int64_t t2 = ((tw * quant[ac]) >> 16) + tw;
      q = (int)((t2 * quant_shift[ac]) >> (16 - log_scale + QM_BITS));
    }
    qcoeff[rc] = (q ^ sign) - sign;
    const int dq = (dequant[ac] * (iqm ? (int)iqm[rc] : (1 << QM_BITS)) + (1 << (QM_BITS - 1))) >> QM_BITS;
    const int adq = (int)((uint32_t)q * (uint32_t)dq) >> log_scale;
    dqcoeff[rc] = (adq ^ sign) - sign;
    if (q) {
      eob = i;
      if (first == -1) first = i;
    }
  }
  if (eob >= 0 && first == eob) {
    const int rc = scan[eob], ac = (rc != 0);
    if (qcoeff[rc] == 1 || qcoeff[rc] == -1) {
      const int c = coeff[rc] * wt;
      const int add = rpot(dequant[ac] * (325 + 200), 7);
      if (c < (zb[ac] * (1 << QM_BITS) + add) && c > (nzb[ac] * (1 << QM_BITS) - add)) {
        qcoeff[rc] = 0;
        dqcoeff[rc] = 0;
        eob = -1;
      }
    }
  }
  *eob_out = (uint16_t)(eob + 1);
}
#undef wt
void orc_quantize_b_adaptive(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                             const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                             const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale, int highbd) {
  orc_quantize_b_adaptive_qm(coeff, n, zbin, round, quant, quant_shift, qcoeff, dqcoeff, dequant, eob_out, scan, log_scale, highbd, NULL, NULL);
}

/* av1_block_error_c / av1_highbd_block_error_c (av1/encoder/rdopt.c:635-682): transform-domain distortion and the energy
 * of the unquantised coefficients.  The low-bd form multiplies `int` operands (diff * diff, coeff * coeff): 32-bit
 * products, reproduced with unsigned wrap-around + sign extension (what the compiled reference does on this target).
 * bd = 0 selects the low-bd form; 8 / 10 / 12 the highbd form with its 2 * (bd - 8)-bit rounding. */
int64_t orc_block_error(const int32_t *coeff, const int32_t *dqcoeff, intptr_t n, int64_t *ssz, int bd) {
  int64_t error = 0, sq = 0;
  for (intptr_t i = 0; i < n; ++i) {
    const int32_t diff = (int32_t)((uint32_t)coeff[i] - (uint32_t)dqcoeff[i]);
    if (bd == 0) {
      error += (int32_t)((uint32_t)diff * (uint32_t)diff);
      sq += (int32_t)((uint32_t)coeff[i] * (uint32_t)coeff[i]);
    } else {
      error += (int64_t)diff * diff;
      sq += (int64_t)coeff[i] * coeff[i];
    }
  }
  if (bd > 8) {
    const int shift = 2 * (bd - 8);
    error = (error + ((int64_t)1 << (shift - 1))) >> shift;
    sq = (sq + ((int64_t)1 << (shift - 1))) >> shift;
  }
  *ssz = sq;
  return error;
}

/* av1_quantize_fp_no_qmatrix (av1/encoder/av1_quantize.c:36-69, behind av1_quantize_fp{,_32x32,_64x64}_c) and
 * highbd_quantize_fp_helper_c without matrices (:181-207, behind av1_highbd_quantize_fp_c).  round / quant are the
 * caller's round_fp / quant_fp tables. */
void orc_quantize_fp(const int32_t *coeff, intptr_t n, const int16_t *round_fp, const int16_t *quant_fp, int32_t *qcoeff,
                     int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale,
                     int highbd) {
  const int rounding[2] = { (round_fp[0] + ((1 << log_scale) >> 1)) >> log_scale, (round_fp[1] + ((1 << log_scale) >> 1)) >> log_scale };
  int eob = 0;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  for (intptr_t i = 0; i < n; ++i) {
    const int rc = scan[i], ac = rc != 0;
    const int c = coeff[rc], sign = c < 0 ? -1 : 0;
    int64_t a = (int64_t)((c ^ sign) - sign);
    int q = 0;
    if ((a << (1 + log_scale)) >= dequant[ac]) {
      a += rounding[ac];
      if (!highbd && a > INT16_MAX) a = INT16_MAX;
      q = (int)((a * quant_fp[ac]) >> (16 - log_scale));
      if (q) {
        qcoeff[rc] = (q ^ sign) - sign;
        const int32_t adq = (int32_t)((uint32_t)q * (uint32_t)dequant[ac]) >> log_scale;
        dqcoeff[rc] = (adq ^ sign) - sign;
      }
    }
    if (q) eob = (int)i + 1;
  }
  *eob_out = (uint16_t)eob;
}

/* quantize_fp_helper_c / highbd_quantize_fp_helper_c WITH matrices (av1/encoder/av1_quantize.c:92-121,141-169: qm_ptr / iqm_ptr non-NULL): the
 * dead zone is dequant scaled by the weight -- a * wt >= dequant << (AOM_QM_BITS - (1 + log_scale)) --, the level (a + round) * wt * quant
 * >> (16 - log_scale + AOM_QM_BITS), the low-bit-depth form clamps a + round to int16 first; dqcoeff through the weighted dequantiser.
 * Pinned by tests/golden/ref_eval_qm_fp.npz. */
void orc_quantize_fp_qm(const int32_t *coeff, intptr_t n, const int16_t *round_fp, const int16_t *quant_fp, int32_t *qcoeff, int32_t *dqcoeff,
                        const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale, int highbd, const uint8_t *qm,
                        const uint8_t *iqm) {
  const int rounding[2] = { (round_fp[0] + ((1 << log_scale) >> 1)) >> log_scale, (round_fp[1] + ((1 << log_scale) >> 1)) >> log_scale };
  int eob = -1;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  for (intptr_t i = 0; i < n; ++i) {
    const int rc = scan[i], ac = rc != 0;
    const int wt = qm ? qm[rc] : 32, iwt = iqm ? iqm[rc] : 32;
    const int dq = (dequant[ac] * iwt + 16) >> 5;
    const int c = coeff[rc], sign = c < 0 ? -1 : 0;
    int64_t a = (int64_t)((c ^ sign) - sign);
    int q = 0;
    if (a * wt >= (dequant[ac] << (5 - (1 + log_scale)))) {
      a += rounding[ac];
      if (!highbd && a > INT16_MAX) a = INT16_MAX;
      q = (int)((a * wt * quant_fp[ac]) >> (16 - log_scale + 5));
      qcoeff[rc] = (q ^ sign) - sign;
      const int32_t adq = (int32_t)((uint32_t)q * (uint32_t)dq) >> log_scale;
      dqcoeff[rc] = (adq ^ sign) - sign;
    }
    if (q) eob = (int)i;
  }
  *eob_out = (uint16_t)(eob + 1);
}

/* av1_quantize_lp_c (av1/encoder/av1_quantize.c:212-240): the low-precision (int16 coefficients) quantiser of the non-RD mode search --
 * clamp(|c| + round) * quant >> 16, dqcoeff = qcoeff * dequant stored in int16 -- and av1_block_error_lp_c (av1/encoder/rdopt.c:650-660; the
 * products in `int`, as the compiled reference).  Pinned by tests/golden/ref_eval_quant_lp.npz. */
void orc_quantize_lp(const int16_t *coeff, intptr_t n, const int16_t *round_fp, const int16_t *quant_fp, int16_t *qcoeff, int16_t *dqcoeff,
                     const int16_t *dequant, uint16_t *eob_out, const int16_t *scan) {
  int eob = -1;
  memset(qcoeff, 0, (size_t)n * sizeof(*qcoeff));
  memset(dqcoeff, 0, (size_t)n * sizeof(*dqcoeff));
  for (intptr_t i = 0; i < n; ++i) {
    const int rc = scan[i], ac = rc != 0;
    const int c = coeff[rc], sign = c < 0 ? -1 : 0;
    const int a = (c ^ sign) - sign;
    int t = a + round_fp[ac];
    t = t > INT16_MAX ? INT16_MAX : (t < INT16_MIN ? INT16_MIN : t);
    t = (t * quant_fp[ac]) >> 16;
    qcoeff[rc] = (int16_t)((t ^ sign) - sign);
    dqcoeff[rc] = (int16_t)(qcoeff[rc] * dequant[ac]);
    if (t) eob = (int)i;
  }
  *eob_out = (uint16_t)(eob + 1);
}
int64_t orc_block_error_lp(const int16_t *coeff, const int16_t *dqcoeff, intptr_t n) {
  int64_t error = 0;
  for (intptr_t i = 0; i < n; ++i) {
    const int diff = coeff[i] - dqcoeff[i];
    error += (int32_t)((uint32_t)diff * (uint32_t)diff);
  }
  return error;
}
