/*
 * oracle/aomref_rdhelp.c -- the RD helpers of SURVEY 8(f)-3: aom_sse / aom_highbd_sse (aom_dsp/sse.c:19-53), the
 * Hadamard family aom_hadamard_{4x4,8x8,16x16,32x32}, aom_hadamard_lp_{8x8,16x16}, aom_highbd_hadamard_{8x8,16x16,32x32}
 * (aom_dsp/avg.c:110-514), aom_satd / aom_satd_lp (:517-533), av1_txb_init_levels (av1/encoder/encodetxb.c:238-254) and the wedge-mask helpers
 * av1_wedge_sse_from_residuals / av1_wedge_sign_from_residuals / av1_wedge_compute_delta_squares (av1/encoder/wedge_utils.c:52-125; pinned by
 * tests/golden/ref_eval_wedge.npz).
 *
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_rdhelp.npz (the reference's own functions,
 * interpreted where they lie).  The restatement is recursive over the block size instead of the reference's three
 * hand-unrolled levels; every intermediate keeps the reference's storage width (int16_t wrap-around where the
 * reference computes in int16_t).
 */
#include <stdlib.h>
#include <string.h>

#include "aomref.h"

int64_t orc_sse(const void *a, int a_stride, const void *b, int b_stride, int w, int h, int elem16) {
  int64_t sse = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int d = elem16 ? (int)((const uint16_t *)a)[(ptrdiff_t)y * a_stride + x] - (int)((const uint16_t *)b)[(ptrdiff_t)y * b_stride + x]
                           : (int)((const uint8_t *)a)[(ptrdiff_t)y * a_stride + x] - (int)((const uint8_t *)b)[(ptrdiff_t)y * b_stride + x];
      sse += d * d;
    }
  return sse;
}

/* hadamard_col8 (avg.c:155-183) on 8 values; WIDE keeps 32 bits (hadamard_highbd_col8_second_pass, :391-422) */
static void col8(const int32_t in[8], int32_t out[8], int wide) {
#define W16(v) (wide ? (int32_t)(v) : (int32_t)(int16_t)(v))
  const int32_t b0 = W16(in[0] + in[1]), b1 = W16(in[0] - in[1]), b2 = W16(in[2] + in[3]), b3 = W16(in[2] - in[3]);
  const int32_t b4 = W16(in[4] + in[5]), b5 = W16(in[4] - in[5]), b6 = W16(in[6] + in[7]), b7 = W16(in[6] - in[7]);
  const int32_t c0 = W16(b0 + b2), c1 = W16(b1 + b3), c2 = W16(b0 - b2), c3 = W16(b1 - b3);
  const int32_t c4 = W16(b4 + b6), c5 = W16(b5 + b7), c6 = W16(b4 - b6), c7 = W16(b5 - b7);
  out[0] = W16(c0 + c4); out[7] = W16(c1 + c5); out[3] = W16(c2 + c6); out[4] = W16(c3 + c7);
  out[2] = W16(c0 - c4); out[6] = W16(c1 - c5); out[1] = W16(c2 - c6); out[5] = W16(c3 - c7);
#undef W16
}

/* flavour 0: aom_hadamard_NxN (tran_low_t out), 1: aom_hadamard_lp_NxN (int16 out), 2: aom_highbd_hadamard_NxN */
static void had8(const int16_t *src, ptrdiff_t stride, int32_t *coeff, int flavour) {
  int32_t buf[64], buf2[64], in[8];
  for (int c = 0; c < 8; ++c) {
    for (int r = 0; r < 8; ++r) in[r] = src[r * stride + c];
    col8(in, buf + c * 8, 0);
  }
  for (int c = 0; c < 8; ++c) {
    for (int r = 0; r < 8; ++r) in[r] = buf[r * 8 + c];
    col8(in, buf2 + c * 8, flavour == 2);
  }
  if (flavour == 2) {
    memcpy(coeff, buf2, sizeof(buf2));
  } else { /* the transpose that matches the SSE2 output order (avg.c:207-212,237-243) */
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) coeff[i * 8 + j] = buf2[j * 8 + i];
  }
}

static void had_rec(const int16_t *src, ptrdiff_t stride, int32_t *coeff, int n, int flavour) {
  if (n == 8) {
    had8(src, stride, coeff, flavour);
    return;
  }
  const int half = n / 2, q = half * half, sh = n == 16 ? 1 : 2;
  for (int idx = 0; idx < 4; ++idx) had_rec(src + (idx >> 1) * half * stride + (idx & 1) * half, stride, coeff + idx * q, half, flavour);
  for (int i = 0; i < q; ++i) {
    const int32_t a0 = coeff[i], a1 = coeff[i + q], a2 = coeff[i + 2 * q], a3 = coeff[i + 3 * q];
    int32_t b0 = (a0 + a1) >> sh, b1 = (a0 - a1) >> sh, b2 = (a2 + a3) >> sh, b3 = (a2 - a3) >> sh;
    if (flavour == 1) { b0 = (int16_t)b0; b1 = (int16_t)b1; b2 = (int16_t)b2; b3 = (int16_t)b3; }
    coeff[i] = b0 + b2; coeff[i + q] = b1 + b3; coeff[i + 2 * q] = b0 - b2; coeff[i + 3 * q] = b1 - b3;
    if (flavour == 1) for (int k = 0; k < 4; ++k) coeff[i + k * q] = (int16_t)coeff[i + k * q];
  }
  if (n == 16 && flavour == 0) /* "extra shift to match AVX2 output" (avg.c:285-294) */
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 4; ++j) {
        const int32_t t = coeff[i * 16 + 4 + j];
        coeff[i * 16 + 4 + j] = coeff[i * 16 + 8 + j];
        coeff[i * 16 + 8 + j] = t;
      }
}

/* n = 4 (flavour 0 only), 8, 16, 32 (flavours 0 and 2; lp: 8 and 16).  coeff: n * n int32 (lp values are int16-ranged).
 * Returns aom_satd / aom_satd_lp of the result. */
int orc_hadamard(const int16_t *src, ptrdiff_t stride, int n, int flavour, int32_t *coeff) {
  if (n == 4) { /* aom_hadamard_4x4_c (avg.c:110-153): both passes halve after the first butterfly */
    int32_t buf[16], buf2[16];
    for (int pass = 0; pass < 2; ++pass)
      for (int c = 0; c < 4; ++c) {
        int32_t v[4];
        for (int r = 0; r < 4; ++r) v[r] = pass ? buf[r * 4 + c] : src[r * stride + c];
        const int32_t b0 = (int16_t)((v[0] + v[1]) >> 1), b1 = (int16_t)((v[0] - v[1]) >> 1);
        const int32_t b2 = (int16_t)((v[2] + v[3]) >> 1), b3 = (int16_t)((v[2] - v[3]) >> 1);
        int32_t *o = (pass ? buf2 : buf) + c * 4;
        o[0] = (int16_t)(b0 + b2); o[1] = (int16_t)(b1 + b3); o[2] = (int16_t)(b0 - b2); o[3] = (int16_t)(b1 - b3);
      }
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) coeff[i * 4 + j] = buf2[j * 4 + i];
  } else {
    had_rec(src, stride, coeff, n, flavour);
  }
  int satd = 0;
  for (int i = 0; i < n * n; ++i) satd += abs(coeff[i]);
  return satd;
}

/* av1_txb_init_levels_c (encodetxb.c:238-254): TX_PAD_HOR 4, TX_PAD_BOTTOM 4, TX_PAD_END 16 (av1/common/enums.h:191-199).
 * levels: (height + 4) * (width + 4) + 16 bytes. */
void orc_txb_init_levels(const int32_t *coeff, int width, int height, uint8_t *levels) {
  const int stride = height + 4;
  uint8_t *ls = levels;
  memset(levels + stride * width, 0, (size_t)(4 * stride + 16));
  for (int i = 0; i < width; ++i) {
    for (int j = 0; j < height; ++j) {
      const int a = abs(coeff[i * height + j]);
      *ls++ = (uint8_t)(a > 127 ? 127 : a);
    }
    for (int j = 0; j < 4; ++j) *ls++ = 0;
  }
}

/* av1_wedge_sse_from_residuals_c (av1/encoder/wedge_utils.c:52-63): sum of clamp(64 r1 + m d, int16)^2, rounded by 2 * WEDGE_WEIGHT_BITS = 12 */
uint64_t orc_wedge_sse_from_residuals(const int16_t *r1, const int16_t *d, const uint8_t *m, int n) {
  uint64_t csse = 0;
  for (int i = 0; i < n; ++i) {
    int32_t t = 64 * (int32_t)r1[i] + (int32_t)m[i] * d[i];
    t = t < -32768 ? -32768 : (t > 32767 ? 32767 : t);
    csse += (uint64_t)((int64_t)t * t);
  }
  return (csse + 2048) >> 12;
}
/* av1_wedge_sign_from_residuals_c (:96-105): sum(ds * m) > limit */
int orc_wedge_sign_from_residuals(const int16_t *ds, const uint8_t *m, int n, int64_t limit) {
  int64_t acc = 0;
  for (int i = 0; i < n; ++i) acc += (int64_t)ds[i] * m[i];
  return acc > limit;
}
/* av1_wedge_compute_delta_squares_c (:119-125): d = clamp(a^2 - b^2, int16) */
void orc_wedge_compute_delta_squares(int16_t *d, const int16_t *a, const int16_t *b, int n) {
  for (int i = 0; i < n; ++i) {
    const int32_t v = (int32_t)a[i] * a[i] - (int32_t)b[i] * b[i];
    d[i] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
  }
}

/* av1_get_nz_map_contexts_c (av1/encoder/encodetxb.c:222-267) with get_nz_mag / get_nz_map_ctx_from_stats (av1/common/txb_common.h:150-224): the
 * context of every coefficient before the end of block from the magnitudes of its causal neighbours in the padded level map.  The 2-D position
 * offsets are Nz_Map's rule (av1_nz_map_ctx_offset, txb_common.c:18-360; oracle/gen_tables.py checks the rule against all 19 tables).
 * tx_w / tx_h: the transform's own size (64 included); tx_class 0 2D, 1 HORIZ, 2 VERT.  Pinned by tests/golden/ref_eval_nzmap.npz. */
static int clip3(int v) { return v > 3 ? 3 : v; }
void orc_get_nz_map_contexts(const uint8_t *levels, const int16_t *scan, int eob, int tx_w, int tx_h, int tx_class, int8_t *coeff_contexts) {
  const int w = tx_w > 32 ? 32 : tx_w, h = tx_h > 32 ? 32 : tx_h;   /* av1_get_adjusted_tx_size */
  int bhl = 0;
  while ((1 << bhl) < h) ++bhl;
  for (int i = 0; i < eob; ++i) {
    const int pos = scan[i];
    int ctx;
    if (i == eob - 1) {
      ctx = i == 0 ? 0 : (i <= (w << bhl) / 8 ? 1 : (i <= (w << bhl) / 4 ? 2 : 3));
    } else {
      const uint8_t *lv = levels + pos + ((pos >> bhl) << 2);   /* get_padded_idx, TX_PAD_HOR_LOG2 */
      int mag = clip3(lv[(1 << bhl) + 4]) + clip3(lv[1]);
      if (tx_class == 0) mag += clip3(lv[(1 << bhl) + 4 + 1]) + clip3(lv[(2 << bhl) + (2 << 2)]) + clip3(lv[2]);
      else if (tx_class == 2) mag += clip3(lv[2]) + clip3(lv[3]) + clip3(lv[4]);
      else mag += clip3(lv[(2 << bhl) + (2 << 2)]) + clip3(lv[(3 << bhl) + (3 << 2)]) + clip3(lv[(4 << bhl) + (4 << 2)]);
      const int col = pos >> bhl, row = pos - (col << bhl);
      if ((tx_class | pos) == 0) {
        ctx = 0;
      } else {
        ctx = (mag + 1) >> 1;
        ctx = ctx > 4 ? 4 : ctx;
        if (tx_class == 0) {
          int off;
          if (tx_w < tx_h && row < 2) off = 11;
          else if (tx_w > tx_h && col < 2) off = 16;
          else if (row + col < 2) off = 1;
          else if (row + col < 4) off = 6;
          else off = 21;
          ctx += off;
        } else {
          const int k = tx_class == 1 ? col : row;   /* nz_map_ctx_offset_1d: SIG_COEF_CONTEXTS_2D (26) + 0, 5, 10, 10, .. */
          ctx += 26 + (k == 0 ? 0 : (k == 1 ? 5 : 10));
        }
      }
    }
    coeff_contexts[pos] = (int8_t)ctx;
  }
}

/* av1_cost_coeffs_txb (av1/encoder/txb_rdopt.c:450-544,603-622): the rate of a transform block's quantised coefficients under the level-map coder's
 * cost tables -- everything the function adds except get_tx_type_cost (a table look-up on the block's mode, added by the caller).  `costs`: the
 * LV_MAP_COEFF_COST of (transform-size context, plane type) as its 944 ints in declaration order (av1/encoder/block.h:173-195: txb_skip_cost[13][2],
 * base_eob_cost[4][3], base_cost[42][8], eob_extra_cost[9][2], dc_sign_cost[3][2], lps_cost[21][26]) followed by LV_MAP_EOB_COST.eob_cost[2][11] of
 * (eob_multi_size, plane type).  Pinned by tests/golden/ref_eval_txb_cost.npz. */
enum { OFF_SKIP = 0, OFF_BASE_EOB = 26, OFF_BASE = 38, OFF_EOB_EXTRA = 374, OFF_DC_SIGN = 392, OFF_LPS = 398, OFF_EOB = 944 };
static int golomb_cost(int abs_qc) {
  if (abs_qc >= 1 + 2 + 12) {   /* NUM_BASE_LEVELS, COEFF_BASE_RANGE */
    const int r = abs_qc - 12 - 2;
    int length = 0;
    while ((r >> length) > 0) ++length;   /* get_msb(r) + 1 */
    return (2 * length - 1) * 512;        /* av1_cost_literal: 1 << AV1_PROB_COST_SHIFT */
  }
  return 0;
}
static int br_cost(int level, const int *lps) {
  const int base_range = level - 1 - 2 < 12 ? level - 1 - 2 : 12;
  return lps[base_range] + golomb_cost(level);
}
static int br_ctx(const uint8_t *levels, int c, int bhl, int tx_class) {
  const int col = c >> bhl, row = c - (col << bhl), stride = (1 << bhl) + 4, pos = col * stride + row;
  int mag = levels[pos + 1] + levels[pos + stride];
  if (tx_class == 0) {
    mag += levels[pos + stride + 1];
    mag = (mag + 1) >> 1 < 6 ? (mag + 1) >> 1 : 6;
    if (c == 0) return mag;
    if (row < 2 && col < 2) return mag + 7;
  } else if (tx_class == 1) {
    mag += levels[pos + (stride << 1)];
    mag = (mag + 1) >> 1 < 6 ? (mag + 1) >> 1 : 6;
    if (c == 0) return mag;
    if (col == 0) return mag + 7;
  } else {
    mag += levels[pos + 2];
    mag = (mag + 1) >> 1 < 6 ? (mag + 1) >> 1 : 6;
    if (c == 0) return mag;
    if (row == 0) return mag + 7;
  }
  return mag + 14;
}
int orc_cost_coeffs_txb(const int32_t *qcoeff, int eob, int tx_w, int tx_h, int tx_class, const int16_t *scan, int txb_skip_ctx, int dc_sign_ctx,
                        const int32_t *costs) {
  if (eob == 0) return costs[OFF_SKIP + txb_skip_ctx * 2 + 1];
  const int w = tx_w > 32 ? 32 : tx_w, h = tx_h > 32 ? 32 : tx_h;
  int bhl = 0;
  while ((1 << bhl) < h) ++bhl;
  uint8_t levels[(32 + 4) * (32 + 4) + 16];
  int8_t ctxs[1024];
  orc_txb_init_levels(qcoeff, w, h, levels);
  orc_get_nz_map_contexts(levels, scan, eob, tx_w, tx_h, tx_class, ctxs);
  int cost = costs[OFF_SKIP + txb_skip_ctx * 2 + 0];
  {   /* get_eob_cost (txb_rdopt_utils.h:66-83) with av1_get_eob_pos_token (encodetxb.c:100-130) */
    static const int group_start[12] = { 0, 1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 513 }, offset_bits[12] = { 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9 };
    int t = 0;
    while (t < 11 && eob >= group_start[t + 1]) ++t;   /* eob_to_pos_small / _large: the group whose start is the last one <= eob */
    const int extra = eob - group_start[t];
    cost += costs[OFF_EOB + (tx_class == 0 ? 0 : 1) * 11 + t - 1];
    if (offset_bits[t] > 0) {
      const int bit = (extra >> (offset_bits[t] - 1)) & 1;
      cost += costs[OFF_EOB_EXTRA + (t - 3) * 2 + bit];
      if (offset_bits[t] > 1) cost += (offset_bits[t] - 1) * 512;
    }
  }
  int c = eob - 1;
  {
    const int pos = scan[c], v = qcoeff[pos], level = abs(v), coeff_ctx = ctxs[pos];
    cost += costs[OFF_BASE_EOB + coeff_ctx * 3 + (level < 3 ? level : 3) - 1];
    if (v) {
      if (level > 2) {
        const int col = pos >> bhl, row = pos - (col << bhl);
        const int ctx = pos == 0 ? 0 : (((tx_class == 0 && row < 2 && col < 2) || (tx_class == 1 && col == 0) || (tx_class == 2 && row == 0)) ? 7 : 14);
        cost += br_cost(level, costs + OFF_LPS + ctx * 26);
      }
      if (c) cost += 512;
      else return cost + costs[OFF_DC_SIGN + dc_sign_ctx * 2 + (v < 0)];
    }
  }
  for (c = eob - 2; c >= 1; --c) {
    const int pos = scan[c], v = qcoeff[pos], level = abs(v);
    cost += costs[OFF_BASE + ctxs[pos] * 8 + (level < 3 ? level : 3)];
    if (v) {
      cost += 512;
      if (level > 2) cost += br_cost(level, costs + OFF_LPS + br_ctx(levels, pos, bhl, tx_class) * 26);
    }
  }
  {
    const int pos = scan[0], v = qcoeff[pos], level = abs(v);
    cost += costs[OFF_BASE + ctxs[pos] * 8 + (level < 3 ? level : 3)];
    if (v) {
      cost += costs[OFF_DC_SIGN + dc_sign_ctx * 2 + (v < 0)];
      if (level > 2) cost += br_cost(level, costs + OFF_LPS + br_ctx(levels, pos, bhl, tx_class) * 26);
    }
  }
  return cost;
}

/* av1_cost_coeffs_txb_laplacian with adjust_eob == 0 (av1/encoder/txb_rdopt.c:546-601,624-668): the transform-type search's cheap rate -- the skip and
 * end-of-block terms of the exact form, then per coefficient a table entry by its level (the last one: (|q| - 1) << 11) and a constant per position.
 * Without get_tx_type_cost, like orc_cost_coeffs_txb; the same `costs` layout. */
#include "aomref_txb_cost.inc"
int orc_cost_coeffs_txb_laplacian(const int32_t *qcoeff, int eob, int tx_class, const int16_t *scan, int txb_skip_ctx, const int32_t *costs) {
  static const int lut[15] = AOMHIP_TXB_COST_LUT;
  if (eob == 0) return costs[OFF_SKIP + txb_skip_ctx * 2 + 1];
  int cost = costs[OFF_SKIP + txb_skip_ctx * 2 + 0];
  {
    static const int group_start[12] = { 0, 1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 513 }, offset_bits[12] = { 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9 };
    int t = 0;
    while (t < 11 && eob >= group_start[t + 1]) ++t;
    const int extra = eob - group_start[t];
    cost += costs[OFF_EOB + (tx_class == 0 ? 0 : 1) * 11 + t - 1];
    if (offset_bits[t] > 0) {
      cost += costs[OFF_EOB_EXTRA + (t - 3) * 2 + ((extra >> (offset_bits[t] - 1)) & 1)];
      if (offset_bits[t] > 1) cost += (offset_bits[t] - 1) * 512;
    }
  }
  cost += (abs(qcoeff[scan[eob - 1]]) - 1) * 2048;   /* << (AV1_PROB_COST_SHIFT + 2) */
  for (int c = eob - 2; c >= 0; --c) {
    const int v = abs(qcoeff[scan[c]]);
    cost += lut[v < 14 ? v : 14];
  }
  return cost + (512 + 739) * (eob - 1);   /* const_term + loge_par */
}

/* av1_get_txb_entropy_context (av1/encoder/encodetxb.c:451-467): what a coded transform block leaves in the above / left entropy contexts -- the sum of
 * its levels saturated at COEFF_CONTEXT_MASK (7), the DC coefficient's sign in the bits above (set_dc_sign, txb_common.h:274-279). */
int orc_get_txb_entropy_context(const int32_t *qcoeff, const int16_t *scan, int eob) {
  if (eob == 0) return 0;
  int cul = 0;
  for (int c = 0; c < eob; ++c) {
    cul += abs(qcoeff[scan[c]]);
    if (cul > 7) break;
  }
  cul = cul < 7 ? cul : 7;
  if (qcoeff[0] < 0) cul |= 1 << 3;
  else if (qcoeff[0] > 0) cul += 2 << 3;
  return cul & 255;
}

/* aom_sum_squares_2d_i16_c / aom_sum_sse_2d_i16_c (aom_dsp/sum_squares.c:16-30,75-90): a residual block's sum of squares (and sum) -- the transform
 * search's skip prediction and per-pixel statistics.  *sum is ADDED to, like the reference's. */
uint64_t orc_sum_sse_2d_i16(const int16_t *src, int src_stride, int width, int height, int *sum) {
  uint64_t ss = 0;
  for (int r = 0; r < height; ++r)
    for (int c = 0; c < width; ++c) {
      const int v = src[(ptrdiff_t)r * src_stride + c];
      ss += (uint64_t)(int64_t)(v * v);
      if (sum) *sum += v;
    }
  return ss;
}

