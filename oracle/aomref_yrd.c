/*
 * oracle/aomref_yrd.c -- av1_estimate_txfm_yrd and the RD-based second-MV choice of av1_single_motion_search.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_yrd.npz (the reference's own av1_estimate_txfm_yrd and the
 * branch of av1_single_motion_search interpreted where they lie, tests/golden/gen_ref_eval_yrd.py).
 *
 *   orc_estimate_txfm_yrd     av1_estimate_txfm_yrd (av1/encoder/tx_search.c:3016-3139) with ref_best_rd = INT64_MAX for an INTER block wholly
 *                             inside the frame: the block's luma residual through ONE transform size (max_txsize_rect_lookup[bsize], DCT_DCT,
 *                             AV1_XFORM_QUANT_B without matrices), per transform block get_txb_ctx (av1/common/txb_common.h:251-460) on the running
 *                             above / left contexts -> av1_cost_coeffs_txb (txb_rdopt.c:604-623) -> dist_block_tx_domain (tx_search.c:1077-1113)
 *                             -> av1_set_txb_context (the value av1_quant left in txb_entropy_ctx, encodemb.c:333-340), then the function's tail
 *                             (skip / no-skip header rates, the forced-skip check).
 *   orc_second_mv_rd_choice   motion_search_facade.c:378-425 with disable_second_mv == 0: rd = RDCOST(rdmult, mv_rate + stats.rate, stats.dist) of
 *                             the two candidates, the second one kept when its rd is SMALLER.
 */
#include <limits.h>
#include <string.h>

#include "aomref.h"

int64_t orc_block_error(const int32_t *coeff, const int32_t *dqcoeff, intptr_t n, int64_t *ssz, int bd);

static int64_t rdcost(int rdmult, int64_t rate, int64_t dist) { return ((rate * rdmult + 256) >> 9) + dist * 128; }   /* RDCOST, rd.h:31-33 */

static int tx_size_of(int w, int h) {
  static const int tw[19] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64 };
  static const int th[19] = { 4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16 };
  for (int i = 0; i < 19; ++i)
    if (tw[i] == w && th[i] == h) return i;
  return -1;
}

/* get_txb_ctx for plane 0 (txb_common.h:251-460; the specialised and the general form compute the same) */
static void txb_ctx_luma(const uint8_t *a, const uint8_t *l, int txw_unit, int txh_unit, int whole_block, int *skip_ctx, int *dc_sign_ctx) {
  static const int8_t signs[3] = { 0, -1, 1 };
  static const uint8_t skip_contexts[5][5] = { { 1, 2, 2, 2, 3 }, { 2, 4, 4, 4, 5 }, { 2, 4, 4, 4, 5 }, { 2, 4, 4, 4, 5 }, { 3, 5, 5, 5, 6 } };
  int dc_sign = 0, top = 0, left = 0;
  for (int k = 0; k < txw_unit; ++k) { dc_sign += signs[a[k] >> 3]; top |= a[k]; }       /* COEFF_CONTEXT_BITS 3 */
  for (int k = 0; k < txh_unit; ++k) { dc_sign += signs[l[k] >> 3]; left |= l[k]; }
  *dc_sign_ctx = dc_sign < 0 ? 1 : (dc_sign > 0 ? 2 : 0);                                  /* dc_sign_contexts[] */
  top &= 7; left &= 7;                                                                      /* COEFF_CONTEXT_MASK */
  if (top > 4) top = 4;
  if (left > 4) left = 4;
  *skip_ctx = whole_block ? 0 : skip_contexts[top][left];
}

/* out: [0] rate, [1] skip_txfm, [2] dist, [3] sse (int64 each); returns rd.  above / left: bw / 4 and bh / 4 entries (read only: the reference
 * works on copies too).  q: zbin, round, quant, quant_shift, dequant x (DC, AC).  costs: the 966 ints of orc_cost_coeffs_txb for this transform
 * size's context and plane type 0. */
int64_t orc_estimate_txfm_yrd(const int16_t *residual, int stride, int bw, int bh, int bd, int is_hbd, const int16_t q[5][2], const uint8_t *above,
                              const uint8_t *left, const int32_t *costs, int tx_type_rate, int tx_size_rate, int no_skip_txfm_rate,
                              int skip_txfm_rate, int rdmult, int lossless, int64_t *out) {
  const int txw = bw > 64 ? 64 : bw, txh = bh > 64 ? 64 : bh;
  const int tx_size = tx_size_of(txw, txh);
  const int pels = txw * txh, scale = (pels > 256) + (pels > 1024);                        /* av1_get_tx_scale */
  const int n = (txw == 64 || txh == 64) ? (txw == 64 && txh == 64 ? 1024 : (txw * txh == 2048 ? 1024 : 512)) : pels;   /* av1_get_max_eob */
  const int shift = (1 - scale) * 2;                                                        /* (MAX_TX_SCALE - scale) * 2 */
  uint8_t ta[32], tl[32];
  memcpy(ta, above, (size_t)(bw / 4)); memcpy(tl, left, (size_t)(bh / 4));
  int16_t scan[1024], iscan[1024];
  orc_get_scan(tx_size, 0, scan, iscan);
  int rate = 0, skip = 1;
  int64_t dist = 0, sse = 0;
  static __thread int32_t coeff[4096], qc[4096], dq[4096];
  for (int by = 0; by < bh; by += txh)
    for (int bx = 0; bx < bw; bx += txw) {
      uint8_t *a = ta + bx / 4, *l = tl + by / 4;
      int skip_ctx, dc_ctx;
      txb_ctx_luma(a, l, txw / 4, txh / 4, bw == txw && bh == txh, &skip_ctx, &dc_ctx);
      orc_fwd_txfm2d(residual + (ptrdiff_t)by * stride + bx, coeff, stride, tx_size, 0, bd);
      uint16_t eob = 0;
      (is_hbd ? orc_highbd_quantize_b : orc_quantize_b)(coeff, n, q[0], q[1], q[2], q[3], qc, dq, q[4], &eob, scan, iscan, scale);
      int r = orc_cost_coeffs_txb(qc, eob, txw, txh, 0, scan, skip_ctx, dc_ctx, costs);
      if (eob) r += tx_type_rate;                                                            /* get_tx_type_cost sits behind the eob == 0 return */
      int64_t ssz, err = orc_block_error(coeff, dq, n, &ssz, is_hbd ? bd : 0);
      err = shift < 0 ? err << -shift : err >> shift;                                        /* RIGHT_SIGNED_SHIFT */
      ssz = shift < 0 ? ssz << -shift : ssz >> shift;
      skip &= !eob;
      rate += r; dist += err; sse += ssz;                                                    /* av1_merge_rd_stats */
      const uint8_t ectx = (uint8_t)orc_get_txb_entropy_context(qc, scan, eob);
      memset(a, ectx, (size_t)(txw / 4)); memset(l, ectx, (size_t)(txh / 4));               /* av1_set_txb_context */
    }
  int64_t rd;
  if (skip) {
    rd = rdcost(rdmult, skip_txfm_rate, sse);
  } else {
    rd = rdcost(rdmult, (int64_t)rate + no_skip_txfm_rate + tx_size_rate, dist);
    rate += tx_size_rate;
  }
  if (!skip && !lossless) {
    const int64_t t = rdcost(rdmult, skip_txfm_rate, sse);
    if (t <= rd) { rd = t; rate = 0; dist = sse; skip = 1; }
  }
  out[0] = rate; out[1] = skip; out[2] = dist; out[3] = sse;
  return rd;
}

/* 1 when the second candidate replaces the first (tmp_rd < rd) */
int orc_second_mv_rd_choice(int rdmult, int mv_rate0, int rate0, int64_t dist0, int mv_rate1, int rate1, int64_t dist1) {
  return rdcost(rdmult, (int64_t)rate1 + mv_rate1, dist1) < rdcost(rdmult, (int64_t)mv_rate0 + rate0, dist0);
}
