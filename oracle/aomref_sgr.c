/*
 * oracle/aomref_sgr.c -- the self-guided restoration filter: av1_selfguided_restoration_c (av1/common/restoration.c:871-915) with
 * calculate_intermediate_result (:672-764), selfguided_restoration_fast_internal (r = 2, every other row, :766-823) and
 * selfguided_restoration_internal (r = 1, :825-869); AV1 spec 7.17.3.
 *
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_sgr.npz (the reference's own function, interpreted where it lies).
 * The reference builds running box sums over the unit extended by 3 pixels and truncates them at that extension's edge; the values it then
 * USES sit at most one pixel outside the unit, whose windows (r <= 2) stay inside the extension -- so every used box sum is the full
 * (2 r + 1)^2 window, which is what this restatement computes directly, for a unit of any size (the reference's buffers hold 64 x 64).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

#include "aomref.h"
#include "aomref_sgr.inc"

static const int k_sgr[16][4] = AOMHIP_SGR_PARAMS;
static const int32_t k_x_by_xplus1[256] = AOMHIP_X_BY_XPLUS1;
static const int32_t k_one_by_x[25] = AOMHIP_ONE_BY_X;

#define PXS(p, i) (elem16 ? (int)((const uint16_t *)(p))[i] : (int)((const uint8_t *)(p))[i])

/* A[] and B[] of calculate_intermediate_result at (i, j) of the unit, i, j in [-1, n] */
static void ab_at(const void *dgd, int elem16, int stride, int i, int j, int r, int s, int bit_depth, int32_t *A, int32_t *B) {
  uint32_t sum = 0, sq = 0;
  for (int y = -r; y <= r; ++y)
    for (int x = -r; x <= r; ++x) {
      const uint32_t v = (uint32_t)PXS(dgd, (ptrdiff_t)(i + y) * stride + (j + x));
      sum += v; sq += v * v;
    }
  const int n = (2 * r + 1) * (2 * r + 1);
  const int sh = bit_depth - 8;
  const uint32_t a = sh ? (sq + ((1u << (2 * sh)) >> 1)) >> (2 * sh) : sq, b = sh ? (sum + ((1u << sh) >> 1)) >> sh : sum;
  const uint32_t p = (a * n < b * b) ? 0 : a * n - b * b;
  const uint32_t z = (p * (uint32_t)s + (1u << 19)) >> 20;   /* SGRPROJ_MTABLE_BITS */
  *A = k_x_by_xplus1[z < 255 ? z : 255];
  *B = (int32_t)(((uint32_t)(256 - *A) * sum * (uint32_t)k_one_by_x[n - 1] + (1u << 11)) >> 12);   /* SGRPROJ_SGR, SGRPROJ_RECIP_BITS */
}

void orc_selfguided_restoration(const void *dgd, int elem16, int width, int height, int stride, int32_t *flt0, int32_t *flt1, int flt_stride,
                                int sgr_params_idx, int bit_depth) {
  const int *prm = k_sgr[sgr_params_idx];
  const int W2 = width + 2;
  int32_t *A = (int32_t *)malloc(sizeof(int32_t) * (size_t)W2 * (height + 2) * 2), *B = A + (size_t)W2 * (height + 2);
  for (int pass = 0; pass < 2; ++pass) {
    const int r = prm[pass], s = prm[2 + pass];
    int32_t *dst = pass ? flt1 : flt0;
    if (r <= 0) continue;
    const int step = pass == 0 ? 2 : 1;   /* the r[0] filter computes the odd rows' A, B only */
    for (int i = -1; i < height + 1; i += step)
      for (int j = -1; j < width + 1; ++j) ab_at(dgd, elem16, stride, i, j, r, s, bit_depth, &A[(i + 1) * W2 + j + 1], &B[(i + 1) * W2 + j + 1]);
#define AT(M, i, j) M[((i) + 1) * W2 + (j) + 1]
    for (int i = 0; i < height; ++i)
      for (int j = 0; j < width; ++j) {
        int32_t a, b;
        int nb;
        if (pass == 0) {
          if (!(i & 1)) {
            nb = 5;
            a = (AT(A, i - 1, j) + AT(A, i + 1, j)) * 6 + (AT(A, i - 1, j - 1) + AT(A, i + 1, j - 1) + AT(A, i - 1, j + 1) + AT(A, i + 1, j + 1)) * 5;
            b = (AT(B, i - 1, j) + AT(B, i + 1, j)) * 6 + (AT(B, i - 1, j - 1) + AT(B, i + 1, j - 1) + AT(B, i - 1, j + 1) + AT(B, i + 1, j + 1)) * 5;
          } else {
            nb = 4;
            a = AT(A, i, j) * 6 + (AT(A, i, j - 1) + AT(A, i, j + 1)) * 5;
            b = AT(B, i, j) * 6 + (AT(B, i, j - 1) + AT(B, i, j + 1)) * 5;
          }
        } else {
          nb = 5;
          a = (AT(A, i, j) + AT(A, i, j - 1) + AT(A, i, j + 1) + AT(A, i - 1, j) + AT(A, i + 1, j)) * 4 +
              (AT(A, i - 1, j - 1) + AT(A, i + 1, j - 1) + AT(A, i - 1, j + 1) + AT(A, i + 1, j + 1)) * 3;
          b = (AT(B, i, j) + AT(B, i, j - 1) + AT(B, i, j + 1) + AT(B, i - 1, j) + AT(B, i + 1, j)) * 4 +
              (AT(B, i - 1, j - 1) + AT(B, i + 1, j - 1) + AT(B, i - 1, j + 1) + AT(B, i + 1, j + 1)) * 3;
        }
        const int32_t v = a * PXS(dgd, (ptrdiff_t)i * stride + j) + b;
        const int sh = 8 + nb - 4;   /* SGRPROJ_SGR_BITS + nb - SGRPROJ_RST_BITS */
        dst[(ptrdiff_t)i * flt_stride + j] = (v + ((1 << sh) >> 1)) >> sh;
      }
#undef AT
  }
  free(A);
}

/* av1_apply_selfguided_restoration_c (av1/common/restoration.c:917-956): the filter, then per pixel u = dat << SGRPROJ_RST_BITS,
 * v = (u << SGRPROJ_PRJ_BITS) + xq0 (flt0 - u) + xq1 (flt1 - u) over the radii in use with (xq0, xq1) = av1_decode_xq(xqd) (:631-643),
 * w = (int16_t)ROUND_POWER_OF_TWO(v, 11), clipped to the bit depth.  Pinned by tests/golden/ref_eval_sgr_apply.npz. */
void orc_apply_selfguided_restoration(const void *dat, int elem16, int width, int height, int stride, int eps, const int *xqd, void *dst, int dst_stride,
                                      int bit_depth) {
  int32_t *flt0 = (int32_t *)malloc(sizeof(int32_t) * (size_t)width * height * 2), *flt1 = flt0 + (size_t)width * height;
  orc_selfguided_restoration(dat, elem16, width, height, stride, flt0, flt1, width, eps, bit_depth);
  const int r0 = k_sgr[eps][0], r1 = k_sgr[eps][1];
  int xq[2];
  if (r0 == 0) { xq[0] = 0; xq[1] = 128 - xqd[1]; }
  else if (r1 == 0) { xq[0] = xqd[0]; xq[1] = 0; }
  else { xq[0] = xqd[0]; xq[1] = 128 - xq[0] - xqd[1]; }
  const int mx = (1 << bit_depth) - 1;
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j) {
      const int k = i * width + j;
      const int32_t u = (int32_t)PXS(dat, (ptrdiff_t)i * stride + j) << 4;
      int32_t v = u << 7;
      if (r0 > 0) v += xq[0] * (flt0[k] - u);
      if (r1 > 0) v += xq[1] * (flt1[k] - u);
      const int16_t w = (int16_t)((v + (1 << 10)) >> 11);
      const int o = w < 0 ? 0 : (w > mx ? mx : w);
      if (elem16) ((uint16_t *)dst)[(ptrdiff_t)i * dst_stride + j] = (uint16_t)o;
      else ((uint8_t *)dst)[(ptrdiff_t)i * dst_stride + j] = (uint8_t)o;
    }
  free(flt0);
}

/* av1_wiener_convolve_add_src_c / av1_highbd_wiener_convolve_add_src_c (av1/common/convolve.c:1093-1257) as wiener_filter_stripe calls them: steps
 * 16 (no scaling), the unit's 7-tap filters stored as 8 taps with the centre tap reduced by 128 -- the functions add the source back
 * ("add_src": + src << FILTER_BITS).  round_0 / round_1 as get_conv_params_wiener(bd) (convolve.h:102-117): 3 / 11, at 12 bits 5 / 9.  The
 * horizontal pass clamps to [0, WIENER_CLAMP_LIMIT), the vertical one removes the offset the horizontal one added and clips to the pixel range.
 * src points at the block's first pixel and is read 3 pixels beyond it on every side (4 to the right / below: tap 7, which is 0).
 * Pinned by tests/golden/ref_eval_lr_apply.npz (the two static passes interpreted where they lie). */
void orc_wiener_convolve_add_src(const void *src, int elem16, int src_stride, void *dst, int dst_stride, const int16_t *filter_x, const int16_t *filter_y, int w,
                                 int h, int bd) {
  if (!elem16) bd = 8;
  const int round_0 = bd == 12 ? 5 : 3, round_1 = 14 - round_0;
  const int limit = 1 << (bd + 1 + 7 - round_0);
  const int mx = (1 << bd) - 1;
  uint16_t *temp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * (h + 8));
  for (int y = 0; y < h + 8; ++y)       /* temp row y = source row y - 3 */
    for (int x = 0; x < w; ++x) {
      int sum = (PXS(src, (ptrdiff_t)(y - 3) * src_stride + x) << 7) + (1 << (bd + 7 - 1));
      for (int k = 0; k < 8; ++k) sum += PXS(src, (ptrdiff_t)(y - 3) * src_stride + x - 3 + k) * filter_x[k];
      int v = (sum + ((1 << round_0) >> 1)) >> round_0;
      temp[y * w + x] = (uint16_t)(v < 0 ? 0 : (v > limit - 1 ? limit - 1 : v));
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int sum = ((int)temp[(y + 3) * w + x] << 7) - (1 << (bd + round_1 - 1));
      for (int k = 0; k < 8; ++k) sum += (int)temp[(y + k) * w + x] * filter_y[k];
      const int v = (sum + ((1 << round_1) >> 1)) >> round_1;
      const int o = v < 0 ? 0 : (v > mx ? mx : v);
      if (elem16) ((uint16_t *)dst)[(ptrdiff_t)y * dst_stride + x] = (uint16_t)o;
      else ((uint8_t *)dst)[(ptrdiff_t)y * dst_stride + x] = (uint8_t)o;
    }
  free(temp);
}
