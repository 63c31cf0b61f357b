/*
 * oracle/aomref_convolve.c -- single-reference, unscaled inter prediction of one block: the sub-pel interpolation
 * behind av1_enc_build_inter_predictor (av1/encoder/reconinter_enc.c:47-51 -> build_one_inter_predictor ->
 * av1_make_inter_predictor -> [highbd_]inter_predictor, av1/common/reconinter.h:252-296 ->
 * av1_[highbd_]convolve_2d_facade, av1/common/convolve.c:495-567,982-1058).
 *
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_convolve.npz (the reference's own
 * av1_[highbd_]convolve_{2d,x,y}_sr_c and the facade, interpreted where they lie).
 */
#include <stdlib.h>

#include "aomref.h"

#define RPOT(v, n) (((v) + ((1 << (n)) >> 1)) >> (n))

static const int16_t k_interp[6][16][8] = {
#include "aomref_interp.inc"
};

/* av1_get_interp_filter_params_with_block_size (av1/common/filter.h:247-253): interp_filter 0 EIGHTTAP_REGULAR,
 * 1 EIGHTTAP_SMOOTH, 2 MULTITAP_SHARP, 3 BILINEAR; a dimension <= 4 switches to the 4-tap sets (sharp -> regular). */
static const int16_t *kernel_of(int interp_filter, int dim, int subpel_qn) {
  int set = interp_filter;
  if (dim <= 4) set = interp_filter == 1 ? 5 : interp_filter == 3 ? 3 : 4;
  return k_interp[set][subpel_qn & 15];
}

static int px(const void *p, int elem16, ptrdiff_t i) { return elem16 ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i]; }
static void put(void *p, int elem16, ptrdiff_t i, int v, int bd) {
  const int mx = (1 << (elem16 ? bd : 8)) - 1;
  v = v < 0 ? 0 : v > mx ? mx : v; /* clip_pixel / clip_pixel_highbd */
  if (elem16) ((uint16_t *)p)[i] = (uint16_t)v; else ((uint8_t *)p)[i] = (uint8_t)v;
}

/* convolve_2d_facade_single / highbd_convolve_2d_facade_single with get_conv_params(0, plane, bd)
 * (av1/common/convolve.h:63-100): round_0 = 3 (5 for 12-bit), round_1 = 14 - round_0.
 * src points at the block's integer position; dst stride in elements. */
void orc_convolve_sr(const void *src, int src_stride, void *dst, int dst_stride, int w, int h, int filter_x, int filter_y,
                     int subpel_x_qn, int subpel_y_qn, int elem16, int bd) {
  const int tbd = elem16 ? bd : 8;
  int round_0 = 3, round_1 = 11;
  if (elem16 && bd + 7 - round_0 + 2 > 16) { /* intbufrange > 16 (12-bit) */
    const int extra = bd + 7 - round_0 + 2 - 16;
    round_0 += extra;
    round_1 -= extra;
  }
  const int16_t *fx = kernel_of(filter_x, w, subpel_x_qn), *fy = kernel_of(filter_y, h, subpel_y_qn);
  if (!subpel_x_qn && !subpel_y_qn) { /* aom_convolve_copy */
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) put(dst, elem16, (ptrdiff_t)y * dst_stride + x, px(src, elem16, (ptrdiff_t)y * src_stride + x), tbd);
  } else if (subpel_x_qn && !subpel_y_qn) { /* av1_[highbd_]convolve_x_sr_c (convolve.c:149-174,569-595) */
    const int bits = 7 - round_0;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int res = 0;
        for (int k = 0; k < 8; ++k) res += fx[k] * px(src, elem16, (ptrdiff_t)y * src_stride + x - 3 + k);
        res = RPOT(res, round_0);
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(res, bits), tbd);
      }
  } else if (!subpel_x_qn) { /* av1_[highbd_]convolve_y_sr_c (convolve.c:128-147,597-615) */
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int res = 0;
        for (int k = 0; k < 8; ++k) res += fy[k] * px(src, elem16, (ptrdiff_t)(y - 3 + k) * src_stride + x);
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(res, 7), tbd);
      }
  } else { /* av1_[highbd_]convolve_2d_sr_c (convolve.c:76-126,617-668) */
    const int im_h = h + 7, bits = 14 - round_0 - round_1, offset_bits = tbd + 14 - round_0;
    int16_t *im = (int16_t *)malloc(sizeof(int16_t) * (size_t)im_h * w);
    for (int y = 0; y < im_h; ++y)
      for (int x = 0; x < w; ++x) {
        int sum = 1 << (tbd + 6);
        for (int k = 0; k < 8; ++k) sum += fx[k] * px(src, elem16, (ptrdiff_t)(y - 3) * src_stride + x - 3 + k);
        im[y * w + x] = (int16_t)RPOT(sum, round_0);
      }
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int sum = 1 << offset_bits;
        for (int k = 0; k < 8; ++k) sum += fy[k] * im[(y + k) * w + x];
        int res = RPOT(sum, round_1) - ((1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1)));
        if (!elem16) res = (int16_t)res; /* the 8-bit function keeps it in an int16_t (convolve.c:119) */
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(res, bits), tbd);
      }
    free(im);
  }
}

/* One luma block of av1_enc_build_inter_predictor for an unscaled reference: mv in 1/8 pel (row, col) ->
 * position in 1/16 pel (init_subpel_params, av1/common/reconinter.h:130-165, is_scaled == 0, ss = 0), integer part
 * selects the source block, the fraction the kernel phase.  ref_origin: pixel (0, 0) of the reference plane. */
void orc_build_inter_pred_block_ss(const void *ref_origin, int ref_stride, void *dst, int dst_stride, int bx, int by, int bw, int bh,
                                   int mv_row, int mv_col, int filter_x, int filter_y, int elem16, int bd, int ss_x, int ss_y) {
  const int pos_x = (bx << 4) + mv_col * (1 << (1 - ss_x)), pos_y = (by << 4) + mv_row * (1 << (1 - ss_y));
  const int esz = elem16 ? 2 : 1;
  const char *src = (const char *)ref_origin + ((ptrdiff_t)(pos_y >> 4) * ref_stride + (pos_x >> 4)) * esz;
  orc_convolve_sr(src, ref_stride, dst, dst_stride, bw, bh, filter_x, filter_y, pos_x & 15, pos_y & 15, elem16, bd);
}

void orc_build_inter_pred_block(const void *ref_origin, int ref_stride, void *dst, int dst_stride, int bx, int by, int bw, int bh,
                                int mv_row, int mv_col, int filter_x, int filter_y, int elem16, int bd) {
  orc_build_inter_pred_block_ss(ref_origin, ref_stride, dst, dst_stride, bx, by, bw, bh, mv_row, mv_col, filter_x, filter_y, elem16, bd, 0, 0);
}

/* Compound prediction of one block from two references: av1_[highbd_]convolve_2d_facade with is_compound = 1, first
 * reference into the CONV_BUF (do_average 0), second averaged in (do_average 1) -- convolve_2d_facade_compound
 * (av1/common/convolve.c:471-493) -> av1_[highbd_]dist_wtd_convolve_{2d_copy,x,y,2d}_c (:176-370,670-868), with
 * get_conv_params_no_round's compound rounding (convolve.h:63-95: round_1 = COMPOUND_ROUND1_BITS 7, round_0 3, or 5 for
 * 12-bit).  The four kernels are restated separately, as the reference has them.  fwd = bck = 0: plain average. */
static void compound_one(const void *src, int src_stride, uint16_t *buf, int w, int h, const int16_t *fx, const int16_t *fy, int sx, int sy,
                         int elem16, int tbd, int round_0) {
  const int round_1 = 7, bits = 14 - round_1 - round_0, offset_bits = tbd + 14 - round_0;
  const int round_offset = (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
  if (!sx && !sy) { /* _2d_copy */
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) buf[y * w + x] = (uint16_t)((px(src, elem16, (ptrdiff_t)y * src_stride + x) << bits) + round_offset);
  } else if (sx && !sy) { /* _x: bits = FILTER_BITS - round_1 = 0 */
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int res = 0;
        for (int k = 0; k < 8; ++k) res += fx[k] * px(src, elem16, (ptrdiff_t)y * src_stride + x - 3 + k);
        buf[y * w + x] = (uint16_t)((1 << (7 - round_1)) * RPOT(res, round_0) + round_offset);
      }
  } else if (!sx) { /* _y: bits = FILTER_BITS - round_0 */
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int res = 0;
        for (int k = 0; k < 8; ++k) res += fy[k] * px(src, elem16, (ptrdiff_t)(y - 3 + k) * src_stride + x);
        res *= 1 << (7 - round_0);
        buf[y * w + x] = (uint16_t)(RPOT(res, round_1) + round_offset);
      }
  } else { /* _2d */
    int16_t *im = (int16_t *)malloc(sizeof(int16_t) * (size_t)(h + 7) * w);
    for (int y = 0; y < h + 7; ++y)
      for (int x = 0; x < w; ++x) {
        int sum = 1 << (tbd + 6);
        for (int k = 0; k < 8; ++k) sum += fx[k] * px(src, elem16, (ptrdiff_t)(y - 3) * src_stride + x - 3 + k);
        im[y * w + x] = (int16_t)RPOT(sum, round_0);
      }
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int sum = 1 << offset_bits;
        for (int k = 0; k < 8; ++k) sum += fy[k] * im[(y + k) * w + x];
        buf[y * w + x] = (uint16_t)RPOT(sum, round_1);
      }
    free(im);
  }
}

/* mask != NULL: aom_lowbd_blend_a64_d16_mask_c / aom_highbd_blend_a64_d16_mask_c (aom_dsp/blend_a64_mask.c) of the two
 * CONV_BUFs -- the masked compound predictor (av1/common/reconinter.c build_masked_compound_no_round); the mask weighs
 * reference 0 and lies at (1 << subw) x (1 << subh) times the block's resolution. */
void orc_convolve_compound_mask(const void *src0, int stride0, int sx0, int sy0, const void *src1, int stride1, int sx1, int sy1, void *dst,
                                int dst_stride, int w, int h, int filter_x, int filter_y, int fwd_offset, int bck_offset, int elem16, int bd,
                                const uint8_t *mask, int mask_stride, int subw, int subh) {
  const int tbd = elem16 ? bd : 8;
  const int round_0 = (elem16 && bd == 12) ? 5 : 3, round_1 = 7;
  const int round_bits = 14 - round_0 - round_1, offset_bits = tbd + 14 - round_0;
  const int round_offset = (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
  uint16_t *b0 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h), *b1 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
  compound_one(src0, stride0, b0, w, h, kernel_of(filter_x, w, sx0), kernel_of(filter_y, h, sy0), sx0, sy0, elem16, tbd, round_0);
  compound_one(src1, stride1, b1, w, h, kernel_of(filter_x, w, sx1), kernel_of(filter_y, h, sy1), sx1, sy1, elem16, tbd, round_0);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int tmp = b0[y * w + x];
      const int res = b1[y * w + x];
      if (mask) {
        int m;
        if (!subw && !subh) m = mask[y * mask_stride + x];
        else if (subw && subh)
          m = RPOT(mask[2 * y * mask_stride + 2 * x] + mask[(2 * y + 1) * mask_stride + 2 * x] + mask[2 * y * mask_stride + 2 * x + 1] +
                       mask[(2 * y + 1) * mask_stride + 2 * x + 1], 2);
        else if (subw) m = RPOT(mask[y * mask_stride + 2 * x] + mask[y * mask_stride + 2 * x + 1], 1);
        else m = RPOT(mask[2 * y * mask_stride + x] + mask[(2 * y + 1) * mask_stride + x], 1);
        tmp = (m * tmp + (64 - m) * res) >> 6;
      } else if (fwd_offset || bck_offset) {
        tmp = (tmp * fwd_offset + res * bck_offset) >> 4;
      } else {
        tmp = (tmp + res) >> 1;
      }
      tmp -= round_offset;
      put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(tmp, round_bits), tbd);
    }
  free(b0);
  free(b1);
}

void orc_convolve_compound(const void *src0, int stride0, int sx0, int sy0, const void *src1, int stride1, int sx1, int sy1, void *dst,
                           int dst_stride, int w, int h, int filter_x, int filter_y, int fwd_offset, int bck_offset, int elem16, int bd) {
  orc_convolve_compound_mask(src0, stride0, sx0, sy0, src1, stride1, sx1, sy1, dst, dst_stride, w, h, filter_x, filter_y, fwd_offset, bck_offset,
                             elem16, bd, NULL, 0, 0, 0);
}

/* COMPOUND_DIFFWTD: av1_build_compound_diffwtd_mask_d16_c (av1/common/reconinter.c:296-328) on the two CONV_BUFs --
 * m = clamp(38 + ROUND_POWER_OF_TWO(|p0 - p1|, round_bits + bd - 8) / 16, 0, 64), inverted for DIFFWTD_38_INV --
 * followed by the d16 blend.  mask_type 0 / 1 = DIFFWTD_38 / DIFFWTD_38_INV; mask_out: w * h bytes, stride w. */
void orc_convolve_compound_diffwtd(const void *src0, int stride0, int sx0, int sy0, const void *src1, int stride1, int sx1, int sy1, void *dst,
                                   int dst_stride, int w, int h, int filter_x, int filter_y, int elem16, int bd, int mask_type,
                                   uint8_t *mask_out) {
  const int tbd = elem16 ? bd : 8;
  const int round_0 = (elem16 && bd == 12) ? 5 : 3, round_1 = 7;
  const int round = 14 - round_0 - round_1 + (tbd - 8);
  uint16_t *b0 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h), *b1 = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)w * h);
  compound_one(src0, stride0, b0, w, h, kernel_of(filter_x, w, sx0), kernel_of(filter_y, h, sy0), sx0, sy0, elem16, tbd, round_0);
  compound_one(src1, stride1, b1, w, h, kernel_of(filter_x, w, sx1), kernel_of(filter_y, h, sy1), sx1, sy1, elem16, tbd, round_0);
  for (int i = 0; i < w * h; ++i) {
    int diff = abs((int)b0[i] - (int)b1[i]);
    diff = RPOT(diff, round);
    int m = 38 + diff / 16;
    m = m < 0 ? 0 : m > 64 ? 64 : m;
    mask_out[i] = (uint8_t)(mask_type ? 64 - m : m);
  }
  free(b0);
  free(b1);
  orc_convolve_compound_mask(src0, stride0, sx0, sy0, src1, stride1, sx1, sy1, dst, dst_stride, w, h, filter_x, filter_y, 0, 0, elem16, bd,
                             mask_out, w, 0, 0);
}

/* aom_[highbd_]blend_a64_vmask_c / _hmask_c (aom_dsp/blend_a64_vmask.c, blend_a64_hmask.c) in place: dst = AOM_BLEND_A64(m, dst,
 * src1) with m = mask[row] (vertical) or mask[column] -- the OBMC blends of build_obmc_inter_pred_above / _left
 * (av1/common/reconinter.c:844-920).  Pinned by tests/golden/ref_eval_obmc_blend.npz. */
void orc_blend_a64_1d(void *dst, int dst_stride, const void *src1, int src1_stride, const uint8_t *mask, int w, int h, int vertical, int elem16) {
  for (int i = 0; i < h; ++i)
    for (int j = 0; j < w; ++j) {
      const int m = mask[vertical ? i : j];
      const int v = RPOT(m * px(dst, elem16, (ptrdiff_t)i * dst_stride + j) + (64 - m) * px(src1, elem16, (ptrdiff_t)i * src1_stride + j), 6);
      if (elem16) ((uint16_t *)dst)[(ptrdiff_t)i * dst_stride + j] = (uint16_t)v; else ((uint8_t *)dst)[(ptrdiff_t)i * dst_stride + j] = (uint8_t)v;
    }
}

/* av1_convolve_2d_scale_c / av1_highbd_convolve_2d_scale_c (av1/common/convolve.c:560-668,1090-1200): the predictor of a SCALED reference
 * (reference scaling / frame resizing; av1_make_inter_predictor's `is_scaled` branch).  src points at the block's integer source position
 * (pos >> SCALE_SUBPEL_BITS), subpel_*_qn / *_step_qn in 1/1024 pel; the kernel phase of a sample is (qn & 1023) >> 6.  conv_params as
 * get_conv_params_no_round gives them: round_0 = 3 (5 at 12 bits), round_1 = 7 for a compound, else 14 - round_0.  is_compound with
 * do_average 0 writes conv (the block's CONV_BUF), do_average 1 blends with it into dst.  Pinned by tests/golden/ref_eval_scale.npz. */
void orc_convolve_2d_scale(const void *src, int src_stride, void *dst, int dst_stride, int w, int h, int filter_x, int filter_y, int subpel_x_qn,
                           int x_step_qn, int subpel_y_qn, int y_step_qn, int elem16, int bd, int is_compound, int do_average, int use_dist_wtd,
                           int fwd_offset, int bck_offset, uint16_t *conv, int conv_stride) {
  if (!elem16) bd = 8;
  const int round_0 = bd == 12 ? 5 : 3, round_1 = is_compound ? 7 : 14 - round_0, bits = 14 - round_0 - round_1;
  const int im_h = (((h - 1) * y_step_qn + subpel_y_qn) >> 10) + 8;
  int16_t *im = (int16_t *)malloc(sizeof(int16_t) * (size_t)im_h * w);
  for (int y = 0; y < im_h; ++y) {
    int x_qn = subpel_x_qn;
    for (int x = 0; x < w; ++x, x_qn += x_step_qn) {
      const ptrdiff_t at = (ptrdiff_t)(y - 3) * src_stride + (x_qn >> 10);
      const int16_t *f = kernel_of(filter_x, w, (x_qn & 1023) >> 6);
      int32_t sum = 1 << (bd + 7 - 1);
      for (int k = 0; k < 8; ++k) sum += f[k] * px(src, elem16, at + k - 3);
      im[y * w + x] = (int16_t)RPOT(sum, round_0);
    }
  }
  const int offset_bits = bd + 14 - round_0;
  for (int x = 0; x < w; ++x) {
    int y_qn = subpel_y_qn;
    for (int y = 0; y < h; ++y, y_qn += y_step_qn) {
      const int16_t *col = im + ((y_qn >> 10) + 3) * w + x;
      const int16_t *f = kernel_of(filter_y, h, (y_qn & 1023) >> 6);
      int32_t sum = 1 << offset_bits;
      for (int k = 0; k < 8; ++k) sum += f[k] * col[(k - 3) * w];
      const int res = RPOT(sum, round_1);
      const int off = (1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1));
      if (is_compound && !do_average) {
        conv[(ptrdiff_t)y * conv_stride + x] = (uint16_t)res;
      } else if (is_compound) {
        int32_t t = conv[(ptrdiff_t)y * conv_stride + x];
        t = use_dist_wtd ? (t * fwd_offset + res * bck_offset) >> 4 : (t + res) >> 1;
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(t - off, bits), bd);
      } else {
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(res - off, bits), bd);
      }
    }
  }
  free(im);
}
