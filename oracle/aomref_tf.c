/*
 * oracle/aomref_tf.c -- the part of the temporal filter that follows its motion search: predictor, pixel weights, accumulation
 * and normalisation of one 32x32 block (av1/encoder/temporal_filter.c: tf_build_predictor :331-392,
 * tf_apply_temporal_filter_self :407-442, compute_square_diff / compute_luma_sq_error_sum :460-512,
 * av1_apply_temporal_filter_c :557-712, tf_normalize_filtered_frame :740-775, and their sequence in
 * av1_tf_do_filtering_row :849-905).
 *
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Pinned by tests/golden/ref_eval_tf_apply.npz: the reference's own functions,
 * interpreted where they lie (tests/golden/gen_ref_eval_tf_apply.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "aomref.h"

#define RPOT(v, n) (((v) + ((1 << (n)) >> 1)) >> (n))

static const int16_t k_interp12[16][12] = {
#include "aomref_interp12.inc"
};

static int px(const void *p, int elem16, ptrdiff_t i) { return elem16 ? ((const uint16_t *)p)[i] : ((const uint8_t *)p)[i]; }
static void put(void *p, int elem16, ptrdiff_t i, int v, int bd) {
  const int mx = (1 << (elem16 ? bd : 8)) - 1;
  v = v < 0 ? 0 : v > mx ? mx : v;
  if (elem16) ((uint16_t *)p)[i] = (uint16_t)v; else ((uint8_t *)p)[i] = (uint8_t)v;
}

/* [highbd_]convolve_2d_facade_single with MULTITAP_SHARP2 on both axes (av1/common/convolve.c:495-515,982-1002): the four
 * functions it selects, stated for taps = 12 (fo = taps / 2 - 1 = 5), get_conv_params(0, plane, bd) rounding
 * (convolve.h:63-100: round_0 3, round_1 11; 12-bit 5 / 9). */
void orc_convolve_sr12(const void *src, int src_stride, void *dst, int dst_stride, int w, int h, int subpel_x_qn, int subpel_y_qn,
                       int elem16, int bd) {
  enum { TAPS = 12, FO = 5 };
  const int tbd = elem16 ? bd : 8;
  int round_0 = 3, round_1 = 11;
  if (elem16 && bd + 7 - round_0 + 2 > 16) {
    const int extra = bd + 7 - round_0 + 2 - 16;
    round_0 += extra;
    round_1 -= extra;
  }
  const int16_t *fx = k_interp12[subpel_x_qn & 15], *fy = k_interp12[subpel_y_qn & 15];
  if (!subpel_x_qn && !subpel_y_qn) { /* aom_[highbd_]convolve_copy */
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) put(dst, elem16, (ptrdiff_t)y * dst_stride + x, px(src, elem16, (ptrdiff_t)y * src_stride + x), tbd);
  } else if (subpel_x_qn && !subpel_y_qn) { /* av1_[highbd_]convolve_x_sr_c (convolve.c:149-174,569-595) */
    const int bits = 7 - round_0;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int res = 0;
        for (int k = 0; k < TAPS; ++k) res += fx[k] * px(src, elem16, (ptrdiff_t)y * src_stride + x - FO + k);
        res = RPOT(res, round_0);
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(res, bits), tbd);
      }
  } else if (!subpel_x_qn) { /* av1_[highbd_]convolve_y_sr_c (convolve.c:128-147,597-615) */
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int res = 0;
        for (int k = 0; k < TAPS; ++k) res += fy[k] * px(src, elem16, (ptrdiff_t)(y - FO + k) * src_stride + x);
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(res, 7), tbd);
      }
  } else { /* av1_[highbd_]convolve_2d_sr_c (convolve.c:76-126,617-668) */
    const int im_h = h + TAPS - 1, bits = 14 - round_0 - round_1, offset_bits = tbd + 14 - round_0;
    int16_t *im = (int16_t *)malloc(sizeof(int16_t) * (size_t)im_h * w);
    for (int y = 0; y < im_h; ++y)
      for (int x = 0; x < w; ++x) {
        int sum = 1 << (tbd + 6);
        for (int k = 0; k < TAPS; ++k) sum += fx[k] * px(src, elem16, (ptrdiff_t)(y - FO) * src_stride + x - FO + k);
        im[y * w + x] = (int16_t)RPOT(sum, round_0);
      }
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        int sum = 1 << offset_bits;
        for (int k = 0; k < TAPS; ++k) sum += fy[k] * im[(y + k) * w + x];
        int res = RPOT(sum, round_1) - ((1 << (offset_bits - round_1)) + (1 << (offset_bits - round_1 - 1)));
        if (!elem16) res = (int16_t)res; /* the 8-bit function keeps it in an int16_t (convolve.c:119) */
        put(dst, elem16, (ptrdiff_t)y * dst_stride + x, RPOT(res, bits), tbd);
      }
    free(im);
  }
}

/* tf_build_predictor for ONE plane of one block (temporal_filter.c:349-391): the plane's four sub-blocks, each with its own MV
 * (1/8 pel luma units; position in 1/16 pel of the plane: init_subpel_params, av1/common/reconinter.h:130-165, unscaled).
 * ref_origin: pixel (0, 0) of the reference plane; pred: plane_w x plane_h, row pitch plane_w. */
void orc_tf_build_predictor_plane(const void *ref_origin, int ref_stride, void *pred, int mb_row, int mb_col, int ss_x, int ss_y,
                                  const int16_t *subblock_mvs, int elem16, int bd) {
  const int plane_h = 32 >> ss_y, plane_w = 32 >> ss_x, plane_y = (32 * mb_row) >> ss_y, plane_x = (32 * mb_col) >> ss_x;
  const int h = plane_h >> 1, w = plane_w >> 1, esz = elem16 ? 2 : 1;
  int idx = 0;
  for (int i = 0; i < plane_h; i += h)
    for (int j = 0; j < plane_w; j += w, ++idx) {
      const int mv_row = subblock_mvs[2 * idx], mv_col = subblock_mvs[2 * idx + 1];
      const int pos_x = ((plane_x + j) << 4) + mv_col * (1 << (1 - ss_x)), pos_y = ((plane_y + i) << 4) + mv_row * (1 << (1 - ss_y));
      const char *src = (const char *)ref_origin + ((ptrdiff_t)(pos_y >> 4) * ref_stride + (pos_x >> 4)) * esz;
      orc_convolve_sr12(src, ref_stride, (char *)pred + ((ptrdiff_t)i * plane_w + j) * esz, plane_w, w, h, pos_x & 15, pos_y & 15, elem16, bd);
    }
}

/* av1_apply_temporal_filter_c (temporal_filter.c:557-712) for one block.  frame_planes[p]: pixel (0, 0) of plane p of the frame to
 * filter; pred: the block's predictors, planes one after the other (plane_w x plane_h each); accum / count likewise. */
void orc_tf_apply_block(const void *const *frame_planes, const int *strides, int frame_w, int frame_h, int num_planes, int ss_x, int ss_y,
                        int mb_row, int mb_col, const double *noise_levels, const int16_t *subblock_mvs, const int32_t *subblock_mses,
                        int q_factor, int filter_strength, const void *pred, uint32_t *accum, uint16_t *count, int elem16, int bd) {
  enum { MBH = 32, MBW = 32, PELS = 1024, WIN = 5 };
  const int min_frame_size = frame_h < frame_w ? frame_h : frame_w;
  const double inv_factor = 1.0 / ((5 + 1) * 20);              /* TF_WINDOW_BLOCK_BALANCE_WEIGHT, TF_SEARCH_ERROR_NORM_WEIGHT */
  const double weight_factor = (double)5 * inv_factor;
  double decay_factor[3] = { 0, 0, 0 };
  double q_decay = pow((double)q_factor / 20, 2);               /* TF_Q_DECAY_THRESHOLD */
  q_decay = q_decay < 1e-5 ? 1e-5 : q_decay > 1 ? 1 : q_decay;
  if (q_factor >= 128) q_decay = 0.5 * pow((double)q_factor / 64, 2);   /* TF_QINDEX_CUTOFF */
  double s_decay = pow((double)filter_strength / 4, 2);         /* TF_STRENGTH_THRESHOLD */
  s_decay = s_decay < 1e-5 ? 1e-5 : s_decay > 1 ? 1 : s_decay;
  for (int plane = 0; plane < num_planes; plane++) {
    const double n_decay = 0.5 + log(2 * noise_levels[plane] + 5.0);
    decay_factor[plane] = 1 / (n_decay * q_decay * s_decay);
  }
  double d_factor[4];
  for (int s = 0; s < 4; s++) {
    const double distance = sqrt(pow(subblock_mvs[2 * s], 2) + pow(subblock_mvs[2 * s + 1], 2));
    double distance_threshold = min_frame_size * 0.1;           /* TF_SEARCH_DISTANCE_THRESHOLD */
    distance_threshold = distance_threshold > 1 ? distance_threshold : 1;
    d_factor[s] = distance / distance_threshold;
    d_factor[s] = d_factor[s] > 1 ? d_factor[s] : 1;
  }
  uint32_t square_diff[PELS], luma_sse_sum[PELS];
  memset(square_diff, 0, sizeof(square_diff));
  memset(luma_sse_sum, 0, sizeof(luma_sse_sum));
  const int half_window = WIN >> 1;
  int plane_offset = 0;
  for (int plane = 0; plane < num_planes; ++plane) {
    const int sy = plane ? ss_y : 0, sx = plane ? ss_x : 0;
    const int h = MBH >> sy, w = MBW >> sx;
    const int frame_stride = strides[plane];
    const ptrdiff_t frame_offset = (ptrdiff_t)mb_row * h * frame_stride + mb_col * w;
    const int num_ref_pixels = WIN * WIN + (plane ? (1 << (sx + sy)) : 0);
    const double inv_num_ref_pixels = 1.0 / num_ref_pixels;
    if (plane == 1) { /* compute_luma_sq_error_sum: once, reused by both chroma planes (:659-661) */
      for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j)
          for (int ii = 0; ii < (1 << sy); ++ii)
            for (int jj = 0; jj < (1 << sx); ++jj) luma_sse_sum[i * w + j] += square_diff[((i << sy) + ii) * (w << sx) + (j << sx) + jj];
    }
    for (int i = 0; i < h; ++i) /* compute_square_diff */
      for (int j = 0; j < w; ++j) {
        const int a = px(frame_planes[plane], elem16, frame_offset + (ptrdiff_t)i * frame_stride + j);
        const int b = px(pred, elem16, plane_offset + i * w + j);
        const uint32_t d = (uint32_t)(a > b ? a - b : b - a);
        square_diff[i * w + j] = d * d;
      }
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j) {
        uint64_t sum_square_diff = 0;
        for (int wi = -half_window; wi <= half_window; ++wi)
          for (int wj = -half_window; wj <= half_window; ++wj) {
            int y = i + wi, x = j + wj;
            y = y < 0 ? 0 : y > h - 1 ? h - 1 : y;
            x = x < 0 ? 0 : x > w - 1 ? w - 1 : x;
            sum_square_diff += square_diff[y * w + x];
          }
        sum_square_diff += luma_sse_sum[i * w + j];
        if (bd > 8) sum_square_diff >>= ((bd - 8) * 2);
        const double window_error = sum_square_diff * inv_num_ref_pixels;
        const int subblock_idx = (i >= h / 2) * 2 + (j >= w / 2);
        const double block_error = (double)subblock_mses[subblock_idx];
        const double combined_error = weight_factor * window_error + block_error * inv_factor;
        double scaled_error = combined_error * d_factor[subblock_idx] * decay_factor[plane];
        scaled_error = scaled_error < 7 ? scaled_error : 7;
        const int weight = (int)(exp(-scaled_error) * 1000);   /* TF_WEIGHT_SCALE */
        const int idx = plane_offset + i * w + j;
        accum[idx] += (uint32_t)(weight * px(pred, elem16, idx));
        count[idx] = (uint16_t)(count[idx] + weight);
      }
    plane_offset += h * w;
  }
}

/* tf_apply_temporal_filter_self (:407-442) */
void orc_tf_apply_self_block(const void *const *frame_planes, const int *strides, int num_planes, int ss_x, int ss_y, int mb_row, int mb_col,
                             uint32_t *accum, uint16_t *count, int elem16) {
  int plane_offset = 0;
  for (int plane = 0; plane < num_planes; ++plane) {
    const int h = 32 >> (plane ? ss_y : 0), w = 32 >> (plane ? ss_x : 0);
    const ptrdiff_t frame_offset = (ptrdiff_t)mb_row * h * strides[plane] + mb_col * w;
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j) {
        const int idx = plane_offset + i * w + j;
        accum[idx] += 1000u * (uint32_t)px(frame_planes[plane], elem16, frame_offset + (ptrdiff_t)i * strides[plane] + j);
        count[idx] = (uint16_t)(count[idx] + 1000);
      }
    plane_offset += h * w;
  }
}

/* tf_normalize_filtered_frame (:740-775); OD_DIVU (aom_dsp/odintrin.h:35-42) is an exact unsigned division */
void orc_tf_normalize_block(void *const *out_planes, const int *strides, int num_planes, int ss_x, int ss_y, int mb_row, int mb_col,
                            const uint32_t *accum, const uint16_t *count, int elem16) {
  int plane_offset = 0;
  for (int plane = 0; plane < num_planes; ++plane) {
    const int h = 32 >> (plane ? ss_y : 0), w = 32 >> (plane ? ss_x : 0);
    const ptrdiff_t frame_offset = (ptrdiff_t)mb_row * h * strides[plane] + mb_col * w;
    for (int i = 0; i < h; ++i)
      for (int j = 0; j < w; ++j) {
        const int idx = plane_offset + i * w + j;
        const uint32_t v = (accum[idx] + (uint32_t)(count[idx] >> 1)) / count[idx];
        if (elem16) ((uint16_t *)out_planes[plane])[frame_offset + (ptrdiff_t)i * strides[plane] + j] = (uint16_t)v;
        else ((uint8_t *)out_planes[plane])[frame_offset + (ptrdiff_t)i * strides[plane] + j] = (uint8_t)v;
      }
    plane_offset += h * w;
  }
}

/* av1_tf_do_filtering_row's loop (:849-905) over all blocks of a frame, after the motion search: n_frames window planes per
 * component (frame_origins[f * 3 + p]; absent frames NULL), MVs / MSEs as aomhip_tf_motion_search_frames lays them out
 * ([(f * n_blocks + i) * 4 + k]).  The blocks cover ceil(h / 32) x ceil(w / 32); the planes must hold them (aligned frame + border). */
void orc_tf_apply_frames(const void *const *frame_origins, const int *strides, int n_frames, int filter_frame, int frame_w, int frame_h,
                         int num_planes, int ss_x, int ss_y, const double *noise_levels, const int16_t *subblock_mvs,
                         const int32_t *subblock_mses, int q_factor, int filter_strength, void *const *out_planes, const int *out_strides,
                         int elem16, int bd, int threads, int block_first, int block_step) {
  const int mb_rows = (frame_h + 31) / 32, mb_cols = (frame_w + 31) / 32, n_blocks = mb_rows * mb_cols;
  (void)threads;
  if (block_step < 1) block_step = 1;
  /* (block_first, block_step: a test may ask for every k-th block only -- blocks are independent) */
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
  for (int b = block_first; b < n_blocks; b += block_step) {
    const int mb_row = b / mb_cols, mb_col = b % mb_cols;
    uint32_t accum[3 * 1024];
    uint16_t count[3 * 1024];
    uint16_t pred16[3 * 1024];
    memset(accum, 0, sizeof(accum));
    memset(count, 0, sizeof(count));
    for (int f = 0; f < n_frames; ++f) {
      if (!frame_origins[f * 3]) continue;
      if (f == filter_frame) {
        orc_tf_apply_self_block(frame_origins + f * 3, strides, num_planes, ss_x, ss_y, mb_row, mb_col, accum, count, elem16);
      } else {
        const int16_t *mvs = subblock_mvs + ((size_t)f * n_blocks + b) * 8;
        const int32_t *mses = subblock_mses + ((size_t)f * n_blocks + b) * 4;
        int plane_offset = 0;
        for (int p = 0; p < num_planes; ++p) {
          const int sx = p ? ss_x : 0, sy = p ? ss_y : 0;
          orc_tf_build_predictor_plane(frame_origins[f * 3 + p], strides[p], (char *)pred16 + (size_t)plane_offset * (elem16 ? 2 : 1), mb_row,
                                       mb_col, sx, sy, mvs, elem16, bd);
          plane_offset += (32 >> sx) * (32 >> sy);
        }
        orc_tf_apply_block(frame_origins + filter_frame * 3, strides, frame_w, frame_h, num_planes, ss_x, ss_y, mb_row, mb_col, noise_levels,
                           mvs, mses, q_factor, filter_strength, pred16, accum, count, elem16, bd);
      }
    }
    orc_tf_normalize_block(out_planes, out_strides, num_planes, ss_x, ss_y, mb_row, mb_col, accum, count, elem16);
  }
}
