/*
 * oracle/aomref.h -- CPU restatement of the aom_dsp / av1 encoder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the CPU baseline
 * ("port") for bench.py.  Nothing under aom-av1-psy_amd/ (the product) may
 * include, link or call it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Every function is a from-scratch plain-C restatement of the reference function
 * named in its comment (file:line relative to /root/reference, libaom v3.5.0
 * "psy" fork).  Block sizes are run-time (w, h) parameters instead of the
 * reference's 22 macro-stamped symbols per family.
 *
 * Pinning status (see DESIGN.md "Oracle"): the reference cannot be compiled here
 * under the project rules (all sources include the cmake-generated
 * config/aom_config.h), so there is no oracle/_ref build.  The oracle is pinned
 * by (a) the known-answer tests the reference's own gtests hold (SAD max,
 * variance Zero/OneQuarter, ...), (b) golden vectors produced by mechanically
 * evaluating the reference's straight-line 1-D transform statements
 * (tests/golden/ref_txfm1d_eval.py), (c) the reference tests' double-precision
 * transform tolerance bounds, and (d) every constant table compared with the
 * initialisers parsed out of the reference sources, and (e) for EVERY family --
 * SAD, variance, subtract, the 2-D transforms, the quantisers and their tables,
 * the loop filters and their thresholds, CDEF (block and filter-block level) and
 * the whole of the motion search, the compound / masked / OBMC table members, the
 * sub-pel / compound / masked / OBMC prediction, the RD helpers, the CDEF and loop-
 * restoration search statistics -- golden vectors obtained by interpreting the
 * reference's own C functions where they lie (tests/golden/ref_c_eval.py, a
 * C-subset interpreter; fixtures tests/golden/ref_eval_*.npz, checked by
 * tests/test_golden_ref_eval.py).  No family is left "parity unpinned".
 */
#ifndef AOMREF_ORACLE_H_
#define AOMREF_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- SAD (aom_dsp/sad.c) ------------------------------------------------ */
/* sad.c:22-36 sad(); SADMXN :41-45 */
unsigned orc_sad(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int w, int h);
/* sad.c:65-69 aom_sad_skip_MxN_c: even rows only, doubled */
unsigned orc_sad_skip(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride, int w, int h);
/* sad.c:94-103 aom_sadMxNx4d_c; x3d (:124-129) forwards to x4d and so reads ref[3] */
void orc_sad_x4d(const uint8_t *src, int src_stride, const uint8_t *const ref[4], int ref_stride, int w, int h,
                 uint32_t out[4]);
void orc_sad_skip_x4d(const uint8_t *src, int src_stride, const uint8_t *const ref[4], int ref_stride, int w, int h,
                      uint32_t out[4]);
/* sad.c:46-53 aom_sadMxN_avg_c with variance.c:306-320 aom_comp_avg_pred_c */
unsigned orc_sad_avg(const uint8_t *src, int src_stride, const uint8_t *ref, int ref_stride,
                     const uint8_t *second_pred, int w, int h);
/* all flavours: plain / dist_wtd (sad.c:57-64, variance.c:322-339), highbd (sad.c:282-297, variance.c:731-766) */
unsigned orc_sad_avg_any(const void *src, int src_stride, const void *ref, int ref_stride, const void *second_pred,
                         int w, int h, int elem16, int bd, int fwd_offset, int bck_offset);

/* highbd: sad.c:240-256,276-332.  Planes are plain uint16_t* here (the
 * CONVERT_TO_SHORTPTR byte-pointer convention, aom_ports/mem.h:79-80, is a host
 * pointer encoding, applied by the caller).  `bd` selects the encoder's vtable
 * wrapper (av1/encoder/encoder_utils.h:155-208): 8 -> raw, 10 -> >>2, 12 -> >>4;
 * bd == 0 returns the raw kernel value. */
unsigned orc_highbd_sad(const uint16_t *src, int src_stride, const uint16_t *ref, int ref_stride, int w, int h,
                        int bd);
unsigned orc_highbd_sad_skip(const uint16_t *src, int src_stride, const uint16_t *ref, int ref_stride, int w, int h,
                             int bd);
void orc_highbd_sad_x4d(const uint16_t *src, int src_stride, const uint16_t *const ref[4], int ref_stride, int w,
                        int h, int bd, uint32_t out[4]);

/* ---- variance (aom_dsp/variance.c) ---------------------------------------- */
/* variance.c:56-73 variance(); VAR :141-148.  returns var, writes *sse (and *sum if non-NULL) */
uint32_t orc_variance(const uint8_t *a, int a_stride, const uint8_t *b, int b_stride, int w, int h, uint32_t *sse,
                      int *sum);
/* variance.c:91-139,150-163 aom_sub_pixel_varianceMxN_c; a = interpolated block, b = source */
uint32_t orc_sub_pixel_variance(const uint8_t *a, int a_stride, int xoff, int yoff, const uint8_t *b, int b_stride,
                                int w, int h, uint32_t *sse);
/* variance.c:342-429 highbd_{8,10,12}_variance */
uint32_t orc_highbd_variance(const uint16_t *a, int a_stride, const uint16_t *b, int b_stride, int w, int h,
                             int bd, uint32_t *sse, int *sum);
/* variance.c:475-561 aom_highbd_{8,10,12}_sub_pixel_varianceMxN_c */
uint32_t orc_highbd_sub_pixel_variance(const uint16_t *a, int a_stride, int xoff, int yoff, const uint16_t *b,
                                       int b_stride, int w, int h, int bd, uint32_t *sse);

/* ---- subtract (aom_dsp/subtract.c:20-53) ---------------------------------- */
void orc_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint8_t *src,
                        ptrdiff_t src_stride, const uint8_t *pred, ptrdiff_t pred_stride);
void orc_highbd_subtract_block(int rows, int cols, int16_t *diff, ptrdiff_t diff_stride, const uint16_t *src,
                               ptrdiff_t src_stride, const uint16_t *pred, ptrdiff_t pred_stride);

/* ---- transforms (av1/common/av1_txfm.[ch], av1/encoder/av1_fwd_txfm{1d,2d}.c,
 *                  av1/common/av1_inv_txfm{1d,2d}.c) ---------------------------- */
/* TX_SIZE / TX_TYPE numbering follows av1/common/enums.h:169-193 and
 * aom_dsp/txfm_common.h:52-68 (values are part of the interface). */
enum {
  ORC_TX_4X4, ORC_TX_8X8, ORC_TX_16X16, ORC_TX_32X32, ORC_TX_64X64, ORC_TX_4X8, ORC_TX_8X4, ORC_TX_8X16,
  ORC_TX_16X8, ORC_TX_16X32, ORC_TX_32X16, ORC_TX_32X64, ORC_TX_64X32, ORC_TX_4X16, ORC_TX_16X4, ORC_TX_8X32,
  ORC_TX_32X8, ORC_TX_16X64, ORC_TX_64X16, ORC_TX_SIZES_ALL
};
enum {
  ORC_DCT_DCT, ORC_ADST_DCT, ORC_DCT_ADST, ORC_ADST_ADST, ORC_FLIPADST_DCT, ORC_DCT_FLIPADST,
  ORC_FLIPADST_FLIPADST, ORC_ADST_FLIPADST, ORC_FLIPADST_ADST, ORC_IDTX, ORC_V_DCT, ORC_H_DCT, ORC_V_ADST,
  ORC_H_ADST, ORC_V_FLIPADST, ORC_H_FLIPADST, ORC_TX_TYPES
};
/* 1-D kinds */
enum { ORC_1D_DCT, ORC_1D_ADST, ORC_1D_IDTX };

extern const int32_t orc_cospi[7][64]; /* av1_txfm.c:18-58 av1_cospi_arr_data */
extern const int32_t orc_sinpi[7][5];  /* av1_txfm.c:62-69 av1_sinpi_arr_data */
extern const int orc_tx_wide[ORC_TX_SIZES_ALL], orc_tx_high[ORC_TX_SIZES_ALL];

/* av1_fwd_txfm1d.c: av1_fdct{4..64}, av1_fadst{4,8,16}, av1_fidentity{4..32}_c.  in/out must not alias. */
void orc_fwd_txfm1d(int kind, int n, const int32_t *in, int32_t *out, int cos_bit);
/* av1_inv_txfm1d.c: av1_idct*, av1_iadst*, av1_iidentity*.  clamp_bit = stage_range value (<=0: none) */
void orc_inv_txfm1d(int kind, int n, const int32_t *in, int32_t *out, int cos_bit, int clamp_bit);
/* 1 if (tx_size, tx_type) is a combination av1_get_fwd_txfm_cfg can serve (no INVALID 1-D type) */
int orc_txfm_valid(int tx_size, int tx_type);
/* av1_fwd_txfm2d.c:56-127 fwd_txfm2d_c + the 19 av1_fwd_txfm2d_WxH_c wrappers (:129-312),
 * including the 64-point zero-out / re-pack.  Output length = min(w,32)*min(h,32) packed
 * for 64-wide sizes exactly as the reference leaves it (full w*h array is written). */
void orc_fwd_txfm2d(const int16_t *input, int32_t *output, int stride, int tx_size, int tx_type, int bd);
/* interval analysis of the same transform (bound mode, aomref_txfm.c) */
void orc_fwd_txfm2d_bounds(int tx_size, int tx_type, int input_max, int64_t *max_operand, int64_t *max_sum);
/* av1_inv_txfm2d.c:234-309 inv_txfm2d_add_c via av1_inv_txfm2d_add_WxH_c; dst is uint16 (highbd) */
void orc_inv_txfm2d_add(const int32_t *input, uint16_t *dst, int stride, int tx_size, int tx_type, int bd);
#define ORC_TX_WHT 16 /* lossless 4x4 Walsh-Hadamard in the batch drivers' tx_type field */
void orc_fwht4x4(const int16_t *input, int32_t *output, int stride);
void orc_iwht4x4_add(const int32_t *input, uint16_t *dst, int stride, int eob, int bd);

/* ---- quantize (aom_dsp/quantize.c, av1/encoder/av1_quantize.c) -------------------- */
/* quantize.c:108-169 aom_quantize_b_helper_c with qm_ptr == iqm_ptr == NULL */
void orc_quantize_lp(const int16_t *coeff, intptr_t n, const int16_t *round_fp, const int16_t *quant_fp, int16_t *qcoeff, int16_t *dqcoeff,
                     const int16_t *dequant, uint16_t *eob_out, const int16_t *scan);
int64_t orc_block_error_lp(const int16_t *coeff, const int16_t *dqcoeff, intptr_t n);
/* the fp quantiser with matrices (aomref_quant.c; quantize_fp_helper_c / highbd_quantize_fp_helper_c, qm_ptr / iqm_ptr non-NULL) */
void orc_quantize_fp_qm(const int32_t *coeff, intptr_t n, const int16_t *round_fp, const int16_t *quant_fp, int32_t *qcoeff, int32_t *dqcoeff,
                        const int16_t *dequant, uint16_t *eob_out, const int16_t *scan, int log_scale, int highbd, const uint8_t *qm,
                        const uint8_t *iqm);
void orc_quantize_b_qm(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                       const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob,
                       const int16_t *scan, int log_scale, const uint8_t *qm, const uint8_t *iqm);
void orc_highbd_quantize_b_qm(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                              const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob,
                              const int16_t *scan, int log_scale, const uint8_t *qm, const uint8_t *iqm);
void orc_set_qm(const uint8_t *qm, const uint8_t *iqm);
void orc_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                    const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                    const int16_t *dequant, uint16_t *eob, const int16_t *scan, const int16_t *iscan,
                    int log_scale);
/* quantize.c:261-316 aom_highbd_quantize_b_helper_c, qm NULL */
void orc_highbd_quantize_b(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                           const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff,
                           int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob, const int16_t *scan,
                           const int16_t *iscan, int log_scale);
/* quantize.c:16-105,173-258 adaptive variants (use_quant_b_adapt, av1_quantize.c:309-341,453-) */
void orc_quantize_b_adaptive_qm(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round, const int16_t *quant,
                                const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff, const int16_t *dequant, uint16_t *eob_out,
                                const int16_t *scan, int log_scale, int highbd, const uint8_t *qm, const uint8_t *iqm);
void orc_quantize_b_adaptive(const int32_t *coeff, intptr_t n, const int16_t *zbin, const int16_t *round,
                             const int16_t *quant, const int16_t *quant_shift, int32_t *qcoeff, int32_t *dqcoeff,
                             const int16_t *dequant, uint16_t *eob, const int16_t *scan, int log_scale, int highbd);
/* av1_quantize.c:580-674 av1_build_quantizer (Y plane, no delta-q, sharpness 0): fills the
 * two (DC, AC) lanes of each table for one qindex.  tables laid out [5][2]:
 * zbin, round, quant, quant_shift, dequant. */
void orc_build_quantizer_y(int bit_depth, int qindex, int16_t tables[5][2]);
int16_t orc_dc_q(int qindex, int delta, int bit_depth); /* av1/common/quant_common.c av1_dc_quant_QTX */
int16_t orc_ac_q(int qindex, int delta, int bit_depth); /* av1_ac_quant_QTX */
/* av1/common/scan.c default (zig-zag) / row / col scans, get_scan (scan.h:41-48).
 * Writes n = w*h (clamped to 32x32 for 64-point sizes) entries; returns n. */
int orc_get_scan(int tx_size, int tx_type, int16_t *scan, int16_t *iscan);

/* ---- loop filter (aom_dsp/loopfilter.c) ------------------------------------------- */
/* aom_lpf_{horizontal,vertical}_{4,6,8,14}_c (:136-511): one 4-px edge unit, in place.
 * vertical != 0 filters a vertical edge (pixels left/right of s). */
void orc_lpf(uint8_t *s, int pitch, int vertical, int len, uint8_t blimit, uint8_t limit, uint8_t thresh);
/* aom_highbd_lpf_* (:602-997) */
void orc_highbd_lpf(uint16_t *s, int pitch, int vertical, int len, uint8_t blimit, uint8_t limit, uint8_t thresh,
                    int bd);
/* av1_loopfilter.c:47-66,118-120: thresholds from (level, sharpness) */
void orc_lpf_thresholds(int level, int sharpness, uint8_t *mblim, uint8_t *lim, uint8_t *hev_thr);

/* ---- CDEF (av1/common/cdef_block.c) ------------------------------------------------ */
int orc_cdef_find_dir(const uint16_t *img, int stride, int32_t *var, int coeff_shift);
/* cdef_filter_block_internal (:139-201): dst8 or dst16 non-NULL selects output type */
void orc_cdef_filter_block(uint8_t *dst8, uint16_t *dst16, int dstride, const uint16_t *in, int pri_strength,
                           int sec_strength, int dir, int pri_damping, int sec_damping, int coeff_shift,
                           int block_w, int block_h, int enable_primary, int enable_secondary);
/* plane drivers: av1_cdef_frame for one plane, out of place (cdef.c:138-345 + cdef_block.c:323-426) */
void orc_cdef_plane_luma(const void *src, void *dst, int stride, int width, int height, int elem16, int bd,
                         const uint8_t *fb_pri, const uint8_t *fb_sec, int fb_stride, const uint8_t *skip, int damping,
                         uint8_t *dir_out, int32_t *var_out);
void orc_cdef_plane_chroma(const void *src, void *dst, int stride, int width, int height, int elem16, int bd, int xdec,
                           int ydec, const uint8_t *dir, const uint8_t *fb_pri, const uint8_t *fb_sec, int fb_stride,
                           const uint8_t *skip, int damping);

/* ---- compound / masked / OBMC table members (aomref_compound.c) ----------- */
uint32_t orc_compound_sub_pixel_variance(const void *a, int a_stride, int xoff, int yoff, const void *b, int b_stride, int w, int h,
                                         int elem16, int bd, int kind, const void *second_pred, int fwd_offset, int bck_offset,
                                         const uint8_t *mask, int mask_stride, int invert_mask, uint32_t *sse);
unsigned orc_masked_sad(const void *src, int src_stride, const void *ref, int ref_stride, const void *second_pred, const uint8_t *mask,
                        int mask_stride, int invert_mask, int w, int h, int elem16, int bd);
unsigned orc_obmc_sad(const void *pre, int pre_stride, const int32_t *wsrc, const int32_t *mask, int w, int h, int elem16, int bd);
uint32_t orc_obmc_variance(const void *pre, int pre_stride, int subpel, int xoff, int yoff, const int32_t *wsrc, const int32_t *mask, int w,
                           int h, int elem16, int bd, uint32_t *sse);

/* ---- inter prediction (aomref_convolve.c) --------------------------------- */
void orc_convolve_sr(const void *src, int src_stride, void *dst, int dst_stride, int w, int h, int filter_x, int filter_y,
                     int subpel_x_qn, int subpel_y_qn, int elem16, int bd);
void orc_build_inter_pred_block_ss(const void *ref_origin, int ref_stride, void *dst, int dst_stride, int bx, int by, int bw, int bh,
                                   int mv_row, int mv_col, int filter_x, int filter_y, int elem16, int bd, int ss_x, int ss_y);
/* aomref_tf.c: the temporal filter after its motion search (temporal_filter.c:331-392,407-442,460-512,557-712,740-775,849-905) */
void orc_convolve_sr12(const void *src, int src_stride, void *dst, int dst_stride, int w, int h, int subpel_x_qn, int subpel_y_qn,
                       int elem16, int bd);
void orc_tf_build_predictor_plane(const void *ref_origin, int ref_stride, void *pred, int mb_row, int mb_col, int ss_x, int ss_y,
                                  const int16_t *subblock_mvs, int elem16, int bd);
void orc_tf_apply_block(const void *const *frame_planes, const int *strides, int frame_w, int frame_h, int num_planes, int ss_x, int ss_y,
                        int mb_row, int mb_col, const double *noise_levels, const int16_t *subblock_mvs, const int32_t *subblock_mses,
                        int q_factor, int filter_strength, const void *pred, uint32_t *accum, uint16_t *count, int elem16, int bd);
void orc_tf_apply_self_block(const void *const *frame_planes, const int *strides, int num_planes, int ss_x, int ss_y, int mb_row, int mb_col,
                             uint32_t *accum, uint16_t *count, int elem16);
void orc_tf_normalize_block(void *const *out_planes, const int *strides, int num_planes, int ss_x, int ss_y, int mb_row, int mb_col,
                            const uint32_t *accum, const uint16_t *count, int elem16);
void orc_tf_apply_frames(const void *const *frame_origins, const int *strides, int n_frames, int filter_frame, int frame_w, int frame_h,
                         int num_planes, int ss_x, int ss_y, const double *noise_levels, const int16_t *subblock_mvs,
                         const int32_t *subblock_mses, int q_factor, int filter_strength, void *const *out_planes, const int *out_strides,
                         int elem16, int bd, int threads, int block_first, int block_step);
void orc_convolve_compound_mask(const void *src0, int stride0, int sx0, int sy0, const void *src1, int stride1, int sx1, int sy1, void *dst,
                                int dst_stride, int w, int h, int filter_x, int filter_y, int fwd_offset, int bck_offset, int elem16, int bd,
                                const uint8_t *mask, int mask_stride, int subw, int subh);
void orc_convolve_compound_diffwtd(const void *src0, int stride0, int sx0, int sy0, const void *src1, int stride1, int sx1, int sy1, void *dst,
                                   int dst_stride, int w, int h, int filter_x, int filter_y, int elem16, int bd, int mask_type,
                                   uint8_t *mask_out);
void orc_blend_a64_1d(void *dst, int dst_stride, const void *src1, int src1_stride, const uint8_t *mask, int w, int h, int vertical, int elem16);

/* ---- RD helpers (aomref_rdhelp.c) and restoration statistics (aomref_lrstats.c) */
int64_t orc_sse(const void *a, int a_stride, const void *b, int b_stride, int w, int h, int elem16);
int orc_hadamard(const int16_t *src, ptrdiff_t stride, int n, int flavour, int32_t *coeff);
int orc_cost_coeffs_txb(const int32_t *qcoeff, int eob, int tx_w, int tx_h, int tx_class, const int16_t *scan, int txb_skip_ctx, int dc_sign_ctx,
                        const int32_t *costs);
int orc_cost_coeffs_txb_laplacian(const int32_t *qcoeff, int eob, int tx_class, const int16_t *scan, int txb_skip_ctx, const int32_t *costs);
int orc_get_txb_entropy_context(const int32_t *qcoeff, const int16_t *scan, int eob);
uint64_t orc_sum_sse_2d_i16(const int16_t *src, int src_stride, int width, int height, int *sum);
void orc_get_nz_map_contexts(const uint8_t *levels, const int16_t *scan, int eob, int tx_w, int tx_h, int tx_class, int8_t *coeff_contexts);
void orc_txb_init_levels(const int32_t *coeff, int width, int height, uint8_t *levels);
void orc_convolve_2d_scale(const void *src, int src_stride, void *dst, int dst_stride, int w, int h, int filter_x, int filter_y, int subpel_x_qn,
                           int x_step_qn, int subpel_y_qn, int y_step_qn, int elem16, int bd, int is_compound, int do_average, int use_dist_wtd,
                           int fwd_offset, int bck_offset, uint16_t *conv, int conv_stride);
/* the self-guided restoration filter (aomref_sgr.c): dgd points at the unit's first pixel and is read 3 pixels beyond it on every side */
void orc_selfguided_restoration(const void *dgd, int elem16, int width, int height, int stride, int32_t *flt0, int32_t *flt1, int flt_stride,
                                int sgr_params_idx, int bit_depth);
void orc_apply_selfguided_restoration(const void *dat, int elem16, int width, int height, int stride, int eps, const int *xqd, void *dst, int dst_stride,
                                      int bit_depth);
void orc_wiener_convolve_add_src(const void *src, int elem16, int src_stride, void *dst, int dst_stride, const int16_t *filter_x, const int16_t *filter_y, int w,
                                 int h, int bd);
void orc_calc_proj_params(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, const int32_t *flt0, int flt0_stride,
                          const int32_t *flt1, int flt1_stride, int elem16, int r0, int r1, int64_t H[4], int64_t C[2]);
int64_t orc_pixel_proj_error(const void *src, int width, int height, int src_stride, const void *dat, int dat_stride, const int32_t *flt0, int flt0_stride,
                             const int32_t *flt1, int flt1_stride, int elem16, int r0, int r1, int xq0, int xq1);
/* the warped-motion predictor of one reference, not compound (aomref_warp.c) */
void orc_warp_affine(const int32_t *mat, const void *ref, int elem16, int width, int height, int stride, void *pred, int p_col, int p_row, int p_width,
                     int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd, int round_0, int alpha, int beta, int gamma, int delta);
void orc_warp_affine_compound(const int32_t *mat, const void *ref, int elem16, int width, int height, int stride, void *pred, int p_col, int p_row, int p_width,
                              int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd, int round_0, int alpha, int beta, int gamma, int delta,
                              int do_average, int use_dist_wtd, int fwd_offset, int bck_offset, uint16_t *conv, int conv_stride);
unsigned orc_int_pro_motion_estimation(const void *src, int src_stride, const void *ref, int ref_stride, int bw, int bh, int bd, const int *limits,
                                       const int16_t *ref_mv, int16_t *out_mv);
void orc_vbp_fill_8x8avg(const void *src, int src_stride, const void *dst, int dst_stride, int x16, int y16, int hbd, int pixels_wide, int pixels_high,
                         int32_t *sum, uint32_t *sse);
int orc_vbp_minmax_8x8(const void *src, int src_stride, const void *dst, int dst_stride, int x16, int y16, int hbd, int pixels_wide, int pixels_high);
void orc_vbp_fill_4x4avg(const void *src, int src_stride, int x8, int y8, int hbd, int pixels_wide, int pixels_high, int border_offset_4x4, int32_t *sum,
                         uint32_t *sse);
int orc_get_shear_params(const int32_t *mat, int16_t *abgd);
/* aomref_warpfit.c: av1_selectSamples / av1_find_projection (1 = no usable model, as the reference returns it) */
int orc_select_samples(int mv_row, int mv_col, int *pts, int *pts_inref, int len, int bw, int bh);
int orc_find_projection(int np, const int *pts1, const int *pts2, int bw, int bh, int mvy, int mvx, int32_t *mat, int16_t *abgd, int mi_row, int mi_col);
int64_t orc_warp_error(const int32_t *mat, const int16_t *abgd, const void *ref, int elem16, int width, int height, int stride, const void *dst, int p_col,
                       int p_row, int p_width, int p_height, int p_stride, int subsampling_x, int subsampling_y, int bd, int64_t best_error,
                       const uint8_t *segment_map, int segment_map_stride);
int64_t orc_segmented_frame_error(const void *ref, int elem16, int stride, const void *dst, int p_width, int p_height, int p_stride, int bd,
                                  const uint8_t *segment_map, int segment_map_stride);
uint64_t orc_wedge_sse_from_residuals(const int16_t *r1, const int16_t *d, const uint8_t *m, int n);
int orc_wedge_sign_from_residuals(const int16_t *ds, const uint8_t *m, int n, int64_t limit);
void orc_wedge_compute_delta_squares(int16_t *d, const int16_t *a, const int16_t *b, int n);
void orc_compute_stats(int wiener_win, const void *dgd, const void *src, int h_start, int h_end, int v_start, int v_end, int dgd_stride,
                       int src_stride, int elem16, int bit_depth, int use_downsampled_wiener_stats, int64_t *M, int64_t *H);

/* the small members of the named files (aomref_misc.c) */
uint32_t orc_get_mb_ss(const int16_t *a);
uint64_t orc_mse_wxh_16bit(const void *dst, int dstride, int dst16, const uint16_t *src, int sstride, int w, int h);
uint64_t orc_mse_16xh_16bit(const uint8_t *dst, int dstride, const uint16_t *src, int w, int h);
void orc_comp_mask_pred(void *comp_pred, const void *pred, int width, int height, const void *ref, int ref_stride, const uint8_t *mask, int mask_stride,
                        int invert_mask, int elem16);
int orc_return_extreme_sub_pixel_mv(const int *limits, int allow_hp, int want_max, int16_t *bestmv);

/* av1_estimate_txfm_yrd and the RD-based second-MV choice (aomref_yrd.c) */
int64_t orc_estimate_txfm_yrd(const int16_t *residual, int stride, int bw, int bh, int bd, int is_hbd, const int16_t q[5][2], const uint8_t *above,
                              const uint8_t *left, const int32_t *costs, int tx_type_rate, int tx_size_rate, int no_skip_txfm_rate,
                              int skip_txfm_rate, int rdmult, int lossless, int64_t *out);
int orc_second_mv_rd_choice(int rdmult, int mv_rate0, int rate0, int64_t dist0, int mv_rate1, int rate1, int64_t dist1);

#ifdef __cplusplus
}
#endif
#endif /* AOMREF_ORACLE_H_ */
