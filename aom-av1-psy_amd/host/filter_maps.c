/*
 * Producers of the in-loop filter parameter planes (plain C99, host only: no GPU call in this file).
 *
 * The deblocking and CDEF kernels of libaomhip take per-unit parameter planes instead of walking the encoder's
 * mode-info grid on the device.  These functions are the host half of that contract: they turn a compact per-4x4
 * description -- what an integrator copies out of MB_MODE_INFO while it walks the grid once per frame -- into
 *   * the edge-parameter plane of aomhip_deblock_plane: {len_v, lvl_v, len_h, lvl_h} per 4x4 unit, following
 *     set_lpf_parameters (av1/common/av1_loopfilter.c:223-328) for both edge directions;
 *   * the per-segment / reference / mode level table of av1_loop_filter_frame_init (:126-195) that
 *     av1_get_filter_level (:68-111) reads when delta_lf is off;
 *   * the CDEF skip map (is_8x8_block_skip / av1_cdef_compute_sb_list, av1/common/cdef.c:24-68) and the per-64x64
 *     strength planes (cdef.c:309-322).
 */
#include <string.h>

#include "aomhip.h"

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* tx_size_wide_unit_log2 / tx_size_high_unit_log2 (av1/common/common_data.h) for the 19 TX_SIZEs */
static const unsigned char k_tx_w_unit_log2[19] = { 0, 1, 2, 3, 4, 0, 1, 1, 2, 2, 3, 3, 4, 0, 2, 1, 3, 2, 4 };
static const unsigned char k_tx_h_unit_log2[19] = { 0, 1, 2, 3, 4, 1, 0, 2, 1, 3, 2, 4, 3, 2, 0, 3, 1, 4, 2 };
static const unsigned char k_len_luma[5] = { 4, 8, 14, 14, 14 }; /* tx_dim_to_filter_length (:219) */

void aomhip_lf_level_table(const aomhip_lf_frame_params *fp, int plane, uint8_t lvl[8][2][8][2]) {
  /* av1_loop_filter_frame_init (:136-195) for one plane */
  const int base[2] = { plane == 0 ? fp->filter_level[0] : (plane == 1 ? fp->filter_level_u : fp->filter_level_v),
                        plane == 0 ? fp->filter_level[1] : (plane == 1 ? fp->filter_level_u : fp->filter_level_v) };
  for (int seg = 0; seg < 8; ++seg)
    for (int dir = 0; dir < 2; ++dir) {
      int lvl_seg = base[dir];
      /* seg_lvl_lf_lut[plane][dir]: SEG_LVL_ALT_LF_Y_V, _Y_H, _U, _V = features 1, 2, 3, 4 (av1/common/seg_common.h) */
      const int feature = plane == 0 ? 1 + dir : (plane == 1 ? 3 : 4);
      if (fp->seg_enabled && (fp->seg_feature_mask[seg] >> feature) & 1)
        lvl_seg = clampi(lvl_seg + fp->seg_feature_data[seg][feature], 0, 63);
      if (!fp->mode_ref_delta_enabled) {
        memset(lvl[seg][dir], lvl_seg, sizeof(lvl[seg][dir]));
      } else {
        const int scale = 1 << (lvl_seg >> 5);
        memset(lvl[seg][dir], 0, sizeof(lvl[seg][dir]));
        lvl[seg][dir][0][0] = (uint8_t)clampi(lvl_seg + fp->ref_deltas[0] * scale, 0, 63); /* INTRA_FRAME: mode delta unused */
        for (int ref = 1; ref < 8; ++ref)
          for (int mode = 0; mode < 2; ++mode)
            lvl[seg][dir][ref][mode] = (uint8_t)clampi(lvl_seg + fp->ref_deltas[ref] * scale + fp->mode_deltas[mode] * scale, 0, 63);
      }
    }
}

static void one_edge(const aomhip_lf_unit *cur, const aomhip_lf_unit *prev, unsigned coord, int vert, int is_chroma,
                     uint8_t *len_out, uint8_t *lvl_out) {
  *len_out = 0;
  *lvl_out = 0;
  const int ts = cur->tx_size;
  if (ts > 18) return; /* TX_INVALID: the mode info of this unit is not set up (:252) */
  const unsigned tx_mask = (4u << (vert ? k_tx_w_unit_log2[ts] : k_tx_h_unit_log2[ts])) - 1;
  if (coord & tx_mask) return;   /* not a transform edge (:262-265) */
  if (!coord || !prev) return;   /* the picture edge is never filtered (:273) */
  if (prev->tx_size > 18) return;
  const unsigned curr_level = vert ? cur->level_v : cur->level_h;
  const unsigned pv_lvl = vert ? prev->level_v : prev->level_h;
  const unsigned pu_mask = (1u << (vert ? cur->pb_w_log2 : cur->pb_h_log2)) - 1;
  const int pu_edge = !(coord & pu_mask);
  if ((curr_level || pv_lvl) && (!prev->skip_inter || !cur->skip_inter || pu_edge)) {
    const int a = vert ? k_tx_w_unit_log2[ts] : k_tx_h_unit_log2[ts];
    const int b = vert ? k_tx_w_unit_log2[prev->tx_size] : k_tx_h_unit_log2[prev->tx_size];
    const int dim = a < b ? a : b;
    *len_out = is_chroma ? (dim == 0 ? 4 : 6) : k_len_luma[dim];
    *lvl_out = (uint8_t)(curr_level ? curr_level : pv_lvl);
  }
}

int aomhip_lf_build_edge_params(const aomhip_lf_unit *units, int units_stride, int plane_width, int plane_height, int is_chroma,
                                uint8_t *edge_params, int edge_stride) {
  if (!units || !edge_params || plane_width <= 0 || plane_height <= 0) return AOMHIP_ERR_INVALID;
  const int ucols = (plane_width + 3) >> 2, urows = (plane_height + 3) >> 2;
  if (units_stride < ucols || edge_stride < ucols) return AOMHIP_ERR_INVALID;
  for (int uy = 0; uy < urows; ++uy)
    for (int ux = 0; ux < ucols; ++ux) {
      const aomhip_lf_unit *cur = units + (size_t)uy * units_stride + ux;
      uint8_t *e = edge_params + ((size_t)uy * edge_stride + ux) * 4;
      one_edge(cur, ux ? cur - 1 : 0, 4u * (unsigned)ux, 1, is_chroma, &e[0], &e[1]);
      one_edge(cur, uy ? cur - units_stride : 0, 4u * (unsigned)uy, 0, is_chroma, &e[2], &e[3]);
      if (e[1] == 0) e[0] = 0; /* a filter of level 0 does nothing (the kernels skip it; lfthr[0] would be all-pass anyway) */
      if (e[3] == 0) e[2] = 0;
    }
  return AOMHIP_OK;
}

int aomhip_cdef_build_skip8x8(const uint8_t *mi_skip_txfm, int mi_stride, int mi_rows, int mi_cols, uint8_t *skip8x8, int skip_stride) {
  if (!mi_skip_txfm || !skip8x8 || mi_rows <= 0 || mi_cols <= 0 || mi_stride < mi_cols || skip_stride < (mi_cols + 1) / 2)
    return AOMHIP_ERR_INVALID;
  for (int r = 0; r < mi_rows; r += 2)
    for (int c = 0; c < mi_cols; c += 2) {
      int all = 1;
      for (int dr = 0; dr < 2; ++dr)
        for (int dc = 0; dc < 2; ++dc) {
          const int rr = r + dr < mi_rows ? r + dr : mi_rows - 1, cc = c + dc < mi_cols ? c + dc : mi_cols - 1;
          all &= mi_skip_txfm[(size_t)rr * mi_stride + cc] != 0;
        }
      skip8x8[(size_t)(r >> 1) * skip_stride + (c >> 1)] = (uint8_t)all;
    }
  return AOMHIP_OK;
}

int aomhip_cdef_build_strengths(const int8_t *fb_strength_index, int n_fb, const int *cdef_strengths, const int *cdef_uv_strengths,
                                uint8_t *fb_pri, uint8_t *fb_sec, uint8_t *fb_uv_pri, uint8_t *fb_uv_sec) {
  if (!fb_strength_index || !cdef_strengths || !fb_pri || !fb_sec || n_fb < 0) return AOMHIP_ERR_INVALID;
  for (int i = 0; i < n_fb; ++i) {
    const int idx = fb_strength_index[i];
    int y = 0, uv = 0;
    if (idx >= 0) { /* cdef_strength == -1: the filter block is skipped (cdef.c:303-307) */
      y = cdef_strengths[idx];
      uv = cdef_uv_strengths ? cdef_uv_strengths[idx] : 0;
    }
    int sec = y % 4;
    sec += sec == 3;
    fb_pri[i] = (uint8_t)(y / 4);
    fb_sec[i] = (uint8_t)sec;
    if (fb_uv_pri && fb_uv_sec) {
      sec = uv % 4;
      sec += sec == 3;
      fb_uv_pri[i] = (uint8_t)(uv / 4);
      fb_uv_sec[i] = (uint8_t)sec;
    }
  }
  return AOMHIP_OK;
}
