/*
 * oracle/aomref_filtermaps.c -- the mode-info walk that produces the in-loop filter parameter planes.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Restates, at the level of the encoder's MB_MODE_INFO grid,
 *   av1_loop_filter_frame_init / av1_get_filter_level   av1/common/av1_loopfilter.c:68-195
 *   get_transform_size / set_lpf_parameters              av1/common/av1_loopfilter.c:197-328
 *   is_8x8_block_skip / av1_cdef_compute_sb_list         av1/common/cdef.c:24-68
 * PINNED by tests/golden/ref_eval_filtermaps.npz: those functions interpreted where they lie on random mode-info grids
 * (luma + 4:2:0 chroma, delta_lf on / off, segment features, reference / mode deltas), checked bit for bit in
 * tests/test_filter_maps.py.  The product's producers (aom-av1-psy_amd/host/filter_maps.c) take the compact per-unit
 * description this file derives from the grid (orc_lf_units) and must give the same planes.
 */
#include <string.h>

#include "aomref.h"
#include "aomref_blocktables.inc"

typedef struct {
  uint8_t bsize, tx_size, inter_tx_size[16], skip_txfm, mode, segment_id;
  int8_t ref_frame0, delta_lf_from_base, delta_lf[4], cdef_strength;
} orc_mbmi; /* the members of MB_MODE_INFO (av1/common/blockd.h) these functions read */

typedef struct {
  int filter_level[2], filter_level_u, filter_level_v, mode_ref_delta_enabled;
  int8_t ref_deltas[8], mode_deltas[2];
  int delta_lf_present_flag, delta_lf_multi;
  int seg_enabled;
  uint8_t seg_feature_mask[8];
  int16_t seg_feature_data[8][8];
} orc_lf_frame;

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static const int k_mode_lf_lut[25] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1 }; /* :41-45 */

void orc_lf_frame_init(const orc_lf_frame *f, uint8_t lvl[3][8][2][8][2]) { /* :126-195 */
  memset(lvl, 0, 3 * 8 * 2 * 8 * 2);
  const int filt[3] = { f->filter_level[0], f->filter_level_u, f->filter_level_v };
  const int filt_r[3] = { f->filter_level[1], f->filter_level_u, f->filter_level_v };
  for (int plane = 0; plane < 3; ++plane) {
    if (plane == 0 && !filt[0] && !filt_r[0]) break;
    if (plane > 0 && !filt[plane]) continue;
    for (int seg = 0; seg < 8; ++seg)
      for (int dir = 0; dir < 2; ++dir) {
        int lvl_seg = dir == 0 ? filt[plane] : filt_r[plane];
        const int feature = plane == 0 ? 1 + dir : 2 + plane; /* seg_lvl_lf_lut */
        if (f->seg_enabled && ((f->seg_feature_mask[seg] >> feature) & 1)) lvl_seg = clampi(lvl_seg + f->seg_feature_data[seg][feature], 0, 63);
        if (!f->mode_ref_delta_enabled) {
          memset(lvl[plane][seg][dir], lvl_seg, 16);
        } else {
          const int scale = 1 << (lvl_seg >> 5);
          lvl[plane][seg][dir][0][0] = (uint8_t)clampi(lvl_seg + f->ref_deltas[0] * scale, 0, 63);
          for (int ref = 1; ref < 8; ++ref)
            for (int mode = 0; mode < 2; ++mode)
              lvl[plane][seg][dir][ref][mode] = (uint8_t)clampi(lvl_seg + f->ref_deltas[ref] * scale + f->mode_deltas[mode] * scale, 0, 63);
        }
      }
  }
}

static int filter_level(const orc_lf_frame *f, const uint8_t lvl[3][8][2][8][2], int dir, int plane, const orc_mbmi *m) { /* :68-111 */
  if (f->delta_lf_present_flag) {
    static const int delta_lf_id_lut[3][2] = { { 0, 1 }, { 2, 2 }, { 3, 3 } };
    const int delta_lf = f->delta_lf_multi ? m->delta_lf[delta_lf_id_lut[plane][dir]] : m->delta_lf_from_base;
    const int base = plane == 0 ? f->filter_level[dir] : (plane == 1 ? f->filter_level_u : f->filter_level_v);
    int lvl_seg = clampi(delta_lf + base, 0, 63);
    const int feature = plane == 0 ? 1 + dir : 2 + plane;
    if (f->seg_enabled && ((f->seg_feature_mask[m->segment_id] >> feature) & 1))
      lvl_seg = clampi(lvl_seg + f->seg_feature_data[m->segment_id][feature], 0, 63);
    if (f->mode_ref_delta_enabled) {
      const int scale = 1 << (lvl_seg >> 5);
      lvl_seg += f->ref_deltas[m->ref_frame0] * scale;
      if (m->ref_frame0 > 0) lvl_seg += f->mode_deltas[k_mode_lf_lut[m->mode]] * scale;
      lvl_seg = clampi(lvl_seg, 0, 63);
    }
    return lvl_seg;
  }
  return lvl[plane][m->segment_id][dir][m->ref_frame0][k_mode_lf_lut[m->mode]];
}

static int adjusted_tx_size(int t) { /* av1_get_adjusted_tx_size (av1/common/blockd.h): 64-point sizes are coded as 32-point */
  switch (t) {
    case 4: case 12: case 11: return 3; /* TX_64X64, TX_64X32, TX_32X64 -> TX_32X32 */
    case 18: return 10;                 /* TX_64X16 -> TX_32X16 */
    case 17: return 9;                  /* TX_16X64 -> TX_16X32 */
    default: return t;
  }
}

static int transform_size(const orc_mbmi *m, int mi_row, int mi_col, int plane, int ssx, int ssy) { /* :197-217, xd == NULL */
  int ts = plane == 0 ? m->tx_size : adjusted_tx_size(k_max_txsize_rect_lookup[k_ss_size_lookup[m->bsize][ssx][ssy]]);
  if (plane == 0 && m->ref_frame0 > 0 && !m->skip_txfm) {
    static const uint8_t tw_w[22] = { 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3, 3, 0, 1, 1, 2, 2, 3 };
    static const uint8_t tw_h[22] = { 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3, 3, 3, 1, 0, 2, 1, 3, 2 };
    static const uint8_t st[22] = { 0, 0, 1, 1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, 2, 2, 0, 1, 0, 1, 0, 1 }; /* av1_get_txb_size_index (blockd.h:1210-1226) */
    const int br = mi_row & (k_mi_size_high[m->bsize] - 1), bc = mi_col & (k_mi_size_wide[m->bsize] - 1);
    ts = m->inter_tx_size[((br >> tw_h[m->bsize]) << st[m->bsize]) + (bc >> tw_w[m->bsize])];
  }
  return ts;
}

static int log2i(int v) { int n = 0; while ((1 << n) < v) ++n; return n; }

/* set_lpf_parameters (:223-328): filter length and level of the edge on the left (dir 0) / top (dir 1) side of the 4x4 unit at
 * (x, y) of the plane; returns the transform size. */
int orc_set_lpf_parameters(const orc_mbmi *const *grid, int mi_stride, const orc_lf_frame *f, const uint8_t lvl[3][8][2][8][2], int dir,
                           unsigned x, unsigned y, int plane, int ssx, int ssy, unsigned width, unsigned height, int *filter_length,
                           int *level) {
  *filter_length = 0;
  *level = 0;
  if (width <= x || height <= y) return 0;
  const int mi_row = ssy | (int)((y << ssy) >> 2), mi_col = ssx | (int)((x << ssx) >> 2);
  const orc_mbmi *const *mi = grid + (size_t)mi_row * mi_stride + mi_col;
  const orc_mbmi *m = mi[0];
  if (!m) return 255;
  const int ts = transform_size(m, mi_row, mi_col, plane, ssx, ssy);
  const unsigned coord = dir == 0 ? x : y;
  const unsigned mask = (dir == 0 ? k_tx_size_wide[ts] : k_tx_size_high[ts]) - 1u;
  if (coord & mask) return ts;
  const int curr_level = filter_level(f, lvl, dir, plane, m);
  const int curr_skipped = m->skip_txfm && m->ref_frame0 > 0;
  if (coord) {
    const ptrdiff_t mode_step = dir == 0 ? (1 << ssx) : ((ptrdiff_t)mi_stride << ssy);
    const orc_mbmi *pv = *(mi - mode_step);
    if (!pv) return 255;
    const int pv_row = dir == 0 ? mi_row : mi_row - (1 << ssy), pv_col = dir == 0 ? mi_col - (1 << ssx) : mi_col;
    const int pv_ts = transform_size(pv, pv_row, pv_col, plane, ssx, ssy);
    const int pv_lvl = filter_level(f, lvl, dir, plane, pv);
    const int pv_skip = pv->skip_txfm && pv->ref_frame0 > 0;
    const int pb = k_ss_size_lookup[m->bsize][ssx][ssy];
    const unsigned pmask = (dir == 0 ? k_block_size_wide[pb] : k_block_size_high[pb]) - 1u;
    const int pu_edge = !(coord & pmask);
    if ((curr_level || pv_lvl) && (!pv_skip || !curr_skipped || pu_edge)) {
      const int a = log2i((dir == 0 ? k_tx_size_wide[ts] : k_tx_size_high[ts]) / 4), b = log2i((dir == 0 ? k_tx_size_wide[pv_ts] : k_tx_size_high[pv_ts]) / 4);
      const int dim = a < b ? a : b;
      static const int len_luma[5] = { 4, 8, 14, 14, 14 };
      *filter_length = plane ? (dim == 0 ? 4 : 6) : len_luma[dim];
      *level = curr_level ? curr_level : pv_lvl;
    }
  }
  return ts;
}

/* The compact per-unit description the product's producer takes (aomhip_lf_unit: tx_size, skip_inter, pb_w_log2, pb_h_log2,
 * level_v, level_h), derived from the grid for every 4x4 unit of the plane. */
void orc_lf_units(const orc_mbmi *const *grid, int mi_stride, const orc_lf_frame *f, const uint8_t lvl[3][8][2][8][2], int plane, int ssx,
                  int ssy, int width, int height, uint8_t *units /* [(h/4)][(w/4)][6] */) {
  const int ucols = (width + 3) / 4, urows = (height + 3) / 4;
  for (int uy = 0; uy < urows; ++uy)
    for (int ux = 0; ux < ucols; ++ux) {
      const int mi_row = ssy | ((4 * uy << ssy) >> 2), mi_col = ssx | ((4 * ux << ssx) >> 2);
      const orc_mbmi *m = grid[(size_t)mi_row * mi_stride + mi_col];
      uint8_t *u = units + ((size_t)uy * ucols + ux) * 6;
      if (!m) { u[0] = 255; u[1] = u[2] = u[3] = u[4] = u[5] = 0; continue; }
      const int pb = k_ss_size_lookup[m->bsize][ssx][ssy];
      u[0] = (uint8_t)transform_size(m, mi_row, mi_col, plane, ssx, ssy);
      u[1] = (uint8_t)(m->skip_txfm && m->ref_frame0 > 0);
      u[2] = (uint8_t)log2i(k_block_size_wide[pb]);
      u[3] = (uint8_t)log2i(k_block_size_high[pb]);
      u[4] = (uint8_t)filter_level(f, lvl, 0, plane, m);
      u[5] = (uint8_t)filter_level(f, lvl, 1, plane, m);
    }
}

/* av1_cdef_compute_sb_list (cdef.c:36-68) for the 64x64 at (mi_row, mi_col): (by, bx) pairs of the non-skipped 8x8 blocks */
int orc_cdef_compute_sb_list(const orc_mbmi *const *grid, int mi_stride, int mi_rows, int mi_cols, int mi_row, int mi_col, uint8_t *list_by_bx) {
  int maxc = mi_cols - mi_col, maxr = mi_rows - mi_row, count = 0;
  if (maxc > 16) maxc = 16;
  if (maxr > 16) maxr = 16;
  for (int r = 0; r < maxr; r += 2)
    for (int c = 0; c < maxc; c += 2) {
      int skip = 1;
      for (int dr = 0; dr < 2; ++dr)
        for (int dc = 0; dc < 2; ++dc)
          if (!grid[(size_t)(mi_row + r + dr) * mi_stride + mi_col + c + dc]->skip_txfm) skip = 0;
      if (!skip) {
        list_by_bx[2 * count] = (uint8_t)(r >> 1);
        list_by_bx[2 * count + 1] = (uint8_t)(c >> 1);
        ++count;
      }
    }
  return count;
}
