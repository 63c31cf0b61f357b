/*
 * oracle/aomref_mcomp.c -- full-pel diamond search and bilinear sub-pel refinement.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Restates, for single-reference, unmasked search:
 *   av1_init_dsmotion_compensation (av1/encoder/mcomp.c:350-389), mvsad_err_cost / mv_err_cost for the
 *   NONE and L1 cost types (:271-339), diamond_search_sad (:1299-1416), full_pixel_diamond (:1421-1470),
 *   get_mvpred_var_cost (:645-664), setup_center_error (:2718-2778), check_better_fast (:2433-2461),
 *   first/second_level_check_fast + two_level_checks_fast (:2503-2624) and
 *   av1_find_best_sub_pixel_tree_pruned_more (:2844-2929, cost_list == NULL, unscaled reference).
 * The kernels are reached through the per-block-size vtable (aom_dsp/variance.h:84-103): sdf / vf / svf
 * with the 10/12-bit wrappers of av1/encoder/encoder_utils.h.
 *
 * PARITY UNPINNED: no reference unit test drives mcomp.c; tests check convergence on
 * content with a known shift and definitional properties only.
 */
#include "aomref.h"

#include <limits.h>
#include <stdlib.h>

typedef struct { int16_t bx, by, start_row, start_col, ref_row, ref_col, row_min, row_max, col_min, col_max; } orc_search_block;
typedef struct { int16_t bx, by, start_row, start_col, ref_row, ref_col, row_min, row_max, col_min, col_max; } orc_subpel_block;

enum { ORC_MV_COST_ENTROPY, ORC_MV_COST_L1_LOWRES, ORC_MV_COST_L1_MIDRES, ORC_MV_COST_L1_HDRES, ORC_MV_COST_NONE };

typedef struct {
  const void *src, *ref; /* pixel (0,0) of the block in src; pixel (0,0)+block origin of the ref plane */
  int src_stride, ref_stride, elem16, bd, w, h, cost_type;
  int ref_row, ref_col; /* ref_mv in 1/8 pel */
} search_ctx;

static unsigned sad_at(const search_ctx *c, int row, int col) { /* ms_params->sdf */
  if (c->elem16)
    return orc_highbd_sad((const uint16_t *)c->src, c->src_stride,
                          (const uint16_t *)c->ref + (ptrdiff_t)row * c->ref_stride + col, c->ref_stride, c->w, c->h,
                          c->bd);
  return orc_sad((const uint8_t *)c->src, c->src_stride, (const uint8_t *)c->ref + (ptrdiff_t)row * c->ref_stride + col,
                 c->ref_stride, c->w, c->h);
}

static int mvsad_cost(const search_ctx *c, int row, int col) { /* mvsad_err_cost_, full_ref_mv = GET_MV_RAWPEL */
  const int frr = (c->ref_row + 3 + (c->ref_row >= 0)) >> 3, frc = (c->ref_col + 3 + (c->ref_col >= 0)) >> 3;
  const int d = abs((row - frr) * 8) + abs((col - frc) * 8);
  switch (c->cost_type) {
    case ORC_MV_COST_L1_LOWRES: return (32 * d) >> 3;
    case ORC_MV_COST_L1_MIDRES: return (15 * d) >> 3;
    case ORC_MV_COST_L1_HDRES: return (8 * d) >> 3;
    default: return 0;
  }
}
static int mv_cost_var(const search_ctx *c, int mrow, int mcol) { /* mv_err_cost_, mv in 1/8 pel */
  const int d = abs(mrow - c->ref_row) + abs(mcol - c->ref_col);
  switch (c->cost_type) {
    case ORC_MV_COST_L1_LOWRES: return (2 * d) >> 3;
    case ORC_MV_COST_L1_MIDRES: return (0 * d) >> 3;
    case ORC_MV_COST_L1_HDRES: return (1 * d) >> 3;
    default: return 0;
  }
}

static int diamond_search(const search_ctx *c, const orc_search_block *b, int level, int search_step, int *num00,
                          int *best_row, int *best_col) {
  /* site table of av1_init_dsmotion_compensation: stage k has radius r_k; stage 10 is the first_step */
  int radius[11], nsteps = 0;
  {
    int stage = 10;
    for (int r = level > 0 ? 1024 / 4 : 1024; r > 0;) {
      radius[stage] = r;
      if (!level || (stage < 9 && level)) r /= 2;
      --stage;
      ++nsteps;
    }
    /* the reference fills stages from index 10 downwards and uses cfg->site[step] for step < nsteps;
     * with 11 stages the two numberings coincide (DIAMOND).  CLAMPED_DIAMOND has more than 11 radii
     * repeats and is not needed by the configs built here. */
    if (nsteps != 11) return INT_MAX;
  }
  static const int8_t dirs[8][2] = { { -1, 0 }, { 1, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, 1 }, { -1, 1 }, { 1, -1 } };
  int row = b->start_row, col = b->start_col;
  row = row < b->row_min ? b->row_min : row > b->row_max ? b->row_max : row; /* clamp_fullmv */
  col = col < b->col_min ? b->col_min : col > b->col_max ? b->col_max : col;
  const int tot_steps = nsteps - search_step;
  *num00 = 0;
  unsigned bestsad = sad_at(c, row, col) + (unsigned)mvsad_cost(c, row, col);
  int is_off_center = 0;
  int next_step_size = tot_steps > 2 ? radius[tot_steps - 2] : 1;
  for (int step = tot_steps - 1; step >= 0; --step) {
    int best_site = 0;
    if (step > 0) next_step_size = radius[step - 1];
    for (int idx = 1; idx <= 8; ++idx) {
      const int r = row + dirs[idx - 1][0] * radius[step], cc = col + dirs[idx - 1][1] * radius[step];
      if (cc < b->col_min || cc > b->col_max || r < b->row_min || r > b->row_max) continue;
      unsigned thissad = sad_at(c, r, cc);
      if (thissad < bestsad) {
        thissad += (unsigned)mvsad_cost(c, r, cc);
        if (thissad < bestsad) {
          bestsad = thissad;
          best_site = idx;
        }
      }
    }
    if (best_site != 0) {
      row += dirs[best_site - 1][0] * radius[step];
      col += dirs[best_site - 1][1] * radius[step];
      is_off_center = 1;
    }
    if (is_off_center == 0) (*num00)++;
    if (best_site == 0) {
      while (next_step_size == radius[step] && step > 2) {
        ++(*num00);
        --step;
        next_step_size = radius[step - 1];
      }
    }
  }
  *best_row = row;
  *best_col = col;
  return (int)bestsad;
}

static int var_cost_at(const search_ctx *c, int row, int col) { /* get_mvpred_var_cost: vf(src, ref) + mv_err_cost_ */
  uint32_t sse;
  unsigned v;
  if (c->elem16)
    v = orc_highbd_variance((const uint16_t *)c->src, c->src_stride,
                            (const uint16_t *)c->ref + (ptrdiff_t)row * c->ref_stride + col, c->ref_stride, c->w, c->h,
                            c->bd, &sse, NULL);
  else
    v = orc_variance((const uint8_t *)c->src, c->src_stride,
                     (const uint8_t *)c->ref + (ptrdiff_t)row * c->ref_stride + col, c->ref_stride, c->w, c->h, &sse,
                     NULL);
  return (int)v + mv_cost_var(c, row * 8, col * 8);
}

static void make_ctx(search_ctx *c, const void *src_origin, int src_stride, const void *ref_origin, int ref_stride,
                     int elem16, int bd, int w, int h, int cost_type, int bx, int by, int ref_row, int ref_col) {
  const size_t e = elem16 ? 2 : 1;
  c->src = (const char *)src_origin + ((ptrdiff_t)by * src_stride + bx) * e;
  c->ref = (const char *)ref_origin + ((ptrdiff_t)by * ref_stride + bx) * e;
  c->src_stride = src_stride; c->ref_stride = ref_stride; c->elem16 = elem16; c->bd = bd; c->w = w; c->h = h;
  c->cost_type = cost_type; c->ref_row = ref_row; c->ref_col = ref_col;
}

void orc_fullpel_diamond_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride,
                               int elem16, int bd, int w, int h, int level, int step_param, int cost_type,
                               const orc_search_block *blocks, int n, int16_t *out_mv, int32_t *out_cost,
                               int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
  for (int i = 0; i < n; ++i) {
    const orc_search_block *b = &blocks[i];
    search_ctx c;
    make_ctx(&c, src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, cost_type, b->bx, b->by, b->ref_row,
             b->ref_col);
    /* full_pixel_diamond (mcomp.c:1421-1470) */
    int n00, num00 = 0, br, bc;
    int bestsme = diamond_search(&c, b, level, step_param, &n00, &br, &bc);
    if (bestsme < INT_MAX) bestsme = var_cost_at(&c, br, bc);
    const int further_steps = 11 - 1 - step_param;
    int nn = n00;
    while (nn < further_steps) {
      ++nn;
      if (num00) {
        num00--;
      } else {
        int tr, tc;
        int thissme = diamond_search(&c, b, level, step_param + nn, &num00, &tr, &tc);
        if (thissme < INT_MAX) thissme = var_cost_at(&c, tr, tc);
        if (thissme < bestsme) {
          bestsme = thissme;
          br = tr;
          bc = tc;
        }
      }
    }
    out_mv[2 * i] = (int16_t)br;
    out_mv[2 * i + 1] = (int16_t)bc;
    out_cost[i] = bestsme;
  }
}

/* ---- exhaustive mesh search: exhaustive_mesh_search (mcomp.c:1474-1543) + full_pixel_exhaustive (:1547-1617) ----
 * Literal restatement including the reference's column handling at step 1: positions are taken four at a time
 * (sdx4df), and the tail group `for (i = 0; i < end_col - c; ++i)` does NOT visit column end_col itself.
 * update_mvs_and_sad (:839-858): skip when this_sad >= best_sad, else add the MV cost and take it on strict <. */
static int mesh_pass(const search_ctx *c, const orc_search_block *b, int *row0, int *col0, int range, int step) {
  int srow = *row0, scol = *col0;
  srow = srow < b->row_min ? b->row_min : srow > b->row_max ? b->row_max : srow; /* clamp_fullmv */
  scol = scol < b->col_min ? b->col_min : scol > b->col_max ? b->col_max : scol;
  int best_row = srow, best_col = scol;
  unsigned best_sad = sad_at(c, srow, scol) + (unsigned)mvsad_cost(c, srow, scol);
  const int col_step = step > 1 ? step : 4;
  const int start_row = -range > b->row_min - srow ? -range : b->row_min - srow;
  const int start_col = -range > b->col_min - scol ? -range : b->col_min - scol;
  const int end_row = range < b->row_max - srow ? range : b->row_max - srow;
  const int end_col = range < b->col_max - scol ? range : b->col_max - scol;
  for (int r = start_row; r <= end_row; r += step) {
    for (int cc = start_col; cc <= end_col; cc += col_step) {
      const int n = step > 1 ? 1 : (cc + 3 <= end_col ? 4 : end_col - cc);
      for (int i = 0; i < n; ++i) {
        const int row = srow + r, col = scol + cc + i;
        const unsigned this_sad = sad_at(c, row, col);
        if (this_sad >= best_sad) continue;
        const unsigned sad = this_sad + (unsigned)mvsad_cost(c, row, col);
        if (sad < best_sad) {
          best_sad = sad;
          best_row = row;
          best_col = col;
        }
      }
    }
  }
  *row0 = best_row;
  *col0 = best_col;
  return (int)best_sad;
}

/* patterns: MAX_MESH_STEP = 4 pairs {range, interval} (av1/encoder/speed_features.c:25-33) */
void orc_mesh_search_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16,
                           int bd, int w, int h, int cost_type, const int *patterns, int fine_search_interval,
                           const orc_search_block *blocks, int n, int16_t *out_mv, int32_t *out_cost, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
  for (int i = 0; i < n; ++i) {
    const orc_search_block *b = &blocks[i];
    search_ctx c;
    make_ctx(&c, src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, cost_type, b->bx, b->by, b->ref_row,
             b->ref_col);
    int interval = patterns[1], range = patterns[0];
    int br = b->start_row, bc = b->start_col, bestsme = INT_MAX;
    if (!(range < 7 || range > 256 || interval < 1 || interval > range)) {
      const int div = range / interval;
      const int m = abs(br) > abs(bc) ? abs(br) : abs(bc);
      range = range > (5 * m) / 4 ? range : (5 * m) / 4;
      range = range < 256 ? range : 256;
      interval = interval > range / div ? interval : range / div;
      if (fine_search_interval) interval = interval < 4 ? interval : 4;
      bestsme = mesh_pass(&c, b, &br, &bc, range, interval);
      if (interval > 1 && range > 7) {
        for (int k = 1; k < 4; ++k) {
          bestsme = mesh_pass(&c, b, &br, &bc, patterns[2 * k], patterns[2 * k + 1]);
          if (patterns[2 * k + 1] == 1) break;
        }
      }
      if (bestsme < INT_MAX) bestsme = var_cost_at(&c, br, bc);
    }
    out_mv[2 * i] = (int16_t)br;
    out_mv[2 * i + 1] = (int16_t)bc;
    out_cost[i] = bestsme;
  }
}

/* ---- bilinear sub-pel: av1_find_best_sub_pixel_tree_pruned_more, cost_list NULL ---- */

typedef struct {
  const search_ctx *c;
  int row_min, row_max, col_min, col_max; /* SubpelMvLimits */
  unsigned besterr, sse1;
  int distortion, best_row, best_col;
} subpel_state;

static unsigned svf_at(const search_ctx *c, int mrow, int mcol, uint32_t *sse) { /* estimated_pref_error */
  const int fr = mrow >> 3, fc = mcol >> 3; /* get_buf_from_mv: floor */
  if (c->elem16)
    return orc_highbd_sub_pixel_variance((const uint16_t *)c->ref + (ptrdiff_t)fr * c->ref_stride + fc, c->ref_stride,
                                         mcol & 7, mrow & 7, (const uint16_t *)c->src, c->src_stride, c->w, c->h,
                                         c->bd, sse);
  return orc_sub_pixel_variance((const uint8_t *)c->ref + (ptrdiff_t)fr * c->ref_stride + fc, c->ref_stride, mcol & 7,
                                mrow & 7, (const uint8_t *)c->src, c->src_stride, c->w, c->h, sse);
}

static unsigned check_better_fast(subpel_state *s, int mrow, int mcol) {
  if (mcol < s->col_min || mcol > s->col_max || mrow < s->row_min || mrow > s->row_max) return INT_MAX;
  uint32_t sse;
  const int thismse = (int)svf_at(s->c, mrow, mcol, &sse);
  const unsigned cost = (unsigned)mv_cost_var(s->c, mrow, mcol) + (unsigned)thismse;
  if (cost < s->besterr) {
    s->besterr = cost;
    s->best_row = mrow;
    s->best_col = mcol;
    s->distortion = thismse;
    s->sse1 = sse;
  }
  return cost;
}

static void two_level_checks_fast(subpel_state *s, int trow, int tcol, int hstep, int iters) {
  /* first_level_check_fast */
  const unsigned left = check_better_fast(s, trow, tcol - hstep);
  const unsigned right = check_better_fast(s, trow, tcol + hstep);
  const unsigned up = check_better_fast(s, trow - hstep, tcol);
  const unsigned down = check_better_fast(s, trow + hstep, tcol);
  const int drow = up <= down ? -hstep : hstep, dcol = left <= right ? -hstep : hstep; /* get_best_diag_step */
  check_better_fast(s, trow + drow, tcol + dcol);
  if (iters <= 1) return;
  /* second_level_check_fast */
  const int br = s->best_row, bc = s->best_col;
  if (trow != br && tcol != bc) {
    check_better_fast(s, br, bc + dcol);
    check_better_fast(s, br + drow, bc);
  } else if (trow == br && tcol != bc) {
    check_better_fast(s, br + hstep, bc + dcol);
    check_better_fast(s, br - hstep, bc + dcol);
    check_better_fast(s, br - drow, bc);
  } else if (trow != br && tcol == bc) {
    check_better_fast(s, br + drow, bc + hstep);
    check_better_fast(s, br + drow, bc - hstep);
    check_better_fast(s, br, bc - dcol);
  }
}

/* start mv = start_row/start_col in 1/8 pel; limits = SubpelMvLimits; forced_stop: 0 EIGHTH 1 QUARTER 2 HALF 3 FULL */
void orc_subpel_bilinear_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride,
                               int elem16, int bd, int w, int h, int cost_type, int iters_per_step, int allow_hp,
                               int forced_stop, const orc_subpel_block *blocks, int n, int16_t *out_mv,
                               uint32_t *out_err, int32_t *out_distortion, uint32_t *out_sse, int threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
  for (int i = 0; i < n; ++i) {
    const orc_subpel_block *b = &blocks[i];
    search_ctx c;
    make_ctx(&c, src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, cost_type, b->bx, b->by, b->ref_row,
             b->ref_col);
    subpel_state s = { &c, b->row_min, b->row_max, b->col_min, b->col_max, INT_MAX, 0, 0, b->start_row, b->start_col };
    /* setup_center_error: vf(ref at the full-pel part, src) -- note the operand order */
    {
      uint32_t sse;
      unsigned v;
      const int fr = b->start_row >> 3, fc = b->start_col >> 3;
      if (elem16)
        v = orc_highbd_variance((const uint16_t *)c.ref + (ptrdiff_t)fr * ref_stride + fc, ref_stride,
                                (const uint16_t *)c.src, src_stride, w, h, bd, &sse, NULL);
      else
        v = orc_variance((const uint8_t *)c.ref + (ptrdiff_t)fr * ref_stride + fc, ref_stride, (const uint8_t *)c.src,
                         src_stride, w, h, &sse, NULL);
      s.distortion = (int)v;
      s.sse1 = sse;
      s.besterr = v + (unsigned)mv_cost_var(&c, b->start_row, b->start_col);
    }
    int hstep = 4; /* INIT_SUBPEL_STEP_SIZE */
    if (forced_stop != 3) {
      two_level_checks_fast(&s, b->start_row, b->start_col, hstep, iters_per_step);
      if (forced_stop < 2) {
        hstep >>= 1;
        two_level_checks_fast(&s, s.best_row, s.best_col, hstep, iters_per_step);
      }
      if (allow_hp && forced_stop == 0) {
        hstep >>= 1;
        two_level_checks_fast(&s, s.best_row, s.best_col, hstep, iters_per_step);
      }
    }
    out_mv[2 * i] = (int16_t)s.best_row;
    out_mv[2 * i + 1] = (int16_t)s.best_col;
    out_err[i] = s.besterr;
    out_distortion[i] = s.distortion;
    out_sse[i] = s.sse1;
  }
}
