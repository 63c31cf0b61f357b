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
 * PINNED by interpreting the reference's mcomp.c itself (full_pixel_diamond, av1_full_pixel_search for all 11 search
 * methods, full_pixel_exhaustive, the sub-pel trees; 8/10-bit; every MV cost type; cost lists; second-best MVs) with
 * its own site builders and its own SAD / variance functions in the vtable: tests/golden/ref_eval_mcomp.npz
 * (generator tests/golden/gen_ref_eval_mcomp.py), checked bit for bit in tests/test_golden_ref_eval.py.
 */
#include "aomref.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int16_t bx, by, start_row, start_col, ref_row, ref_col, row_min, row_max, col_min, col_max; } orc_search_block;
typedef struct { int16_t bx, by, start_row, start_col, ref_row, ref_col, row_min, row_max, col_min, col_max; } orc_subpel_block;

enum { ORC_MV_COST_ENTROPY, ORC_MV_COST_L1_LOWRES, ORC_MV_COST_L1_MIDRES, ORC_MV_COST_L1_HDRES, ORC_MV_COST_NONE };

typedef struct {
  const void *src, *ref; /* pixel (0,0) of the block in src; pixel (0,0)+block origin of the ref plane */
  int src_stride, ref_stride, elem16, bd, w, h, cost_type;
  int ref_row, ref_col; /* ref_mv in 1/8 pel */
  /* MV_COST_ENTROPY inputs (MV_COST_PARAMS, mcomp.h:70-84): joint[4], two tables addressed from their CENTRE */
  const int *mvjcost, *mvcost[2];
  int sad_per_bit, error_per_bit;
  int skip_sad; /* ms_params->sdf is the vtable's sdsf (rows skipped): av1_make_default_fullpel_ms_params, mcomp.c:122-133 */
  /* compound sub-pel search (ms_buffers.second_pred / mask / inv_mask, av1_set_ms_compound_refs): the W x H predictor of the other
   * reference (contiguous) and, for a masked compound, W x H blend weights 0..64; NULL = single reference */
  const void *second_pred;
  const uint8_t *cmask;
  int invert_mask;
  int up_taps; /* SUBPEL_SEARCH_TYPE of the up-sampled prediction error: 1 USE_2_TAPS, 2 USE_4_TAPS, 3 (and 0: default) USE_8_TAPS */
} search_ctx;

static unsigned sad_at(const search_ctx *c, int row, int col) { /* ms_params->sdf */
  if (c->skip_sad) {
    if (c->elem16)
      return orc_highbd_sad_skip((const uint16_t *)c->src, c->src_stride,
                                 (const uint16_t *)c->ref + (ptrdiff_t)row * c->ref_stride + col, c->ref_stride, c->w,
                                 c->h, c->bd);
    return orc_sad_skip((const uint8_t *)c->src, c->src_stride,
                        (const uint8_t *)c->ref + (ptrdiff_t)row * c->ref_stride + col, c->ref_stride, c->w, c->h);
  }
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
    case ORC_MV_COST_ENTROPY: { /* ROUND_POWER_OF_TWO((unsigned)mv_cost(diff) * sad_per_bit, AV1_PROB_COST_SHIFT = 9) */
      const int dr = (row - frr) * 8, dc = (col - frc) * 8;
      const unsigned bits = (unsigned)(c->mvjcost[(dc != 0) | ((dr != 0) << 1)] + c->mvcost[0][dr] + c->mvcost[1][dc]);
      return (int)((bits * (unsigned)c->sad_per_bit + 256u) >> 9);
    }
    case ORC_MV_COST_L1_LOWRES: return (32 * d) >> 3;
    case ORC_MV_COST_L1_MIDRES: return (15 * d) >> 3;
    case ORC_MV_COST_L1_HDRES: return (8 * d) >> 3;
    default: return 0;
  }
}
static int mv_cost_var(const search_ctx *c, int mrow, int mcol) { /* mv_err_cost_, mv in 1/8 pel */
  const int d = abs(mrow - c->ref_row) + abs(mcol - c->ref_col);
  switch (c->cost_type) {
    case ORC_MV_COST_ENTROPY: { /* ROUND_POWER_OF_TWO_64((int64)mv_cost(diff) * error_per_bit, 7 + 9 - 6 + 4) */
      if (!c->mvcost[0]) return 0;
      const int dr = mrow - c->ref_row, dc = mcol - c->ref_col;
      const int64_t bits = c->mvjcost[(dc != 0) | ((dr != 0) << 1)] + c->mvcost[0][dr] + c->mvcost[1][dc];
      return (int)((bits * c->error_per_bit + (1 << 13)) >> 14);
    }
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

/* get_mvpred_compound_sad (mcomp.c:710-731) / get_mvpred_compound_var_cost (:676-708): what diamond_search_sad and full_pixel_diamond call --
 * vfp->msdf / msvf with a mask, vfp->sdaf / svaf with a second predictor only, the plain sdf / vf of a single reference otherwise.  (The mesh
 * passes, the pattern searches and the cost list keep get_mvpred_sad / get_mvpred_var_cost, i.e. the plain forms, even on a compound.) */
static unsigned diamond_sad_at(const search_ctx *c, int row, int col) {
  if (!c->second_pred) return sad_at(c, row, col);
  const void *rp = (const char *)c->ref + ((ptrdiff_t)row * c->ref_stride + col) * (ptrdiff_t)(c->elem16 ? 2 : 1);
  if (c->cmask)
    return orc_masked_sad(c->src, c->src_stride, rp, c->ref_stride, c->second_pred, c->cmask, c->w, c->invert_mask, c->w, c->h, c->elem16, c->bd);
  return orc_sad_avg_any(c->src, c->src_stride, rp, c->ref_stride, c->second_pred, c->w, c->h, c->elem16, c->bd, 0, 0);
}
static int diamond_var_cost_at(const search_ctx *c, int row, int col) {
  if (!c->second_pred) return var_cost_at(c, row, col);
  const void *rp = (const char *)c->ref + ((ptrdiff_t)row * c->ref_stride + col) * (ptrdiff_t)(c->elem16 ? 2 : 1);
  uint32_t sse;
  const uint32_t v = orc_compound_sub_pixel_variance(rp, c->ref_stride, 0, 0, c->src, c->src_stride, c->w, c->h, c->elem16, c->bd, c->cmask ? 2 : 0,
                                                     c->second_pred, 0, 0, c->cmask, c->w, c->invert_mask, &sse);
  return (int)v + mv_cost_var(c, row * 8, col * 8);
}

static void make_ctx(search_ctx *c, const void *src_origin, int src_stride, const void *ref_origin, int ref_stride,
                     int elem16, int bd, int w, int h, int cost_type, int bx, int by, int ref_row, int ref_col) {
  const size_t e = elem16 ? 2 : 1;
  c->src = (const char *)src_origin + ((ptrdiff_t)by * src_stride + bx) * e;
  c->ref = (const char *)ref_origin + ((ptrdiff_t)by * ref_stride + bx) * e;
  c->src_stride = src_stride; c->ref_stride = ref_stride; c->elem16 = elem16; c->bd = bd; c->w = w; c->h = h;
  c->cost_type = cost_type; c->ref_row = ref_row; c->ref_col = ref_col;
  c->mvjcost = NULL; c->mvcost[0] = c->mvcost[1] = NULL; c->sad_per_bit = c->error_per_bit = 0; c->skip_sad = 0;
  c->second_pred = NULL; c->cmask = NULL; c->invert_mask = 0; c->up_taps = 3;
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
static int mesh_pass2(const search_ctx *c, const orc_search_block *b, int *row0, int *col0, int range, int step, int *second) {
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
          if (second) { /* update_mvs_and_sad: *second_best_mv = *best_mv */
            second[0] = best_row;
            second[1] = best_col;
          }
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
static int mesh_pass(const search_ctx *c, const orc_search_block *b, int *row0, int *col0, int range, int step) {
  return mesh_pass2(c, b, row0, col0, range, step, NULL);
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

/* =====================================================================================================
 * av1_full_pixel_search (mcomp.c:1693-1832) for every SEARCH_METHODS value: site tables of all five builders,
 * diamond_search_sad on tables (NSTEP has 12 sites per stage), pattern_search (:998-1226) with its 3-point
 * refinement and cost_list bookkeeping, calc_int_sad_list (:768-821), the mesh follow-up rules and the
 * downsampled-SAD quality re-check.  Pinned against the interpreted reference (tests/golden/ref_eval_mcomp.npz).
 * ===================================================================================================== */

enum { SM_DIAMOND, SM_NSTEP, SM_NSTEP_8PT, SM_CLAMPED_DIAMOND, SM_HEX, SM_BIGDIA, SM_SQUARE, SM_FAST_HEX, SM_FAST_DIAMOND,
       SM_FAST_BIGDIA, SM_VFAST_DIAMOND, SM_COUNT, /* SEARCH_METHODS, mcomp_structs.h:50-83 */
       SM_NSTEP_FPF = SM_COUNT, /* NSTEP on the first-pass site table (av1_init_motion_fpf, mcomp.c:391-431; firstpass.c:261-299) */
       SM_ALL };

/* search_site_config (mcomp_structs.h:36-48) without the stride-dependent offsets */
typedef struct {
  int num_search_steps;
  int searches_per_step[22];
  int radius[22];
  int16_t mv[22][17][2]; /* row, col */
} orc_sites;

static void site_set(orc_sites *s, int stage, int idx, int row, int col) {
  s->mv[stage][idx][0] = (int16_t)row;
  s->mv[stage][idx][1] = (int16_t)col;
}

/* The 13-entry site list shared by the diamond / n-step builders (mcomp.c:366-370,405-419,452-466): centre, the four
 * axis points, then pairs at (+-radius, +-t) and (+-t, +-radius); with t == radius the first eight are the diamond's. */
static void step_sites(orc_sites *s, int stage, int radius, int t, int npts) {
  const int m[13][2] = { { 0, 0 },        { -radius, 0 }, { radius, 0 },  { 0, -radius }, { 0, radius },
                         { -radius, -t }, { radius, t },  { -t, radius }, { t, -radius }, { -radius, t },
                         { radius, -t },  { t, radius },  { -t, -radius } };
  for (int i = 0; i <= npts; ++i) site_set(s, stage, i, m[i][0], m[i][1]);
  s->searches_per_step[stage] = npts;
  s->radius[stage] = radius;
}

void orc_init_search_sites(int method, orc_sites *s) {
  static const uint8_t lookup[SM_ALL] = { SM_DIAMOND, SM_NSTEP,  SM_NSTEP_8PT, SM_CLAMPED_DIAMOND, SM_HEX,   SM_BIGDIA,
                                            SM_SQUARE,  SM_HEX,    SM_BIGDIA,    SM_BIGDIA,          SM_BIGDIA, SM_NSTEP_FPF }; /* mcomp.h:199-211 */
  memset(s, 0, sizeof(*s));
  switch (lookup[method]) {
    case SM_DIAMOND:
    case SM_CLAMPED_DIAMOND: { /* av1_init_dsmotion_compensation, level = CLAMPED */
      const int level = lookup[method] == SM_CLAMPED_DIAMOND;
      int stage = 10, n = 0;
      for (int radius = level ? 1024 / 4 : 1024; radius > 0;) {
        /* the diamond's diagonal sites are (-r,-r), (r,r), (-r,r), (r,-r): the n-step list with t = radius */
        step_sites(s, stage, radius, radius, 8);
        if (!level || (stage < 9 && level)) radius /= 2;
        --stage;
        ++n;
      }
      s->num_search_steps = n;
      break;
    }
    case SM_NSTEP_FPF: { /* av1_init_motion_fpf: radius 1024 .. 1 from stage 10 down, 12 sites (8 at radius 1) */
      int stage = 10, n = 0;
      for (int radius = 1024; radius > 0; radius /= 2, --stage, ++n) {
        int t = (int)(0.41 * radius);
        if (t < 1) t = 1;
        step_sites(s, stage, radius, t, radius == 1 ? 8 : 12);
      }
      s->num_search_steps = n;
      break;
    }
    case SM_NSTEP:
    case SM_NSTEP_8PT: { /* av1_init_motion_compensation_nstep, level = NSTEP_8PT */
      const int level = lookup[method] == SM_NSTEP_8PT;
      const int num_stages = level ? 16 : 15;
      int radius = 1;
      for (int stage = 0; stage < num_stages; ++stage) {
        int t = (int)(0.41 * radius), npts = 12;
        if (t < 1) t = 1;
        if (radius <= 5 || level) {
          t = radius;
          npts = 8;
        }
        step_sites(s, stage, radius, t, npts);
        if (stage < 12) {
          const double grown = radius * 1.5 + 0.5;
          radius = (int)(grown > radius + 1 ? grown : radius + 1);
        }
      }
      s->num_search_steps = num_stages;
      break;
    }
    default: { /* the three pattern shapes; the largest step of scale i is 2^i (mcomp.c:476-633) */
      const int shape = lookup[method];
      int radius = 1;
      for (int i = 0; i < 11; ++i, radius *= 2) {
        const int r = radius, hf = radius / 2;
        int n;
        if (shape == SM_SQUARE || (shape == SM_HEX && i == 0)) {
          const int m[8][2] = { { -r, -r }, { 0, -r }, { r, -r }, { r, 0 }, { r, r }, { 0, r }, { -r, r }, { -r, 0 } };
          for (int j = 0; j < 8; ++j) site_set(s, i, j, m[j][0], m[j][1]);
          n = 8;
        } else if (shape == SM_HEX) {
          const int m[6][2] = { { -hf, -r }, { hf, -r }, { r, 0 }, { hf, r }, { -hf, r }, { -r, 0 } };
          for (int j = 0; j < 6; ++j) site_set(s, i, j, m[j][0], m[j][1]);
          n = 6;
        } else if (i == 0) { /* BIGDIA: the four closest points */
          const int m[4][2] = { { 0, -1 }, { 1, 0 }, { 0, 1 }, { -1, 0 } };
          for (int j = 0; j < 4; ++j) site_set(s, i, j, m[j][0], m[j][1]);
          n = 4;
        } else {
          const int m[8][2] = { { -hf, -hf }, { 0, -r }, { hf, -hf }, { r, 0 }, { hf, hf }, { 0, r }, { -hf, hf }, { -r, 0 } };
          for (int j = 0; j < 8; ++j) site_set(s, i, j, m[j][0], m[j][1]);
          n = 8;
        }
        s->searches_per_step[i] = n;
        s->radius[i] = radius;
      }
      s->num_search_steps = 11;
    }
  }
}

static int mv_in_range(const orc_search_block *b, int row, int col) { /* av1_is_fullmv_in_range */
  return col >= b->col_min && col <= b->col_max && row >= b->row_min && row <= b->row_max;
}
static int check_bounds(const orc_search_block *b, int row, int col, int range) { /* mcomp.c:637-643 */
  return (row - range) >= b->row_min && (row + range) <= b->row_max && (col - range) >= b->col_min && (col + range) <= b->col_max;
}
/* Number of candidates a full scan of scale `stage` really visits (mcomp.c:1046-1066,1095-1113).  Inside the limits
 * the reference takes groups of four through sdx4df and then calls calc_sad_update_bestmv(num_candidates = n % 4,
 * cand_start = 4 * (n / 4)), whose loop `for (i = cand_start; i < num_candidates; i++)` is EMPTY for n = 6: the last
 * two HEX candidates are only visited by the per-candidate path taken near the limits.  Reproduced as is. */
static int scan_count(const orc_search_block *b, int br, int bc, int stage, int n) {
  if (!check_bounds(b, br, bc, 1 << stage)) return n;
  const int groups = 4 * (n >> 2);
  return (n % 4) > groups ? (n % 4) : groups;
}

/* diamond_search_sad (mcomp.c:1299-1416) on a site table; second[2] (may be NULL) follows *second_best_mv */
static int diamond_search_sites(const search_ctx *c, const orc_search_block *b, const orc_sites *s, int search_step, int *num00,
                                int *best_row, int *best_col, int *second) {
  int row = b->start_row, col = b->start_col;
  row = row < b->row_min ? b->row_min : row > b->row_max ? b->row_max : row; /* clamp_fullmv */
  col = col < b->col_min ? b->col_min : col > b->col_max ? b->col_max : col;
  const int tot_steps = s->num_search_steps - search_step;
  *num00 = 0;
  unsigned bestsad = diamond_sad_at(c, row, col) + (unsigned)mvsad_cost(c, row, col);
  int is_off_center = 0;
  int next_step_size = tot_steps > 2 ? s->radius[tot_steps - 2] : 1;
  for (int step = tot_steps - 1; step >= 0; --step) {
    int best_site = 0;
    if (step > 0) next_step_size = s->radius[step - 1];
    /* (the all_in / sdx4df branch and the per-site branch apply the same two-stage comparison in the same order) */
    for (int idx = 1; idx <= s->searches_per_step[step]; ++idx) {
      const int r = row + s->mv[step][idx][0], cc = col + s->mv[step][idx][1];
      if (!mv_in_range(b, r, cc)) continue;
      unsigned thissad = diamond_sad_at(c, r, cc);
      if (thissad < bestsad) {
        thissad += (unsigned)mvsad_cost(c, r, cc);
        if (thissad < bestsad) {
          bestsad = thissad;
          best_site = idx;
        }
      }
    }
    if (best_site != 0) {
      if (second) {
        second[0] = row;
        second[1] = col;
      }
      row += s->mv[step][best_site][0];
      col += s->mv[step][best_site][1];
      is_off_center = 1;
    }
    if (is_off_center == 0) (*num00)++;
    if (best_site == 0) {
      while (next_step_size == s->radius[step] && step > 2) {
        ++(*num00);
        --step;
        next_step_size = s->radius[step - 1];
      }
    }
  }
  *best_row = row;
  *best_col = col;
  return (int)bestsad;
}

/* calc_int_sad_list (mcomp.c:768-821): cost_list[0] centre, then left, bottom, right, top */
static void sad_cost_list(const search_ctx *c, const orc_search_block *b, int br, int bc, int *cost_list, int has_sad) {
  static const int8_t nb[4][2] = { { 0, -1 }, { 1, 0 }, { 0, 1 }, { -1, 0 } };
  if (!has_sad) {
    cost_list[0] = (int)sad_at(c, br, bc);
    for (int i = 0; i < 4; ++i) {
      const int r = br + nb[i][0], cc = bc + nb[i][1];
      cost_list[i + 1] = mv_in_range(b, r, cc) ? (int)sad_at(c, r, cc) : INT_MAX;
    }
  }
  cost_list[0] += mvsad_cost(c, br, bc);
  for (int i = 0; i < 4; ++i)
    if (cost_list[i + 1] != INT_MAX) cost_list[i + 1] += mvsad_cost(c, br + nb[i][0], bc + nb[i][1]);
}

/* full_pixel_diamond (mcomp.c:1421-1470) */
static int full_pixel_diamond_sites(const search_ctx *c, const orc_search_block *b, const orc_sites *s, int step_param,
                                    int *cost_list, int *best_row, int *best_col, int *second) {
  int n, num00 = 0, br, bc;
  int bestsme = diamond_search_sites(c, b, s, step_param, &n, &br, &bc, second);
  if (bestsme < INT_MAX) bestsme = diamond_var_cost_at(c, br, bc);
  const int further_steps = s->num_search_steps - 1 - step_param;
  while (n < further_steps) {
    ++n;
    if (num00) {
      num00--;
    } else {
      int tr, tc;
      int thissme = diamond_search_sites(c, b, s, step_param + n, &num00, &tr, &tc, second);
      if (thissme < INT_MAX) thissme = diamond_var_cost_at(c, tr, tc);
      if (thissme < bestsme) {
        bestsme = thissme;
        br = tr;
        bc = tc;
      }
    }
  }
  if (cost_list) sad_cost_list(c, b, br, bc, cost_list, 0);
  *best_row = br;
  *best_col = bc;
  return bestsme;
}

typedef struct {
  const search_ctx *c;
  const orc_search_block *b;
  const orc_sites *s;
  unsigned bestsad, raw_bestsad;
} pat_state;

/* calc_sad4 / calc_sad / calc_sad3 / calc_sad_..._with_indices (mcomp.c:862-992) in one routine: candidates idx[0..n)
 * of scale `stage` around (br, bc) are tried in order with update_mvs_and_sad (:839-858).  Returns the winner as a
 * POSITION in idx[] (report_pos, the 3-point forms) or as the candidate index itself; -1 = none.  A candidate
 * outside the limits is skipped; where a cost_list is kept its entry is INT_MAX (written by the indexed form,
 * left at its initial INT_MAX by the range form). */
static int pat_eval(pat_state *p, int br, int bc, int stage, const int *idx, int n, int report_pos, int *cost_list) {
  int best = -1;
  for (int i = 0; i < n; ++i) {
    const int index = idx[i];
    const int r = br + p->s->mv[stage][index][0], cc = bc + p->s->mv[stage][index][1];
    if (!mv_in_range(p->b, r, cc)) {
      if (cost_list) cost_list[index + 1] = INT_MAX;
      continue;
    }
    const unsigned this_sad = sad_at(p->c, r, cc);
    if (cost_list) cost_list[index + 1] = (int)this_sad;
    if (this_sad >= p->bestsad) continue;
    const unsigned sad = this_sad + (unsigned)mvsad_cost(p->c, r, cc);
    if (sad < p->bestsad) {
      p->raw_bestsad = this_sad;
      p->bestsad = sad;
      best = report_pos ? i : index;
    }
  }
  return best;
}

static int pattern_search(const search_ctx *c, const orc_search_block *b, const orc_sites *s, int search_step, int do_init_search,
                          int *cost_list, int *best_row, int *best_col) {
  static const int all[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
  const int *num_candidates = s->searches_per_step;
  const int last_is_4 = num_candidates[0] == 4;
  pat_state p = { c, b, s, UINT_MAX, UINT_MAX };
  int k = -1, st;
  if (search_step > 10) search_step = 10;
  int best_init_s = 10 - search_step; /* search_steps[] */
  int br = b->start_row, bc = b->start_col;
  br = br < b->row_min ? b->row_min : br > b->row_max ? b->row_max : br;
  bc = bc < b->col_min ? b->col_min : bc > b->col_max ? b->col_max : bc;
  if (cost_list) cost_list[0] = cost_list[1] = cost_list[2] = cost_list[3] = cost_list[4] = INT_MAX;
  int costlist_has_sad = 0;
  p.raw_bestsad = sad_at(c, br, bc);
  p.bestsad = p.raw_bestsad + (unsigned)mvsad_cost(c, br, bc);

  if (do_init_search) {
    st = best_init_s;
    best_init_s = -1;
    for (int t = 0; t <= st; ++t) {
      const int best_site = pat_eval(&p, br, bc, t, all, scan_count(b, br, bc, t, num_candidates[t]), 0, NULL);
      if (best_site == -1) continue;
      best_init_s = t;
      k = best_site;
    }
    if (best_init_s != -1) {
      br += s->mv[best_init_s][k][0];
      bc += s->mv[best_init_s][k][1];
    }
  }

  if (best_init_s != -1) {
    const int last_s = (last_is_4 && cost_list != NULL);
    int best_site = -1;
    st = best_init_s;
    for (; st >= last_s; st--) {
      if (!do_init_search || st != best_init_s) {
        best_site = pat_eval(&p, br, bc, st, all, scan_count(b, br, bc, st, num_candidates[st]), 0, NULL);
        if (best_site == -1) continue;
        br += s->mv[st][best_site][0];
        bc += s->mv[st][best_site][1];
        k = best_site;
      }
      do {
        int chk[3];
        chk[0] = (k == 0) ? num_candidates[st] - 1 : k - 1;
        chk[1] = k;
        chk[2] = (k == num_candidates[st] - 1) ? 0 : k + 1;
        best_site = pat_eval(&p, br, bc, st, chk, 3, 1, NULL);
        if (best_site != -1) {
          k = chk[best_site];
          br += s->mv[st][k][0];
          bc += s->mv[st][k][1];
        }
      } while (best_site != -1);
    }
    if (st == 0) { /* only reached with a cost_list (last_s == 1) */
      cost_list[0] = (int)p.raw_bestsad;
      costlist_has_sad = 1;
      if (!do_init_search || st != best_init_s) {
        best_site = pat_eval(&p, br, bc, 0, all, 4, 0, cost_list);
        if (best_site != -1) {
          br += s->mv[0][best_site][0];
          bc += s->mv[0][best_site][1];
          k = best_site;
        }
      }
      while (best_site != -1) {
        int chk[3];
        chk[0] = (k == 0) ? num_candidates[0] - 1 : k - 1;
        chk[1] = k;
        chk[2] = (k == num_candidates[0] - 1) ? 0 : k + 1;
        cost_list[1] = cost_list[2] = cost_list[3] = cost_list[4] = INT_MAX;
        cost_list[((k + 2) % 4) + 1] = cost_list[0];
        cost_list[0] = (int)p.raw_bestsad;
        best_site = pat_eval(&p, br, bc, 0, chk, 3, 1, cost_list);
        if (best_site != -1) {
          k = chk[best_site];
          br += s->mv[0][k][0];
          bc += s->mv[0][k][1];
        }
      }
    }
  }
  *best_row = br;
  *best_col = bc;
  if (cost_list) sad_cost_list(c, b, br, bc, cost_list, costlist_has_sad);
  return var_cost_at(c, br, bc);
}

typedef struct {
  int32_t search_method, step_param, cost_type, sad_per_bit, error_per_bit;
  int32_t skip_sad;               /* ms_params->sdf / sdx4df / sdx3df are the row-skipping forms */
  int32_t run_mesh_search, prune_mesh_search, mesh_search_mv_diff_threshold, force_mesh_thresh;
  int32_t fine_search_interval;
  int32_t mesh_patterns[8];       /* {range, interval} x MAX_MESH_STEP */
  int32_t no_cost_list;           /* the caller passes cost_list == NULL (changes pattern_search's last scale, :1077) */
} orc_search_params;

static int mesh_pass2(const search_ctx *c, const orc_search_block *b, int *row0, int *col0, int range, int step, int *second);

static int full_pixel_exhaustive(const search_ctx *c, const orc_search_block *b, const orc_search_params *q, int srow, int scol,
                                 int *cost_list, int *best_row, int *best_col, int *second) {
  int interval = q->mesh_patterns[1], range = q->mesh_patterns[0];
  int br = srow, bc = scol, bestsme;
  *best_row = br;
  *best_col = bc;
  if (range < 7 || range > 256 || interval < 1 || interval > range) return INT_MAX;
  const int div = range / interval;
  const int m = abs(br) > abs(bc) ? abs(br) : abs(bc);
  range = range > (5 * m) / 4 ? range : (5 * m) / 4;
  range = range < 256 ? range : 256;
  interval = interval > range / div ? interval : range / div;
  if (q->fine_search_interval) interval = interval < 4 ? interval : 4;
  bestsme = mesh_pass2(c, b, &br, &bc, range, interval, second);
  if (interval > 1 && range > 7) {
    for (int k = 1; k < 4; ++k) {
      bestsme = mesh_pass2(c, b, &br, &bc, q->mesh_patterns[2 * k], q->mesh_patterns[2 * k + 1], second);
      if (q->mesh_patterns[2 * k + 1] == 1) break;
    }
  }
  if (bestsme < INT_MAX) bestsme = var_cost_at(c, br, bc);
  if (cost_list) sad_cost_list(c, b, br, bc, cost_list, 0);
  *best_row = br;
  *best_col = bc;
  return bestsme;
}

static int log2_area_mi(int w, int h) { /* mi_size_wide_log2 + mi_size_high_log2 (4-pixel units) */
  int l = 0;
  for (int v = w / 4; v > 1; v >>= 1) ++l;
  for (int v = h / 4; v > 1; v >>= 1) ++l;
  return l;
}

static int full_pixel_search(search_ctx *c, const orc_search_block *b, const orc_search_params *q, const orc_sites *s,
                             int *cost_list, int *best_row, int *best_col, int *second) {
  int var = 0, br = -32768, bc = -32768; /* MARK_MV_INVALID */
  second[0] = second[1] = -32768;
  if (cost_list)
    for (int i = 0; i < 5; ++i) cost_list[i] = INT_MAX;
  const int m = q->search_method, sp = q->step_param;
  switch (m) {
    case SM_FAST_BIGDIA: var = pattern_search(c, b, s, sp > 8 ? sp : 8, 0, cost_list, &br, &bc); break;
    case SM_VFAST_DIAMOND: var = pattern_search(c, b, s, sp > 10 ? sp : 10, 0, cost_list, &br, &bc); break;
    case SM_FAST_DIAMOND: var = pattern_search(c, b, s, sp > 9 ? sp : 9, 0, cost_list, &br, &bc); break;
    case SM_FAST_HEX: var = pattern_search(c, b, s, sp > 9 ? sp : 9, 0, cost_list, &br, &bc); break;
    case SM_HEX:
    case SM_SQUARE:
    case SM_BIGDIA: var = pattern_search(c, b, s, sp, 1, cost_list, &br, &bc); break;
    default: var = full_pixel_diamond_sites(c, b, s, sp, cost_list, &br, &bc, second); break;
  }
  int run_mesh_search = q->run_mesh_search;
  if (!run_mesh_search && (m == SM_NSTEP || m == SM_NSTEP_8PT || m == SM_NSTEP_FPF)) {
    int thr = q->force_mesh_thresh;
    thr >>= 10 - log2_area_mi(c->w, c->h);
    if (var > thr) run_mesh_search = 1;
  }
  if (q->prune_mesh_search) { /* (is_intra_mode == 0 here) */
    const int dr = abs(b->start_row - br), dc = abs(b->start_col - bc);
    if ((dr > dc ? dr : dc) <= q->mesh_search_mv_diff_threshold) run_mesh_search = 0;
  }
  if (c->skip_sad) { /* ms_params->sdf != ms_params->vfp->sdf */
    search_ctx full = *c;
    full.skip_sad = 0;
    const int sad = (int)sad_at(&full, br, bc), skip_sad = (int)sad_at(c, br, bc);
    const int thresh = 1 << log2_area_mi(c->w, c->h);
    if (sad > thresh && abs(skip_sad - sad) * 10 >= (sad > 1 ? sad : 1) * 9)
      return full_pixel_search(&full, b, q, s, cost_list, best_row, best_col, second);
  }
  if (run_mesh_search) {
    int er, ec;
    const int var_ex = full_pixel_exhaustive(c, b, q, br, bc, cost_list, &er, &ec, second);
    if (var_ex < var) {
      var = var_ex;
      br = er;
      bc = ec;
    }
  }
  *best_row = br;
  *best_col = bc;
  return var;
}

/* outputs per block: mv[2], cost, cost_list[5], second_best[2] */
void orc_full_pixel_search_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16,
                                 int bd, int w, int h, const orc_search_params *q, const int *mvjcost, const int *mvcost0,
                                 const int *mvcost1, const orc_search_block *blocks, int n, int16_t *out_mv, int32_t *out_cost,
                                 int32_t *out_cost_list, int16_t *out_second, int threads) {
  orc_sites sites;
  orc_init_search_sites(q->search_method, &sites);
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
  for (int i = 0; i < n; ++i) {
    const orc_search_block *b = &blocks[i];
    search_ctx c;
    make_ctx(&c, src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, q->cost_type, b->bx, b->by, b->ref_row,
             b->ref_col);
    c.mvjcost = mvjcost; c.mvcost[0] = mvcost0; c.mvcost[1] = mvcost1; /* table centres */
    c.sad_per_bit = q->sad_per_bit; c.error_per_bit = q->error_per_bit; c.skip_sad = q->skip_sad;
    int br, bc, second[2], cl[5];
    for (int k = 0; k < 5; ++k) cl[k] = INT_MAX;
    const int var = full_pixel_search(&c, b, q, &sites, q->no_cost_list ? NULL : cl, &br, &bc, second);
    out_mv[2 * i] = (int16_t)br; out_mv[2 * i + 1] = (int16_t)bc;
    out_cost[i] = var;
    for (int k = 0; k < 5; ++k) out_cost_list[5 * i + k] = cl[k];
    out_second[2 * i] = (int16_t)second[0]; out_second[2 * i + 1] = (int16_t)second[1];
  }
}

/* av1_full_pixel_search with ms_buffers.second_pred [/ mask / inv_mask] set (av1_set_ms_compound_refs): the full-pel step of
 * av1_joint_motion_search when disable_extensive_joint_motion_search is 0 (motion_search_facade.c:613-619; speed 0).  second_pred: n x (w*h)
 * pixels; cmask: n x (w*h) bytes or NULL.  cost_list is NULL there; skip_sad is not a compound form (the diamond ignores ms_params->sdf). */
void orc_compound_full_pixel_search_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16, int bd, int w,
                                          int h, const orc_search_params *q, const int *mvjcost, const int *mvcost0, const int *mvcost1,
                                          const orc_search_block *blocks, int n, const void *second_pred, const uint8_t *cmask, int invert_mask,
                                          int16_t *out_mv, int32_t *out_cost, int16_t *out_second, int threads) {
  orc_sites sites;
  orc_init_search_sites(q->search_method, &sites);
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 8)
  for (int i = 0; i < n; ++i) {
    const orc_search_block *b = &blocks[i];
    search_ctx c;
    make_ctx(&c, src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, q->cost_type, b->bx, b->by, b->ref_row, b->ref_col);
    c.mvjcost = mvjcost; c.mvcost[0] = mvcost0; c.mvcost[1] = mvcost1;
    c.sad_per_bit = q->sad_per_bit; c.error_per_bit = q->error_per_bit;
    c.second_pred = (const char *)second_pred + (size_t)i * w * h * (elem16 ? 2 : 1);
    c.cmask = cmask ? cmask + (size_t)i * w * h : NULL;
    c.invert_mask = invert_mask;
    int br, bc, second[2];
    out_cost[i] = full_pixel_search(&c, b, q, &sites, NULL, &br, &bc, second);
    out_mv[2 * i] = (int16_t)br; out_mv[2 * i + 1] = (int16_t)bc;
    out_second[2 * i] = (int16_t)second[0]; out_second[2 * i + 1] = (int16_t)second[1];
  }
}

/* site tables for the tests: flat (row, col) pairs of the USED sites per stage */
int orc_search_sites_dump(int method, int *num_steps, int *per_step, int *radius, int16_t *mv /* [22][17][2] */) {
  orc_sites s;
  orc_init_search_sites(method, &s);
  *num_steps = s.num_search_steps;
  memcpy(per_step, s.searches_per_step, sizeof(s.searches_per_step));
  memcpy(radius, s.radius, sizeof(s.radius));
  memcpy(mv, s.mv, sizeof(s.mv));
  return 0;
}

/* ---- bilinear sub-pel: av1_find_best_sub_pixel_tree_pruned_more, cost_list NULL ---- */

typedef struct {
  const search_ctx *c;
  int row_min, row_max, col_min, col_max; /* SubpelMvLimits */
  unsigned besterr, sse1;
  int distortion, best_row, best_col;
  int upsampled; /* 0: estimated_pref_error (bilinear svf); 1: upsampled_pref_error with the 8-tap regular filter */
} subpel_state;

static unsigned svf_at(const search_ctx *c, int mrow, int mcol, uint32_t *sse) { /* estimated_pref_error */
  const int fr = mrow >> 3, fc = mcol >> 3; /* get_buf_from_mv: floor */
  if (c->second_pred) { /* vfp->svaf / msvf (:2311-2337) */
    const void *rp = (const char *)c->ref + ((ptrdiff_t)fr * c->ref_stride + fc) * (ptrdiff_t)(c->elem16 ? 2 : 1);
    return orc_compound_sub_pixel_variance(rp, c->ref_stride, mcol & 7, mrow & 7, c->src, c->src_stride, c->w, c->h, c->elem16, c->bd, c->cmask ? 2 : 0,
                                           c->second_pred, 0, 0, c->cmask, c->w, c->invert_mask, sse);
  }
  if (c->elem16)
    return orc_highbd_sub_pixel_variance((const uint16_t *)c->ref + (ptrdiff_t)fr * c->ref_stride + fc, c->ref_stride,
                                         mcol & 7, mrow & 7, (const uint16_t *)c->src, c->src_stride, c->w, c->h,
                                         c->bd, sse);
  return orc_sub_pixel_variance((const uint8_t *)c->ref + (ptrdiff_t)fr * c->ref_stride + fc, c->ref_stride, mcol & 7,
                                mrow & 7, (const uint8_t *)c->src, c->src_stride, c->w, c->h, sse);
}

/* EIGHTTAP_REGULAR sub-pel kernels (AV1 spec, av1/common/filter.h:124-141); a 1/8-pel offset s uses row 2 * s */
static const int16_t k_sub_pel_8[16][8] = {
  { 0, 0, 0, 128, 0, 0, 0, 0 },      { 0, 2, -6, 126, 8, -2, 0, 0 },    { 0, 2, -10, 122, 18, -4, 0, 0 },
  { 0, 2, -12, 116, 28, -8, 2, 0 },  { 0, 2, -14, 110, 38, -10, 2, 0 }, { 0, 2, -14, 102, 48, -12, 2, 0 },
  { 0, 2, -16, 94, 58, -12, 2, 0 },  { 0, 2, -14, 84, 66, -12, 2, 0 },  { 0, 2, -14, 76, 76, -14, 2, 0 },
  { 0, 2, -12, 66, 84, -14, 2, 0 },  { 0, 2, -12, 58, 94, -16, 2, 0 },  { 0, 2, -12, 48, 102, -14, 2, 0 },
  { 0, 2, -10, 38, 110, -14, 2, 0 }, { 0, 2, -8, 28, 116, -12, 2, 0 },  { 0, 0, -4, 18, 122, -10, 2, 0 },
  { 0, 0, -2, 8, 126, -6, 2, 0 }
};
const int16_t *orc_sub_pel_filters_8(void) { return &k_sub_pel_8[0][0]; }
/* av1_get_filter(subpel_search_type) (av1/common/filter.h:270-279): USE_2_TAPS -> av1_interp_4tap[BILINEAR] = av1_bilinear_filters (:111-121),
 * USE_4_TAPS -> av1_interp_4tap[EIGHTTAP_REGULAR] = av1_sub_pel_filters_4 (:205-215), USE_8_TAPS -> av1_sub_pel_filters_8; all stored as 8 taps */
static const int16_t k_sub_pel_4[16][8] = {
  { 0, 0, 0, 128, 0, 0, 0, 0 },     { 0, 0, -4, 126, 8, -2, 0, 0 },   { 0, 0, -8, 122, 18, -4, 0, 0 },  { 0, 0, -10, 116, 28, -6, 0, 0 },
  { 0, 0, -12, 110, 38, -8, 0, 0 }, { 0, 0, -12, 102, 48, -10, 0, 0 }, { 0, 0, -14, 94, 58, -10, 0, 0 }, { 0, 0, -12, 84, 66, -10, 0, 0 },
  { 0, 0, -12, 76, 76, -12, 0, 0 }, { 0, 0, -10, 66, 84, -12, 0, 0 }, { 0, 0, -10, 58, 94, -14, 0, 0 }, { 0, 0, -10, 48, 102, -12, 0, 0 },
  { 0, 0, -8, 38, 110, -12, 0, 0 }, { 0, 0, -6, 28, 116, -10, 0, 0 }, { 0, 0, -4, 18, 122, -8, 0, 0 },  { 0, 0, -2, 8, 126, -4, 0, 0 }
};
static const int16_t k_bilinear_8[16][8] = { /* av1_bilinear_filters (:111-121) */
  { 0, 0, 0, 128, 0, 0, 0, 0 }, { 0, 0, 0, 120, 8, 0, 0, 0 }, { 0, 0, 0, 112, 16, 0, 0, 0 }, { 0, 0, 0, 104, 24, 0, 0, 0 }, { 0, 0, 0, 96, 32, 0, 0, 0 }, { 0, 0, 0, 88, 40, 0, 0, 0 }, { 0, 0, 0, 80, 48, 0, 0, 0 }, { 0, 0, 0, 72, 56, 0, 0, 0 }, { 0, 0, 0, 64, 64, 0, 0, 0 }, { 0, 0, 0, 56, 72, 0, 0, 0 }, { 0, 0, 0, 48, 80, 0, 0, 0 }, { 0, 0, 0, 40, 88, 0, 0, 0 }, { 0, 0, 0, 32, 96, 0, 0, 0 }, { 0, 0, 0, 24, 104, 0, 0, 0 }, { 0, 0, 0, 16, 112, 0, 0, 0 }, { 0, 0, 0, 8, 120, 0, 0, 0 }
};
static const int16_t *up_kernel(const search_ctx *c, int phase8) { /* the kernel of a 1/8-pel offset: row 2 * offset of the 16-phase table */
  return c->up_taps == 1 ? k_bilinear_8[2 * phase8] : c->up_taps == 2 ? k_sub_pel_4[2 * phase8] : k_sub_pel_8[2 * phase8];
}

static int ref_px(const search_ctx *c, int row, int col) {
  return c->elem16 ? ((const uint16_t *)c->ref)[(ptrdiff_t)row * c->ref_stride + col]
                   : ((const uint8_t *)c->ref)[(ptrdiff_t)row * c->ref_stride + col];
}

/* aom_[highbd_]upsampled_pred_c (av1/encoder/reconinter_enc.c:424-505,562-640), unscaled reference, USE_8_TAPS:
 * aom_convolve8_horiz then _vert (aom_dsp/aom_convolve.c:36-108,181-253), each pass rounded by FILTER_BITS = 7 and
 * clipped to the pixel range; a zero offset in one direction skips that pass.  pred: w * h, row pitch w. */
static void upsampled_pred8(const search_ctx *c, int mrow, int mcol, uint16_t *pred) {
  const int fr = mrow >> 3, fc = mcol >> 3, sx = mcol & 7, sy = mrow & 7;
  const int16_t *kx = up_kernel(c, sx), *ky = up_kernel(c, sy);
  const int mx = c->elem16 ? (1 << c->bd) - 1 : 255;
  const int w = c->w, h = c->h;
  if (!sx && !sy) {
    for (int r = 0; r < h; ++r)
      for (int x = 0; x < w; ++x) pred[r * w + x] = (uint16_t)ref_px(c, fr + r, fc + x);
    return;
  }
  uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)(h + 7) * w); /* rows -3 .. h + 3 */
  for (int r = -3; r < h + 4; ++r)
    for (int x = 0; x < w; ++x) {
      int v;
      if (sx) {
        int sum = 0;
        for (int k = 0; k < 8; ++k) sum += ref_px(c, fr + r, fc + x - 3 + k) * kx[k];
        v = (sum + 64) >> 7;
        v = v < 0 ? 0 : v > mx ? mx : v;
      } else {
        v = ref_px(c, fr + r, fc + x);
      }
      tmp[(r + 3) * w + x] = (uint16_t)v;
    }
  for (int r = 0; r < h; ++r)
    for (int x = 0; x < w; ++x) {
      int v;
      if (sy) {
        int sum = 0;
        for (int k = 0; k < 8; ++k) sum += tmp[(r + k) * w + x] * ky[k];
        v = (sum + 64) >> 7;
        v = v < 0 ? 0 : v > mx ? mx : v;
      } else {
        v = tmp[(r + 3) * w + x];
      }
      pred[r * w + x] = (uint16_t)v;
    }
  free(tmp);
}

static unsigned upsampled_err(const search_ctx *c, int mrow, int mcol, uint32_t *sse) { /* upsampled_pref_error */
  uint16_t *pred = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)c->w * c->h);
  upsampled_pred8(c, mrow, mcol, pred);
  if (c->second_pred) { /* aom_[highbd_]comp_avg_upsampled_pred / comp_mask_upsampled_pred (reconinter_enc.c:507-560, 642-...): the up-sampled
                         * block blended with second_pred as aom_comp_avg_pred / aom_comp_mask_pred do */
    for (int i = 0; i < c->w * c->h; ++i) {
      const int r = pred[i], p = c->elem16 ? ((const uint16_t *)c->second_pred)[i] : ((const uint8_t *)c->second_pred)[i];
      int v;
      if (!c->cmask) v = (p + r + 1) >> 1;
      else {
        const int m = c->cmask[i];
        v = c->invert_mask ? (m * p + (64 - m) * r + 32) >> 6 : (m * r + (64 - m) * p + 32) >> 6;
      }
      pred[i] = (uint16_t)v;
    }
  }
  unsigned v;
  if (c->elem16) {
    v = orc_highbd_variance(pred, c->w, (const uint16_t *)c->src, c->src_stride, c->w, c->h, c->bd, sse, NULL);
  } else {
    uint8_t *p8 = (uint8_t *)malloc((size_t)c->w * c->h);
    for (int i = 0; i < c->w * c->h; ++i) p8[i] = (uint8_t)pred[i];
    v = orc_variance(p8, c->w, (const uint8_t *)c->src, c->src_stride, c->w, c->h, sse, NULL);
    free(p8);
  }
  free(pred);
  return v;
}

static unsigned check_better_fast2(subpel_state *s, int mrow, int mcol, int *is_better) {
  if (mcol < s->col_min || mcol > s->col_max || mrow < s->row_min || mrow > s->row_max) return INT_MAX;
  uint32_t sse;
  const int thismse = s->upsampled ? (int)upsampled_err(s->c, mrow, mcol, &sse) : (int)svf_at(s->c, mrow, mcol, &sse);
  const unsigned cost = (unsigned)mv_cost_var(s->c, mrow, mcol) + (unsigned)thismse;
  if (cost < s->besterr) {
    s->besterr = cost;
    s->best_row = mrow;
    s->best_col = mcol;
    s->distortion = thismse;
    s->sse1 = sse;
    *is_better |= 1;
  }
  return cost;
}
static unsigned check_better_fast(subpel_state *s, int mrow, int mcol) {
  int dummy = 0;
  return check_better_fast2(s, mrow, mcol, &dummy);
}

/* first_level_check_fast (mcomp.c:2503-2543): four cardinal candidates, then the diagonal they point at */
static void first_level_check_fast(subpel_state *s, int trow, int tcol, int hstep, int *drow_out, int *dcol_out) {
  const unsigned left = check_better_fast(s, trow, tcol - hstep);
  const unsigned right = check_better_fast(s, trow, tcol + hstep);
  const unsigned up = check_better_fast(s, trow - hstep, tcol);
  const unsigned down = check_better_fast(s, trow + hstep, tcol);
  const int drow = up <= down ? -hstep : hstep, dcol = left <= right ? -hstep : hstep; /* get_best_diag_step */
  check_better_fast(s, trow + drow, tcol + dcol);
  *drow_out = drow;
  *dcol_out = dcol;
}

/* second_level_check_v2 (mcomp.c:2665-2716), bilinear branch */
static void second_level_check_v2(subpel_state *s, int trow, int tcol, int drow, int dcol) {
  if (trow == s->best_row && tcol == s->best_col) return;
  if (trow == s->best_row) drow = -drow;
  else if (tcol == s->best_col) dcol = -dcol;
  const int br = s->best_row, bc = s->best_col;
  int has_better = 0;
  check_better_fast2(s, br + drow, bc, &has_better);
  check_better_fast2(s, br, bc + dcol, &has_better);
  if (has_better) check_better_fast2(s, br + drow, bc + dcol, &has_better);
}

static void two_level_checks_fast(subpel_state *s, int trow, int tcol, int hstep, int iters) {
  int drow, dcol;
  first_level_check_fast(s, trow, tcol, hstep, &drow, &dcol);
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
    subpel_state s = { &c, b->row_min, b->row_max, b->col_min, b->col_max, INT_MAX, 0, 0, b->start_row, b->start_col, 0 };
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

/* ---- the three bilinear sub-pel trees with an optional cost list and every MV cost type ----
 * tree: 0 av1_find_best_sub_pixel_tree_pruned_more (mcomp.c:2844-2929), 1 _pruned (:2931-3067), 2 _tree (:3069-3133),
 * subpel_search_type USE_2_TAPS_ORIG (0) or, for the tree, USE_8_TAPS (3: up-sampled prediction error); unscaled
 * reference, last_mv_search_list == NULL.
 * cost_lists: 5 ints per block (what av1_full_pixel_search returned) or NULL. */
static int divide_and_round(int n, int d) { return ((n < 0) ^ (d < 0)) ? ((n - d / 2) / d) : ((n + d / 2) / d); }

/* check_repeated_mv_and_update (mcomp.c:2818-2828) on one block's last_mv_search_list: 3 x (row, col), INVALID_MV = (-32768, -32768) */
static int repeated_mv(int16_t *list, int row, int col, int iter) {
  if (!list) return 0;
  if (list[2 * iter] == row && list[2 * iter + 1] == col) return 1;
  list[2 * iter] = (int16_t)row;
  list[2 * iter + 1] = (int16_t)col;
  return 0;
}

/* mv_lists: last_mv_search_list per block (n x 3 x 2 int16, read and updated) or NULL.  A search that finds the centre of one of its
 * iterations equal to the list's entry for that iteration returns INT_MAX there, leaving bestmv / distortion / sse1 as they stand. */
static void subpel_tree_core(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16, int bd,
                             int w, int h, int tree, int subpel_search_type, int cost_type, int error_per_bit, const int *mvjcost, const int *mvcost0,
                             const int *mvcost1, int iters_per_step, int allow_hp, int forced_stop,
                             const orc_subpel_block *blocks, const int32_t *cost_lists, int n, int16_t *out_mv, uint32_t *out_err,
                             int32_t *out_distortion, uint32_t *out_sse, int threads, int16_t *mv_lists, const void *second_pred,
                             const uint8_t *cmask, int invert_mask) {
  if (threads < 1) threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 16)
  for (int i = 0; i < n; ++i) {
    const orc_subpel_block *b = &blocks[i];
    search_ctx c;
    make_ctx(&c, src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, cost_type, b->bx, b->by, b->ref_row,
             b->ref_col);
    c.mvjcost = mvjcost; c.mvcost[0] = mvcost0; c.mvcost[1] = mvcost1; c.error_per_bit = error_per_bit;
    if (second_pred) {
      c.second_pred = (const char *)second_pred + (size_t)i * w * h * (elem16 ? 2 : 1);
      c.cmask = cmask ? cmask + (size_t)i * w * h : NULL;
      c.invert_mask = invert_mask;
    }
    /* only av1_find_best_sub_pixel_tree measures with the up-sampled prediction (check_better / first_level_check,
     * :2465-2663); the pruned trees always use the bilinear estimate on an unscaled reference (check_better_fast) */
    const int upsampled = tree == 2 && subpel_search_type != 0; /* != USE_2_TAPS_ORIG (mcomp.c:2689, :3093, :3112) */
    c.up_taps = subpel_search_type;
    subpel_state s = { &c, b->row_min, b->row_max, b->col_min, b->col_max, INT_MAX, 0, 0, b->start_row, b->start_col, upsampled };
    { /* setup_center_error (:2718-2778) / upsampled_setup_center_error: vf(ref at the full-pel part, src) */
      uint32_t sse;
      unsigned v;
      const int fr = b->start_row >> 3, fc = b->start_col >> 3;
      if (c.second_pred) /* svaf / msvf at offset 0, or the up-sampled form, at the start MV (:2746-2777, :2780-2795) */
        v = upsampled ? upsampled_err(&c, b->start_row, b->start_col, &sse) : svf_at(&c, b->start_row, b->start_col, &sse);
      else if (elem16)
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
    const int sr = b->start_row, sc = b->start_col;
    int16_t *ml = mv_lists ? mv_lists + 6 * i : NULL;
    if (tree == 2) {
      int round = 3 - forced_stop; /* FULL_PEL - forced_stop */
      if (round > 3 - !allow_hp) round = 3 - !allow_hp;
      for (int iter = 0; iter < round; ++iter) {
        const int cr = s.best_row, cc = s.best_col;
        int drow, dcol;
        if (repeated_mv(ml, cr, cc, iter)) { s.besterr = INT_MAX; break; }   /* :3106-3109 */
        first_level_check_fast(&s, cr, cc, hstep, &drow, &dcol);
        if (!(cr == s.best_row && cc == s.best_col) && iters_per_step > 1) second_level_check_v2(&s, cr, cc, drow, dcol);
        hstep >>= 1;
      }
    } else if (forced_stop != 3 && repeated_mv(ml, s.best_row, s.best_col, 0)) {   /* :2874-2876, :2959-2961 */
      s.besterr = INT_MAX;
    } else if (forced_stop != 3) {
      const int32_t *cl = cost_lists ? cost_lists + 5 * i : NULL;
      const int usable = cl && cl[0] != INT_MAX && cl[1] != INT_MAX && cl[2] != INT_MAX && cl[3] != INT_MAX && cl[4] != INT_MAX;
      if (tree == 0 && usable && cl[0] < cl[1] && cl[0] < cl[2] && cl[0] < cl[3] && cl[0] < cl[4]) {
        /* get_cost_surf_min(bits = 1): minimum of the fitted paraboloid in half-pel units */
        const int ic = divide_and_round((cl[1] - cl[3]) * 1, cl[1] - 2 * cl[0] + cl[3]);
        const int ir = divide_and_round((cl[4] - cl[2]) * 1, cl[4] - 2 * cl[0] + cl[2]);
        if (ir != 0 || ic != 0) check_better_fast(&s, sr + ir * hstep, sc + ic * hstep);
      } else if (tree == 1 && usable) {
        const unsigned whichdir = (cl[1] < cl[3] ? 0 : 1) + (cl[2] < cl[4] ? 0 : 2);
        const int dc = (whichdir & 1) ? hstep : -hstep, dr = (whichdir & 2) ? -hstep : hstep; /* right : left, top : bottom */
        check_better_fast(&s, sr, sc + dc);
        check_better_fast(&s, sr + dr, sc);
        check_better_fast(&s, sr + dr, sc + dc);
      } else {
        two_level_checks_fast(&s, sr, sc, hstep, iters_per_step);
      }
      int live = 1;
      if (forced_stop < 2) {
        if (repeated_mv(ml, s.best_row, s.best_col, 1)) { s.besterr = INT_MAX; live = 0; }   /* :2899-2903 */
        else {
          hstep >>= 1;
          two_level_checks_fast(&s, s.best_row, s.best_col, hstep, iters_per_step);
        }
      }
      if (live && allow_hp && forced_stop == 0) {
        /* the third check uses the running `iter`: 2 after a quarter-pel level (the only way here: forced_stop == 0 < HALF_PEL) */
        if (repeated_mv(ml, s.best_row, s.best_col, 2)) s.besterr = INT_MAX;                  /* :2912-2916 */
        else {
          hstep >>= 1;
          two_level_checks_fast(&s, s.best_row, s.best_col, hstep, iters_per_step);
        }
      }
    }
    out_mv[2 * i] = (int16_t)s.best_row;
    out_mv[2 * i + 1] = (int16_t)s.best_col;
    out_err[i] = s.besterr;
    out_distortion[i] = s.distortion;
    out_sse[i] = s.sse1;
  }
}

void orc_subpel_tree_batch_list(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16, int bd,
                                int w, int h, int tree, int subpel_search_type, int cost_type, int error_per_bit, const int *mvjcost, const int *mvcost0,
                                const int *mvcost1, int iters_per_step, int allow_hp, int forced_stop,
                                const orc_subpel_block *blocks, const int32_t *cost_lists, int n, int16_t *out_mv, uint32_t *out_err,
                                int32_t *out_distortion, uint32_t *out_sse, int threads, int16_t *mv_lists) {
  subpel_tree_core(src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, tree, subpel_search_type, cost_type, error_per_bit, mvjcost, mvcost0,
                   mvcost1, iters_per_step, allow_hp, forced_stop, blocks, cost_lists, n, out_mv, out_err, out_distortion, out_sse, threads, mv_lists, NULL,
                   NULL, 0);
}

/* The sub-pel trees of a COMPOUND search (av1_joint_motion_search / av1_compound_single_motion_search, motion_search_facade.c:496-870: the
 * find_fractional_mv_step call with ms_buffers.second_pred [/ mask / inv_mask]): every error is vfp->svaf or msvf (bilinear form) or the
 * comp_avg / comp_mask up-sampled prediction + vf (tree 2 with USE_8_TAPS).  second_pred: n x (w*h) pixels; cmask: n x (w*h) bytes or NULL. */
void orc_compound_subpel_tree_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16, int bd, int w, int h,
                                    int tree, int subpel_search_type, int cost_type, int error_per_bit, const int *mvjcost, const int *mvcost0,
                                    const int *mvcost1, int iters_per_step, int allow_hp, int forced_stop, const orc_subpel_block *blocks, int n,
                                    const void *second_pred, const uint8_t *cmask, int invert_mask, int16_t *out_mv, uint32_t *out_err,
                                    int32_t *out_distortion, uint32_t *out_sse, int threads) {
  subpel_tree_core(src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, tree, subpel_search_type, cost_type, error_per_bit, mvjcost, mvcost0,
                   mvcost1, iters_per_step, allow_hp, forced_stop, blocks, NULL, n, out_mv, out_err, out_distortion, out_sse, threads, NULL, second_pred,
                   cmask, invert_mask);
}

void orc_subpel_tree_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16, int bd,
                           int w, int h, int tree, int subpel_search_type, int cost_type, int error_per_bit, const int *mvjcost, const int *mvcost0,
                           const int *mvcost1, int iters_per_step, int allow_hp, int forced_stop,
                           const orc_subpel_block *blocks, const int32_t *cost_lists, int n, int16_t *out_mv, uint32_t *out_err,
                           int32_t *out_distortion, uint32_t *out_sse, int threads) {
  orc_subpel_tree_batch_list(src_origin, src_stride, ref_origin, ref_stride, elem16, bd, w, h, tree, subpel_search_type, cost_type, error_per_bit, mvjcost,
                             mvcost0, mvcost1, iters_per_step, allow_hp, forced_stop, blocks, cost_lists, n, out_mv, out_err, out_distortion, out_sse,
                             threads, NULL);
}

/* =====================================================================================================================
 * Compound-reference and OBMC full-pel searches (the last search call sites of the RD path).
 *   av1_refining_search_8p_c        av1/encoder/mcomp.c:1621-1691  (+ get_mvpred_compound_sad :710-731)
 *   av1_get_mvpred_compound_var     :3679-3693 (get_mvpred_av_var :3650-3663, get_mvpred_mask_var :3665-3677)
 *   obmc_refining_search_sad        :2127-2171     obmc_diamond_search_sad :2173-2234     obmc_full_pixel_diamond :2236-2270
 *   av1_obmc_full_pixel_search      :2272-2285     get_obmc_mvpred_var :2110-2125
 * PINNED by interpreting those functions themselves: tests/golden/ref_eval_compound_search.npz
 * (generator tests/golden/gen_ref_eval_compound_search.py), checked in tests/test_golden_compound_search.py.
 * second_pred: the W x H predictor of the OTHER reference, contiguous (stride W); mask: W x H blend weights 0..64 (stride W) or NULL.
 * ===================================================================================================================== */
typedef struct {
  search_ctx c;
  const void *second_pred;
  const uint8_t *mask;
  int invert_mask;
} compound_ctx;

static unsigned compound_sad_at(const compound_ctx *cc, int row, int col) { /* get_mvpred_compound_sad: msdf / sdaf */
  const search_ctx *c = &cc->c;
  const size_t es = c->elem16 ? 2 : 1;
  const void *rp = (const char *)c->ref + ((ptrdiff_t)row * c->ref_stride + col) * (ptrdiff_t)es;
  if (cc->mask)
    return orc_masked_sad(c->src, c->src_stride, rp, c->ref_stride, cc->second_pred, cc->mask, c->w, cc->invert_mask, c->w, c->h, c->elem16, c->bd);
  return orc_sad_avg_any(c->src, c->src_stride, rp, c->ref_stride, cc->second_pred, c->w, c->h, c->elem16, c->bd, 0, 0);
}
static int compound_var_at(const compound_ctx *cc, int row, int col) { /* av1_get_mvpred_compound_var: msvf / svaf at offset 0, + mv_err_cost_ */
  const search_ctx *c = &cc->c;
  const size_t es = c->elem16 ? 2 : 1;
  const void *rp = (const char *)c->ref + ((ptrdiff_t)row * c->ref_stride + col) * (ptrdiff_t)es;
  uint32_t sse;
  const uint32_t v = orc_compound_sub_pixel_variance(rp, c->ref_stride, 0, 0, c->src, c->src_stride, c->w, c->h, c->elem16, c->bd, cc->mask ? 2 : 0,
                                                     cc->second_pred, 0, 0, cc->mask, c->w, cc->invert_mask, &sse);
  return (int)v + mv_cost_var(c, row * 8, col * 8);
}

static int refining_search_8p(const compound_ctx *cc, const orc_search_block *b, int *best_row, int *best_col) {
  static const int8_t nb[8][2] = { { -1, 0 }, { 0, -1 }, { 0, 1 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
  enum { RANGE = 3, STRIDE = 2 * RANGE + 1 }; /* SEARCH_RANGE_8P, SEARCH_GRID_STRIDE_8P (mcomp_structs.h:26-29) */
  uint8_t visited[STRIDE * STRIDE] = { 0 };
  int grid_center = RANGE * STRIDE + RANGE;
  int row = b->start_row, col = b->start_col;
  row = row < b->row_min ? b->row_min : row > b->row_max ? b->row_max : row; /* clamp_fullmv */
  col = col < b->col_min ? b->col_min : col > b->col_max ? b->col_max : col;
  unsigned best_sad = compound_sad_at(cc, row, col) + (unsigned)mvsad_cost(&cc->c, row, col);
  visited[grid_center] = 1;
  for (int i = 0; i < RANGE; ++i) {
    int best_site = -1;
    for (int j = 0; j < 8; ++j) {
      const int gc = grid_center + nb[j][0] * STRIDE + nb[j][1];
      if (visited[gc]) continue;
      const int r = row + nb[j][0], c = col + nb[j][1];
      visited[gc] = 1;
      if (c < b->col_min || c > b->col_max || r < b->row_min || r > b->row_max) continue;
      unsigned sad = compound_sad_at(cc, r, c);
      if (sad < best_sad) {
        sad += (unsigned)mvsad_cost(&cc->c, r, c);
        if (sad < best_sad) {
          best_sad = sad;
          best_site = j;
        }
      }
    }
    if (best_site == -1) break;
    row += nb[best_site][0];
    col += nb[best_site][1];
    grid_center += nb[best_site][0] * STRIDE + nb[best_site][1];
  }
  *best_row = row;
  *best_col = col;
  return (int)best_sad;
}

/* blocks[n]; second_pred: n x (w*h) pixels; mask: n x (w*h) bytes or NULL.  Outputs: best_mv[n][2], best_sad[n] (the return value of
 * av1_refining_search_8p_c), best_var[n] (av1_get_mvpred_compound_var at best_mv). */
void orc_refining_search_8p_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16, int bd, int w, int h,
                                  const void *blocks_v, int n, int cost_type, int sad_per_bit, int error_per_bit, const int *mvjcost,
                                  const int *mvcost0, const int *mvcost1, const void *second_pred, const uint8_t *mask, int invert_mask,
                                  int16_t *best_mv, int32_t *best_sad, int32_t *best_var, int threads) {
  const orc_search_block *blocks = (const orc_search_block *)blocks_v;
  const size_t es = elem16 ? 2 : 1;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 8)
  for (int i = 0; i < n; ++i) {
    const orc_search_block *b = &blocks[i];
    compound_ctx cc;
    memset(&cc, 0, sizeof(cc));
    cc.c.src = (const char *)src_origin + ((ptrdiff_t)b->by * src_stride + b->bx) * (ptrdiff_t)es;
    cc.c.ref = (const char *)ref_origin + ((ptrdiff_t)b->by * ref_stride + b->bx) * (ptrdiff_t)es;
    cc.c.src_stride = src_stride; cc.c.ref_stride = ref_stride; cc.c.elem16 = elem16; cc.c.bd = bd; cc.c.w = w; cc.c.h = h;
    cc.c.cost_type = cost_type; cc.c.ref_row = b->ref_row; cc.c.ref_col = b->ref_col;
    cc.c.mvjcost = mvjcost; cc.c.mvcost[0] = mvcost0; cc.c.mvcost[1] = mvcost1; cc.c.sad_per_bit = sad_per_bit; cc.c.error_per_bit = error_per_bit;
    cc.second_pred = (const char *)second_pred + (size_t)i * w * h * es;
    cc.mask = mask ? mask + (size_t)i * w * h : NULL;
    cc.invert_mask = invert_mask;
    int r, c;
    best_sad[i] = refining_search_8p(&cc, b, &r, &c);
    best_mv[2 * i] = (int16_t)r; best_mv[2 * i + 1] = (int16_t)c;
    best_var[i] = compound_var_at(&cc, r, c);
  }
}

typedef struct {
  search_ctx c; /* (src unused) */
  const int32_t *wsrc, *omask;
} obmc_ctx;
static unsigned obmc_sad_at(const obmc_ctx *oc, int row, int col) { /* vfp->osdf */
  const search_ctx *c = &oc->c;
  const void *rp = (const char *)c->ref + ((ptrdiff_t)row * c->ref_stride + col) * (ptrdiff_t)(c->elem16 ? 2 : 1);
  return orc_obmc_sad(rp, c->ref_stride, oc->wsrc, oc->omask, c->w, c->h, c->elem16, c->bd);
}
static int obmc_var_at(const obmc_ctx *oc, int row, int col) { /* get_obmc_mvpred_var */
  const search_ctx *c = &oc->c;
  const void *rp = (const char *)c->ref + ((ptrdiff_t)row * c->ref_stride + col) * (ptrdiff_t)(c->elem16 ? 2 : 1);
  uint32_t sse;
  return (int)orc_obmc_variance(rp, c->ref_stride, 0, 0, 0, oc->wsrc, oc->omask, c->w, c->h, c->elem16, c->bd, &sse) + mv_cost_var(c, row * 8, col * 8);
}
static int obmc_diamond(const obmc_ctx *oc, const orc_search_block *b, const orc_sites *s, int search_step, int *num00, int *best_row, int *best_col) {
  const int tot_steps = s->num_search_steps - search_step;
  int row = b->start_row, col = b->start_col;
  row = row < b->row_min ? b->row_min : row > b->row_max ? b->row_max : row;
  col = col < b->col_min ? b->col_min : col > b->col_max ? b->col_max : col;
  const int init_row = row, init_col = col;
  *num00 = 0;
  int best_sad = (int)(obmc_sad_at(oc, row, col) + (unsigned)mvsad_cost(&oc->c, row, col));
  for (int step = tot_steps - 1; step >= 0; --step) {
    int best_site = 0;
    for (int idx = 1; idx <= s->searches_per_step[step]; ++idx) {
      const int r = row + s->mv[step][idx][0], c = col + s->mv[step][idx][1];
      if (c < b->col_min || c > b->col_max || r < b->row_min || r > b->row_max) continue;
      int sad = (int)obmc_sad_at(oc, r, c); /* `int sad`, `int best_sad`: signed comparisons in this function (mcomp.c:2206-2215) */
      if (sad < best_sad) {
        sad += mvsad_cost(&oc->c, r, c);
        if (sad < best_sad) {
          best_sad = sad;
          best_site = idx;
        }
      }
    }
    if (best_site != 0) {
      row += s->mv[step][best_site][0];
      col += s->mv[step][best_site][1];
    } else if (row == init_row && col == init_col) { /* best_address == init_ref */
      (*num00)++;
    }
  }
  *best_row = row;
  *best_col = col;
  return best_sad;
}
static int obmc_full_pixel_search(const obmc_ctx *oc, const orc_search_block *b, const orc_sites *s, int step_param, int fast, int *best_row, int *best_col) {
  if (!fast) { /* obmc_full_pixel_diamond */
    int n, num00 = 0, tr, tc;
    int bestsme = obmc_diamond(oc, b, s, step_param, &n, &tr, &tc);
    if (bestsme < INT_MAX) bestsme = obmc_var_at(oc, tr, tc);
    *best_row = tr; *best_col = tc;
    const int further_steps = s->num_search_steps - 1 - step_param;
    while (n < further_steps) {
      ++n;
      if (num00) {
        num00--;
      } else {
        int thissme = obmc_diamond(oc, b, s, step_param + n, &num00, &tr, &tc);
        if (thissme < INT_MAX) thissme = obmc_var_at(oc, tr, tc);
        if (thissme < bestsme) {
          bestsme = thissme;
          *best_row = tr; *best_col = tc;
        }
      }
    }
    return bestsme;
  }
  /* obmc_refining_search_sad from the clamped start */
  static const int8_t nb[4][2] = { { -1, 0 }, { 0, -1 }, { 0, 1 }, { 1, 0 } };
  int row = b->start_row, col = b->start_col;
  row = row < b->row_min ? b->row_min : row > b->row_max ? b->row_max : row;
  col = col < b->col_min ? b->col_min : col > b->col_max ? b->col_max : col;
  unsigned best_sad = obmc_sad_at(oc, row, col) + (unsigned)mvsad_cost(&oc->c, row, col);
  for (int i = 0; i < 8; ++i) {
    int best_site = -1;
    for (int j = 0; j < 4; ++j) {
      const int r = row + nb[j][0], c = col + nb[j][1];
      if (c < b->col_min || c > b->col_max || r < b->row_min || r > b->row_max) continue;
      unsigned sad = obmc_sad_at(oc, r, c);
      if (sad < best_sad) {
        sad += (unsigned)mvsad_cost(&oc->c, r, c);
        if (sad < best_sad) {
          best_sad = sad;
          best_site = j;
        }
      }
    }
    if (best_site == -1) break;
    row += nb[best_site][0];
    col += nb[best_site][1];
  }
  *best_row = row; *best_col = col;
  int thissme = (int)best_sad;
  if (thissme < INT_MAX) thissme = obmc_var_at(oc, row, col);
  return thissme;
}
/* wsrc / obmc_mask: n x (w*h) int32 (calc_target_weighted_pred's outputs).  Outputs: best_mv[n][2], best_cost[n] (the return value). */
void orc_obmc_full_pixel_search_batch(const void *ref_origin, int ref_stride, int elem16, int bd, int w, int h, const void *blocks_v, int n, int method,
                                      int step_param, int fast_obmc_search, int cost_type, int sad_per_bit, int error_per_bit, const int *mvjcost,
                                      const int *mvcost0, const int *mvcost1, const int32_t *wsrc, const int32_t *omask, int16_t *best_mv,
                                      int32_t *best_cost, int threads) {
  const orc_search_block *blocks = (const orc_search_block *)blocks_v;
  orc_sites sites;
  orc_init_search_sites(method, &sites);
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 8)
  for (int i = 0; i < n; ++i) {
    const orc_search_block *b = &blocks[i];
    obmc_ctx oc;
    memset(&oc, 0, sizeof(oc));
    oc.c.ref = (const char *)ref_origin + ((ptrdiff_t)b->by * ref_stride + b->bx) * (ptrdiff_t)(elem16 ? 2 : 1);
    oc.c.ref_stride = ref_stride; oc.c.elem16 = elem16; oc.c.bd = bd; oc.c.w = w; oc.c.h = h;
    oc.c.cost_type = cost_type; oc.c.ref_row = b->ref_row; oc.c.ref_col = b->ref_col;
    oc.c.mvjcost = mvjcost; oc.c.mvcost[0] = mvcost0; oc.c.mvcost[1] = mvcost1; oc.c.sad_per_bit = sad_per_bit; oc.c.error_per_bit = error_per_bit;
    oc.wsrc = wsrc + (size_t)i * w * h; oc.omask = omask + (size_t)i * w * h;
    int r, c;
    best_cost[i] = obmc_full_pixel_search(&oc, b, &sites, step_param, fast_obmc_search, &r, &c);
    best_mv[2 * i] = (int16_t)r; best_mv[2 * i + 1] = (int16_t)c;
  }
}

/* ---- av1_find_best_obmc_sub_pixel_tree_up (mcomp.c:3588-3633) ----
 * subpel_search_type USE_2_TAPS_ORIG: the centre by setup_obmc_center_error (:3359-3374 -- vfp->ovf at ms_buffers->ref->buf, i.e. at MV 0
 * whatever the start MV is: the reference's own TODO) and candidates by obmc_check_better_fast (:3416-3442: vfp->osvf + estimate_obmc_mvcost,
 * :3390-3412, whose difference MV is multiplied by 8 and whose ENTROPY form rounds by 13 bits); USE_8_TAPS: centre and candidates by
 * upsampled_obmc_pref_error (:3314-3357: aom_[highbd_]upsampled_pred, then vfp->ovf on the w-pitch prediction) + mv_err_cost_. */
typedef struct {
  const obmc_ctx *oc;
  int upsampled;
  int row_min, row_max, col_min, col_max;
  unsigned besterr;
  int best_row, best_col, distortion;
  uint32_t sse1;
} obmc_subpel_state;
static unsigned obmc_upsampled_err(const obmc_ctx *oc, int mrow, int mcol, uint32_t *sse) {
  const search_ctx *c = &oc->c;
  const int n = c->w * c->h;
  uint16_t *pred = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n);
  upsampled_pred8(c, mrow, mcol, pred);
  unsigned v;
  if (c->elem16) {
    v = orc_obmc_variance(pred, c->w, 0, 0, 0, oc->wsrc, oc->omask, c->w, c->h, 1, c->bd, sse);
  } else {
    uint8_t *p8 = (uint8_t *)malloc((size_t)n);
    for (int i = 0; i < n; ++i) p8[i] = (uint8_t)pred[i];
    v = orc_obmc_variance(p8, c->w, 0, 0, 0, oc->wsrc, oc->omask, c->w, c->h, 0, c->bd, sse);
    free(p8);
  }
  free(pred);
  return v;
}
static int estimate_obmc_mvcost(const search_ctx *c, int mrow, int mcol) {
  if (c->cost_type != 0) return 0; /* MV_COST_NONE: 0; the L1 types: assert(0) in the reference, 0 in a release build */
  const int dr = (mrow - c->ref_row) * 8, dc = (mcol - c->ref_col) * 8; /* GET_MV_SUBPEL of a 1/8-pel difference */
  const int bits = c->mvjcost[(dc != 0) | ((dr != 0) << 1)] + c->mvcost[0][dr] + c->mvcost[1][dc];
  return (int)(((unsigned)bits * (unsigned)c->error_per_bit + 4096u) >> 13);
}
static unsigned obmc_check_better(obmc_subpel_state *s, int mrow, int mcol, int *has_better) {
  if (mcol < s->col_min || mcol > s->col_max || mrow < s->row_min || mrow > s->row_max) return INT_MAX;
  const search_ctx *c = &s->oc->c;
  uint32_t sse;
  int thismse;
  unsigned cost;
  if (s->upsampled) {
    thismse = (int)obmc_upsampled_err(s->oc, mrow, mcol, &sse);
    cost = (unsigned)mv_cost_var(c, mrow, mcol);
  } else {
    const void *rp = (const char *)c->ref + ((ptrdiff_t)(mrow >> 3) * c->ref_stride + (mcol >> 3)) * (ptrdiff_t)(c->elem16 ? 2 : 1);
    thismse = (int)orc_obmc_variance(rp, c->ref_stride, 1, mcol & 7, mrow & 7, s->oc->wsrc, s->oc->omask, c->w, c->h, c->elem16, c->bd, &sse);
    cost = (unsigned)estimate_obmc_mvcost(c, mrow, mcol);
  }
  cost += (unsigned)thismse;
  if (cost < s->besterr) {
    s->besterr = cost; s->best_row = mrow; s->best_col = mcol; s->distortion = thismse; s->sse1 = sse;
    *has_better |= 1;
  }
  return cost;
}
void orc_obmc_subpel_tree_batch(const void *ref_origin, int ref_stride, int elem16, int bd, int w, int h, const void *blocks_v, int n, int cost_type,
                                int error_per_bit, const int *mvjcost, const int *mvcost0, const int *mvcost1, int iters_per_step, int allow_hp,
                                int forced_stop, int upsampled, const int32_t *wsrc, const int32_t *omask, int16_t *best_mv, uint32_t *best_err,
                                int32_t *distortion, uint32_t *sse1, int threads) {
  const orc_subpel_block *blocks = (const orc_subpel_block *)blocks_v;
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(dynamic, 4)
  for (int i = 0; i < n; ++i) {
    const orc_subpel_block *b = &blocks[i];
    obmc_ctx oc;
    memset(&oc, 0, sizeof(oc));
    oc.c.ref = (const char *)ref_origin + ((ptrdiff_t)b->by * ref_stride + b->bx) * (ptrdiff_t)(elem16 ? 2 : 1);
    oc.c.ref_stride = ref_stride; oc.c.elem16 = elem16; oc.c.bd = bd; oc.c.w = w; oc.c.h = h;
    oc.c.cost_type = cost_type; oc.c.ref_row = b->ref_row; oc.c.ref_col = b->ref_col;
    oc.c.mvjcost = mvjcost; oc.c.mvcost[0] = mvcost0; oc.c.mvcost[1] = mvcost1; oc.c.error_per_bit = error_per_bit;
    oc.wsrc = wsrc + (size_t)i * w * h; oc.omask = omask + (size_t)i * w * h;
    obmc_subpel_state s;
    memset(&s, 0, sizeof(s));
    s.oc = &oc; s.upsampled = upsampled; oc.c.up_taps = upsampled; /* `upsampled` = the SUBPEL_SEARCH_TYPE, 0 = USE_2_TAPS_ORIG */
    s.row_min = b->row_min; s.row_max = b->row_max; s.col_min = b->col_min; s.col_max = b->col_max;
    s.best_row = b->start_row; s.best_col = b->start_col;
    if (upsampled) {
      s.distortion = (int)obmc_upsampled_err(&oc, s.best_row, s.best_col, &s.sse1);
    } else { /* setup_obmc_center_error: ovf at ref->buf, NOT at the start MV */
      s.distortion = (int)orc_obmc_variance(oc.c.ref, ref_stride, 0, 0, 0, oc.wsrc, oc.omask, w, h, elem16, bd, &s.sse1);
    }
    s.besterr = (unsigned)s.distortion + (unsigned)mv_cost_var(&oc.c, s.best_row, s.best_col);
    int hstep = 4; /* INIT_SUBPEL_STEP_SIZE */
    const int round = (3 - forced_stop) < (3 - !allow_hp) ? (3 - forced_stop) : (3 - !allow_hp);
    for (int iter = 0; iter < round; ++iter) {
      const int tr = s.best_row, tc = s.best_col;
      int dummy = 0;
      /* obmc_first_level_check (:3471-3533) */
      const unsigned left = obmc_check_better(&s, tr, tc - hstep, &dummy);
      const unsigned right = obmc_check_better(&s, tr, tc + hstep, &dummy);
      const unsigned up = obmc_check_better(&s, tr - hstep, tc, &dummy);
      const unsigned down = obmc_check_better(&s, tr + hstep, tc, &dummy);
      int drow = up <= down ? -hstep : hstep, dcol = left <= right ? -hstep : hstep;
      obmc_check_better(&s, tr + drow, tc + dcol, &dummy);
      if ((tr != s.best_row || tc != s.best_col) && iters_per_step > 1) { /* obmc_second_level_check_v2 (:3535-3586) */
        if (tr == s.best_row) drow = -drow;
        else if (tc == s.best_col) dcol = -dcol;
        const int br = s.best_row, bc = s.best_col;
        int has_better = 0;
        obmc_check_better(&s, br + drow, bc, &has_better);
        obmc_check_better(&s, br, bc + dcol, &has_better);
        if (has_better) obmc_check_better(&s, br + drow, bc + dcol, &has_better);
      }
      hstep >>= 1;
    }
    best_mv[2 * i] = (int16_t)s.best_row; best_mv[2 * i + 1] = (int16_t)s.best_col;
    best_err[i] = s.besterr; distortion[i] = s.distortion; sse1[i] = s.sse1;
  }
}
