/*
 * oracle/aomref_lpf.c -- AV1 deblocking filter taps and a whole-plane driver.
 * TEST INFRASTRUCTURE ONLY (see aomref.h).  Restates aom_dsp/loopfilter.c (8-bit :20-511, highbd
 * :515-997) and the threshold rule of av1/common/av1_loopfilter.c:47-66,118-120.
 *
 * One code path serves every bit depth: the 8-bit functions are the highbd ones with bd = 8
 * (`x ^ 0x80` == x - 128, thresholds << 0).  The flat filters are written as what they are --
 * box filters with a doubled centre over an edge-replicated window:
 *   filter6 : 5 taps [1 2 2 2 1]          on p2..q2, >> 3
 *   filter8 : 7 taps [1 1 1 2 1 1 1]      on p3..q3, >> 3
 *   filter14: 13 taps [1 1 1 1 1 2 2 2 1 1 1 1 1] on p6..q6, >> 4
 * tests/test_oracle_lpf_cdef.py checks these against the reference's literal tap listings by
 * evaluating the reference's own statements.
 *
 * PINNED by interpreting the reference's aom_[highbd_]lpf_{horizontal,vertical}_{4,6,8,14}_c (720 cases, 8/10/12-bit)
 * and update_sharpness (every level x sharpness): tests/golden/ref_eval_lpf.npz, ref_eval_tables.npz, checked bit for
 * bit in tests/test_golden_ref_eval.py.  (The reference's own lpf gtests are SIMD-vs-C only, test/lpf_test.cc.)
 */
#include "aomref.h"

#include <stdlib.h>
#include <string.h>

static int iabs(int v) { return v < 0 ? -v : v; }
static int clampi(int v, int lo, int hi) { return v < lo ? lo : v > hi ? hi : v; }

/* x points at q0; x[-1] = p0 ... x[-7] = p6, x[6] = q6 (only the taps `len` needs are touched) */
static void lpf_taps(int *x, int len, int blimit, int limit, int thresh, int bd) {
  const int sh = bd - 8;
  const int lim = limit << sh, blim = blimit << sh, thr = thresh << sh, one = 1 << sh;
  const int p1 = x[-2], p0 = x[-1], q0 = x[0], q1 = x[1];
  /* filter_mask2 / filter_mask3_chroma / filter_mask: "apply any filter at all" */
  int mask = !(iabs(p1 - p0) > lim || iabs(q1 - q0) > lim || iabs(p0 - q0) * 2 + iabs(p1 - q1) / 2 > blim);
  const int reach = len == 4 ? 2 : len == 6 ? 3 : 4; /* pixels per side examined by the mask */
  for (int k = 2; k < reach; ++k)
    if (iabs(x[-k - 1] - x[-k]) > lim || iabs(x[k] - x[k - 1]) > lim) mask = 0;
  /* flat_mask3_chroma / flat_mask4 with thresh 1 */
  int flat = 0, flat2 = 0;
  if (len >= 6) {
    flat = 1;
    for (int k = 1; k < reach; ++k)
      if (iabs(x[-k - 1] - p0) > one || iabs(x[k] - q0) > one) flat = 0;
  }
  if (len == 14) {
    flat2 = 1;
    for (int k = 4; k < 7; ++k)
      if (iabs(x[-k - 1] - p0) > one || iabs(x[k] - q0) > one) flat2 = 0;
  }
  if (len == 14 && flat2 && flat && mask) {
    int o[12];
    for (int i = -6; i <= 5; ++i) {
      int s = 8;
      for (int k = -6; k <= 6; ++k) s += ((k >= -1 && k <= 1) ? 2 : 1) * x[clampi(i + k, -7, 6)];
      o[i + 6] = s >> 4;
    }
    for (int i = -6; i <= 5; ++i) x[i] = o[i + 6];
  } else if (len >= 8 && flat && mask) {
    int o[6];
    for (int i = -3; i <= 2; ++i) {
      int s = 4 + x[i];
      for (int k = -3; k <= 3; ++k) s += x[clampi(i + k, -4, 3)];
      o[i + 3] = s >> 3;
    }
    for (int i = -3; i <= 2; ++i) x[i] = o[i + 3];
  } else if (len == 6 && flat && mask) {
    int o[4];
    for (int i = -2; i <= 1; ++i) {
      int s = 4;
      for (int k = -2; k <= 2; ++k) s += ((k >= -1 && k <= 1) ? 2 : 1) * x[clampi(i + k, -3, 2)];
      o[i + 2] = s >> 3;
    }
    for (int i = -2; i <= 1; ++i) x[i] = o[i + 2];
  } else {
    /* filter4 / highbd_filter4 in the offset domain */
    const int off = 0x80 << sh, lo = -(128 << sh), hi = (128 << sh) - 1;
    const int ps1 = p1 - off, ps0 = p0 - off, qs0 = q0 - off, qs1 = q1 - off;
    const int hev = (iabs(p1 - p0) > thr || iabs(q1 - q0) > thr);
    int f = hev ? clampi(ps1 - qs1, lo, hi) : 0;
    f = mask ? clampi(f + 3 * (qs0 - ps0), lo, hi) : 0;
    const int f1 = clampi(f + 4, lo, hi) >> 3;
    const int f2 = clampi(f + 3, lo, hi) >> 3;
    x[0] = clampi(qs0 - f1, lo, hi) + off;
    x[-1] = clampi(ps0 + f2, lo, hi) + off;
    const int f3 = hev ? 0 : ((f1 + 1) >> 1);
    x[1] = clampi(qs1 - f3, lo, hi) + off;
    x[-2] = clampi(ps1 + f3, lo, hi) + off;
  }
}

void orc_lpf_thresholds(int level, int sharpness, uint8_t *mblim, uint8_t *lim, uint8_t *hev_thr) {
  int inside = level >> ((sharpness > 0) + (sharpness > 4));
  if (sharpness > 0 && inside > 9 - sharpness) inside = 9 - sharpness;
  if (inside < 1) inside = 1;
  *lim = (uint8_t)inside;
  *mblim = (uint8_t)(2 * (level + 2) + inside);
  *hev_thr = (uint8_t)(level >> 4);
}

static void lpf_unit(void *s, int pitch, int elem16, int vertical, int len, int blimit, int limit, int thresh,
                     int bd) {
  const int reach = len == 14 ? 7 : len == 8 ? 4 : len == 6 ? 3 : 2;
  for (int i = 0; i < 4; ++i) {
    int w[14];
    int *x = w + 7;
    const ptrdiff_t along = vertical ? (ptrdiff_t)i * pitch : i, across = vertical ? 1 : pitch;
    for (int k = -reach; k < reach; ++k)
      x[k] = elem16 ? ((uint16_t *)s)[along + k * across] : ((uint8_t *)s)[along + k * across];
    lpf_taps(x, len, blimit, limit, thresh, bd);
    for (int k = -reach; k < reach; ++k) {
      if (elem16)
        ((uint16_t *)s)[along + k * across] = (uint16_t)x[k];
      else
        ((uint8_t *)s)[along + k * across] = (uint8_t)x[k];
    }
  }
}

void orc_lpf(uint8_t *s, int pitch, int vertical, int len, uint8_t blimit, uint8_t limit, uint8_t thresh) {
  lpf_unit(s, pitch, 0, vertical, len, blimit, limit, thresh, 8);
}
void orc_highbd_lpf(uint16_t *s, int pitch, int vertical, int len, uint8_t blimit, uint8_t limit, uint8_t thresh,
                    int bd) {
  lpf_unit(s, pitch, 1, vertical, len, blimit, limit, thresh, bd);
}

/* Whole-plane driver.  `params` has one entry per 4x4 unit of the plane, {len_v, lvl_v, len_h, lvl_h}:
 * the filter length / level of the vertical edge on the unit's LEFT side and of the horizontal edge on its
 * TOP side (what set_lpf_parameters, av1_loopfilter.c:223-328, derives from the mode info).
 * order 0: the reference's single-thread order -- per 64-row superblock row, every vertical edge of the row
 *          (superblock by superblock), then every horizontal edge of the row (thread_common.c:251-322,375-395);
 * order 1: every vertical edge of the plane, then every horizontal edge (what the HIP path does). */
void orc_deblock_plane(void *plane, int stride, int width, int height, int elem16, int bd, const uint8_t *params,
                       int units_stride, int sharpness, int order) {
  const int ucols = (width + 3) / 4, urows = (height + 3) / 4;
  const size_t esz = elem16 ? 2 : 1;
#define EDGE(dir, uy, ux)                                                                                       \
  do {                                                                                                          \
    const uint8_t *e = params + ((size_t)(uy)*units_stride + (ux)) * 4 + ((dir) ? 2 : 0);                       \
    if (e[0] && e[1]) {                                                                                         \
      uint8_t mbl, lim, hev;                                                                                    \
      orc_lpf_thresholds(e[1], sharpness, &mbl, &lim, &hev);                                                    \
      lpf_unit((char *)plane + ((size_t)(uy)*4 * stride + (size_t)(ux)*4) * esz, stride, elem16, !(dir), e[0],  \
               mbl, lim, hev, bd);                                                                              \
    }                                                                                                           \
  } while (0)
  if (order == 1) {
    for (int uy = 0; uy < urows; ++uy)
      for (int ux = 0; ux < ucols; ++ux) EDGE(0, uy, ux);
    for (int uy = 0; uy < urows; ++uy)
      for (int ux = 0; ux < ucols; ++ux) EDGE(1, uy, ux);
  } else {
    const int sbu = 16; /* 64 / 4 */
    for (int sy = 0; sy < urows; sy += sbu) {
      const int y1 = sy + sbu < urows ? sy + sbu : urows;
      for (int dir = 0; dir < 2; ++dir)                /* loop_filter_rows: dir 0 then dir 1 per SB row */
        for (int sx = 0; sx < ucols; sx += sbu)        /* av1_thread_loop_filter_rows: superblocks left to right */
          for (int uy = sy; uy < y1; ++uy)
            for (int ux = sx; ux < sx + sbu && ux < ucols; ++ux) EDGE(dir, uy, ux);
    }
  }
#undef EDGE
}
