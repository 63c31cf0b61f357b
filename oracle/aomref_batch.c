/*
 * oracle/aomref_batch.c -- whole-work-list drivers over the scalar oracle functions.
 * TEST INFRASTRUCTURE ONLY (see aomref.h): used as the checker for the batched HIP
 * entry points and as bench.py's `cpu_baseline` ("port", OpenMP over candidates,
 * mirroring the reference's static tile-thread partition, av1/encoder/ethread.c).
 */
#include <omp.h>

#include "aomref.h"

typedef struct { int16_t sx, sy, rx, ry; } orc_cand;            /* == aomhip_sad_cand */
typedef struct { int16_t sx, sy, rx[4], ry[4]; } orc_x4d_group;  /* == aomhip_sad_x4d_cand */

int orc_max_threads(void) { return omp_get_max_threads(); }

/* origin pointers address pixel (0,0) of a bordered plane; elem16 selects uint16 planes. */
void orc_sad_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16,
                   int bd, int w, int h, int skip, const orc_cand *c, int n, uint32_t *out, int threads, int reps) {
  if (threads < 1) threads = 1;
  if (reps < 1) reps = 1;
  /* reps > 1: the CPU-baseline leg walks the same list `reps` times inside one parallel loop */
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long long it = 0; it < (long long)n * reps; ++it) {
    const int i = (int)(it % n);
    if (!elem16) {
      const uint8_t *s = (const uint8_t *)src_origin + (ptrdiff_t)c[i].sy * src_stride + c[i].sx;
      const uint8_t *r = (const uint8_t *)ref_origin + (ptrdiff_t)c[i].ry * ref_stride + c[i].rx;
      out[i] = skip ? orc_sad_skip(s, src_stride, r, ref_stride, w, h) : orc_sad(s, src_stride, r, ref_stride, w, h);
    } else {
      const uint16_t *s = (const uint16_t *)src_origin + (ptrdiff_t)c[i].sy * src_stride + c[i].sx;
      const uint16_t *r = (const uint16_t *)ref_origin + (ptrdiff_t)c[i].ry * ref_stride + c[i].rx;
      out[i] = skip ? orc_highbd_sad_skip(s, src_stride, r, ref_stride, w, h, bd)
                    : orc_highbd_sad(s, src_stride, r, ref_stride, w, h, bd);
    }
  }
}

void orc_sad_x4d_batch(const void *src_origin, int src_stride, const void *ref_origin, int ref_stride, int elem16,
                       int bd, int w, int h, int skip, const orc_x4d_group *g, int n, uint32_t *out, int threads, int reps) {
  if (threads < 1) threads = 1;
  if (reps < 1) reps = 1;
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long long it = 0; it < (long long)n * reps; ++it) {
    const int i = (int)(it % n);
    for (int k = 0; k < 4; ++k) {
      if (!elem16) {
        const uint8_t *s = (const uint8_t *)src_origin + (ptrdiff_t)g[i].sy * src_stride + g[i].sx;
        const uint8_t *r = (const uint8_t *)ref_origin + (ptrdiff_t)g[i].ry[k] * ref_stride + g[i].rx[k];
        out[4 * i + k] =
            skip ? orc_sad_skip(s, src_stride, r, ref_stride, w, h) : orc_sad(s, src_stride, r, ref_stride, w, h);
      } else {
        const uint16_t *s = (const uint16_t *)src_origin + (ptrdiff_t)g[i].sy * src_stride + g[i].sx;
        const uint16_t *r = (const uint16_t *)ref_origin + (ptrdiff_t)g[i].ry[k] * ref_stride + g[i].rx[k];
        out[4 * i + k] = skip ? orc_highbd_sad_skip(s, src_stride, r, ref_stride, w, h, bd)
                              : orc_highbd_sad(s, src_stride, r, ref_stride, w, h, bd);
      }
    }
  }
}
