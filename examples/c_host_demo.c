/* A plain-C99 host program against the C ABI (include/aomhip.h), the way the reference -- a C code base -- would bind
 * it: no C++, no Python, no torch.  It keeps a frame pair resident in HBM, evaluates the 16x16 SAD of every block at
 * MV (0,0) in one launch (aomhip_sad_batch), runs the reference's motion search on the device for the same blocks
 * (aomhip_full_pixel_search_batch, NSTEP) and checks both against a scalar loop written here.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_host_demo.c -Laom-av1-psy_amd/lib -laomhip -Wl,-rpath,$PWD/aom-av1-psy_amd/lib -o build/c_host_demo
 *
 * Exit code 0 = everything matched; 2 = no GPU visible (the library has no CPU fallback). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aomhip.h"

#define W 256
#define H 128
#define BORDER 64

#define CHECK(call)                                                            \
  do {                                                                         \
    if ((call) != AOMHIP_OK) {                                                 \
      fprintf(stderr, "%s failed: %s\n", #call, aomhip_last_error());          \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static unsigned sad16(const uint8_t *a, const uint8_t *b, int stride) {
  unsigned s = 0;
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 16; ++c) s += (unsigned)abs(a[r * stride + c] - b[r * stride + c]);
  return s;
}

/* The host-only helpers (no GPU call): tile columns + exchange plan of a 4K frame on 8 ranks, the temporal filter's block list and
 * parameters.  Returns 0 when they say what the reference's rules say. */
static int host_helpers(void) {
  int bounds[8][2];
  if (aomhip_tile_column_bounds(3840, 8, 64, bounds) != 8 || bounds[0][1] != 512 || bounds[7][0] != 3584 || bounds[7][1] != 3840) return 1;
  aomhip_exchange_item send[8], recv[8];
  if (aomhip_recon_exchange_plan(8, 3, (const int(*)[2])bounds, 3840, 132, send, recv) != AOMHIP_OK) return 2;
  /* rank 3 owns [1536, 2048): with a 132-pixel halo it sends its two edge bands to its neighbours only */
  if (send[2].x0 != 1536 || send[2].x1 != 1536 + 132 || send[4].x0 != 2048 - 132 || send[4].x1 != 2048 || send[0].x1 != send[0].x0) return 3;
  if (recv[2].x0 != 1536 - 132 || recv[2].x1 != 1536 || recv[4].x0 != 2048 || recv[4].x1 != 2048 + 132) return 4;
  const int n = aomhip_tf_block_list(1920, 1080, 160, NULL);
  if (n != 60 * 34) return 5;
  static aomhip_search_block blocks[60 * 34];
  if (aomhip_tf_block_list(1920, 1080, 160, blocks) != n || blocks[61].bx != 32 || blocks[61].by != 32) return 6;
  const int mesh[8] = { 64, 8, 28, 4, 15, 1, 7, 1 };
  aomhip_tf_params tp;
  aomhip_tf_default_params(1920, 1080, 10, 30, 1, mesh, 2, 2, 1, 0, 0, 0, &tp);
  if (tp.full.search_method != AOMHIP_SEARCH_NSTEP || !tp.full.run_mesh_search || !tp.full.prune_mesh_search ||
      tp.full.mv_cost_type != AOMHIP_MV_COST_L1_HDRES || tp.sub.subpel_search_type != 3 || tp.mse_thresh != (12 << 2)) return 7;
  return 0;
}

int main(void) {
  {
    const int rc = host_helpers();
    if (rc) {
      fprintf(stderr, "host helper check %d failed\n", rc);
      return 3;
    }
  }
  if (aomhip_device_count() <= 0) {
    fprintf(stderr, "no GPU visible: %s\n", aomhip_last_error());
    return 2;
  }
  aomhip_ctx *ctx = NULL;
  CHECK(aomhip_ctx_create(0, NULL, &ctx));
  static uint8_t src[H][W], ref[H][W];
  uint32_t lcg = 12345u;
  static uint8_t grid[H / 16 + 2][W / 16 + 2];
  for (int y = 0; y < H / 16 + 2; ++y)
    for (int x = 0; x < W / 16 + 2; ++x) {
      lcg = lcg * 1664525u + 1013904223u;
      grid[y][x] = (uint8_t)(lcg >> 24);
    }
  for (int y = 0; y < H; ++y) /* a smooth random texture (bilinear interpolation of a coarse grid) plus a little noise */
    for (int x = 0; x < W; ++x) {
      const int gy = y >> 4, gx = x >> 4, fy = y & 15, fx = x & 15;
      const int top = grid[gy][gx] * (16 - fx) + grid[gy][gx + 1] * fx, bot = grid[gy + 1][gx] * (16 - fx) + grid[gy + 1][gx + 1] * fx;
      lcg = lcg * 1664525u + 1013904223u;
      ref[y][x] = (uint8_t)(((top * (16 - fy) + bot * fy) >> 8) * 7 / 8 + ((lcg >> 29) & 3));
    }
  for (int y = 0; y < H; ++y) /* the source is the reference moved by (+2, -3) plus a little noise */
    for (int x = 0; x < W; ++x) {
      const int ry = y + 2 < 0 ? 0 : y + 2 >= H ? H - 1 : y + 2, rx = x - 3 < 0 ? 0 : x - 3 >= W ? W - 1 : x - 3;
      lcg = lcg * 1664525u + 1013904223u;
      src[y][x] = (uint8_t)(ref[ry][rx] + ((lcg >> 30) & 1));
    }
  aomhip_planes ps, pr;
  CHECK(aomhip_planes_alloc(ctx, W, H, BORDER, 8, 1, &ps));
  CHECK(aomhip_planes_alloc(ctx, W, H, BORDER, 8, 1, &pr));
  CHECK(aomhip_planes_upload(ctx, &ps, 0, src, W));
  CHECK(aomhip_planes_upload(ctx, &pr, 0, ref, W));

  enum { NB = (W / 16) * (H / 16) };
  aomhip_sad_cand cands[NB];
  aomhip_search_block blocks[NB];
  for (int i = 0; i < NB; ++i) {
    const int bx = (i % (W / 16)) * 16, by = (i / (W / 16)) * 16;
    cands[i].sx = cands[i].rx = (int16_t)bx;
    cands[i].sy = cands[i].ry = (int16_t)by;
    memset(&blocks[i], 0, sizeof(blocks[i]));
    blocks[i].bx = (int16_t)bx;
    blocks[i].by = (int16_t)by;
    /* av1_set_mv_limits-style limits: the block may leave the frame by BORDER - 16 - AOM_INTERP_EXTEND pixels */
    blocks[i].row_min = (int16_t)(-by - (BORDER - 20));
    blocks[i].row_max = (int16_t)(H - 16 - by + (BORDER - 20));
    blocks[i].col_min = (int16_t)(-bx - (BORDER - 20));
    blocks[i].col_max = (int16_t)(W - 16 - bx + (BORDER - 20));
  }
  void *d_cands, *d_out, *d_blocks, *d_mv, *d_cost;
  CHECK(aomhip_malloc(ctx, sizeof(cands), &d_cands));
  CHECK(aomhip_malloc(ctx, NB * 4, &d_out));
  CHECK(aomhip_malloc(ctx, sizeof(blocks), &d_blocks));
  CHECK(aomhip_malloc(ctx, NB * 4, &d_mv));
  CHECK(aomhip_malloc(ctx, NB * 4, &d_cost));
  CHECK(aomhip_memcpy_h2d(ctx, d_cands, cands, sizeof(cands)));
  CHECK(aomhip_memcpy_h2d(ctx, d_blocks, blocks, sizeof(blocks)));

  CHECK(aomhip_sad_batch(ctx, &ps, &pr, 0, 1, 16, 16, 0, (const aomhip_sad_cand *)d_cands, NB, 0, (uint32_t *)d_out));
  uint32_t sads[NB];
  CHECK(aomhip_memcpy_d2h(ctx, sads, d_out, sizeof(sads)));
  int bad = 0;
  for (int i = 0; i < NB; ++i) bad += sads[i] != sad16(&src[cands[i].sy][cands[i].sx], &ref[cands[i].ry][cands[i].rx], W);

  aomhip_search_params sp;
  memset(&sp, 0, sizeof(sp));
  sp.search_method = AOMHIP_SEARCH_NSTEP;
  sp.step_param = 5;
  sp.mv_cost_type = 4; /* MV_COST_NONE */
  CHECK(aomhip_full_pixel_search_batch(ctx, &ps, &pr, 0, 16, 16, &sp, NULL, NULL, NULL, (const aomhip_search_block *)d_blocks, NB,
                                       (int16_t *)d_mv, (int32_t *)d_cost, NULL, NULL));
  int16_t mv[NB][2];
  CHECK(aomhip_memcpy_d2h(ctx, mv, d_mv, sizeof(mv)));
  /* the search may stop in a local minimum (it is the reference's greedy search), but it must never end worse than it
   * started: SAD at the returned MV <= SAD at MV (0, 0), with the reference block read through the replicated border */
  int found = 0, interior = 0, worse = 0;
  for (int i = 0; i < NB; ++i) {
    const int bx = blocks[i].bx, by = blocks[i].by;
    unsigned s_mv = 0;
    for (int r = 0; r < 16; ++r)
      for (int c = 0; c < 16; ++c) {
        int ry = by + mv[i][0] + r, rx = bx + mv[i][1] + c;
        ry = ry < 0 ? 0 : ry >= H ? H - 1 : ry;
        rx = rx < 0 ? 0 : rx >= W ? W - 1 : rx;
        s_mv += (unsigned)abs(src[by + r][bx + c] - ref[ry][rx]);
      }
    worse += s_mv > sads[i];
    if (bx < 16 || by < 16 || bx >= W - 32 || by >= H - 32) continue; /* away from the clamped edges of the synthetic shift */
    ++interior;
    found += mv[i][0] == 2 && mv[i][1] == -3;
  }
  printf("c_host_demo: %d blocks, %d SAD mismatches, %d searches ended worse than they started, the (+2,-3) shift found in %d of %d interior blocks\n",
         NB, bad, worse, found, interior);
  aomhip_free(ctx, d_cands); aomhip_free(ctx, d_out); aomhip_free(ctx, d_blocks); aomhip_free(ctx, d_mv); aomhip_free(ctx, d_cost);
  aomhip_planes_free(ctx, &ps);
  aomhip_planes_free(ctx, &pr);
  aomhip_ctx_destroy(ctx);
  return (bad == 0 && worse == 0 && found * 2 >= interior) ? 0 : 1;
}
