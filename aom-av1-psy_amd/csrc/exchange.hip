// The one exchange step of the tile-column encoder, in the product and over RCCL directly: after frame t is
// reconstructed and filtered, GPU r holds valid pixels only in its own tile column [x0_r, x1_r) of the reconstruction
// (uniform tile columns, av1/common/tile_common.c:76-110), while motion vectors of frame t + 1 are limited by FRAME-relative
// limits (av1/encoder/mcomp.h:216-247): every GPU needs its neighbours' pixels as reference.  aomhip_allgather_recon
// moves them: column strips are packed to contiguous staging buffers, exchanged with ONE ncclGroupStart / ncclGroupEnd of
// per-peer ncclSend / ncclRecv on the context's stream (on the fully connected xGMI mesh every pair has its own link, so
// the N simultaneous point-to-point transfers are the direct all-gather; bytes, because RCCL has no 16-bit integer type),
// unpacked, and the plane's borders re-extended.  halo >= 0 exchanges only what a rank can reference: its own column
// widened by `halo` pixels (search range + AOM_INTERP_EXTEND) on each side.
//
// The plan (who sends which columns to whom) is a pure host function, aomhip_recon_exchange_plan, so that the protocol
// is testable without a GPU: what rank a sends to b must be exactly what b expects from a.
#include <rccl/rccl.h>

#include "common.h"

constexpr int kMaxRanks = 64;

struct aomhip_comm {
  ncclComm_t comm;
  int rank, n_ranks;
  void *d_stage;       // send strips then receive strips, contiguous
  size_t stage_bytes;
};

namespace aomhip {

// rows x row_bytes between a strided plane region and a contiguous buffer (16-byte lanes where alignment allows)
__global__ __launch_bounds__(256) void strip_copy_kernel(char *plane, int64_t pitch_b, char *packed, int row_bytes, int rows, int to_packed) {
  const int y = blockIdx.y;
  if (y >= rows) return;
  char *p = plane + (int64_t)y * pitch_b, *q = packed + (int64_t)y * row_bytes;
  for (int x = (blockIdx.x * 256 + threadIdx.x) * 4; x < row_bytes; x += gridDim.x * 256 * 4) {
    if (x + 4 <= row_bytes && ((((uintptr_t)p + x) | ((uintptr_t)q + x)) & 3) == 0) {
      if (to_packed) *reinterpret_cast<uint32_t *>(q + x) = *reinterpret_cast<const uint32_t *>(p + x);
      else *reinterpret_cast<uint32_t *>(p + x) = *reinterpret_cast<const uint32_t *>(q + x);
    } else {
      for (int k = x; k < row_bytes && k < x + 4; ++k) {
        if (to_packed) q[k] = p[k]; else p[k] = q[k];
      }
    }
  }
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_tile_column_bounds(int width, int n_cols, int sb_size, int (*bounds)[2]) {
  // av1_get_uniform_tile_size / av1_calculate_tile_cols (tile_common.c:76-110): size_sb = ceil(sb_cols / n_cols)
  if (width <= 0 || n_cols <= 0 || sb_size <= 0 || !bounds) return 0;
  const int sb_cols = (width + sb_size - 1) / sb_size, size_sb = (sb_cols + n_cols - 1) / n_cols;
  int n = 0;
  for (int s = 0; s < sb_cols && n < n_cols; s += size_sb, ++n) {
    bounds[n][0] = s * sb_size;
    bounds[n][1] = (s + size_sb) * sb_size < width ? (s + size_sb) * sb_size : width;
  }
  for (int i = n; i < n_cols; ++i) bounds[i][0] = bounds[i][1] = 0;  // idle ranks (fewer columns than ranks)
  return n;
}

int aomhip_tile_column_bounds_widths(int width, int sb_size, const int *tile_widths_sb, int n_widths, int max_width_sb, int n_cols, int (*bounds)[2]) {
  // set_tile_info's explicit form (av1/encoder/encoder.c:303-312): the width list is walked cyclically, every width clipped to max_width_sb
  if (width <= 0 || sb_size <= 0 || !tile_widths_sb || n_widths <= 0 || n_cols <= 0 || !bounds) return 0;
  const int sb_cols = (width + sb_size - 1) / sb_size;
  int n = 0, j = 0;
  for (int s = 0; s < sb_cols && n < n_cols;) {
    int size_sb = tile_widths_sb[j++];
    if (j >= n_widths) j = 0;
    if (max_width_sb > 0 && size_sb > max_width_sb) size_sb = max_width_sb;
    if (size_sb <= 0) return 0;
    bounds[n][0] = s * sb_size;
    bounds[n][1] = (s + size_sb) * sb_size < width ? (s + size_sb) * sb_size : width;
    s += size_sb;
    ++n;
    if (n == n_cols && s < sb_cols) bounds[n - 1][1] = width;   // (the reference closes the last tile at the frame edge: col_start_sb[cols] = sb_cols)
  }
  for (int i = n; i < n_cols; ++i) bounds[i][0] = bounds[i][1] = 0;
  return n;
}

int aomhip_tile_column_bounds_balanced(int width, int log2_cols, int sb_size, int max_width_sb, int (*bounds)[2]) {
  // auto_tile_size_balancing (av1/encoder/encoder.c:247-275; tile_widths[0] < 0): floor(sb_cols / 2^k) superblocks per column, the LAST
  // (sb_cols mod 2^k) columns one wider -- every rank gets a column even where the uniform rule leaves the last one short or empty
  if (width <= 0 || log2_cols < 0 || log2_cols > 6 || sb_size <= 0 || !bounds) return 0;
  const int sb_cols = (width + sb_size - 1) / sb_size, n_cols = 1 << log2_cols;
  int size_sb = sb_cols >> log2_cols;
  const int res = sb_cols - (size_sb << log2_cols), inc_index = n_cols - res;
  // The reference's loop runs over tile indices i < MAX_TILE_COLS and records every index, zero-width ones too (fewer superblocks than
  // columns: the leading `inc_index` tiles have size_sb == 0).  A rank cannot own an empty column, so the output keeps the columns that
  // have pixels, in order: i walks the reference's tile indices, n counts the columns written.
  int n = 0, s = 0;
  for (int i = 0; s < sb_cols && i < n_cols; ++i) {
    if (i == inc_index) ++size_sb;
    const int w = max_width_sb > 0 && size_sb > max_width_sb ? max_width_sb : size_sb;
    if (w <= 0) continue;
    bounds[n][0] = s * sb_size;
    bounds[n][1] = (s + w) * sb_size < width ? (s + w) * sb_size : width;
    s += w;
    ++n;
  }
  // max_width_sb clipped the columns short of the frame edge: the reference keeps opening tiles up to MAX_TILE_COLS; with one column per
  // rank and 2^log2_cols ranks the last column is closed at the frame edge instead (as col_start_sb[cols] = num_sbs closes the last tile)
  if (n > 0 && s < sb_cols) bounds[n - 1][1] = width;
  for (int i = n; i < n_cols; ++i) bounds[i][0] = bounds[i][1] = 0;
  return n;
}

int aomhip_recon_exchange_plan(int n_ranks, int rank, const int (*col_bounds)[2], int width, int halo, aomhip_exchange_item *send,
                               aomhip_exchange_item *recv) {
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks || !col_bounds || !send || !recv) return AOMHIP_ERR_INVALID;
  for (int r = 0; r < n_ranks; ++r) {
    send[r].x0 = send[r].x1 = recv[r].x0 = recv[r].x1 = 0;
    if (r == rank) continue;
    // what `to` needs from `from`: from's column, clipped to to's column widened by the halo (everything when halo < 0)
    for (int dir = 0; dir < 2; ++dir) {
      const int from = dir == 0 ? rank : r, to = dir == 0 ? r : rank;
      int x0 = col_bounds[from][0], x1 = col_bounds[from][1];
      if (x1 <= x0 || col_bounds[to][1] <= col_bounds[to][0]) { x0 = x1 = 0; }  // an idle rank owns nothing and references nothing
      else if (halo >= 0) {
        const int lo = col_bounds[to][0] - halo, hi = col_bounds[to][1] + halo;
        if (x0 < lo) x0 = lo;
        if (x1 > hi) x1 = hi;
        if (x1 <= x0) x0 = x1 = 0;
      }
      if (x0 < 0) x0 = 0;
      if (x1 > width) x1 = width;
      aomhip_exchange_item *it = dir == 0 ? &send[r] : &recv[r];
      it->x0 = x0; it->x1 = x1;
    }
  }
  return AOMHIP_OK;
}

int aomhip_comm_unique_id(uint8_t id[128]) {
  ncclUniqueId u;
  static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
  if (ncclGetUniqueId(&u) != ncclSuccess) {
    set_error("ncclGetUniqueId failed");
    return AOMHIP_ERR_HIP;
  }
  memcpy(id, &u, 128);
  return AOMHIP_OK;
}

int aomhip_comm_init(aomhip_ctx *ctx, const uint8_t id[128], int rank, int n_ranks, aomhip_comm **out) {
  if (!ctx || !id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return AOMHIP_ERR_INVALID;
  *out = nullptr;
  if (n_ranks > kMaxRanks) {  // (the exchange plans live in fixed arrays; one node has 8 GPUs)
    set_error("aomhip_comm_init: at most %d ranks (%d asked for)", kMaxRanks, n_ranks);
    return AOMHIP_ERR_INVALID;
  }
  AOMHIP_TRY(hipSetDevice(ctx->device));
  aomhip_comm *c = static_cast<aomhip_comm *>(calloc(1, sizeof(aomhip_comm)));
  if (!c) return AOMHIP_ERR_NOMEM;
  ncclUniqueId u;
  memcpy(&u, id, 128);
  const ncclResult_t r = ncclCommInitRank(&c->comm, n_ranks, u, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, n_ranks, ncclGetErrorString(r));
    free(c);
    return AOMHIP_ERR_HIP;
  }
  c->rank = rank;
  c->n_ranks = n_ranks;
  *out = c;
  return AOMHIP_OK;
}

int aomhip_comm_info(aomhip_comm *c, int *rank, int *n_ranks) {
  if (!c || !rank || !n_ranks) return AOMHIP_ERR_INVALID;
  if (ncclCommUserRank(c->comm, rank) != ncclSuccess || ncclCommCount(c->comm, n_ranks) != ncclSuccess) {
    set_error("aomhip_comm_info: RCCL query failed");
    return AOMHIP_ERR_HIP;
  }
  return AOMHIP_OK;
}

void aomhip_comm_destroy(aomhip_comm *c) {
  if (!c) return;
  if (c->d_stage) (void)hipFree(c->d_stage);
  (void)ncclCommDestroy(c->comm);
  free(c);
}

// pack -> one group of sends / receives -> unpack -> borders; send[r] / recv[r] are the column ranges exchanged with peer r
static int exchange(aomhip_ctx *ctx, aomhip_comm *c, const aomhip_planes *p, int frame, int n, const aomhip_exchange_item *send,
                    const aomhip_exchange_item *recv) {
  const size_t esz = p->bit_depth == 8 ? 1 : 2;
  const int rows = p->height;
  if (n > kMaxRanks) {
    set_error("exchange: at most %d ranks", kMaxRanks);
    return AOMHIP_ERR_INVALID;
  }
  size_t off[2][kMaxRanks], total = 0;
  for (int dir = 0; dir < 2; ++dir)
    for (int r = 0; r < n; ++r) {
      const aomhip_exchange_item &it = dir == 0 ? send[r] : recv[r];
      if (it.x0 < 0 || it.x1 > p->width || it.x1 < it.x0) {
        set_error("aomhip_allgather_recon: column range [%d, %d) outside the plane", it.x0, it.x1);
        return AOMHIP_ERR_INVALID;
      }
      off[dir][r] = total;
      total += (((size_t)(it.x1 - it.x0) * esz * rows) + 255) & ~(size_t)255;
    }
  AOMHIP_TRY(hipSetDevice(ctx->device));
  if (total > c->stage_bytes) {
    if (c->d_stage) {
      AOMHIP_TRY(hipStreamSynchronize(ctx->stream));  // an earlier exchange may still be reading it
      AOMHIP_TRY(hipFree(c->d_stage));
    }
    c->d_stage = nullptr;
    c->stage_bytes = 0;
    AOMHIP_TRY(hipMalloc(&c->d_stage, total));
    c->stage_bytes = total;
  }
  char *stage = static_cast<char *>(c->d_stage);
  char *origin = static_cast<char *>(p->base) + ((size_t)frame * p->frame_stride + (size_t)p->border * p->stride + p->border) * esz;
  const int64_t pitch_b = (int64_t)p->stride * (int64_t)esz;
  auto copy = [&](const aomhip_exchange_item &it, size_t o, int to_packed) {
    const int row_bytes = (int)((it.x1 - it.x0) * esz);
    if (row_bytes <= 0) return;
    const int gx = (row_bytes / 4 + 255) / 256;
    const dim3 grid((unsigned)(gx > 0 ? gx : 1), (unsigned)rows);
    hipLaunchKernelGGL(strip_copy_kernel, grid, dim3(256), 0, ctx->stream, origin + (size_t)it.x0 * esz, pitch_b, stage + o, row_bytes, rows, to_packed);
  };
  for (int r = 0; r < n; ++r) copy(send[r], off[0][r], 1);
  AOMHIP_LAUNCH_CHECK();
  ncclResult_t nr = ncclGroupStart();
  for (int r = 0; r < n && nr == ncclSuccess; ++r) {
    const size_t sb = (size_t)(send[r].x1 - send[r].x0) * esz * rows, rb = (size_t)(recv[r].x1 - recv[r].x0) * esz * rows;
    if (sb) nr = ncclSend(stage + off[0][r], sb, ncclUint8, r, c->comm, ctx->stream);
    if (rb && nr == ncclSuccess) nr = ncclRecv(stage + off[1][r], rb, ncclUint8, r, c->comm, ctx->stream);
  }
  const ncclResult_t ne = ncclGroupEnd();
  if (nr != ncclSuccess || ne != ncclSuccess) {
    // (the pack kernels are already queued and the staging buffer's receive half is undefined: drain the stream so that nothing of this
    // exchange is still running when the caller sees the error; the plane itself has not been written)
    (void)hipStreamSynchronize(ctx->stream);
    set_error("aomhip_allgather_recon: RCCL send / recv failed: %s", ncclGetErrorString(nr != ncclSuccess ? nr : ne));
    return AOMHIP_ERR_HIP;
  }
  for (int r = 0; r < n; ++r) copy(recv[r], off[1][r], 0);
  AOMHIP_LAUNCH_CHECK();
  return aomhip_planes_extend_borders(ctx, p, frame, 1);
}

int aomhip_allgather_recon(aomhip_ctx *ctx, aomhip_comm *c, const aomhip_planes *p, int frame, const int (*col_bounds)[2], int halo) {
  if (!ctx || !c || !p || !p->base || !col_bounds || frame < 0 || frame >= p->n_frames) {
    set_error("aomhip_allgather_recon: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (c->n_ranks > 64) {
    set_error("aomhip_allgather_recon: at most 64 ranks");
    return AOMHIP_ERR_INVALID;
  }
  aomhip_exchange_item send[64], recv[64];
  const int rc = aomhip_recon_exchange_plan(c->n_ranks, c->rank, col_bounds, p->width, halo, send, recv);
  if (rc != AOMHIP_OK) return rc;
  return exchange(ctx, c, p, frame, c->n_ranks, send, recv);
}

int aomhip_exchange_loopback(aomhip_ctx *ctx, aomhip_comm *c, const aomhip_planes *p, int frame, int x0, int x1, int dst_x0) {
  if (!ctx || !c || !p || !p->base || frame < 0 || frame >= p->n_frames || x1 < x0) {
    set_error("aomhip_exchange_loopback: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  aomhip_exchange_item send[64] = {}, recv[64] = {};
  send[c->rank].x0 = x0; send[c->rank].x1 = x1;
  recv[c->rank].x0 = dst_x0; recv[c->rank].x1 = dst_x0 + (x1 - x0);
  return exchange(ctx, c, p, frame, c->n_ranks, send, recv);
}

}  // extern "C"
