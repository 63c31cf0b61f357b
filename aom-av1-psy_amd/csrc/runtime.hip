// libaomhip runtime: contexts, streams, device memory, HBM-resident YV12 plane rings.
// Host side of include/aomhip.h "context" and "planes in HBM".
#include <atomic>
#include <cstdarg>

#include "common.h"

namespace aomhip {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<int> g_sticky{ AOMHIP_OK };
static std::atomic<long> g_failures{ 0 };

void note_failure(const char *what, int status) {
  int expected = AOMHIP_OK;
  g_sticky.compare_exchange_strong(expected, status);  // the first failure wins
  if (g_failures.fetch_add(1) == 0) {
    fprintf(stderr, "libaomhip: %s failed: %s (sticky status %d; later failures are counted, not printed)\n", what, g_err, status);
    fflush(stderr);
  }
  static const bool abort_on_error = [] { const char *e = getenv("AOMHIP_ABORT_ON_ERROR"); return e && atoi(e) != 0; }();
  if (abort_on_error) abort();
}

static thread_local aomhip_ctx *g_default_ctx = nullptr;
static thread_local bool g_default_ctx_failed = false;  // a failed creation is remembered: not retried (and re-reported) by every later call

aomhip_ctx *default_ctx() {
  if (!g_default_ctx && !g_default_ctx_failed) {
    const char *dev = getenv("AOMHIP_DEVICE");
    if (aomhip_ctx_create(dev ? atoi(dev) : 0, nullptr, &g_default_ctx) != AOMHIP_OK) {
      note_failure("default context", AOMHIP_ERR_NO_DEVICE);
      g_default_ctx = nullptr;
      g_default_ctx_failed = true;
    }
  }
  return g_default_ctx;
}

hipStream_t side_stream(aomhip_ctx *ctx) {
  if (ctx->side_stream) return ctx->side_stream;
  {  // not while ctx->stream is being captured into a graph: creating a stream there is not a capturable operation -- the caller stays serial
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
      (void)hipGetLastError();
      return nullptr;
    }
  }
  hipStream_t st = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  // the side stream carries throughput work beside a latency chain on ctx->stream (first pass: the golden leg beside the row chain; temporal
  // filter: a frame's sub-block searches beside the next frame's block search): the lowest priority the device offers
  int prio_lo = 0, prio_hi = 0;
  if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { prio_lo = 0; (void)hipGetLastError(); }
  static const int env_prio = [] { const char *e = getenv("AOMHIP_SIDE_PRIORITY"); return e ? atoi(e) : 1; }();
  if ((env_prio ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_lo) : hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess || hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess) {
    if (b) (void)hipEventDestroy(b);
    if (a) (void)hipEventDestroy(a);
    if (st) (void)hipStreamDestroy(st);
    (void)hipGetLastError();
    return nullptr;
  }
  ctx->side_stream = st; ctx->ev_fork = a; ctx->ev_join = b;
  return st;
}

void *scratch(aomhip_ctx *ctx, size_t bytes) {
  if (ctx->d_scratch_bytes < bytes) {
    ++ctx->buf_generation;
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
    if (hipMalloc(&ctx->d_scratch, cap) != hipSuccess) {
      ctx->d_scratch = nullptr;
      ctx->d_scratch_bytes = 0;
      set_error("hipMalloc(%zu) for scratch failed", cap);
      return nullptr;
    }
    ctx->d_scratch_bytes = cap;
  }
  return ctx->d_scratch;
}

void *work(aomhip_ctx *ctx, size_t bytes) {
  if (ctx->d_work_bytes < bytes) {
    ++ctx->buf_generation;
    if (ctx->d_work) {
      (void)hipStreamSynchronize(ctx->stream);  // kernels of an earlier call may still use it
      (void)hipFree(ctx->d_work);
    }
    ctx->d_work = nullptr;
    ctx->d_work_bytes = 0;
    const size_t cap = bytes + bytes / 4;
    if (hipMalloc(&ctx->d_work, cap) != hipSuccess) {
      ctx->d_work = nullptr;
      set_error("hipMalloc(%zu) for work memory failed", cap);
      return nullptr;
    }
    ctx->d_work_bytes = cap;
  }
  return ctx->d_work;
}

void *pinned(aomhip_ctx *ctx, size_t bytes) {
  if (ctx->h_pinned_bytes < bytes) {
    ++ctx->buf_generation;
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    size_t cap = bytes < (1u << 20) ? (1u << 20) : bytes;
    if (hipHostMalloc(&ctx->h_pinned, cap, hipHostMallocDefault) != hipSuccess) {
      ctx->h_pinned = nullptr;
      ctx->h_pinned_bytes = 0;
      set_error("hipHostMalloc(%zu) failed", cap);
      return nullptr;
    }
    ctx->h_pinned_bytes = cap;
  }
  return ctx->h_pinned;
}

// Replicate the visible edge pixels of each frame into its border: every element
// outside the visible rectangle takes visible[clamp(y)][clamp(x)], which is what
// aom_extend_frame_borders_c produces (aom_scale/generic/yv12extend.c:22-221).
template <typename T>
__global__ void extend_borders_kernel(T *base, int64_t frame_stride, int first_frame, int width, int height,
                                      int stride, int border) {
  const int rows = ((height + 7) & ~7) + 2 * border;
  T *frame = base + (int64_t)(first_frame + blockIdx.z) * frame_stride;
  const int y = blockIdx.y - border;  // one row per blockIdx.y
  if ((int)blockIdx.y >= rows) return;
  const int cy = y < 0 ? 0 : (y >= height ? height - 1 : y);
  const T *src_row = frame + (int64_t)(cy + border) * stride + border;
  T *dst_row = frame + (int64_t)(y + border) * stride + border;
  const bool interior_row = (y == cy);
  for (int xi = blockIdx.x * blockDim.x + threadIdx.x; xi < stride; xi += gridDim.x * blockDim.x) {
    const int x = xi - border;
    if (interior_row && x >= 0 && x < width) continue;
    const int cx = x < 0 ? 0 : (x >= width ? width - 1 : x);
    dst_row[x] = src_row[cx];
  }
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_abi_version(void) { return AOMHIP_ABI_VERSION; }

int aomhip_status(void) { return g_sticky.load(); }
long aomhip_failure_count(void) { return g_failures.load(); }
void aomhip_status_clear(void) {
  g_sticky.store(AOMHIP_OK);
  g_failures.store(0);
  g_default_ctx_failed = false;  // (the calling thread may try to create its default context again)
}

const char *aomhip_last_error(void) { return g_err; }

int aomhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int aomhip_ctx_create(int device, void *stream, aomhip_ctx **out) {
  if (!out) return AOMHIP_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    set_error("no HIP device visible (libaomhip has no CPU fallback)");
    return AOMHIP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    set_error("device %d out of range [0,%d)", device, n);
    return AOMHIP_ERR_INVALID;
  }
  AOMHIP_TRY(hipSetDevice(device));
  aomhip_ctx *c = static_cast<aomhip_ctx *>(calloc(1, sizeof(aomhip_ctx)));
  if (!c) return AOMHIP_ERR_NOMEM;
  c->device = device;
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
    c->own_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
      free(c);
      return AOMHIP_ERR_HIP;
    }
    c->own_stream = true;
  }
  // (every resource taken so far is released on a failed creation: the stream it owns, the events, the status words)
  auto fail = [&](int rc) {
    if (c->h_status) (void)hipHostFree(c->h_status);
    if (c->d_status) (void)hipFree(c->d_status);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    free(c);
    return rc;
  };
  if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
    set_error("hipEventCreate failed");
    return fail(AOMHIP_ERR_HIP);
  }
  if (hipMalloc(reinterpret_cast<void **>(&c->d_status), sizeof(int)) != hipSuccess || hipMemset(c->d_status, 0, sizeof(int)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void **>(&c->h_status), sizeof(int), hipHostMallocDefault) != hipSuccess) {
    set_error("allocating the device status word / its pinned mirror failed");
    return fail(AOMHIP_ERR_NOMEM);
  }
  *c->h_status = 0;
  *out = c;
  return AOMHIP_OK;
}

void aomhip_ctx_destroy(aomhip_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->d_work) (void)hipFree(ctx->d_work);
  if (ctx->d_status) (void)hipFree(ctx->d_status);
  if (ctx->h_status) (void)hipHostFree(ctx->h_status);
  if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
  (void)hipEventDestroy(ctx->ev0);
  (void)hipEventDestroy(ctx->ev1);
  if (ctx->side_stream) {
    (void)hipStreamSynchronize(ctx->side_stream);
    (void)hipStreamDestroy(ctx->side_stream);
    (void)hipEventDestroy(ctx->ev_fork);
    (void)hipEventDestroy(ctx->ev_join);
  }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  if (g_default_ctx == ctx) g_default_ctx = nullptr;
  free(ctx);
}

int aomhip_ctx_sync(aomhip_ctx *ctx) {
  if (!ctx) return AOMHIP_ERR_INVALID;
  // the status word travels behind the queued work on the same stream: one synchronise, no second blocking copy
  AOMHIP_TRY(hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  AOMHIP_TRY(hipStreamSynchronize(ctx->stream));
  const int st = *ctx->h_status;
  if (st) {
    AOMHIP_TRY(hipMemset(ctx->d_status, 0, sizeof(int)));
    set_error("a batched call since the last synchronisation was given work-list entries it cannot process (device status 0x%x:%s); "
              "its outputs are undefined", st, (st & kStatusBadTxType) ? " an aomhip_txb tx_type that does not exist for the transform size" : "");
    return AOMHIP_ERR_INVALID;
  }
  return AOMHIP_OK;
}

void *aomhip_ctx_stream(aomhip_ctx *ctx) { return ctx ? ctx->stream : nullptr; }

// ---- replaying a sequence of batched calls as one hipGraph ----------------------------------------------------------------------
// The per-frame chain of an encoder (search -> predict -> transform -> filters) is a dozen dependent launches, some of them 7-20 us long:
// enqueued one by one, each pays the queue's dispatch latency behind its predecessor.  Between capture_begin and capture_end the batched
// entry points are RECORDED instead of run (everything they enqueue goes to the context's stream); the graph replays them in one launch.
// The sequence must have run once un-captured before (work buffers grow on first use, and an allocation cannot be captured), and its
// arguments -- device pointers, frame indices, list lengths -- are frozen into the graph.
struct aomhip_graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  const aomhip_ctx *ctx;     // the context it was captured on ...
  unsigned buf_generation;   // ... and the state of that context's internal buffers, whose addresses the graph froze
};

int aomhip_graph_capture_begin(aomhip_ctx *ctx) {
  if (!ctx) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  AOMHIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
  return AOMHIP_OK;
}

int aomhip_graph_capture_end(aomhip_ctx *ctx, aomhip_graph **out) {
  if (!ctx || !out) return AOMHIP_ERR_INVALID;
  *out = nullptr;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  if (e != hipSuccess || !g) {
    set_error("aomhip_graph_capture_end: %s (a call inside the capture allocated, synchronised or failed: run the sequence once before capturing it)",
              hipGetErrorString(e));
    // leave the stream usable: an invalidated capture keeps failing every later call until it has been ended and the error state read
    for (int k = 0; k < 4; ++k) {
      (void)hipGetLastError();
      hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(ctx->stream, &st) == hipSuccess && st == hipStreamCaptureStatusNone) break;
      hipGraph_t g2 = nullptr;
      (void)hipStreamEndCapture(ctx->stream, &g2);
      if (g2) (void)hipGraphDestroy(g2);
    }
    (void)hipGetLastError();
    return AOMHIP_ERR_HIP;
  }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
    return AOMHIP_ERR_HIP;
  }
  aomhip_graph *gr = static_cast<aomhip_graph *>(malloc(sizeof(aomhip_graph)));
  if (!gr) { (void)hipGraphExecDestroy(x); (void)hipGraphDestroy(g); return AOMHIP_ERR_NOMEM; }
  gr->graph = g; gr->exec = x; gr->ctx = ctx; gr->buf_generation = ctx->buf_generation;
  *out = gr;
  return AOMHIP_OK;
}

int aomhip_graph_launch(aomhip_ctx *ctx, aomhip_graph *g) {
  if (!ctx || !g) return AOMHIP_ERR_INVALID;
  if (g->ctx != ctx || g->buf_generation != ctx->buf_generation) {
    // the composite entry points keep their intermediates in the context's work / scratch memory, which grows (free + malloc) when a
    // later call needs more: the graph would read and write freed memory
    set_error(g->ctx != ctx ? "aomhip_graph_launch: the graph was captured on another context"
                            : "aomhip_graph_launch: a larger call on this context reallocated its work buffers after the capture: capture again");
    return AOMHIP_ERR_INVALID;
  }
  AOMHIP_TRY(hipGraphLaunch(g->exec, ctx->stream));
  return AOMHIP_OK;
}

int aomhip_graph_destroy(aomhip_graph *g) {
  if (!g) return AOMHIP_OK;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  free(g);
  return AOMHIP_OK;
}

int aomhip_timer_begin(aomhip_ctx *ctx) {
  if (!ctx) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  return AOMHIP_OK;
}

int aomhip_timer_end(aomhip_ctx *ctx, float *elapsed_ms) {
  if (!ctx || !elapsed_ms) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  AOMHIP_TRY(hipEventSynchronize(ctx->ev1));
  AOMHIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
  return AOMHIP_OK;
}

int aomhip_malloc(aomhip_ctx *ctx, size_t bytes, void **dptr) {
  if (!ctx || !dptr) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
  if (e != hipSuccess) {
    set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? AOMHIP_ERR_NOMEM : AOMHIP_ERR_HIP;
  }
  return AOMHIP_OK;
}

int aomhip_free(aomhip_ctx *ctx, void *dptr) {
  if (!ctx) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipStreamSynchronize(ctx->stream));
  AOMHIP_TRY(hipFree(dptr));
  return AOMHIP_OK;
}

int aomhip_memcpy_h2d(aomhip_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  AOMHIP_TRY(hipStreamSynchronize(ctx->stream));  // src may be pageable and reused by the caller
  return AOMHIP_OK;
}

int aomhip_memcpy_d2h(aomhip_ctx *ctx, void *dst, const void *src, size_t bytes) {
  if (!ctx) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  AOMHIP_TRY(hipStreamSynchronize(ctx->stream));
  return AOMHIP_OK;
}

int aomhip_memset(aomhip_ctx *ctx, void *dst, int value, size_t bytes) {
  if (!ctx) return AOMHIP_ERR_INVALID;
  AOMHIP_TRY(hipMemsetAsync(dst, value, bytes, ctx->stream));
  return AOMHIP_OK;
}

int aomhip_calc_stride(int width, int border) {
  const int aligned_width = (width + 7) & ~7;
  return ((aligned_width + 2 * border) + 31) & ~31;
}

int aomhip_planes_alloc(aomhip_ctx *ctx, int width, int height, int border, int bit_depth, int n_frames,
                        aomhip_planes *out) {
  if (!ctx || !out || width <= 0 || height <= 0 || border < 0 || n_frames <= 0 ||
      (bit_depth != 8 && bit_depth != 10 && bit_depth != 12)) {
    set_error("aomhip_planes_alloc: bad geometry");
    return AOMHIP_ERR_INVALID;
  }
  memset(out, 0, sizeof(*out));
  const int stride = aomhip_calc_stride(width, border);
  const int aligned_height = (height + 7) & ~7;
  const int64_t rows = (int64_t)aligned_height + 2 * border;
  // Round the per-frame size up to 256 elements so every frame starts 256-byte aligned.
  const int64_t frame_elems = (rows * stride + 255) & ~(int64_t)255;
  const size_t esz = bit_depth == 8 ? 1 : 2;
  void *base = nullptr;
  // + one extra row of slack: the sub-pixel kernels read (W+1) x (H+1) like the reference does.
  int rc = aomhip_malloc(ctx, (size_t)(frame_elems * n_frames + stride + 64) * esz, &base);
  if (rc != AOMHIP_OK) return rc;
  out->base = base;
  out->frame_stride = frame_elems;
  out->width = width;
  out->height = height;
  out->stride = stride;
  out->border = border;
  out->bit_depth = bit_depth;
  out->n_frames = n_frames;
  return AOMHIP_OK;
}

int aomhip_planes_free(aomhip_ctx *ctx, aomhip_planes *p) {
  if (!ctx || !p) return AOMHIP_ERR_INVALID;
  int rc = AOMHIP_OK;
  if (p->base) rc = aomhip_free(ctx, p->base);
  memset(p, 0, sizeof(*p));
  return rc;
}

int aomhip_planes_extend_borders(aomhip_ctx *ctx, const aomhip_planes *p, int first_frame, int n_frames) {
  if (!ctx || !p || !p->base || first_frame < 0 || n_frames <= 0 || first_frame + n_frames > p->n_frames)
    return AOMHIP_ERR_INVALID;
  if (p->border == 0) return AOMHIP_OK;
  const int rows = ((p->height + 7) & ~7) + 2 * p->border;  // aligned_height + borders, yv12config.c:138-170
  dim3 grid((p->stride + 255) / 256, rows, n_frames), block(256);
  if (p->bit_depth == 8)
    hipLaunchKernelGGL(extend_borders_kernel<uint8_t>, grid, block, 0, ctx->stream, static_cast<uint8_t *>(p->base),
                       p->frame_stride, first_frame, p->width, p->height, p->stride, p->border);
  else
    hipLaunchKernelGGL(extend_borders_kernel<uint16_t>, grid, block, 0, ctx->stream,
                       static_cast<uint16_t *>(p->base), p->frame_stride, first_frame, p->width, p->height,
                       p->stride, p->border);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_planes_upload(aomhip_ctx *ctx, const aomhip_planes *p, int frame, const void *host_pixels,
                         int host_stride) {
  if (!ctx || !p || !p->base || !host_pixels || frame < 0 || frame >= p->n_frames || host_stride < p->width)
    return AOMHIP_ERR_INVALID;
  const size_t esz = p->bit_depth == 8 ? 1 : 2;
  char *dst = static_cast<char *>(p->base) +
              ((size_t)frame * p->frame_stride + (size_t)p->border * p->stride + p->border) * esz;
  AOMHIP_TRY(hipMemcpy2DAsync(dst, (size_t)p->stride * esz, host_pixels, (size_t)host_stride * esz,
                              (size_t)p->width * esz, p->height, hipMemcpyHostToDevice, ctx->stream));
  AOMHIP_TRY(hipStreamSynchronize(ctx->stream));
  return aomhip_planes_extend_borders(ctx, p, frame, 1);
}

int aomhip_planes_download(aomhip_ctx *ctx, const aomhip_planes *p, int frame, void *host_bordered) {
  if (!ctx || !p || !p->base || !host_bordered || frame < 0 || frame >= p->n_frames) return AOMHIP_ERR_INVALID;
  const size_t esz = p->bit_depth == 8 ? 1 : 2;
  const size_t n = (size_t)p->stride * (p->height + 2 * p->border) * esz;
  return aomhip_memcpy_d2h(ctx, host_bordered, static_cast<char *>(p->base) + (size_t)frame * p->frame_stride * esz,
                           n);
}

}  // extern "C"
