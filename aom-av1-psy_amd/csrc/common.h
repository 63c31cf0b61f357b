// Internal helpers shared by the libaomhip translation units (not part of the ABI).
#ifndef AOMHIP_CSRC_COMMON_H_
#define AOMHIP_CSRC_COMMON_H_

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "aomhip.h"

struct aomhip_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
  hipEvent_t ev0, ev1;
  // scratch for the rtcd-signature conformance entry points (host pointers in/out)
  void *d_scratch;
  size_t d_scratch_bytes;
  void *h_pinned;
  size_t h_pinned_bytes;
  // work memory of the device-side composites (aomhip_tf_motion_search_frames): intermediate lists between batched kernels
  void *d_work;
  size_t d_work_bytes;
  // device-side status word: kernels that find a work-list entry they cannot process (e.g. an aomhip_txb whose tx_type does not
  // exist for the transform size) OR a code into it; aomhip_ctx_sync reports and clears it
  int *d_status;
  // bumped whenever scratch / work / pinned memory is reallocated: an aomhip_graph captured on this context froze the old addresses and
  // must not be replayed after that (aomhip_graph_launch compares)
  unsigned buf_generation;
  int *h_status;  // pinned mirror: aomhip_ctx_sync reads the word with an async copy ordered before its one stream synchronise
  // a second stream for the independent halves of a composite call (the first pass's golden-frame leg beside its chain kernel): forked from
  // and joined back into `stream` with the two events inside the call, so callers -- and a graph capture of `stream` -- see one stream
  hipStream_t side_stream;
  hipEvent_t ev_fork, ev_join;
};

namespace aomhip {

void set_error(const char *fmt, ...);
// rtcd-signature paths (the reference's signatures have no error return): record the failure in the process-wide
// sticky status (aomhip_status()) and let the caller return its defined "failed" value -- a failed candidate must LOSE the encoder's
// search, 0 would win it: kFailedCost (UINT32_MAX) for SAD, whose consumers compare unsigned (mcomp.c:1350-1395), and kFailedVarCost
// (0x3FFFFFFF) for variance / sub-pixel variance / sse and their *sse, whose consumers convert to int and add an MV cost in 32 bits
// (check_better_fast `int thismse = svf(..); cost += thismse`, mcomp.c:2441-2448; get_mvpred_var_cost :645-664; av1_get_mvpred_sse
// :3661-3677): UINT32_MAX would read as -1 there and WIN, 0x3FFFFFFF stays positive and cannot wrap when a cost is added -- and
// zeroed outputs for everything else (coefficients, eob, filtered pixels untouched).  Never
// aborts, never longjmps (the encoder's only error path is its own, av1/encoder/encoder.c:947-952) -- unless
// AOMHIP_ABORT_ON_ERROR=1 asks for the old fail-stop behaviour.  There is still no CPU fallback: a failed call
// computes nothing.
void note_failure(const char *what, int status = AOMHIP_ERR_HIP);
constexpr uint32_t kFailedCost = 0xFFFFFFFFu;
constexpr uint32_t kFailedVarCost = 0x3FFFFFFFu;
aomhip_ctx *default_ctx();                  // lazily created per-thread context for the rtcd-signature paths; nullptr on failure
void *scratch(aomhip_ctx *ctx, size_t bytes);
hipStream_t side_stream(aomhip_ctx *ctx);   // created on first use; nullptr on failure (the caller then stays on ctx->stream)
void *pinned(aomhip_ctx *ctx, size_t bytes);
void *work(aomhip_ctx *ctx, size_t bytes);
// enqueue the check of a per-block transform list (tx_type valid for tx_size; `wht_ok`: AOMHIP_TX_WHT allowed) on the context's stream
int validate_txb_list(aomhip_ctx *ctx, const aomhip_txb *d_blocks, int n_blocks, int tx_size, bool wht_ok, bool any_type_0_15);
constexpr int kStatusBadTxType = 1;
bool tx_type_ok(int tx_size, int tx_type);  // (size, type) pairs av1_get_fwd_txfm_cfg serves, + AOMHIP_TX_WHT for TX_4X4

#define AOMHIP_TRY(expr)                                                                       \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      aomhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return AOMHIP_ERR_HIP;                                                                   \
    }                                                                                          \
  } while (0)

#define AOMHIP_LAUNCH_CHECK()                                                                  \
  do {                                                                                         \
    hipError_t e_ = hipGetLastError();                                                         \
    if (e_ != hipSuccess) {                                                                    \
      aomhip::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return AOMHIP_ERR_HIP;                                                                   \
    }                                                                                          \
  } while (0)

// MI355X: 8 XCDs, workgroup b is dispatched to XCD b % 8 (speed only, never correctness).
// Remap a linear workgroup index so that each XCD walks one contiguous 1/8 of the
// work list: neighbouring blocks of a frame then share that XCD's private L2.
__device__ __forceinline__ unsigned xcd_chunked_index(unsigned b, unsigned n) {
  constexpr unsigned kXcd = 8;
  const unsigned q = n / kXcd, r = n % kXcd;
  const unsigned xcd = b % kXcd, idx = b / kXcd;
  return xcd < r ? xcd * (q + 1) + idx : r * (q + 1) + (xcd - r) * q + idx;
}

// Device view of aomhip_planes for one element type.
template <typename T>
struct PlaneView {
  const T *origin;  // pixel (0,0) of frame 0
  int64_t frame_stride;
  int stride;
};
template <typename T>
inline PlaneView<T> view_of(const aomhip_planes &p) {
  PlaneView<T> v;
  v.origin = static_cast<const T *>(p.base) + (int64_t)p.border * p.stride + p.border;
  v.frame_stride = p.frame_stride;
  v.stride = p.stride;
  return v;
}

// The reference's 22 block sizes (av1/common/enums.h:99-124).
inline bool valid_block(int w, int h) {
  auto p2 = [](int v) { return v >= 4 && v <= 128 && (v & (v - 1)) == 0; };
  if (!p2(w) || !p2(h)) return false;
  const int r = w > h ? w / h : h / w;
  if (r > 4) return false;
  if (r == 4 && (w == 128 || h == 128)) return false;  // no 128x32 / 32x128
  return true;
}

}  // namespace aomhip
#endif  // AOMHIP_CSRC_COMMON_H_
