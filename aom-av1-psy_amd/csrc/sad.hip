// Batched block SAD on gfx950: aom_sadWxH / aom_sad_skip_WxH / aom_sadWxHx4d and their
// highbd forms (reference: aom_dsp/sad.c:22-129,240-332; vtable wrappers
// av1/encoder/encoder_utils.h:155-208).
//
// Mapping.  A candidate is W x H pixels of a source plane against W x H pixels of a
// reference plane at an arbitrary (unaligned) position.  The work is cut into "row
// units" of up to 16 bytes (one global_load_dwordx4 per plane); a candidate's units
// are spread over TPC adjacent lanes of a 64-wide wavefront, each lane accumulates
// with v_sad_u8 / v_sad_u16 and the TPC partial sums are folded with DPP row
// rotations (within 16 lanes) and lane permutes (beyond).  There is no LDS staging:
// every byte is used exactly once by exactly one lane, so the kernel is a pure
// stream over (src rows, ref rows); what bounds it is the number of distinct cache
// lines a wavefront touches per load instruction, not arithmetic.
//
// Workgroups are re-indexed so that each of the 8 XCDs walks one contiguous 1/8 of a
// frame's candidate list (raster order from the host batching layer): the rows a
// band of blocks reads then stay in that XCD's private 4 MiB L2.
#include <type_traits>

#include "common.h"

namespace aomhip {

struct __attribute__((packed, aligned(1))) U128 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) U64 { uint32_t v[2]; };
struct __attribute__((packed, aligned(1))) U32 { uint32_t v[1]; };

template <int BYTES> struct UnitLoad;
template <> struct UnitLoad<16> { using type = U128; };
template <> struct UnitLoad<8> { using type = U64; };
template <> struct UnitLoad<4> { using type = U32; };

template <typename T> __device__ __forceinline__ uint32_t sad_dword(uint32_t a, uint32_t b, uint32_t acc);
template <> __device__ __forceinline__ uint32_t sad_dword<uint8_t>(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_sad_u8(a, b, acc);
}
template <> __device__ __forceinline__ uint32_t sad_dword<uint16_t>(uint32_t a, uint32_t b, uint32_t acc) {
  return __builtin_amdgcn_sad_u16(a, b, acc);
}

// Sum over the TPC lanes that share a candidate (TPC a power of two, groups aligned).
template <int TPC> __device__ __forceinline__ uint32_t group_sum(uint32_t v) {
  // DPP butterflies that never leave the TPC-lane group: xor 1, xor 2 inside a quad, then
  // mirror inside 8 lanes, then mirror inside the 16-lane row.
  if constexpr (TPC >= 2) v += __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  if constexpr (TPC >= 4) v += __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  if constexpr (TPC >= 8) v += __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  if constexpr (TPC >= 16) v += __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xf, 0xf, false); // row_mirror
  if constexpr (TPC >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (TPC >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// Geometry of one block size for element type T.
template <typename T, int W, int H, bool SKIP, int UPL = 2> struct SadGeom {
  static constexpr int kRowBytes = W * (int)sizeof(T);
  static constexpr int kUnitBytes = kRowBytes < 16 ? kRowBytes : 16;
  static constexpr int kUnitElems = kUnitBytes / (int)sizeof(T);
  static constexpr int kUnitsPerRow = kRowBytes / kUnitBytes;
  static constexpr int kRows = SKIP ? H / 2 : H;
  static constexpr int kUnits = kUnitsPerRow * kRows;
  // UPL units per lane where the block is big enough; never more than a wavefront
  static constexpr int kTpcRaw = kUnits >= UPL ? kUnits / UPL : 1;
  static constexpr int kTpc = kTpcRaw > 64 ? 64 : kTpcRaw;
  static constexpr int kUnitsPerLane = kUnits / kTpc;
  static constexpr int kRowStep = SKIP ? 2 : 1;
};

// Fetch BYTES (4/8/16) from an arbitrarily aligned address.
//  VARIANT 0: one unaligned global_load (the TCP splits a misaligned dwordx4 into ~3.4 cache
//             accesses per lane -- measured, profiles/r01_pmc_*).
//  VARIANT 1: one or two 16-byte ALIGNED loads covering the bytes, realigned in registers with
//             v_cndmask / v_alignbyte (at most 2 cache accesses per lane, 1 when aligned).
template <int BYTES, int VARIANT>
__device__ __forceinline__ typename UnitLoad<BYTES>::type load_unit(const void *p) {
  using L = typename UnitLoad<BYTES>::type;
  if constexpr (VARIANT == 0) {
    return *reinterpret_cast<const L *>(p);
  } else {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    const uint4 *base = reinterpret_cast<const uint4 *>(a & ~(uintptr_t)15);
    const unsigned sh = (unsigned)(a & 15);
    const uint4 lo = base[0];
    uint4 hi = make_uint4(0, 0, 0, 0);
    if (sh + BYTES > 16) hi = base[1];
    uint32_t d[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
    constexpr int kN = BYTES / 4 + 1;  // dwords needed after the dword-granular shift
    if (sh & 8) {
#pragma unroll
      for (int i = 0; i < kN + 1 && i + 2 < 8; ++i) d[i] = d[i + 2];
    }
    if (sh & 4) {
#pragma unroll
      for (int i = 0; i < kN && i + 1 < 8; ++i) d[i] = d[i + 1];
    }
    L out;
#pragma unroll
    for (int i = 0; i < BYTES / 4; ++i) out.v[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh & 3);
    return out;
  }
}

template <typename T, int BYTES, int VARIANT>
__device__ __forceinline__ uint32_t unit_sad(const T *s, const T *r, uint32_t acc) {
  const auto a = load_unit<BYTES, VARIANT>(s);
  const auto b = load_unit<BYTES, VARIANT>(r);
#pragma unroll
  for (int i = 0; i < BYTES / 4; ++i) acc = sad_dword<T>(a.v[i], b.v[i], acc);
  return acc;
}

constexpr int kBlockThreads = 256;

// One candidate list entry per TPC lanes.
template <typename T, int W, int H, bool SKIP, int VARIANT>
__global__ __launch_bounds__(kBlockThreads) void sad_cand_kernel(PlaneView<T> src, PlaneView<T> ref, int first_frame,
                                                                  const aomhip_sad_cand *__restrict__ cands,
                                                                  int n_cands, int64_t cand_frame_stride,
                                                                  uint32_t *__restrict__ out, int blocks_per_frame8,
                                                                  int shift) {
  using G = SadGeom<T, W, H, SKIP, (VARIANT & 2) ? 1 : 2>;
  constexpr int kCpb = kBlockThreads / G::kTpc;  // candidates per workgroup
  const unsigned b = blockIdx.x;
  const unsigned f_rel = b / blocks_per_frame8;
  const unsigned x = xcd_chunked_index(b % blocks_per_frame8, blocks_per_frame8);
  const int lane_in_cand = threadIdx.x % G::kTpc;
  const int ci = x * kCpb + threadIdx.x / G::kTpc;
  if (ci >= n_cands) return;
  const aomhip_sad_cand c = cands[(int64_t)f_rel * cand_frame_stride + ci];
  const int64_t fo = (int64_t)(first_frame + f_rel);
  const T *sp = src.origin + fo * src.frame_stride + (int64_t)c.sy * src.stride + c.sx;
  const T *rp = ref.origin + fo * ref.frame_stride + (int64_t)c.ry * ref.stride + c.rx;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < G::kUnitsPerLane; ++k) {
    const int u = lane_in_cand + k * G::kTpc;
    const int row = (u / G::kUnitsPerRow) * G::kRowStep;
    const int col = (u % G::kUnitsPerRow) * G::kUnitElems;
    acc = unit_sad<T, G::kUnitBytes, (VARIANT & 1)>(sp + (int64_t)row * src.stride + col, rp + (int64_t)row * ref.stride + col, acc);
  }
  acc = group_sum<G::kTpc>(acc);
  if (lane_in_cand == 0) out[(int64_t)f_rel * n_cands + ci] = (SKIP ? 2u * acc : acc) >> shift;
}

// One x4d group (shared source block, four reference positions) per TPC lanes.
template <typename T, int W, int H, bool SKIP, int VARIANT>
__global__ __launch_bounds__(kBlockThreads) void sad_x4d_kernel(PlaneView<T> src, PlaneView<T> ref, int first_frame,
                                                                 const aomhip_sad_x4d_cand *__restrict__ groups,
                                                                 int n_groups, int64_t group_frame_stride,
                                                                 uint32_t *__restrict__ out, int blocks_per_frame8,
                                                                 int shift) {
  using G = SadGeom<T, W, H, SKIP, (VARIANT & 2) ? 1 : 2>;
  using L = typename UnitLoad<G::kUnitBytes>::type;
  constexpr int kGpb = kBlockThreads / G::kTpc;
  const unsigned b = blockIdx.x;
  const unsigned f_rel = b / blocks_per_frame8;
  const unsigned x = xcd_chunked_index(b % blocks_per_frame8, blocks_per_frame8);
  const int lane_in_cand = threadIdx.x % G::kTpc;
  const int gi = x * kGpb + threadIdx.x / G::kTpc;
  if (gi >= n_groups) return;
  const aomhip_sad_x4d_cand c = groups[(int64_t)f_rel * group_frame_stride + gi];
  const int64_t fo = (int64_t)(first_frame + f_rel);
  const T *sp = src.origin + fo * src.frame_stride + (int64_t)c.sy * src.stride + c.sx;
  const T *rbase = ref.origin + fo * ref.frame_stride;
  uint32_t acc[4] = { 0, 0, 0, 0 };
#pragma unroll
  for (int k = 0; k < G::kUnitsPerLane; ++k) {
    const int u = lane_in_cand + k * G::kTpc;
    const int row = (u / G::kUnitsPerRow) * G::kRowStep;
    const int col = (u % G::kUnitsPerRow) * G::kUnitElems;
    const L a = load_unit<G::kUnitBytes, (VARIANT & 1)>(sp + (int64_t)row * src.stride + col);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const L r = load_unit<G::kUnitBytes, (VARIANT & 1)>(rbase + (int64_t)(c.ry[j] + row) * ref.stride + c.rx[j] + col);
#pragma unroll
      for (int i = 0; i < G::kUnitBytes / 4; ++i) acc[j] = sad_dword<T>(a.v[i], r.v[i], acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = group_sum<G::kTpc>(acc[j]);
  if (lane_in_cand == 0) {
    uint4 o;
    o.x = (SKIP ? 2u * acc[0] : acc[0]) >> shift;
    o.y = (SKIP ? 2u * acc[1] : acc[1]) >> shift;
    o.z = (SKIP ? 2u * acc[2] : acc[2]) >> shift;
    o.w = (SKIP ? 2u * acc[3] : acc[3]) >> shift;
    reinterpret_cast<uint4 *>(out)[(int64_t)f_rel * n_groups + gi] = o;
  }
}

// Compound-average SAD (the vtable's sdaf / jsdaf: aom_sadWxH_avg, aom_dist_wtd_sadWxH_avg and their highbd forms,
// aom_dsp/sad.c:50-64,282-297): the candidate is compared with comp = round((second_pred + ref) / 2), or with the
// distance-weighted blend (pred * bck + ref * fwd + 8) >> 4 (aom_dsp/variance.c:306-339,731-766).  second_pred is a
// W-contiguous block; one block per candidate, picked by pred_index (NULL: block 0).  Same lane mapping as
// sad_cand_kernel; the blend is done per element after unpacking.
template <typename T, int W, int H>
__global__ __launch_bounds__(kBlockThreads) void sad_avg_kernel(PlaneView<T> src, PlaneView<T> ref, int first_frame,
                                                                 const aomhip_sad_cand *__restrict__ cands, int n_cands,
                                                                 int64_t cand_frame_stride, const T *__restrict__ preds,
                                                                 const uint32_t *__restrict__ pred_index, int fwd,
                                                                 int bck, uint32_t *__restrict__ out, int shift) {
  using G = SadGeom<T, W, H, false, 2>;
  using L = typename UnitLoad<G::kUnitBytes>::type;
  constexpr int kCpb = kBlockThreads / G::kTpc;
  constexpr int kEPD = 4 / (int)sizeof(T);  // elements per dword
  constexpr uint32_t kMask = sizeof(T) == 1 ? 0xffu : 0xffffu;
  const unsigned f_rel = blockIdx.y;
  const int lane_in_cand = threadIdx.x % G::kTpc;
  const int ci = blockIdx.x * kCpb + threadIdx.x / G::kTpc;
  if (ci >= n_cands) return;
  const aomhip_sad_cand c = cands[(int64_t)f_rel * cand_frame_stride + ci];
  const int64_t fo = (int64_t)(first_frame + f_rel);
  const T *sp = src.origin + fo * src.frame_stride + (int64_t)c.sy * src.stride + c.sx;
  const T *rp = ref.origin + fo * ref.frame_stride + (int64_t)c.ry * ref.stride + c.rx;
  const T *pp = preds + (int64_t)(pred_index ? pred_index[(int64_t)f_rel * n_cands + ci] : 0u) * (W * H);
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < G::kUnitsPerLane; ++k) {
    const int u = lane_in_cand + k * G::kTpc;
    const int row = u / G::kUnitsPerRow, col = (u % G::kUnitsPerRow) * G::kUnitElems;
    const L a = *reinterpret_cast<const L *>(sp + (int64_t)row * src.stride + col);
    const L b = *reinterpret_cast<const L *>(rp + (int64_t)row * ref.stride + col);
    const L p = *reinterpret_cast<const L *>(pp + row * W + col);
#pragma unroll
    for (int i = 0; i < G::kUnitBytes / 4; ++i) {
      uint32_t comp = 0;
#pragma unroll
      for (int e = 0; e < kEPD; ++e) {
        const int sh = e * 8 * (int)sizeof(T);
        const int rv = (int)((b.v[i] >> sh) & kMask), pv = (int)((p.v[i] >> sh) & kMask);
        const int cv = (fwd | bck) ? (pv * bck + rv * fwd + 8) >> 4 : (pv + rv + 1) >> 1;
        comp |= ((uint32_t)cv & kMask) << sh;
      }
      acc = sad_dword<T>(a.v[i], comp, acc);
    }
  }
  acc = group_sum<G::kTpc>(acc);
  if (lane_in_cand == 0) out[(int64_t)f_rel * n_cands + ci] = acc >> shift;
}

struct SadLaunch {
  hipStream_t stream;
  int first_frame, n_frames;
  int flags;
  int shift;  // vtable wrapper: 0 / 2 (10-bit) / 4 (12-bit)
};

template <typename T, int W, int H, bool SKIP, int VARIANT>
static int launch_cand_v(const SadLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, const aomhip_sad_cand *c,
                         int n, int64_t cfs, uint32_t *out) {
  using G = SadGeom<T, W, H, SKIP, (VARIANT & 2) ? 1 : 2>;
  constexpr int kCpb = kBlockThreads / G::kTpc;
  const int bpf = (n + kCpb - 1) / kCpb;
  const int bpf8 = (bpf + 7) & ~7;
  hipLaunchKernelGGL((sad_cand_kernel<T, W, H, SKIP, VARIANT>), dim3((unsigned)bpf8 * l.n_frames), dim3(kBlockThreads),
                     0, l.stream, s, r, l.first_frame, c, n, cfs, out, bpf8, l.shift);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
template <typename T, int W, int H, bool SKIP, int VARIANT>
static int launch_x4d_v(const SadLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r,
                        const aomhip_sad_x4d_cand *g, int n, int64_t gfs, uint32_t *out) {
  using G = SadGeom<T, W, H, SKIP, (VARIANT & 2) ? 1 : 2>;
  constexpr int kGpb = kBlockThreads / G::kTpc;
  const int bpf = (n + kGpb - 1) / kGpb;
  const int bpf8 = (bpf + 7) & ~7;
  hipLaunchKernelGGL((sad_x4d_kernel<T, W, H, SKIP, VARIANT>), dim3((unsigned)bpf8 * l.n_frames), dim3(kBlockThreads),
                     0, l.stream, s, r, l.first_frame, g, n, gfs, out, bpf8, l.shift);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

// Only variant 0 is instantiated.  Measured on MI355X at 4K Mode A (profiles/r01_sad_variants.md; that file
// also records the LDS-staged reference-window kernel that was tried and removed -- 2x slower):
// aligned-window loads cut TCP cache accesses by 39 % but raise L2 requests by 37 % and run 25-30 %
// slower at either occupancy; one unit per lane is neutral for x4d and 13 % slower for single SADs.
template <typename T, int W, int H, bool SKIP>
static int launch_cand(const SadLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, const aomhip_sad_cand *c, int n,
                       int64_t cfs, uint32_t *out) {
  return launch_cand_v<T, W, H, SKIP, 0>(l, s, r, c, n, cfs, out);
}
template <typename T, int W, int H, bool SKIP>
static int launch_x4d(const SadLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, const aomhip_sad_x4d_cand *g,
                      int n, int64_t gfs, uint32_t *out) {
  return launch_x4d_v<T, W, H, SKIP, 0>(l, s, r, g, n, gfs, out);
}

// Dispatch over the reference's 22 block sizes (av1/common/enums.h:99-124).
#define AOMHIP_FOR_BLOCK_SIZES(X)                                                                                \
  X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) \
  X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)

template <typename T, typename CandT>
static int dispatch(bool x4d, const SadLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r, int bw, int bh,
                    const CandT *c, int n, int64_t cfs, uint32_t *out) {
  const bool skip = (l.flags & AOMHIP_SAD_SKIP_ROWS) != 0;
#define X(W, H)                                                                                             \
  if (bw == W && bh == H) {                                                                                 \
    if constexpr (std::is_same<CandT, aomhip_sad_cand>::value) {                                            \
      return skip ? launch_cand<T, W, H, (H >= 2)>(l, s, r, c, n, cfs, out)                                 \
                  : launch_cand<T, W, H, false>(l, s, r, c, n, cfs, out);                                   \
    } else {                                                                                                \
      return skip ? launch_x4d<T, W, H, (H >= 2)>(l, s, r, c, n, cfs, out)                                  \
                  : launch_x4d<T, W, H, false>(l, s, r, c, n, cfs, out);                                    \
    }                                                                                                       \
  }
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  (void)x4d;
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

static int check_args(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame, int n_frames,
                      int bw, int bh, const void *list, int n, const void *out) {
  if (!ctx || !src || !ref || !src->base || !ref->base || !out || (n > 0 && !list)) {
    set_error("null argument");
    return AOMHIP_ERR_INVALID;
  }
  if (!valid_block(bw, bh)) {
    set_error("unsupported block size %dx%d", bw, bh);
    return AOMHIP_ERR_INVALID;
  }
  if (src->bit_depth != ref->bit_depth && !(src->bit_depth > 8 && ref->bit_depth > 8)) {
    set_error("src/ref element types differ");
    return AOMHIP_ERR_INVALID;
  }
  if (n < 0 || n_frames < 0 || first_frame < 0 || first_frame + n_frames > src->n_frames ||
      first_frame + n_frames > ref->n_frames) {
    set_error("frame range out of bounds");
    return AOMHIP_ERR_INVALID;
  }
  return AOMHIP_OK;
}

static int wrapper_shift(int bit_depth) { return bit_depth == 10 ? 2 : bit_depth == 12 ? 4 : 0; }

// ---- rtcd-signature conformance path: host pointers -> tiny device planes -> same kernels ----

template <typename T>
static void host_sad_multi(const T *src, int src_stride, const T *const refs[], int n_refs, int ref_stride, int bw,
                           int bh, int flags, int shift, uint32_t *result) {
  for (int k = 0; k < n_refs; ++k) result[k] = kFailedCost;  // the defined result of a failed call: a LOSING score (0 would win every search)
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  if (!valid_block(bw, bh)) {
    set_error("unsupported block size %dx%d", bw, bh);
    return note_failure("aomhip_sad", AOMHIP_ERR_INVALID);
  }
  const size_t blk = (size_t)bw * bh * sizeof(T);
  const size_t in_bytes = blk * (1 + n_refs);
  const size_t cand_off = (in_bytes + 15) & ~(size_t)15;
  const size_t out_off = cand_off + sizeof(aomhip_sad_cand) * 4;
  const size_t total = out_off + sizeof(uint32_t) * 4;
  char *h = static_cast<char *>(pinned(ctx, total));
  char *d = static_cast<char *>(scratch(ctx, total));
  if (!h || !d) return note_failure("aomhip_sad scratch", AOMHIP_ERR_NOMEM);
  T *hs = reinterpret_cast<T *>(h);
  for (int r = 0; r < bh; ++r) memcpy(hs + (size_t)r * bw, src + (size_t)r * src_stride, (size_t)bw * sizeof(T));
  for (int k = 0; k < n_refs; ++k) {
    T *hr = reinterpret_cast<T *>(h + blk * (1 + k));
    for (int r = 0; r < bh; ++r) memcpy(hr + (size_t)r * bw, refs[k] + (size_t)r * ref_stride, (size_t)bw * sizeof(T));
  }
  aomhip_sad_cand *hc = reinterpret_cast<aomhip_sad_cand *>(h + cand_off);
  for (int k = 0; k < n_refs; ++k) hc[k] = aomhip_sad_cand{ 0, 0, 0, (int16_t)(k * bh) };
  if (hipMemcpyAsync(d, h, out_off, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
    set_error("H2D failed");
    return note_failure("aomhip_sad");
  }
  PlaneView<T> sv{ reinterpret_cast<const T *>(d), 0, bw };
  PlaneView<T> rv{ reinterpret_cast<const T *>(d + blk), 0, bw };
  SadLaunch l{ ctx->stream, 0, 1, flags, shift };
  if (dispatch<T, aomhip_sad_cand>(false, l, sv, rv, bw, bh, reinterpret_cast<const aomhip_sad_cand *>(d + cand_off),
                                   n_refs, 0, reinterpret_cast<uint32_t *>(d + out_off)) != AOMHIP_OK)
    return note_failure("aomhip_sad launch");
  if (hipMemcpyAsync(h + out_off, d + out_off, sizeof(uint32_t) * n_refs, hipMemcpyDeviceToHost, ctx->stream) !=
          hipSuccess ||
      hipStreamSynchronize(ctx->stream) != hipSuccess) {
    set_error("D2H / sync failed: %s", hipGetErrorString(hipGetLastError()));
    return note_failure("aomhip_sad");
  }
  memcpy(result, h + out_off, sizeof(uint32_t) * n_refs);
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_sad_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame, int n_frames,
                     int bw, int bh, int flags, const aomhip_sad_cand *d_cands, int n_cands, int64_t cand_frame_stride,
                     uint32_t *d_out) {
  int rc = check_args(ctx, src, ref, first_frame, n_frames, bw, bh, d_cands, n_cands, d_out);
  if (rc != AOMHIP_OK) return rc;
  if (n_cands == 0 || n_frames == 0) return AOMHIP_OK;
  SadLaunch l{ ctx->stream, first_frame, n_frames, flags, wrapper_shift(src->bit_depth) };
  if (src->bit_depth == 8)
    return dispatch<uint8_t, aomhip_sad_cand>(false, l, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), bw, bh, d_cands,
                                              n_cands, cand_frame_stride, d_out);
  return dispatch<uint16_t, aomhip_sad_cand>(false, l, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), bw, bh,
                                             d_cands, n_cands, cand_frame_stride, d_out);
}

int aomhip_sad_x4d_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame,
                         int n_frames, int bw, int bh, int flags, const aomhip_sad_x4d_cand *d_groups, int n_groups,
                         int64_t group_frame_stride, uint32_t *d_out) {
  int rc = check_args(ctx, src, ref, first_frame, n_frames, bw, bh, d_groups, n_groups, d_out);
  if (rc != AOMHIP_OK) return rc;
  if (n_groups == 0 || n_frames == 0) return AOMHIP_OK;
  SadLaunch l{ ctx->stream, first_frame, n_frames, flags, wrapper_shift(src->bit_depth) };
  if (src->bit_depth == 8)
    return dispatch<uint8_t, aomhip_sad_x4d_cand>(true, l, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), bw, bh,
                                                  d_groups, n_groups, group_frame_stride, d_out);
  return dispatch<uint16_t, aomhip_sad_x4d_cand>(true, l, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), bw, bh,
                                                 d_groups, n_groups, group_frame_stride, d_out);
}

int aomhip_sad_avg_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame, int n_frames,
                         int bw, int bh, const aomhip_sad_cand *d_cands, int n_cands, int64_t cand_frame_stride,
                         const void *d_second_pred, const uint32_t *d_pred_index, int fwd_offset, int bck_offset,
                         uint32_t *d_out) {
  int rc = check_args(ctx, src, ref, first_frame, n_frames, bw, bh, d_cands, n_cands, d_out);
  if (rc != AOMHIP_OK) return rc;
  if (!d_second_pred || fwd_offset < 0 || bck_offset < 0 || fwd_offset > 16 || bck_offset > 16 ||
      ((fwd_offset | bck_offset) && fwd_offset + bck_offset != 16)) {
    set_error("aomhip_sad_avg_batch: second_pred missing or weights not 0/0 (plain average) or summing to 16");
    return AOMHIP_ERR_INVALID;
  }
  if (n_cands == 0 || n_frames == 0) return AOMHIP_OK;
  const int shift = wrapper_shift(src->bit_depth);
#define X(W, H)                                                                                                          \
  if (bw == W && bh == H) {                                                                                              \
    if (src->bit_depth == 8) {                                                                                           \
      using G = SadGeom<uint8_t, W, H, false, 2>;                                                                        \
      constexpr int kCpb = kBlockThreads / G::kTpc;                                                                      \
      hipLaunchKernelGGL((sad_avg_kernel<uint8_t, W, H>), dim3((n_cands + kCpb - 1) / kCpb, n_frames), dim3(kBlockThreads), \
                         0, ctx->stream, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), first_frame, d_cands, n_cands,  \
                         cand_frame_stride, static_cast<const uint8_t *>(d_second_pred), d_pred_index, fwd_offset,       \
                         bck_offset, d_out, shift);                                                                      \
    } else {                                                                                                             \
      using G = SadGeom<uint16_t, W, H, false, 2>;                                                                       \
      constexpr int kCpb = kBlockThreads / G::kTpc;                                                                      \
      hipLaunchKernelGGL((sad_avg_kernel<uint16_t, W, H>), dim3((n_cands + kCpb - 1) / kCpb, n_frames),                   \
                         dim3(kBlockThreads), 0, ctx->stream, view_of<uint16_t>(*src), view_of<uint16_t>(*ref),          \
                         first_frame, d_cands, n_cands, cand_frame_stride, static_cast<const uint16_t *>(d_second_pred), \
                         d_pred_index, fwd_offset, bck_offset, d_out, shift);                                            \
    }                                                                                                                    \
    AOMHIP_LAUNCH_CHECK();                                                                                               \
    return AOMHIP_OK;                                                                                                    \
  }
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

unsigned int aomhip_sad(const uint8_t *src_ptr, int src_stride, const uint8_t *ref_ptr, int ref_stride, int bw, int bh) {
  uint32_t r = 0;
  const uint8_t *refs[1] = { ref_ptr };
  host_sad_multi<uint8_t>(src_ptr, src_stride, refs, 1, ref_stride, bw, bh, 0, 0, &r);
  return r;
}

unsigned int aomhip_sad_skip(const uint8_t *src_ptr, int src_stride, const uint8_t *ref_ptr, int ref_stride, int bw,
                             int bh) {
  uint32_t r = 0;
  const uint8_t *refs[1] = { ref_ptr };
  host_sad_multi<uint8_t>(src_ptr, src_stride, refs, 1, ref_stride, bw, bh, AOMHIP_SAD_SKIP_ROWS, 0, &r);
  return r;
}

void aomhip_sad_x4d(const uint8_t *src_ptr, int src_stride, const uint8_t *const ref_ptr[4], int ref_stride,
                    uint32_t sad_array[4], int bw, int bh) {
  host_sad_multi<uint8_t>(src_ptr, src_stride, ref_ptr, 4, ref_stride, bw, bh, 0, 0, sad_array);
}

void aomhip_sad_skip_x4d(const uint8_t *src_ptr, int src_stride, const uint8_t *const ref_ptr[4], int ref_stride,
                         uint32_t sad_array[4], int bw, int bh) {
  host_sad_multi<uint8_t>(src_ptr, src_stride, ref_ptr, 4, ref_stride, bw, bh, AOMHIP_SAD_SKIP_ROWS, 0, sad_array);
}

unsigned int aomhip_sad16x16(const uint8_t *src_ptr, int src_stride, const uint8_t *ref_ptr, int ref_stride) {
  return aomhip_sad(src_ptr, src_stride, ref_ptr, ref_stride, 16, 16);
}

void aomhip_sad16x16x4d(const uint8_t *src_ptr, int src_stride, const uint8_t *const ref_ptr[4], int ref_stride,
                        uint32_t sad_array[4]) {
  aomhip_sad_x4d(src_ptr, src_stride, ref_ptr, ref_stride, sad_array, 16, 16);
}

unsigned int aomhip_highbd_sad(const uint8_t *src8, int src_stride, const uint8_t *ref8, int ref_stride, int bw, int bh,
                               int bd) {
  // CONVERT_TO_SHORTPTR, aom_ports/mem.h:79
  const uint16_t *s = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(src8) << 1);
  const uint16_t *r = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(ref8) << 1);
  uint32_t v = 0;
  const uint16_t *refs[1] = { r };
  host_sad_multi<uint16_t>(s, src_stride, refs, 1, ref_stride, bw, bh, 0, wrapper_shift(bd), &v);
  return v;
}

// aom_highbd_sad_skip_{W}x{H} and the x4d forms (aom_dsp/sad.c:276-332) with the _bits wrappers of
// av1/encoder/encoder_utils.h:413-470 (skip) / :155-208 (x4d) folded in through bd.
unsigned int aomhip_highbd_sad_skip(const uint8_t *src8, int src_stride, const uint8_t *ref8, int ref_stride, int bw, int bh, int bd) {
  const uint16_t *s = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(src8) << 1);
  const uint16_t *r = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(ref8) << 1);
  uint32_t v = 0;
  const uint16_t *refs[1] = { r };
  host_sad_multi<uint16_t>(s, src_stride, refs, 1, ref_stride, bw, bh, AOMHIP_SAD_SKIP_ROWS, wrapper_shift(bd), &v);
  return v;
}

void aomhip_highbd_sad_x4d(const uint8_t *src8, int src_stride, const uint8_t *const ref8[4], int ref_stride, uint32_t sad_array[4], int bw,
                           int bh, int bd, int skip_rows) {
  const uint16_t *s = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(src8) << 1);
  const uint16_t *refs[4];
  for (int k = 0; k < 4; ++k) refs[k] = reinterpret_cast<const uint16_t *>(reinterpret_cast<uintptr_t>(ref8[k]) << 1);
  host_sad_multi<uint16_t>(s, src_stride, refs, 4, ref_stride, bw, bh, skip_rows ? AOMHIP_SAD_SKIP_ROWS : 0, wrapper_shift(bd), sad_array);
}

}  // extern "C"
