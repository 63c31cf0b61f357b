// Superblock-bucketed block SAD on gfx950: the same arithmetic as sad.hip (aom_sadWxH, aom_sadWxHx4d,
// _skip_ and highbd forms; reference aom_dsp/sad.c:22-129,240-332), organised the way the encoder issues
// it -- superblock by superblock with every motion vector inside the superblock's search range
// (av1/encoder/encodeframe.c:1069 encode_sb_row; mv limits av1/encoder/mcomp.c:101 av1_set_mv_search_range).
//
// Why a second kernel.  In the direct kernels (sad.hip) every lane pulls its own 16-byte row out of a
// different 128-byte cache line, so a 16x16 candidate costs 16 full-line L2->L1 fills for 256 useful bytes;
// the measured bound is that fill path, not HBM (profiles/r01_sad_variants.md).  Here one workgroup owns one
// bucket (an sb_w x sb_h cell of source blocks), pulls the (sb_w + 2*range) x (sb_h + 2*range) reference
// window into LDS once with full-line coalesced loads (each byte crosses L2->L1 once per workgroup), and
// evaluates every candidate of the bucket from LDS: dword reads + v_alignbyte for the arbitrary byte
// alignment, v_sad_u8 / v_sad_u16, DPP group reduction.  Two 68 KB workgroups fit a CU's 160 KB LDS, so one
// fills while the other computes.  A candidate whose reference block is not wholly inside the window (the
// caller exceeded `range`) is still evaluated, straight from global memory -- slower, never wrong.
#include <type_traits>

#include "common.h"

namespace aomhip {
namespace sb {

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

template <int TPC> __device__ __forceinline__ uint32_t group_sum(uint32_t v) {
  if constexpr (TPC >= 2) v += __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  if constexpr (TPC >= 4) v += __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  if constexpr (TPC >= 8) v += __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
  if constexpr (TPC >= 16) v += __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xf, 0xf, false); // row_mirror
  if constexpr (TPC >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (TPC >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

template <typename T, int W, int H, bool SKIP> struct Geom {
  static constexpr int kRowBytes = W * (int)sizeof(T);
  static constexpr int kUnitBytes = kRowBytes < 16 ? kRowBytes : 16;
  static constexpr int kUnitElems = kUnitBytes / (int)sizeof(T);
  static constexpr int kUnitsPerRow = kRowBytes / kUnitBytes;
  static constexpr int kRows = SKIP ? H / 2 : H;
  static constexpr int kUnits = kUnitsPerRow * kRows;
  static constexpr int kTpcRaw = kUnits >= 2 ? kUnits / 2 : 1;
  static constexpr int kTpc = kTpcRaw > 64 ? 64 : kTpcRaw;
  static constexpr int kUnitsPerLane = kUnits / kTpc;
  static constexpr int kRowStep = SKIP ? 2 : 1;
};

constexpr int kThreads = 512;
constexpr int kLdsPadBytes = 16;  // row pitch = window bytes + 16: 16 consecutive rows start in 16 distinct bank quads

// BYTES at an arbitrary byte offset of the LDS window: BYTES/4 + 1 aligned dword reads, realigned in registers.
template <int BYTES>
__device__ __forceinline__ typename UnitLoad<BYTES>::type lds_unit(const uint32_t *lds, unsigned byte_off) {
  const uint32_t *p = lds + (byte_off >> 2);
  const unsigned sh = byte_off & 3;
  uint32_t d[BYTES / 4 + 1];
#pragma unroll
  for (int i = 0; i <= BYTES / 4; ++i) d[i] = p[i];
  typename UnitLoad<BYTES>::type out;
#pragma unroll
  for (int i = 0; i < BYTES / 4; ++i) out.v[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
  return out;
}

struct SbArgs {
  int first_frame;
  int sb_w, sb_h, range, cells_per_row;
  int xmin, xmax, ymin, ymax;  // readable pixel range of a plane, relative to the visible origin
  int buckets8;                // buckets per frame rounded up to a multiple of 8
  int n_buckets;
  int n_items;                 // buckets8 * n_frames
  int pitch;                   // LDS row pitch in bytes
  int shift;
};

// What a workgroup needs to know about one (frame, bucket) work item.
struct Item {
  int next;  // item index to continue the walk from
  int f_rel, g0, g1, c0, c1;
  int wx0, wx1, wy0, wy1, cpr;  // window in pixels (x1 / y1 exclusive); 16-byte chunks per window row
  bool valid;
};

// Persistent workgroups: the launch has as many workgroups as the chip holds at once (two 68 KB windows per CU)
// and each walks the (frame, bucket) items blockIdx + k * gridDim.  While the candidates of item i are evaluated
// out of LDS, the window of item i+1 is already in flight into registers (8 x 16 bytes per lane) and so are the
// source rows of its first round of candidates; after a barrier the registers are dropped into LDS.  HBM / L2
// latency and workgroup turnover overlap with the SAD arithmetic instead of adding to it (measured: the
// one-shot form of this kernel spent 0.19 of its 0.45 ms per launch in turnover + source latency alone).
template <typename T, int W, int H, bool SKIP>
__global__ __launch_bounds__(kThreads, 4) void sad_sb_kernel(PlaneView<T> src, PlaneView<T> ref, SbArgs a,
                                                          const aomhip_sad_x4d_cand *__restrict__ groups,
                                                          const int32_t *__restrict__ group_off, int n_groups,
                                                          int64_t group_frame_stride, uint32_t *__restrict__ out4,
                                                          const aomhip_sad_cand *__restrict__ cands,
                                                          const int32_t *__restrict__ cand_off, int n_cands,
                                                          int64_t cand_frame_stride, uint32_t *__restrict__ out1) {
  using G = Geom<T, W, H, SKIP>;
  using L = typename UnitLoad<G::kUnitBytes>::type;
  extern __shared__ uint32_t lds[];
  constexpr int kRegChunks = 8;                        // window chunks a lane keeps in flight in registers
  constexpr int kPerWg = kThreads / G::kTpc;           // candidates (or groups) evaluated side by side
  constexpr bool kPrefetch = G::kUnitsPerLane <= 4;    // source rows of the first round ride along with the window
  const int lane_in_cand = threadIdx.x % G::kTpc;
  const int slot = (int)threadIdx.x / G::kTpc;

  auto decode = [&](int item) {
    Item it;
    it.valid = false;
    for (; item < a.n_items; item += (int)gridDim.x) {
      const int bucket = (int)xcd_chunked_index((unsigned)(item % a.buckets8), (unsigned)a.buckets8);
      if (bucket >= a.n_buckets) continue;
      it.g0 = groups ? group_off[bucket] : 0;
      it.g1 = groups ? group_off[bucket + 1] : 0;
      it.c0 = cands ? cand_off[bucket] : 0;
      it.c1 = cands ? cand_off[bucket + 1] : 0;
      if (it.g0 == it.g1 && it.c0 == it.c1) continue;
      it.f_rel = item / a.buckets8;
      const int cell_x = bucket % a.cells_per_row, cell_y = bucket / a.cells_per_row;
      it.wx0 = max(cell_x * a.sb_w - a.range, a.xmin);
      // whole 16-byte chunks only: a ragged tail (clamped window) is served by the direct path instead
      it.cpr = ((min(cell_x * a.sb_w + a.sb_w + a.range, a.xmax) - it.wx0) * (int)sizeof(T)) >> 4;
      it.wx1 = it.wx0 + it.cpr * (16 / (int)sizeof(T));
      it.wy0 = max(cell_y * a.sb_h - a.range, a.ymin);
      it.wy1 = min(cell_y * a.sb_h + a.sb_h + a.range, a.ymax);
      it.valid = true;
      break;
    }
    it.next = item + (int)gridDim.x;
    return it;
  };

  auto src_rows = [&](int f_rel, int sx, int sy, L (&s)[G::kUnitsPerLane]) {
    const T *sp = src.origin + (int64_t)(a.first_frame + f_rel) * src.frame_stride + (int64_t)sy * src.stride + sx;
#pragma unroll
    for (int k = 0; k < G::kUnitsPerLane; ++k) {
      const int u = lane_in_cand + k * G::kTpc;
      const int row = (u / G::kUnitsPerRow) * G::kRowStep;
      const int col = (u % G::kUnitsPerRow) * G::kUnitElems;
      s[k] = *reinterpret_cast<const L *>(sp + (int64_t)row * src.stride + col);
    }
  };

  auto src_unit = [&](int f_rel, int sx, int sy, int k) {
    const int u = lane_in_cand + k * G::kTpc;
    const int row = (u / G::kUnitsPerRow) * G::kRowStep;
    const int col = (u % G::kUnitsPerRow) * G::kUnitElems;
    return *reinterpret_cast<const L *>(src.origin + (int64_t)(a.first_frame + f_rel) * src.frame_stride +
                                        (int64_t)(sy + row) * src.stride + sx + col);
  };

  // Everything of an item that is requested ahead of time.
  struct Ahead {
    U128 win[kRegChunks];
    aomhip_sad_x4d_cand gdesc;
    aomhip_sad_cand cdesc;
    L gsrc[kPrefetch ? G::kUnitsPerLane : 1];
  };
  // Chunk q = threadIdx + k * kThreads of a window sits at (row, chunk-in-row) = (q / cpr, q % cpr); the pair is
  // advanced incrementally (one division per item per lane).
  auto request = [&](const Item &it, Ahead &h) {
    const char *g = reinterpret_cast<const char *>(ref.origin + (int64_t)(a.first_frame + it.f_rel) * ref.frame_stride +
                                                   (int64_t)it.wy0 * ref.stride + it.wx0);
    const int gpitch = ref.stride * (int)sizeof(T);
    const int total = (it.wy1 - it.wy0) * it.cpr;
    const int dr = kThreads / it.cpr, dc = kThreads - dr * it.cpr;
    int r = (int)threadIdx.x / it.cpr, c = (int)threadIdx.x - r * it.cpr;
#pragma unroll
    for (int k = 0; k < kRegChunks; ++k) {
      if ((int)threadIdx.x + k * kThreads < total) h.win[k] = *reinterpret_cast<const U128 *>(g + (int64_t)r * gpitch + c * 16);
      r += dr; c += dc;
      if (c >= it.cpr) { c -= it.cpr; ++r; }
    }
    if constexpr (kPrefetch) {
      if (it.g0 + slot < it.g1) {
        h.gdesc = groups[(int64_t)it.f_rel * group_frame_stride + it.g0 + slot];
        src_rows(it.f_rel, h.gdesc.sx, h.gdesc.sy, h.gsrc);
      }
      if (it.c0 + slot < it.c1) h.cdesc = cands[(int64_t)it.f_rel * cand_frame_stride + it.c0 + slot];
    }
  };
  // Registers -> LDS; chunks beyond kRegChunks per lane (windows larger than 64 KB) are copied synchronously.
  auto commit = [&](const Item &it, const Ahead &h) {
    const int total = (it.wy1 - it.wy0) * it.cpr;
    const int dr = kThreads / it.cpr, dc = kThreads - dr * it.cpr;
    int r = (int)threadIdx.x / it.cpr, c = (int)threadIdx.x - r * it.cpr;
#pragma unroll
    for (int k = 0; k < kRegChunks; ++k) {
      if ((int)threadIdx.x + k * kThreads < total)
        *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(lds) + r * a.pitch + c * 16) =
            make_uint4(h.win[k].v[0], h.win[k].v[1], h.win[k].v[2], h.win[k].v[3]);
      r += dr; c += dc;
      if (c >= it.cpr) { c -= it.cpr; ++r; }
    }
    if (total > kRegChunks * kThreads) {
      const char *g = reinterpret_cast<const char *>(ref.origin + (int64_t)(a.first_frame + it.f_rel) * ref.frame_stride +
                                                     (int64_t)it.wy0 * ref.stride + it.wx0);
      const int gpitch = ref.stride * (int)sizeof(T);
      for (int q0 = (int)threadIdx.x + kRegChunks * kThreads; q0 < total; q0 += 4 * kThreads) {
        U128 t[4];
        int lo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (q0 + k * kThreads < total) {
            t[k] = *reinterpret_cast<const U128 *>(g + (int64_t)r * gpitch + c * 16);
            lo[k] = r * a.pitch + c * 16;
          }
          r += dr; c += dc;
          if (c >= it.cpr) { c -= it.cpr; ++r; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (q0 + k * kThreads < total)
            *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(lds) + lo[k]) = make_uint4(t[k].v[0], t[k].v[1], t[k].v[2], t[k].v[3]);
      }
    }
  };

  // One reference unit: from the LDS window when the block lies inside it, else from global memory.
  auto ref_unit = [&](const Item &it, const T *rbase, bool in, unsigned loff, int rx, int ry, int row, int col) {
    if (in) return lds_unit<G::kUnitBytes>(lds, loff + (unsigned)(row * a.pitch + col * (int)sizeof(T)));
    return *reinterpret_cast<const L *>(rbase + (int64_t)(ry + row) * ref.stride + rx + col);
  };

  Ahead cur;
  Item it = decode((int)blockIdx.x);
  if (it.valid) request(it, cur);
  while (it.valid) {
    commit(it, cur);
    // first-round descriptors / source rows of this item move out of the way of the next request
    aomhip_sad_x4d_cand gdesc = cur.gdesc;
    aomhip_sad_cand cdesc = cur.cdesc;
    L gsrc[kPrefetch ? G::kUnitsPerLane : 1];
    if constexpr (kPrefetch) {
#pragma unroll
      for (int k = 0; k < G::kUnitsPerLane; ++k) gsrc[k] = cur.gsrc[k];
    }
    __syncthreads();
    const Item nxt = decode(it.next);
    if (nxt.valid) request(nxt, cur);

    const T *rbase = ref.origin + (int64_t)(a.first_frame + it.f_rel) * ref.frame_stride;
    // ---- x4d groups
    for (int gi = it.g0 + slot; gi < it.g1; gi += kPerWg) {
      if (!kPrefetch || gi != it.g0 + slot) {
        gdesc = groups[(int64_t)it.f_rel * group_frame_stride + gi];
        if constexpr (kPrefetch) src_rows(it.f_rel, gdesc.sx, gdesc.sy, gsrc);
      }
      uint32_t acc[4] = { 0, 0, 0, 0 };
      bool in[4];
      unsigned loff[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        in[j] = gdesc.rx[j] >= it.wx0 && gdesc.rx[j] + W <= it.wx1 && gdesc.ry[j] >= it.wy0 && gdesc.ry[j] + H <= it.wy1;
        loff[j] = (unsigned)((gdesc.ry[j] - it.wy0) * a.pitch + (gdesc.rx[j] - it.wx0) * (int)sizeof(T));
      }
#pragma unroll(kPrefetch ? G::kUnitsPerLane : 2)
      for (int k = 0; k < G::kUnitsPerLane; ++k) {
        const int u = lane_in_cand + k * G::kTpc;
        const int row = (u / G::kUnitsPerRow) * G::kRowStep;
        const int col = (u % G::kUnitsPerRow) * G::kUnitElems;
        L sv;
        if constexpr (kPrefetch) sv = gsrc[k]; else sv = src_unit(it.f_rel, gdesc.sx, gdesc.sy, k);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const L r = ref_unit(it, rbase, in[j], loff[j], gdesc.rx[j], gdesc.ry[j], row, col);
#pragma unroll
          for (int i = 0; i < G::kUnitBytes / 4; ++i) acc[j] = sad_dword<T>(sv.v[i], r.v[i], acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = group_sum<G::kTpc>(acc[j]);
      if (lane_in_cand == 0) {
        uint4 o;
        o.x = (SKIP ? 2u * acc[0] : acc[0]) >> a.shift;
        o.y = (SKIP ? 2u * acc[1] : acc[1]) >> a.shift;
        o.z = (SKIP ? 2u * acc[2] : acc[2]) >> a.shift;
        o.w = (SKIP ? 2u * acc[3] : acc[3]) >> a.shift;
        reinterpret_cast<uint4 *>(out4)[(int64_t)it.f_rel * n_groups + gi] = o;
      }
    }
    // ---- single candidates
    for (int ci = it.c0 + slot; ci < it.c1; ci += kPerWg) {
      if (!kPrefetch || ci != it.c0 + slot) cdesc = cands[(int64_t)it.f_rel * cand_frame_stride + ci];
      const bool in = cdesc.rx >= it.wx0 && cdesc.rx + W <= it.wx1 && cdesc.ry >= it.wy0 && cdesc.ry + H <= it.wy1;
      const unsigned loff = (unsigned)((cdesc.ry - it.wy0) * a.pitch + (cdesc.rx - it.wx0) * (int)sizeof(T));
      uint32_t acc = 0;
#pragma unroll(kPrefetch ? G::kUnitsPerLane : 4)
      for (int k = 0; k < G::kUnitsPerLane; ++k) {
        const int u = lane_in_cand + k * G::kTpc;
        const int row = (u / G::kUnitsPerRow) * G::kRowStep;
        const int col = (u % G::kUnitsPerRow) * G::kUnitElems;
        // (in Mode-A style lists these rows were fetched for the block's x4d group a moment ago: L1 / L2 hits)
        const L sv = src_unit(it.f_rel, cdesc.sx, cdesc.sy, k);
        const L r = ref_unit(it, rbase, in, loff, cdesc.rx, cdesc.ry, row, col);
#pragma unroll
        for (int i = 0; i < G::kUnitBytes / 4; ++i) acc = sad_dword<T>(sv.v[i], r.v[i], acc);
      }
      acc = group_sum<G::kTpc>(acc);
      if (lane_in_cand == 0) out1[(int64_t)it.f_rel * n_cands + ci] = (SKIP ? 2u * acc : acc) >> a.shift;
    }
    __syncthreads();  // every lane is done reading this window
    it = nxt;
  }
}

struct SbLaunch {
  hipStream_t stream;
  int n_frames;
  int grid;
  size_t lds_bytes;
  SbArgs a;
  const aomhip_sad_x4d_cand *groups;
  const int32_t *group_off;
  int n_groups;
  int64_t gfs;
  uint32_t *out4;
  const aomhip_sad_cand *cands;
  const int32_t *cand_off;
  int n_cands;
  int64_t cfs;
  uint32_t *out1;
};

template <typename T, int W, int H, bool SKIP>
static int launch(const SbLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r) {
  auto k = sad_sb_kernel<T, W, H, SKIP>;
  static thread_local size_t granted = 0;  // per instantiation
  if (l.lds_bytes > granted) {
    AOMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)l.lds_bytes));
    granted = l.lds_bytes;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)l.grid), dim3(kThreads), l.lds_bytes, l.stream, s, r, l.a,
                     l.groups, l.group_off, l.n_groups, l.gfs, l.out4, l.cands, l.cand_off, l.n_cands, l.cfs, l.out1);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

#define AOMHIP_FOR_BLOCK_SIZES(X)                                                                                \
  X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) \
  X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)

template <typename T>
static int dispatch(const SbLaunch &l, bool skip, const PlaneView<T> &s, const PlaneView<T> &r, int bw, int bh) {
#define X(W, H) \
  if (bw == W && bh == H) return skip ? launch<T, W, H, (H >= 2)>(l, s, r) : launch<T, W, H, false>(l, s, r);
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

}  // namespace sb
}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_sad_sb_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame,
                                   int n_frames, int bw, int bh, int flags, int sb_w, int sb_h, int range,
                                   int n_buckets, const aomhip_sad_x4d_cand *d_groups,
                                   const int32_t *d_group_bucket_offsets, int n_groups, int64_t group_frame_stride,
                                   uint32_t *d_out_groups, const aomhip_sad_cand *d_cands,
                                   const int32_t *d_cand_bucket_offsets, int n_cands, int64_t cand_frame_stride,
                                   uint32_t *d_out_cands) {
  if (!ctx || !src || !ref || !src->base || !ref->base) {
    set_error("null argument");
    return AOMHIP_ERR_INVALID;
  }
  if ((d_groups && (!d_group_bucket_offsets || !d_out_groups)) || (d_cands && (!d_cand_bucket_offsets || !d_out_cands)) ||
      (!d_groups && !d_cands)) {
    set_error("a list needs its bucket offsets and its output array; at least one list is required");
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
  if (n_groups < 0 || n_cands < 0 || n_frames < 0 || first_frame < 0 || first_frame + n_frames > src->n_frames ||
      first_frame + n_frames > ref->n_frames) {
    set_error("frame range out of bounds");
    return AOMHIP_ERR_INVALID;
  }
  if (sb_w < 1 || sb_h < 1 || range < 0 || n_buckets < 0) {
    set_error("bad bucket geometry");
    return AOMHIP_ERR_INVALID;
  }
  const int cells_per_row = (src->width + sb_w - 1) / sb_w;
  const int cell_rows = (src->height + sb_h - 1) / sb_h;
  if (n_buckets != cells_per_row * cell_rows) {
    set_error("n_buckets %d != %d x %d cells of %dx%d over a %dx%d plane", n_buckets, cells_per_row, cell_rows, sb_w, sb_h,
              src->width, src->height);
    return AOMHIP_ERR_INVALID;
  }
  if (n_buckets == 0 || n_frames == 0) return AOMHIP_OK;
  const int es = ref->bit_depth > 8 ? 2 : 1;
  const int pitch = (((sb_w + 2 * range) * es + 15) & ~15) + sb::kLdsPadBytes;
  const size_t lds_bytes = (size_t)pitch * (sb_h + 2 * range) + 32;
  if (lds_bytes > 160 * 1024) {
    set_error("reference window %d x %d (%zu bytes) exceeds the 160 KB LDS of a CU", sb_w + 2 * range, sb_h + 2 * range,
              lds_bytes);
    return AOMHIP_ERR_INVALID;
  }
  sb::SbLaunch l;
  l.stream = ctx->stream;
  l.n_frames = n_frames;
  l.lds_bytes = lds_bytes;
  l.a.first_frame = first_frame;
  l.a.sb_w = sb_w; l.a.sb_h = sb_h; l.a.range = range; l.a.cells_per_row = cells_per_row;
  l.a.xmin = -ref->border; l.a.xmax = ref->width + ref->border;
  l.a.ymin = -ref->border; l.a.ymax = ref->height + ref->border;
  l.a.n_buckets = n_buckets;
  l.a.buckets8 = (n_buckets + 7) & ~7;
  l.a.pitch = pitch;
  l.a.shift = src->bit_depth == 10 ? 2 : src->bit_depth == 12 ? 4 : 0;
  l.a.n_items = l.a.buckets8 * n_frames;
  {  // persistent grid: what the chip holds at once, a multiple of 8 so that item % 8 keeps naming one XCD
    static thread_local int cus = 0;
    if (!cus) {
      hipDeviceProp_t prop;
      AOMHIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
      cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int per_cu = (int)((160 * 1024) / lds_bytes) < 1 ? 1 : (int)((160 * 1024) / lds_bytes);
    int grid = cus * (per_cu > 2 ? 2 : per_cu);
    if (const char *e = getenv("AOMHIP_SB_GRID")) grid = atoi(e);
    grid &= ~7;
    if (grid < 8) grid = 8;
    if (grid > l.a.n_items) grid = l.a.n_items;
    l.grid = grid;
  }
  l.groups = d_groups; l.group_off = d_group_bucket_offsets; l.n_groups = n_groups; l.gfs = group_frame_stride;
  l.out4 = d_out_groups;
  l.cands = d_cands; l.cand_off = d_cand_bucket_offsets; l.n_cands = n_cands; l.cfs = cand_frame_stride;
  l.out1 = d_out_cands;
  const bool skip = (flags & AOMHIP_SAD_SKIP_ROWS) != 0;
  if (src->bit_depth == 8) return sb::dispatch<uint8_t>(l, skip, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), bw, bh);
  return sb::dispatch<uint16_t>(l, skip, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), bw, bh);
}
