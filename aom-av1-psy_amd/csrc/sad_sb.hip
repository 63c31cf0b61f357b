// Superblock-bucketed block SAD on gfx950: the same arithmetic as sad.hip (aom_sadWxH, aom_sadWxHx4d,
// _skip_ and highbd forms; reference aom_dsp/sad.c:22-129,240-332), organised the way the encoder issues
// it -- superblock by superblock with every motion vector inside the superblock's search range
// (av1/encoder/encodeframe.c:1069 encode_sb_row; mv limits av1/encoder/mcomp.c:101 av1_set_mv_search_range).
//
// Why a second kernel.  In the direct kernels (sad.hip) every lane pulls its own 16-byte row out of a
// different 128-byte cache line, so a 16x16 candidate costs 16 full-line L2->L1 fills for 256 useful bytes;
// the measured bound is that fill path, not HBM (profiles/r01_sad_variants.md).  Here a workgroup takes one
// bucket (an sb_w x sb_h cell of source blocks) at a time, pulls the (sb_w + 2*range) x (sb_h + 2*range) reference
// window into LDS once with full-line coalesced loads (each byte crosses L2->L1 once per bucket), and evaluates
// every candidate of the bucket from LDS: aligned dword reads + v_alignbyte per 16 reference bytes, v_sad_u8 / v_sad_u16,
// DPP group reduction.  A candidate whose reference block is not wholly inside the window (the caller exceeded
// `range`) is still evaluated, straight from global memory -- slower, never wrong.
// Measured on one MI355X, Mode A 16x16, ring of 64 4K frame pairs (profiles/r01_sad_sb.md): 2.29e10 candidates/s
// for 8-bit (direct kernels 1.10e10), 1.14e10 for 10-bit (7.2e9).
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

constexpr int kLdsPadBytes = 16;  // row pitch = window bytes + 16: 16 consecutive rows start in 16 distinct bank quads

// BYTES at an arbitrary byte offset of the LDS window: BYTES/4 + 1 aligned dword reads, realigned in registers with
// v_alignbyte.  (gfx950 does execute a ds_read_b128 at any byte address, but a misaligned one runs at 1/12 of the
// aligned rate -- tools/lds_unaligned_probe.hip: 0.61 vs 7.4 T lane-reads/s -- which made the LDS the bottleneck.)
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
  int dummy_off;               // 16 spare bytes behind the window
  int shift;
};

// What a workgroup needs to know about one (frame, bucket) work item.
struct Item {
  int next;  // item index to continue the walk from
  int f_rel, g0, g1, c0, c1;
  int wx0, wx1, wy0, wy1, cpr;  // window in pixels (x1 / y1 exclusive); 16-byte chunks per window row
  int valid;
};

// Persistent workgroups: the launch has as many workgroups as the chip holds at once (two 68 KB windows per CU)
// and each walks the (frame, bucket) items blockIdx + k * gridDim.  While the candidates of item i are evaluated
// out of LDS, the window of item i+1 is already in flight into registers (8 x 16 bytes per lane) and so are the
// descriptors and source rows of its first round of candidates; after a barrier the registers are dropped into
// LDS.  HBM / L2 latency and workgroup turnover overlap with the SAD arithmetic instead of adding to it
// (measured: the one-shot form of this kernel spent 0.19 of its 0.45 ms per launch in turnover + source latency).
//
// gfx9 returns vector-memory loads in order (one vmcnt counter), so the overlap only exists if the evaluation of
// item i waits on NO load younger than the prefetch of item i+1.  Hence two evaluation paths, chosen per wavefront:
//   fast    -- first round, every reference block inside the window, single candidates that re-use the source rows
//              of their lane's x4d group (Mode-A style lists): LDS reads and ALU only;
//   generic -- anything else (later rounds of a crowded bucket, blocks outside the window, unrelated single
//              candidates, large blocks): loads what it needs, waits for it, still exact.
template <typename T, int W, int H, bool SKIP, int kThreads>
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
  constexpr int kRegChunks = 8;                      // window chunks a lane keeps in flight in registers
  constexpr int kPerWg = kThreads / G::kTpc;         // candidates (or groups) evaluated side by side
  constexpr bool kPrefetch = G::kUnitsPerLane <= 4;  // source rows of the first round ride along with the window
  constexpr int kES = (int)sizeof(T);
  const int lane_in_cand = threadIdx.x % G::kTpc;
  const int slot = (int)threadIdx.x / G::kTpc;

  auto decode = [&](int item) {
    Item it;
    it.valid = 0;
    it.f_rel = it.g0 = it.g1 = it.c0 = it.c1 = it.wx0 = it.wx1 = it.wy0 = it.wy1 = 0;
    it.cpr = 1;
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
      // whole 16-byte chunks only: a ragged tail (clamped window) is served by the generic path instead
      it.cpr = ((min(cell_x * a.sb_w + a.sb_w + a.range, a.xmax) - it.wx0) * kES) >> 4;
      it.wx1 = it.wx0 + it.cpr * (16 / kES);
      it.wy0 = max(cell_y * a.sb_h - a.range, a.ymin);
      it.wy1 = min(cell_y * a.sb_h + a.sb_h + a.range, a.ymax);
      it.valid = 1;
      break;
    }
    it.next = item + (int)gridDim.x;
    return it;
  };

  auto unit_pos = [&](int k, int &row, int &col) {
    const int u = lane_in_cand + k * G::kTpc;
    row = (u / G::kUnitsPerRow) * G::kRowStep;
    col = (u % G::kUnitsPerRow) * G::kUnitElems;
  };
  auto src_unit = [&](int f_rel, int sx, int sy, int k) {
    int row, col;
    unit_pos(k, row, col);
    return *reinterpret_cast<const L *>(src.origin + (int64_t)(a.first_frame + f_rel) * src.frame_stride +
                                        (int64_t)(sy + row) * src.stride + sx + col);
  };

  // Everything of an item that is requested ahead of time.
  struct Ahead {
    U128 win[kRegChunks];
    L gsrc[kPrefetch ? G::kUnitsPerLane : 1];
  };
  // First-round descriptors of an item.  They are requested TWO items ahead: the source-row addresses depend on
  // them, and a dependent load issued after the window loads would have to wait for all of those (in-order vmcnt).
  // (kept as plain dwords: halfword h of a descriptor is word h / 2, shifted by 16 * (h & 1))
  struct Desc {
    uint32_t g[5];  // aomhip_sad_x4d_cand: sx, sy, rx[4], ry[4]
    uint32_t c[2];  // aomhip_sad_cand: sx, sy, rx, ry
  };
  struct __attribute__((packed, aligned(4))) W5 { uint32_t v[5]; };
  struct __attribute__((packed, aligned(4))) W2 { uint32_t v[2]; };
  auto load_desc = [&](const Item &it, Desc &d) {
    // (no data-dependent branch: an index past the bucket is clamped into the list and the entry ignored later)
    if constexpr (kPrefetch) {
      if (groups) {
        const W5 w = *reinterpret_cast<const W5 *>(groups + (int64_t)it.f_rel * group_frame_stride +
                                                   max(min(it.g0 + slot, n_groups - 1), 0));
#pragma unroll
        for (int i = 0; i < 5; ++i) d.g[i] = w.v[i];
      }
      if (cands) {
        const W2 w = *reinterpret_cast<const W2 *>(cands + (int64_t)it.f_rel * cand_frame_stride +
                                                   max(min(it.c0 + slot, n_cands - 1), 0));
        d.c[0] = w.v[0];
        d.c[1] = w.v[1];
      }
    }
  };
  auto half = [](const uint32_t *w, int h) { return (int)(int16_t)(w[h >> 1] >> (16 * (h & 1))); };
  // Chunk q = threadIdx + k * kThreads of a window sits at (row, chunk-in-row) = (q / cpr, q % cpr); the pair is
  // advanced incrementally (one division per item per lane).  Every load is unconditional (indices clamped to the
  // last chunk / last list entry) so that the code stays straight-line and the compiler's vmcnt bookkeeping exact.
  auto request = [&](const Item &it, const Desc &d, Ahead &h) {
    const char *g = reinterpret_cast<const char *>(ref.origin + (int64_t)(a.first_frame + it.f_rel) * ref.frame_stride +
                                                   (int64_t)it.wy0 * ref.stride + it.wx0);
    const int gpitch = ref.stride * kES;
    const int rows = it.wy1 - it.wy0;
    const int dr = kThreads / it.cpr, dc = kThreads - dr * it.cpr;
    int r = (int)threadIdx.x / it.cpr, c = (int)threadIdx.x - r * it.cpr;
#pragma unroll
    for (int k = 0; k < kRegChunks; ++k) {
      const int rr = min(r, rows - 1);
      h.win[k] = *reinterpret_cast<const U128 *>(g + (int64_t)rr * gpitch + c * 16);
      r += dr; c += dc;
      if (c >= it.cpr) { c -= it.cpr; ++r; }
    }
    if constexpr (kPrefetch) {
      if (groups) {
#pragma unroll
        for (int k = 0; k < G::kUnitsPerLane; ++k) h.gsrc[k] = src_unit(it.f_rel, half(d.g, 0), half(d.g, 1), k);
      }
    }
  };
  // Registers -> LDS; chunks beyond kRegChunks per lane (windows larger than 64 KB) are copied synchronously.
  // Straight-line on purpose: a lane without a chunk stores to a dummy slot behind the window instead of branching,
  // so that the compiler's wait-count bookkeeping sees every prefetched register consumed on every path (a skipped
  // conditional store leaves "maybe pending" state behind and turns later waits into vmcnt(0)).
  auto commit = [&](const Item &it, const Ahead &h) {
    const int total = (it.wy1 - it.wy0) * it.cpr;
    const int dr = kThreads / it.cpr, dc = kThreads - dr * it.cpr;
    int r = (int)threadIdx.x / it.cpr, c = (int)threadIdx.x - r * it.cpr;
#pragma unroll
    for (int k = 0; k < kRegChunks; ++k) {
      const int off = (int)threadIdx.x + k * kThreads < total ? r * a.pitch + c * 16 : a.dummy_off;
      *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(lds) + off) =
          make_uint4(h.win[k].v[0], h.win[k].v[1], h.win[k].v[2], h.win[k].v[3]);
      r += dr; c += dc;
      if (c >= it.cpr) { c -= it.cpr; ++r; }
    }
    if (total > kRegChunks * kThreads) {
      const char *g = reinterpret_cast<const char *>(ref.origin + (int64_t)(a.first_frame + it.f_rel) * ref.frame_stride +
                                                     (int64_t)it.wy0 * ref.stride + it.wx0);
      const int gpitch = ref.stride * kES;
      const int rows = it.wy1 - it.wy0;
      for (int q0 = (int)threadIdx.x + kRegChunks * kThreads; q0 - (int)threadIdx.x < total; q0 += 4 * kThreads) {
        U128 t[4];
        int lo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          t[k] = *reinterpret_cast<const U128 *>(g + (int64_t)min(r, rows - 1) * gpitch + c * 16);
          lo[k] = q0 + k * kThreads < total ? r * a.pitch + c * 16 : a.dummy_off;
          r += dr; c += dc;
          if (c >= it.cpr) { c -= it.cpr; ++r; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
          *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(lds) + lo[k]) = make_uint4(t[k].v[0], t[k].v[1], t[k].v[2], t[k].v[3]);
      }
    }
  };

  auto inside = [&](const Item &it, int rx, int ry) {
    return rx >= it.wx0 && rx + W <= it.wx1 && ry >= it.wy0 && ry + H <= it.wy1;
  };
  auto lds_off = [&](const Item &it, int rx, int ry) {
    return (unsigned)((ry - it.wy0) * a.pitch + (rx - it.wx0) * kES);
  };
  auto store4 = [&](const Item &it, int gi, uint32_t (&acc)[4]) {
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
  };
  auto store1 = [&](const Item &it, int ci, uint32_t acc) {
    acc = group_sum<G::kTpc>(acc);
    if (lane_in_cand == 0) out1[(int64_t)it.f_rel * n_cands + ci] = (SKIP ? 2u * acc : acc) >> a.shift;
  };
  // generic evaluation of one reference block against source rows fetched here
  auto generic_ref = [&](const Item &it, int sx, int sy, int rx, int ry) {
    const T *rbase = ref.origin + (int64_t)(a.first_frame + it.f_rel) * ref.frame_stride;
    const bool in = inside(it, rx, ry);
    const unsigned loff = lds_off(it, rx, ry);
    uint32_t acc = 0;
#pragma unroll(kPrefetch ? G::kUnitsPerLane : 4)
    for (int k = 0; k < G::kUnitsPerLane; ++k) {
      int row, col;
      unit_pos(k, row, col);
      const L sv = src_unit(it.f_rel, sx, sy, k);
      L r;
      if (in)
        r = lds_unit<G::kUnitBytes>(lds, loff + (unsigned)(row * a.pitch + col * kES));
      else
        r = *reinterpret_cast<const L *>(rbase + (int64_t)(ry + row) * ref.stride + rx + col);
#pragma unroll
      for (int i = 0; i < G::kUnitBytes / 4; ++i) acc = sad_dword<T>(sv.v[i], r.v[i], acc);
    }
    return acc;
  };
  auto generic_group = [&](const Item &it, int gi, const aomhip_sad_x4d_cand &d) {
    // one copy of generic_ref in the code, no dynamically indexed arrays (they would live in scratch)
    uint64_t rx4, ry4;
    memcpy(&rx4, d.rx, 8);
    memcpy(&ry4, d.ry, 8);
    uint32_t acc[4] = { 0, 0, 0, 0 };
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
      const uint32_t v = generic_ref(it, d.sx, d.sy, (int16_t)(rx4 >> (16 * j)), (int16_t)(ry4 >> (16 * j)));
      acc[0] = j == 0 ? v : acc[0];
      acc[1] = j == 1 ? v : acc[1];
      acc[2] = j == 2 ? v : acc[2];
      acc[3] = j == 3 ? v : acc[3];
    }
    store4(it, gi, acc);
  };

  Ahead cur;
  Desc d_it, d_n1, d_n2;
#pragma unroll
  for (int i = 0; i < 5; ++i) d_it.g[i] = d_n1.g[i] = d_n2.g[i] = 0;
  d_it.c[0] = d_it.c[1] = d_n1.c[0] = d_n1.c[1] = d_n2.c[0] = d_n2.c[1] = 0;
  Item it = decode((int)blockIdx.x);
  Item n1 = it;
  n1.valid = 0;
  if (it.valid) {
    load_desc(it, d_it);
    n1 = decode(it.next);
    load_desc(n1.valid ? n1 : it, d_n1);
    request(it, d_it, cur);
  }
  while (it.valid) {
    commit(it, cur);
    // source rows of this item move out of the way of the next request
    aomhip_sad_x4d_cand gdesc;
    aomhip_sad_cand cdesc;
    gdesc.sx = (int16_t)half(d_it.g, 0); gdesc.sy = (int16_t)half(d_it.g, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { gdesc.rx[j] = (int16_t)half(d_it.g, 2 + j); gdesc.ry[j] = (int16_t)half(d_it.g, 6 + j); }
    cdesc.sx = (int16_t)half(d_it.c, 0); cdesc.sy = (int16_t)half(d_it.c, 1);
    cdesc.rx = (int16_t)half(d_it.c, 2); cdesc.ry = (int16_t)half(d_it.c, 3);
    L gsrc[kPrefetch ? G::kUnitsPerLane : 1];
    if constexpr (kPrefetch) {
#pragma unroll
      for (int k = 0; k < G::kUnitsPerLane; ++k) gsrc[k] = cur.gsrc[k];
    }
    __syncthreads();
    // Requests are unconditional (past the end of the walk they re-fetch the current item): a conditional request
    // would merge "old" and "new" register values at the join and make the compiler wait for the loads right there.
    const Item rq = n1.valid ? n1 : it;
    Desc d_rq;
#pragma unroll
    for (int i = 0; i < 5; ++i) d_rq.g[i] = n1.valid ? d_n1.g[i] : d_it.g[i];
    request(rq, d_rq, cur);  // first, so that the window is in flight while the scalar loads of decode() return
    const Item n2 = decode(n1.next);  // (an invalid n1 carries next >= n_items, so n2 is invalid too)
    load_desc(n2.valid ? n2 : rq, d_n2);  // consumed at the end of this iteration, after the window has landed anyway

    int gi = it.g0 + slot, ci = it.c0 + slot;
    if constexpr (kPrefetch) {
      // ---- first round, x4d groups
      const bool g_act = gi < it.g1;
      bool g_in = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) g_in = g_in && inside(it, gdesc.rx[j], gdesc.ry[j]);
      if (__all(!g_act || g_in)) {
        if (g_act) {
          uint32_t acc[4] = { 0, 0, 0, 0 };
          unsigned loff[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) loff[j] = lds_off(it, gdesc.rx[j], gdesc.ry[j]);
#pragma unroll
          for (int k = 0; k < G::kUnitsPerLane; ++k) {
            int row, col;
            unit_pos(k, row, col);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const L r = lds_unit<G::kUnitBytes>(lds, loff[j] + (unsigned)(row * a.pitch + col * kES));
#pragma unroll
              for (int i = 0; i < G::kUnitBytes / 4; ++i) acc[j] = sad_dword<T>(gsrc[k].v[i], r.v[i], acc[j]);
            }
          }
          store4(it, gi, acc);
        }
      } else if (g_act) {
        generic_group(it, gi, gdesc);
      }
      gi += kPerWg;
      // ---- first round, single candidates
      const bool c_act = ci < it.c1;
      const bool c_fast = g_act && cdesc.sx == gdesc.sx && cdesc.sy == gdesc.sy && inside(it, cdesc.rx, cdesc.ry);
      if (__all(!c_act || c_fast)) {
        if (c_act) {
          const unsigned loff = lds_off(it, cdesc.rx, cdesc.ry);
          uint32_t acc = 0;
#pragma unroll
          for (int k = 0; k < G::kUnitsPerLane; ++k) {
            int row, col;
            unit_pos(k, row, col);
            const L r = lds_unit<G::kUnitBytes>(lds, loff + (unsigned)(row * a.pitch + col * kES));
#pragma unroll
            for (int i = 0; i < G::kUnitBytes / 4; ++i) acc = sad_dword<T>(gsrc[k].v[i], r.v[i], acc);
          }
          store1(it, ci, acc);
        }
      } else if (c_act) {
        store1(it, ci, generic_ref(it, cdesc.sx, cdesc.sy, cdesc.rx, cdesc.ry));
      }
      ci += kPerWg;
    }
    // ---- remaining rounds (crowded buckets; every round for large blocks)
    for (; gi < it.g1; gi += kPerWg) generic_group(it, gi, groups[(int64_t)it.f_rel * group_frame_stride + gi]);
    for (; ci < it.c1; ci += kPerWg) {
      const aomhip_sad_cand d = cands[(int64_t)it.f_rel * cand_frame_stride + ci];
      store1(it, ci, generic_ref(it, d.sx, d.sy, d.rx, d.ry));
    }
    __syncthreads();  // every lane is done reading this window
    it = n1;
    d_it = d_n1;
    n1 = n2;
    d_n1 = d_n2;
  }
}

struct SbLaunch {
  hipStream_t stream;
  int n_frames;
  int grid;
  int threads;
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

template <typename T, int W, int H, bool SKIP, int kThreads>
static int launch_nt(const SbLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r) {
  auto k = sad_sb_kernel<T, W, H, SKIP, kThreads>;
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

// 512-lane workgroups when two windows fit a CU (they overlap each other), 1024-lane ones when only one does.
template <typename T, int W, int H, bool SKIP>
static int launch(const SbLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r) {
  // blocks with 64 lanes per candidate keep 512 lanes: 1024 would only add idle slots for the few blocks of a cell
  if (l.threads == 1024 && Geom<T, W, H, SKIP>::kTpc < 64) return launch_nt<T, W, H, SKIP, 1024>(l, s, r);
  return launch_nt<T, W, H, SKIP, 512>(l, s, r);
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
  const size_t lds_bytes = (size_t)pitch * (sb_h + 2 * range) + 48;
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
  l.a.dummy_off = pitch * (sb_h + 2 * range) + 32;  // (the 32 bytes before it absorb the 5th dword of edge reads)
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
    l.threads = per_cu >= 2 ? 512 : 1024;
    if (const char *e = getenv("AOMHIP_SB_THREADS")) l.threads = atoi(e);
    if (const char *e = getenv("AOMHIP_SB_GRID")) grid = atoi(e);
    grid &= ~7;
    if (grid < 8) grid = 8;
    if (grid > l.a.n_items) grid = l.a.n_items;
    l.grid = grid;
  }
  l.groups = n_groups > 0 ? d_groups : nullptr; l.group_off = d_group_bucket_offsets; l.n_groups = n_groups; l.gfs = group_frame_stride;
  l.out4 = d_out_groups;
  l.cands = n_cands > 0 ? d_cands : nullptr; l.cand_off = d_cand_bucket_offsets; l.n_cands = n_cands; l.cfs = cand_frame_stride;
  l.out1 = d_out_cands;
  const bool skip = (flags & AOMHIP_SAD_SKIP_ROWS) != 0;
  if (src->bit_depth == 8) return sb::dispatch<uint8_t>(l, skip, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), bw, bh);
  return sb::dispatch<uint16_t>(l, skip, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), bw, bh);
}
