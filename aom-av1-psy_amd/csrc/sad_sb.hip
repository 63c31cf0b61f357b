// Superblock-bucketed block SAD on gfx950: the same arithmetic as sad.hip (aom_sadWxH, aom_sadWxHx4d,
// _skip_ and highbd forms; reference aom_dsp/sad.c:22-129,240-332), organised the way the encoder issues
// it -- superblock by superblock with every motion vector inside the superblock's search range
// (av1/encoder/encodeframe.c:1069 encode_sb_row; mv limits av1/encoder/mcomp.c:101 av1_set_mv_search_range).
//
// Structure (round 2; the round-1 form staged one (sb_w + 2 range) x (sb_h + 2 range) window per cell and so
// pulled every reference byte through L2 -> LDS 2.7 times and through the fabric 1.4-1.9 times):
//   * A persistent workgroup owns one STRIP -- a column of cells of one frame -- and walks it top to bottom.
//     The reference window lives in an LDS ring of R = 2 sb_h + 2 range rows indexed by (y - ymin) mod R, so
//     going from one cell to the next brings in only the sb_h NEW rows: every reference byte of the strip
//     crosses L2 -> LDS once; what is left of the halo is the 2 range columns shared with the neighbour strips.
//   * Wave specialisation: 8 EVALUATING wavefronts + 8 LOADER wavefronts per workgroup.  The loaders bring in
//     everything a step needs -- the new ring rows, the step's SOURCE cell (double buffered) and its slice of the
//     two work lists -- through registers: the loads for step cy + 2 are in flight (in the loaders' VGPRs, so
//     they need no LDS space yet) while step cy is evaluated and step cy + 1's data is written to LDS.  The
//     evaluating wavefronts never issue a vector-memory load -- gfx9 returns those in order (one vmcnt), a
//     single demand load would wait for everything older -- only ds_read, VALU and the result stores; one
//     workgroup barrier per step.  (LDS-DMA was tried first and dropped: a global_load_lds holds its wavefront
//     in the issue stage ~260 cycles, and data in flight must already own LDS space, which forces the memory
//     pipeline to drain at every step -- profiles/r02_sad_strip.md.)
//   * Strips of one frame go to workgroups of one XCD at the same time (workgroup b runs on XCD b % 8; speed
//     only), so the halo columns two neighbour strips both read are served by that XCD's L2.
//   * Any entry whose blocks are not inside the staged source cell / reference window (the caller exceeded
//     `range`, or put an entry in the wrong bucket) is still evaluated, straight from global memory -- slower,
//     never wrong.  Lists of any shape go through the same rounds (no "first round only" fast path).
#include <type_traits>

#include "common.h"
#include "variance_device.h"
#undef AOMHIP_FOR_BLOCK_SIZES   // (variance_device.h's list; this file has its own below)

namespace aomhip {
namespace sb {

// What a candidate accumulates.  SAD: one sum.  VARIANCE (round 6: aom_varianceWxH through the same strip walk, aom_dsp/variance.c:56-163,
// 383-420): S(r), S(r^2), S(s r) per reference and S(s), S(s^2) of the source block -- sum = S(s) - S(r), sse = S(s^2) + S(r^2) - 2 S(s r)
// in 32-bit modular arithmetic, exact because the true sse of a block of <= 256 pixels is < 2^32 at every bit depth (256 x 4095^2); the
// packed dot products take 4 (8-bit) or 2 (16-bit) pixels per instruction.
template <bool VAR> struct Ac;
template <> struct Ac<false> { uint32_t a = 0; };
template <> struct Ac<true> { uint32_t a = 0, rr = 0, xr = 0, ssum = 0, ss = 0; };   // a = S(r)
template <bool VAR> struct Res;
template <> struct Res<false> { uint32_t v; };
template <> struct Res<true> { uint32_t v, sse; };
template <typename T> __device__ __forceinline__ uint32_t dot_dword(uint32_t a, uint32_t b, uint32_t acc) {
  if constexpr (sizeof(T) == 1) return __builtin_amdgcn_udot4(a, b, acc, false);
  else return __builtin_amdgcn_udot2(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned short)))) unsigned short, a),
                                     __builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned short)))) unsigned short, b), acc, false);
}

struct __attribute__((packed, aligned(1))) U128 { uint32_t v[4]; };
struct __attribute__((packed, aligned(1))) U64 { uint32_t v[2]; };
struct __attribute__((packed, aligned(1))) U32 { uint32_t v[1]; };
template <int BYTES> struct UnitLoad;
template <> struct UnitLoad<16> { using type = U128; using aligned_type = uint4; };
template <> struct UnitLoad<8> { using type = U64; using aligned_type = uint2; };
template <> struct UnitLoad<4> { using type = U32; using aligned_type = uint32_t; };

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

// UPL = target number of 16-byte row units a lane owns (2: 8 lanes per 16x16 8-bit block; 1: 16 lanes); 0: 16 lanes per
// block whatever its size (fewer for blocks of fewer than 16 units).
template <typename T, int W, int H, bool SKIP, int UPL> struct Geom {
  static constexpr int kRowBytes = W * (int)sizeof(T);
  static constexpr int kUnitBytes = kRowBytes < 16 ? kRowBytes : 16;
  static constexpr int kUnitElems = kUnitBytes / (int)sizeof(T);
  static constexpr int kUnitsPerRow = kRowBytes / kUnitBytes;
  static constexpr int kRows = SKIP ? H / 2 : H;
  static constexpr int kUnits = kUnitsPerRow * kRows;
  static constexpr int kTpcRaw = UPL > 0 ? (kUnits >= UPL ? kUnits / UPL : 1) : (kUnits < 16 ? kUnits : 16);
  static constexpr int kTpc = kTpcRaw > 64 ? 64 : kTpcRaw;
  static constexpr int kUnitsPerLane = kUnits / kTpc;
  static constexpr int kRowStep = SKIP ? 2 : 1;
};

// BYTES at an arbitrary byte offset of LDS: BYTES/4 + 1 aligned dword reads, realigned in registers with v_alignbyte.
// (Any LDS read off its natural alignment -- ds_read_b32 / _b64 / _b96 / _b128 alike -- returns the right bytes but is replayed at ~64
// cycles per wave-instruction on gfx950: tools/lds_ubench.hip, profiles/r03_ubench.log: a 20-byte row at a random byte offset costs
// 134 ns as five misaligned b32, 54 ns as b128 + b32, against 17 ns for five aligned dwords; v_qsad_pk_u16_u8 / v_mqsad_* issue at
// 1/3.6 of v_sad_u8's rate, so the quad-SAD forms lose to v_alignbyte + v_sad_u8 as well.)
template <int BYTES>
__device__ __forceinline__ typename UnitLoad<BYTES>::type lds_unit(const char *lds, unsigned byte_off) {
  const uint32_t *p = reinterpret_cast<const uint32_t *>(lds) + (byte_off >> 2);
  const unsigned sh = byte_off & 3;
  uint32_t d[BYTES / 4 + 1];
#pragma unroll
  for (int i = 0; i <= BYTES / 4; ++i) d[i] = p[i];
  typename UnitLoad<BYTES>::type out;
#pragma unroll
  for (int i = 0; i < BYTES / 4; ++i) out.v[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
  return out;
}
template <int BYTES>
__device__ __forceinline__ typename UnitLoad<BYTES>::type lds_unit_aligned(const char *lds, unsigned byte_off) {
  using A = typename UnitLoad<BYTES>::aligned_type;
  const A t = *reinterpret_cast<const A *>(lds + byte_off);
  typename UnitLoad<BYTES>::type out;
  memcpy(&out, &t, BYTES);
  return out;
}

struct StripArgs {
  int first_frame, n_frames;
  int sb_w, sb_h, range, cells_per_row, cell_rows;
  // reference plane: readable pixel range relative to the visible origin, allocated row end, border
  int xmin, xmax, ymin, ymax, row_end, border;
  // source plane
  int s_xmax, s_ymax, s_row_end, s_border;
  int cpr, pitch, R;           // ring: 16-byte chunks per row, row pitch in bytes, rows
  int scpr, spitch;            // source cell: chunks per row, pitch
  unsigned magic_cpr, magic_scpr, magic_R;
  int ring_off, src_off[2], gdesc_off[2], cdesc_off[2], misc_off, seg_off;  // LDS byte offsets
  int gcap, ccap;              // list entries per descriptor buffer
  int shift;
  int dbg;  // AOMHIP_SB_DBG (timing ablations only, results invalid): 1 = no evaluation, 2 = no steady-state DMA
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

#ifndef AOMHIP_SB_DBG_KNOBS
#define AOMHIP_SB_DBG_KNOBS 0
#endif
#ifdef AOMHIP_SB_PROF  // phase timing of wave 0 of every workgroup (tools/gpu_sb_prof.sh builds a library with it)
__device__ unsigned long long g_sb_prof[32];  // [0..7] first evaluating wavefront, [8..15] first loader wavefront, [16..31] barrier wait of wavefront w
#define SB_T(v) const long long v = (long long)__builtin_readcyclecounter()
#define SB_ACC(i, t1, t0) prof_acc[i] += (unsigned long long)((t1) - (t0))
#define SB_DECL unsigned long long prof_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, prof_wait = 0
#define SB_WAIT(t1, t0) prof_wait += (unsigned long long)((t1) - (t0))
#define SB_FLUSH if (tid == 0 || tid == kThreads) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_sb_prof[(tid ? 8 : 0) + i_], prof_acc[i_]); } \
  if (lane == 0) atomicAdd(&g_sb_prof[16 + wave], prof_wait);
#else
#define SB_DECL
#define SB_FLUSH
#define SB_T(v)
#define SB_ACC(i, t1, t0)
#define SB_WAIT(t1, t0)
#endif

// Ring rows that are stored twice: a block of height <= kMirrorMax + 1 starting in any ring slot then reads consecutive
// LDS rows (slots R .. R + H - 2 repeat slots 0 .. H - 2), so the evaluation needs no per-row wrap arithmetic.
constexpr int kMirrorMax = 15;
__host__ __device__ constexpr int mirror_rows(int h) { return h - 1 <= kMirrorMax ? h - 1 : 0; }

// A workgroup = C::kThreads evaluating lanes + C::kLoaders loader wavefronts (the last ones), 1024 lanes in all (the 128-VGPR budget).
// kRingN / kSrcN / kGN / kCN: 16-byte chunks / dwords one loader lane has in flight for its step; the loader wavefronts work in two
// groups of kLT lanes that take alternate steps.
template <int EVAL, int UPL_, int LOADERS, int RING_N, int SRC_N, int G_N, int C_N> struct Cfg {
  static constexpr int kThreads = EVAL, kUPL = UPL_, kLoaders = LOADERS, kRingN = RING_N, kSrcN = SRC_N, kGN = G_N, kCN = C_N;
  static constexpr int kLT = 64 * LOADERS / 2;
  static constexpr int kAll = EVAL + 64 * LOADERS;
};
// 8 evaluating wavefronts with two row units per lane + 8 loader wavefronts.
using CfgWide = Cfg<512, 2, 8, 6, 4, 2, 1>;
// 12 evaluating wavefronts with ONE row unit per lane + 4 loader wavefronts that keep twice the chunks in flight each (experiment,
// see use_deep()).
using CfgDeep = Cfg<768, 1, 4, 9, 6, 4, 2>;
template <typename T, int W, int H, bool SKIP, typename C, bool VAR = false>
__global__ __launch_bounds__(C::kAll) void sad_strip_kernel(PlaneView<T> src, PlaneView<T> ref, StripArgs a,
                                                                  const aomhip_sad_x4d_cand *__restrict__ groups,
                                                                  const int32_t *__restrict__ group_off, int n_groups,
                                                                  int64_t group_frame_stride, uint32_t *__restrict__ out4,
                                                                  const aomhip_sad_cand *__restrict__ cands,
                                                                  const int32_t *__restrict__ cand_off, int n_cands,
                                                                  int64_t cand_frame_stride, uint32_t *__restrict__ out1,
                                                                  uint32_t *__restrict__ out4b = nullptr, uint32_t *__restrict__ out1b = nullptr) {
  static_assert(!VAR || (!SKIP && W * H <= 256), "the variance form: whole blocks of at most 256 pixels (32-bit sums)");
  constexpr int kThreads = C::kThreads, UPL = C::kUPL, kRingN = C::kRingN, kSrcN = C::kSrcN, kGN = C::kGN,
                kCN = C::kCN, kLT = C::kLT;
  using G = Geom<T, W, H, SKIP, UPL>;
  using L = typename UnitLoad<G::kUnitBytes>::type;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int kES = (int)sizeof(T);
  constexpr int kEpc = 16 / kES;  // elements per 16-byte chunk
  constexpr int kPerWg = kThreads / G::kTpc;
  constexpr int kWaves = kThreads / 64;       // evaluating wavefronts; wavefront kWaves is the loader
  constexpr int kAll = C::kAll;
  constexpr int kGenUnroll = G::kUnitsPerLane <= 4 ? G::kUnitsPerLane : 4;
  constexpr int kMirror = mirror_rows(H);
  const int tid = (int)threadIdx.x;
  // AOMHIP_SB_DBG (timing ablations, results invalid) is honoured only by the experiment / profiling builds (make exp, make prof): the
  // product kernel carries none of those tests in its loops.
#if AOMHIP_SB_DBG_KNOBS
  const int dbg = a.dbg;
#else
  constexpr int dbg = 0;
#endif
  const int lane = tid & 63;
  const int wave = uni(tid >> 6);
  const bool is_loader = wave >= kWaves;
  const int grp = uni((tid - kThreads) / kLT);  // loader group (0 / 1)
  const int lt = tid - kThreads - grp * kLT;    // lane index inside the group
  const int lane_in_cand = tid % G::kTpc;
  const int slot = tid / G::kTpc;

  // ---- which strips have work at all (the bucket offsets are shared by all frames): a rank of a tile-column
  // partition, or a caller with a partial list, only walks its own strips.
  int *misc = reinterpret_cast<int *>(lds + a.misc_off);  // [0] = n_active, [8 .. 15] = flag words, then uint8 active[256]
  uint8_t *active = reinterpret_cast<uint8_t *>(misc + 72);
  for (int i = tid; i < 72; i += kAll) misc[i] = 0;
  __syncthreads();
  {
    const int n_buckets = a.cells_per_row * a.cell_rows;
    for (int b = tid; b < n_buckets; b += kAll) {
      int n = 0;
      if (groups) n += group_off[b + 1] - group_off[b];
      if (cands) n += cand_off[b + 1] - cand_off[b];
      if (n > 0) {
        const int cx = b % a.cells_per_row;
        atomicOr(reinterpret_cast<unsigned *>(&misc[8 + (cx >> 5)]), 1u << (cx & 31));
      }
    }
    __syncthreads();
    if (tid < 64) {
      int count = 0;
      for (int base = 0; base < a.cells_per_row; base += 64) {
        const int cx = base + tid;
        const bool f = cx < a.cells_per_row && ((misc[8 + (cx >> 5)] >> (cx & 31)) & 1);
        const unsigned long long m = __ballot(f);
        if (f) active[count + __popcll(m & ((1ull << tid) - 1))] = (uint8_t)cx;
        count += __popcll(m);
      }
      if (tid == 0) misc[0] = count;
    }
    __syncthreads();
  }
  const int n_active = uni(misc[0]);
  if (n_active == 0) return;
  // The double-buffer offsets as selects over values read ONCE: indexing the by-value argument struct with the step's parity
  // compiles to scalar loads from the kernel-argument segment in every step, and a scalar-cache miss costs 1000-3000 cycles while
  // the loaders keep the memory system saturated (ISA of the first version: three s_load_dword + s_waitcnt at the top of every step).
  const int src_off0 = a.src_off[0], src_off1 = a.src_off[1], gdesc_off0 = a.gdesc_off[0], gdesc_off1 = a.gdesc_off[1];
  const int cdesc_off0 = a.cdesc_off[0], cdesc_off1 = a.cdesc_off[1];
  auto src_off_of = [&](int buf) { return buf ? src_off1 : src_off0; };
  auto gdesc_off_of = [&](int buf) { return buf ? gdesc_off1 : gdesc_off0; };
  auto cdesc_off_of = [&](int buf) { return buf ? cdesc_off1 : cdesc_off0; };

  // The two roles run two separate instantiations of everything below: values only one role needs (the evaluation's
  // unit tables, the loaders' staging registers and chunk offsets) then never share a live range with the other role's
  // -- with one copy of the code the loaders' loop spilled, and a scratch reload behind a batch of staged loads waits
  // for all of them (vector-memory operations return in order).
  auto run = [&](auto role) {
    constexpr bool kLoader = decltype(role)::value;
    // per-lane constants of the block-unit mapping: row of the lane's k-th unit, its byte offset in a ring row walk and
    // in a source-cell walk
    // Tables (registers) while a lane owns few units; a lane with many units walks them in partly unrolled loops, where a table would
    // be indexed dynamically and therefore live in scratch (272-720 bytes per lane for the 64x128 ... 128x128 instantiations): those
    // work the unit's row / column out of its index instead (the units-per-row count is a power of two).
    constexpr bool kTab = G::kUnitsPerLane <= 4;
    constexpr int kTabN = kTab ? G::kUnitsPerLane : 1;
    int unit_row_t[kTabN], unit_colb_t[kTabN];
    unsigned unit_roff_t[kTabN], unit_soff_t[kTabN];
    if constexpr (kTab) {
  #pragma unroll
      for (int k = 0; k < G::kUnitsPerLane; ++k) {
        const int u = lane_in_cand + k * G::kTpc;
        unit_row_t[k] = (u / G::kUnitsPerRow) * G::kRowStep;
        unit_colb_t[k] = (u % G::kUnitsPerRow) * G::kUnitBytes;
        unit_roff_t[k] = (unsigned)(unit_row_t[k] * a.pitch + unit_colb_t[k]);
        unit_soff_t[k] = (unsigned)(unit_row_t[k] * a.spitch + unit_colb_t[k]);
      }
    }
    auto unit_row = [&](int k) -> int {
      if constexpr (kTab) return unit_row_t[k]; else return ((lane_in_cand + k * G::kTpc) / G::kUnitsPerRow) * G::kRowStep;
    };
    auto unit_colb = [&](int k) -> int {
      if constexpr (kTab) return unit_colb_t[k]; else return ((lane_in_cand + k * G::kTpc) % G::kUnitsPerRow) * G::kUnitBytes;
    };
    auto unit_roff = [&](int k) -> unsigned {
      if constexpr (kTab) return unit_roff_t[k]; else return (unsigned)(unit_row(k) * a.pitch + unit_colb(k));
    };
    auto unit_soff = [&](int k) -> unsigned {
      if constexpr (kTab) return unit_soff_t[k]; else return (unsigned)(unit_row(k) * a.spitch + unit_colb(k));
    };

    // ---- the walk over (frame, strip) items
    const bool affine = a.n_frames >= 8;  // strips of one frame side by side on one XCD
    const int xcd = affine ? (int)(blockIdx.x & 7) : 0;
    const int wg_j = affine ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int wg_n = affine ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    const int frames_here = affine ? (a.n_frames - xcd + 7) >> 3 : a.n_frames;
    const int n_items = frames_here * n_active;

    SB_DECL;
    for (int item = wg_j; item < n_items; item += wg_n) {
      const int fi = item / n_active;
      const int f_rel = affine ? xcd + 8 * fi : fi;
      const int cx = uni((int)active[item - fi * n_active]);
      const int cell_x0 = cx * a.sb_w;
      // window / source-cell columns: start rounded down to a 16-byte boundary of the plane row
      const int wx0 = max(((cell_x0 - a.range + a.border) & ~(kEpc - 1)) - a.border, a.xmin);
      const int colmax = min(a.cpr - 1, (a.row_end - wx0) / kEpc - 1);  // last chunk that stays inside the row's allocation
      const int wx1 = min(wx0 + (colmax + 1) * kEpc, a.xmax);
      const int sx0 = ((cell_x0 + a.s_border) & ~(kEpc - 1)) - a.s_border;
      const int scolmax = min(a.scpr - 1, (a.s_row_end - sx0) / kEpc - 1);
      const int sx1 = min(sx0 + (scolmax + 1) * kEpc, a.s_xmax);
      const char *ref_frame = reinterpret_cast<const char *>(ref.origin + (int64_t)(a.first_frame + f_rel) * ref.frame_stride);
      const char *src_frame = reinterpret_cast<const char *>(src.origin + (int64_t)(a.first_frame + f_rel) * src.frame_stride);
      const int gpitch = ref.stride * kES, sgpitch = src.stride * kES;
      const aomhip_sad_x4d_cand *glist = groups ? groups + (int64_t)f_rel * group_frame_stride : nullptr;
      const aomhip_sad_cand *clist = cands ? cands + (int64_t)f_rel * cand_frame_stride : nullptr;

      // ---- transport (loader lanes): one BATCH = what step `cy` adds to LDS: new ring rows [ya, yb), source cell cy,
      // the list slices of cell cy.  request() starts the loads into registers, commit() writes them to LDS a step later.
      struct Batch { int ya, yb, sy0, ns, g0, ng, c0, nc, buf; unsigned first; };  // first: ring byte offset of row ya
      // The staged 16-byte chunks are NATIVE 4 x 32-bit vectors, moved as a whole from the load to the ds_write: as a struct of four
      // scalars the register allocator was free to park single components elsewhere (it did: `v_mov v68, v23` behind an
      // `s_waitcnt vmcnt(6)` at the end of request(), i.e. the loader waited for half of the loads it had just issued, every step).
      typedef uint32_t V4 __attribute__((ext_vector_type(4)));
      struct Stage { V4 ring[kRingN]; V4 srcv[kSrcN]; uint32_t g[kGN]; uint32_t c[kCN]; };
      // Every load is unconditional and straight-line (lanes past the end of a batch re-read its last chunk; an unused
      // register slot re-reads chunk 0): a conditional load would leave "maybe pending" registers behind at the join
      // and the compiler's wait-count pass would then drain vmcnt(0) at every later use.  `me` of `n` lanes take part.
      const uint32_t *gwords = groups ? reinterpret_cast<const uint32_t *>(glist) : reinterpret_cast<const uint32_t *>(ref_frame);
      const uint32_t *cwords = cands ? reinterpret_cast<const uint32_t *>(clist) : reinterpret_cast<const uint32_t *>(ref_frame);
      const int gwords_n = groups ? n_groups * 5 : 1, cwords_n = cands ? n_cands * 2 : 1;
      auto request_gen = [&](const Batch &b, Stage &st, int me, int n) {
        const int total_r = (b.yb - b.ya) * a.cpr, total_s = b.ns * a.scpr;
        const char *rb = ref_frame + (int64_t)b.ya * gpitch + (int64_t)wx0 * kES;
        const char *sb_ = src_frame + (int64_t)b.sy0 * sgpitch + (int64_t)sx0 * kES;
  #pragma unroll
        for (int i = 0; i < kRingN; ++i) {
          const unsigned q = (unsigned)max(min(me + i * n, total_r - 1), 0);
          const unsigned row = __umulhi(q, a.magic_cpr), col = q - row * (unsigned)a.cpr;
          st.ring[i] = *reinterpret_cast<const V4 *>(rb + (row * (unsigned)gpitch + (unsigned)min((int)col, colmax) * 16u));
        }
  #pragma unroll
        for (int i = 0; i < kSrcN; ++i) {
          const unsigned q = (unsigned)max(min(me + i * n, total_s - 1), 0);
          const unsigned row = __umulhi(q, a.magic_scpr), col = q - row * (unsigned)a.scpr;
          st.srcv[i] = *reinterpret_cast<const V4 *>(sb_ + (row * (unsigned)sgpitch + (unsigned)min((int)col, scolmax) * 16u));
        }
  #pragma unroll
        for (int i = 0; i < kGN; ++i) st.g[i] = gwords[(unsigned)max(min(b.g0 * 5 + min(me + i * n, b.ng * 5 - 1), gwords_n - 1), 0)];
  #pragma unroll
        for (int i = 0; i < kCN; ++i) st.c[i] = cwords[(unsigned)max(min(b.c0 * 2 + min(me + i * n, b.nc * 2 - 1), cwords_n - 1), 0)];
      };
      auto commit_gen = [&](const Batch &b, const Stage &st, int me, int n) {
        const int total_r = (b.yb - b.ya) * a.cpr, total_s = b.ns * a.scpr;
        const unsigned s_first = (unsigned)((b.ya - a.ymin) % a.R);
  #pragma unroll
        for (int i = 0; i < kRingN; ++i) {
          const unsigned q = (unsigned)(me + i * n);
          if ((int)q < total_r) {
            const unsigned row = __umulhi(q, a.magic_cpr), col = q - row * (unsigned)a.cpr;
            unsigned sl = s_first + row;
            sl = min(sl, sl - (unsigned)a.R);
            const V4 v = st.ring[i];
            *reinterpret_cast<V4 *>(lds + a.ring_off + sl * a.pitch + col * 16) = v;
            if (kMirror > 0 && sl < (unsigned)kMirror)
              *reinterpret_cast<V4 *>(lds + a.ring_off + (sl + a.R) * a.pitch + col * 16) = v;
          }
        }
  #pragma unroll
        for (int i = 0; i < kSrcN; ++i) {
          const unsigned q = (unsigned)(me + i * n);
          if ((int)q < total_s) {
            const unsigned row = __umulhi(q, a.magic_scpr), col = q - row * (unsigned)a.scpr;
            *reinterpret_cast<V4 *>(lds + src_off_of(b.buf) + row * a.spitch + col * 16) = st.srcv[i];
          }
        }
  #pragma unroll
        for (int i = 0; i < kGN; ++i)
          if (me + i * n < b.ng * 5) *reinterpret_cast<uint32_t *>(lds + gdesc_off_of(b.buf) + (me + i * n) * 4) = st.g[i];
  #pragma unroll
        for (int i = 0; i < kCN; ++i)
          if (me + i * n < b.nc * 2) *reinterpret_cast<uint32_t *>(lds + cdesc_off_of(b.buf) + (me + i * n) * 4) = st.c[i];
      };
      auto win_y0 = [&](int cy) { return max(cy * a.sb_h - a.range, a.ymin); };
      auto win_y1 = [&](int cy) { return min(cy * a.sb_h + a.sb_h + a.range, a.ymax); };
      struct Seg { int g0, g1, c0, c1; };
      // The strip's bucket bounds come out of LDS (filled in the prologue): a scalar load from global memory at the top of
      // a step took 1000-3000 cycles while the loaders keep the memory system saturated.
      const int4 *segs = reinterpret_cast<const int4 *>(lds + a.seg_off);
      int4 *wins = reinterpret_cast<int4 *>(lds + a.seg_off) + a.cell_rows, *bats = wins + a.cell_rows;
      auto seg_of = [&](int cy) {
        const int4 v = segs[cy];
        Seg s;
        s.g0 = uni(v.x); s.g1 = uni(v.y); s.c0 = uni(v.z); s.c1 = uni(v.w);
        return s;
      };
      // ---- evaluation of list entries out of LDS.  Per step: window rows [wy0, wy0 + wh), ring slot of row wy0, source
      // cell rows [sy0, sy0 + sh); per strip: window columns [wx0, wx0 + ww), source columns [sx0, sx0 + sw).
      struct Win { int wy0, s0; unsigned wh_ok, sy0, sh_ok; };  // *_ok: largest admissible block offset (unsigned compare)
      const unsigned ww_ok = (unsigned)(wx1 - wx0 - W), sw_ok = (unsigned)(sx1 - sx0 - W);
      const bool strip_ok = wx1 - wx0 >= W && sx1 - sx0 >= W;
      auto global_unit = [&](const char *frame, int pitch_b, int x, int y, int k) {
        return *reinterpret_cast<const L *>(frame + (int64_t)(y + unit_row(k)) * pitch_b + (int64_t)x * kES + unit_colb(k));
      };
      using AcT = Ac<VAR>;
      using ResT = Res<VAR>;
      // one reference dword against one source dword (the fast path hoists the source's own sums out of its five references)
      auto add_ref = [&](uint32_t sd, uint32_t rd, AcT &acc) {
        if constexpr (VAR) {
          acc.a = sad_dword<T>(rd, 0u, acc.a);
          acc.rr = dot_dword<T>(rd, rd, acc.rr);
          acc.xr = dot_dword<T>(sd, rd, acc.xr);
        } else {
          acc.a = sad_dword<T>(sd, rd, acc.a);
        }
      };
      auto add_src = [&](uint32_t sd, uint32_t &ssum, uint32_t &ss) {
        ssum = sad_dword<T>(sd, 0u, ssum);
        ss = dot_dword<T>(sd, sd, ss);
      };
      auto sad_unit = [&](const L &s, const L &r, AcT acc) {
  #pragma unroll
        for (int i = 0; i < G::kUnitBytes / 4; ++i) {
          add_ref(s.v[i], r.v[i], acc);
          if constexpr (VAR) add_src(s.v[i], acc.ssum, acc.ss);
        }
        return acc;
      };
      // the lane group's partial sums -> the candidate's result (every lane of the group ends up with it)
      auto reduce = [&](const AcT &acc) {
        ResT r;
        if constexpr (VAR) {
          const uint32_t dsum = group_sum<G::kTpc>(acc.ssum - acc.a);                  // S(s) - S(r), two's complement
          const uint32_t q = group_sum<G::kTpc>(acc.ss + acc.rr - 2u * acc.xr);
          finish<(sizeof(T) > 1), ilog2v(W * H)>((int64_t)(int32_t)dsum, (uint64_t)q, 8 + a.shift, &r.v, &r.sse);   // (a.shift 0 / 2 / 4 <-> 8 / 10 / 12 bits)
        } else {
          r.v = ((SKIP ? 2u * group_sum<G::kTpc>(acc.a) : group_sum<G::kTpc>(acc.a)) >> a.shift);
        }
        return r;
      };
      auto store_group = [&](int64_t idx, const ResT &r0, const ResT &r1, const ResT &r2, const ResT &r3) {
        reinterpret_cast<uint4 *>(out4)[idx] = make_uint4(r0.v, r1.v, r2.v, r3.v);
        if constexpr (VAR) reinterpret_cast<uint4 *>(out4b)[idx] = make_uint4(r0.sse, r1.sse, r2.sse, r3.sse);
      };
      auto store_cand = [&](int64_t idx, const ResT &r) {
        out1[idx] = r.v;
        if constexpr (VAR) out1b[idx] = r.sse;
      };
      auto generic_ref = [&](int sx, int sy, int rx, int ry) {  // both blocks straight from global memory
        AcT acc;
  #pragma unroll kGenUnroll
        for (int k = 0; k < G::kUnitsPerLane; ++k)
          acc = sad_unit(global_unit(src_frame, sgpitch, sx, sy, k), global_unit(ref_frame, gpitch, rx, ry, k), acc);
        return acc;
      };
      // LDS byte offset of the reference block's first row / first byte; false when the block is not wholly in the window
      auto ring_pos = [&](const Win &w, int rx, int ry, unsigned &base, int &slot0) {
        const unsigned dx = (unsigned)(rx - wx0), dy = (unsigned)(ry - w.wy0);
        unsigned t = (unsigned)w.s0 + dy;
        t = min(t, t - (unsigned)a.R);  // one wrap at most: t - R underflows to a huge value unless t >= R
        slot0 = (int)t;
        // (t < R <= 2^9 and the pitch < 2^12 bytes: v_mul_u32_u24, full rate -- the plain product is v_mul_lo_u32, a quarter-rate instruction on
        // the evaluating wavefronts' critical chain; an entry outside the window yields a garbage base that `ok` discards)
        base = (unsigned)a.ring_off + __umul24(t, (unsigned)a.pitch) + dx * kES;
        return dx <= ww_ok && dy <= w.wh_ok;
      };
      auto ring_unit = [&](unsigned base, int slot0, int k) {
        if constexpr (kMirror > 0) {
          return lds_unit<G::kUnitBytes>(lds, base + unit_roff(k));
        } else {
          unsigned s = (unsigned)(slot0 + unit_row(k));
          const unsigned wrap = s >= (unsigned)a.R ? (unsigned)(a.R * a.pitch) : 0u;
          return lds_unit<G::kUnitBytes>(lds, base + unit_roff(k) - wrap);
        }
      };
      auto src_pos = [&](const Win &w, int sx, int sy, unsigned &soff) {
        const unsigned dx = (unsigned)(sx - sx0), dy = (unsigned)sy - w.sy0;
        soff = __umul24(dy, (unsigned)a.spitch) + dx * kES;
        return dx <= sw_ok && dy <= w.sh_ok;
      };

      // One loop over the two list slices: iteration i handles group i and single candidate i of the slice (in a Mode-A
      // style list they belong to the same source block, whose rows are then read once for all five references).
      // The 4 reference positions of a group are decoded and located by 4 different lanes of the block's lane group
      // (lane & 3 = reference index) and then shared with quad broadcasts, instead of every lane redoing all of it:
      // the kernel is bound by the number of wave-instructions issued (PMC: 56 % of wave time waiting with two
      // wavefronts per SIMD, the rest issuing), so per-entry bookkeeping counts as much as the SAD arithmetic.
      constexpr bool kCoop = G::kTpc >= 4;
      const int my_j = lane & 3;                      // the reference this lane decodes (kCoop)
      const int my_w = my_j >> 1, my_sh = 16 * (my_j & 1);
      auto quad_bcast = [](unsigned v, int jj) {
        switch (jj) {
          case 0: return (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0x00, 0xf, 0xf, false);
          case 1: return (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0x55, 0xf, 0xf, false);
          case 2: return (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0xAA, 0xf, 0xf, false);
          default: return (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0xFF, 0xf, 0xf, false);
        }
      };
#if AOMHIP_SB_DBG_KNOBS
      const unsigned grid_magic = (unsigned)((0x100000000ull + (unsigned)(a.sb_w / W) - 1) / (unsigned)(a.sb_w / W));
#endif
      auto eval = [&](const Win &w, int buf, int g0, int ng, int c0, int nc) {
        const uint32_t *gd = reinterpret_cast<const uint32_t *>(lds + gdesc_off_of(buf));
        const uint32_t *cd = reinterpret_cast<const uint32_t *>(lds + cdesc_off_of(buf));
        const char *sbuf = lds + src_off_of(buf);
        const int n_it = max(ng, nc);
        for (int i = slot; i < n_it; i += kPerWg) {
          const bool has_g = i < ng && !(dbg & 8), has_c = i < nc && !(dbg & 4);
          if constexpr (kCoop && G::kUnitsPerLane <= 4 && kMirror > 0) {
            // The common case as ONE basic block: a group and a single candidate of the same source block, everything
            // inside the staged cell / window, source rows unit aligned (a Mode-A style list, a diamond step ...).
            // With no branch between the five references the compiler overlaps their LDS reads, realignments and
            // SADs; the general code below is a chain of small dependent blocks, and with two evaluating wavefronts
            // per SIMD that chain's latency -- not LDS or VALU throughput -- set the step time.
            const bool both = has_g && has_c;
            const int ii = both ? i : 0;  // (lanes without a pair read entry 0: no out-of-range LDS address, no per-word select)
#if AOMHIP_SB_DBG_KNOBS
            uint32_t d0 = gd[ii * 5], c0w = cd[ii * 2], c1w = cd[ii * 2 + 1];
            int mrx = __builtin_amdgcn_sbfe((int)gd[ii * 5 + 1 + my_w], my_sh, 16);
            int mry = __builtin_amdgcn_sbfe((int)gd[ii * 5 + 3 + my_w], my_sh, 16);
            if (dbg & 8192) {  // (timing ablation = upper bound of a grid-mode entry point: no list travels, the entry is a function of its index)
              const unsigned bpr = (unsigned)a.sb_w / W, row = __umulhi((unsigned)ii, grid_magic), col = (unsigned)ii - row * bpr;
              const int gx = cell_x0 + (int)col * W, gy = (int)w.sy0 + (int)row * H;
              d0 = c0w = c1w = ((uint32_t)gx & 0xffffu) | ((uint32_t)gy << 16);
              mrx = gx + (int)((ii * 29u + my_j * 13u) & 31u) - 16;
              mry = gy + (int)((ii * 11u + my_j * 7u) & 31u) - 16;
            }
#else
            const uint32_t d0 = gd[ii * 5], c0w = cd[ii * 2], c1w = cd[ii * 2 + 1];
            const int mrx = __builtin_amdgcn_sbfe((int)gd[ii * 5 + 1 + my_w], my_sh, 16);
            const int mry = __builtin_amdgcn_sbfe((int)gd[ii * 5 + 3 + my_w], my_sh, 16);
#endif
            const int fsx = (int16_t)d0, fsy = (int16_t)(d0 >> 16);
            unsigned soff, mbase, cbase, base[5];
            int unused;
            bool ok = src_pos(w, fsx, fsy, soff);
            ok = ring_pos(w, mrx, mry, mbase, unused) && ok;
            ok = ring_pos(w, (int16_t)c1w, (int16_t)(c1w >> 16), cbase, unused) && ok;
            ok = ok && c0w == d0 && (soff & (G::kUnitBytes - 1)) == 0;
            unsigned okv = ok ? 1u : 0u;
            okv &= (unsigned)__builtin_amdgcn_update_dpp(0u, okv, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
            okv &= (unsigned)__builtin_amdgcn_update_dpp(0u, okv, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
            if (__all((both && okv != 0) || (!has_g && !has_c))) {
              if (dbg & 32) continue;  // (timing ablation: decode and tests only)
              // (`both` is uniform over a block's lanes: the select goes BEFORE the quad broadcasts, which then run unconditionally --
              // selecting after them compiled to four exec-masked DPP moves, each inside its own branch)
              mbase = both ? mbase : (unsigned)a.ring_off;
#pragma unroll
              for (int j = 0; j < 4; ++j) base[j] = quad_bcast(mbase, j);
              base[4] = both ? cbase : (unsigned)a.ring_off;
              if (!both) soff = 0;
              AcT acc[5];
              [[maybe_unused]] uint32_t src_sum = 0, src_ss = 0;   // (variance: the source block's own sums, once for its five references)
              // The single candidate of a Mode-A style pair is the zero-MV one: 16-byte aligned in the ring whenever the cell grid is.  When
              // that holds for the whole wavefront its rows are read like the source's, with one aligned 16-byte load and no
              // realignment (and without the 4-way bank conflict of eight aligned rows x four aligned blocks read dword by dword).
              const bool cand_aligned = __all((base[4] & 15u) == 0);
              auto body = [&](auto al) {
                constexpr bool kAl = decltype(al)::value && G::kUnitBytes == 16;
                constexpr int kNr = kAl ? 4 : 5;  // references read through the realigning path
                {
                  // All LDS reads of a row unit's references are issued before the first realignment (left to itself the compiler
                  // reads one reference, waits, computes, reads the next: ten LDS round trips in the one iteration a wavefront runs
                  // per step).  8-bit: 1080p -4.6 %, 4K -2 %; 16-bit planes: -4 % (it was +5 % while the kernel sat at the 128-VGPR cap).
                  constexpr int kDw = G::kUnitBytes / 4;
                  // (row pitches and unit offsets are multiples of 16 bytes: a reference's dword base and byte shift are the same
                  // for all of its rows)
                  unsigned sh[5], bdw[5];
#pragma unroll
                  for (int j = 0; j < 5; ++j) { sh[j] = base[j] & 3u; bdw[j] = base[j] & ~3u; }
#pragma unroll
                  for (int k = 0; k < G::kUnitsPerLane; ++k) {
                    const L sv = lds_unit_aligned<G::kUnitBytes>(sbuf, soff + unit_soff(k));
                    uint32_t raw[5][kDw + 1];
                    L c4 = sv;
#pragma unroll
                    for (int j = 0; j < kNr; ++j) {
                      const uint32_t *p = reinterpret_cast<const uint32_t *>(lds + (bdw[j] + unit_roff(k)));
#if AOMHIP_SB_DBG_KNOBS
                      if (dbg & 1024) {  // (timing ablation: the arithmetic without the references' LDS reads)
#pragma unroll
                        for (int i = 0; i <= kDw; ++i) raw[j][i] = bdw[j] + i + k;
                        continue;
                      }
#endif
#pragma unroll
                      for (int i = 0; i <= kDw; ++i) raw[j][i] = p[i];
                    }
                    if constexpr (kAl) c4 = lds_unit_aligned<G::kUnitBytes>(lds, base[4] + unit_roff(k));
                    __builtin_amdgcn_sched_barrier(0);
#if AOMHIP_SB_DBG_KNOBS
                    if (dbg & 512) {  // (timing ablation: the LDS reads without the realign + SAD arithmetic)
#pragma unroll
                      for (int j = 0; j < kNr; ++j)
#pragma unroll
                        for (int i = 0; i <= kDw; ++i) acc[j].a ^= raw[j][i];
                      acc[4].a ^= c4.v[0] ^ c4.v[3] ^ sv.v[1];
                      continue;
                    }
#endif
#pragma unroll
                    for (int j = 0; j < kNr; ++j)
#pragma unroll
                      for (int i = 0; i < kDw; ++i) add_ref(sv.v[i], __builtin_amdgcn_alignbyte(raw[j][i + 1], raw[j][i], sh[j]), acc[j]);
                    if constexpr (kAl) {
#pragma unroll
                      for (int i = 0; i < kDw; ++i) add_ref(sv.v[i], c4.v[i], acc[4]);
                    }
                    if constexpr (VAR) {
#pragma unroll
                      for (int i = 0; i < kDw; ++i) add_src(sv.v[i], src_sum, src_ss);
                    }
                  }
                }
              };
              if (cand_aligned) body(std::true_type{}); else body(std::false_type{});
              if (dbg & 128) {  // (timing ablation: no reduction, no stores)
                if ((acc[0].a & acc[1].a & acc[2].a & acc[3].a & acc[4].a) == 0xFFFFFFFFu) out1[0] = 0;
                continue;
              }
              ResT res[5];
#pragma unroll
              for (int j = 0; j < 5; ++j) {
                if constexpr (VAR) { acc[j].ssum = src_sum; acc[j].ss = src_ss; }
                res[j] = reduce(acc[j]);
              }
              if (lane_in_cand == 0 && both) {
                store_group((int64_t)f_rel * n_groups + g0 + i, res[0], res[1], res[2], res[3]);
                store_cand((int64_t)f_rel * n_cands + c0 + i, res[4]);
              }
              continue;
            }
          }
          int gsx = 0, gsy = 0;
          constexpr bool kKeepSrc = G::kUnitsPerLane <= 4;  // larger blocks do not keep their source rows in registers
          L gsrc[kKeepSrc ? G::kUnitsPerLane : 1];
          bool gsrc_ok = false;  // gsrc holds the rows of block (gsx, gsy)
          if (has_g) {
            const uint32_t d0 = gd[i * 5];
            gsx = (int16_t)d0; gsy = (int16_t)(d0 >> 16);
            unsigned soff, base[4];
            int slot0[4];
            bool in = src_pos(w, gsx, gsy, soff);
            int rx[4], ry[4];  // all four only on the non-cooperative / fallback paths
            if constexpr (kCoop) {
              const int mrx = __builtin_amdgcn_sbfe((int)gd[i * 5 + 1 + my_w], my_sh, 16);
              const int mry = __builtin_amdgcn_sbfe((int)gd[i * 5 + 3 + my_w], my_sh, 16);
              unsigned mbase;
              int mslot;
              unsigned ok = ring_pos(w, mrx, mry, mbase, mslot) ? 1u : 0u;
              ok &= (unsigned)__builtin_amdgcn_update_dpp(0u, ok, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
              ok &= (unsigned)__builtin_amdgcn_update_dpp(0u, ok, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
              in = in && ok != 0;
  #pragma unroll
              for (int j = 0; j < 4; ++j) {
                base[j] = quad_bcast(mbase, j);
                slot0[j] = kMirror > 0 ? 0 : (int)quad_bcast((unsigned)mslot, j);
                rx[j] = mrx; ry[j] = mry;  // (own reference only; the fallback below re-broadcasts)
              }
            } else {
              uint32_t d[5];
              d[0] = d0;
  #pragma unroll
              for (int q = 1; q < 5; ++q) d[q] = gd[i * 5 + q];
  #pragma unroll
              for (int j = 0; j < 4; ++j) {  // halfword h of the record: word h / 2, shifted by 16 (h & 1); rx = halves 2..5, ry = 6..9
                rx[j] = (int16_t)(d[(2 + j) >> 1] >> (16 * ((2 + j) & 1)));
                ry[j] = (int16_t)(d[(6 + j) >> 1] >> (16 * ((6 + j) & 1)));
                in = ring_pos(w, rx[j], ry[j], base[j], slot0[j]) && in;
              }
            }
            AcT acc[4];
            if (in && !(dbg & 16)) {
              const bool s_al = (soff & (G::kUnitBytes - 1)) == 0;
              if (__all(s_al)) {
  #pragma unroll
                for (int k = 0; k < G::kUnitsPerLane; ++k) {
                  const L sv = lds_unit_aligned<G::kUnitBytes>(sbuf, soff + unit_soff(k));
                  if constexpr (kKeepSrc) gsrc[k] = sv;
  #pragma unroll
                  for (int j = 0; j < 4; ++j) acc[j] = sad_unit(sv, ring_unit(base[j], slot0[j], k), acc[j]);
                }
              } else {
  #pragma unroll kGenUnroll
                for (int k = 0; k < G::kUnitsPerLane; ++k) {
                  const L sv = lds_unit<G::kUnitBytes>(sbuf, soff + unit_soff(k));
                  if constexpr (kKeepSrc) gsrc[k] = sv;
  #pragma unroll
                  for (int j = 0; j < 4; ++j) acc[j] = sad_unit(sv, ring_unit(base[j], slot0[j], k), acc[j]);
                }
              }
              gsrc_ok = kKeepSrc;
            } else {
  #pragma unroll 1
              for (int j = 0; j < 4; ++j) {
                int jrx = rx[j], jry = ry[j];
                if constexpr (kCoop) {  // every lane of the block's lane group is in this branch together
                  jrx = (int)quad_bcast((unsigned)rx[0], j);
                  jry = (int)quad_bcast((unsigned)ry[0], j);
                }
                const AcT v = generic_ref(gsx, gsy, jrx, jry);
#pragma unroll
                for (int q = 0; q < 4; ++q) {   // (j is a loop counter here, the accumulators are registers: select, do not index)
                  acc[q].a = j == q ? v.a : acc[q].a;
                  if constexpr (VAR) {
                    acc[q].rr = j == q ? v.rr : acc[q].rr; acc[q].xr = j == q ? v.xr : acc[q].xr;
                    acc[q].ssum = j == q ? v.ssum : acc[q].ssum; acc[q].ss = j == q ? v.ss : acc[q].ss;
                  }
                }
              }
            }
            const ResT r0 = reduce(acc[0]), r1 = reduce(acc[1]), r2 = reduce(acc[2]), r3 = reduce(acc[3]);
            if (lane_in_cand == 0) store_group((int64_t)f_rel * n_groups + g0 + i, r0, r1, r2, r3);
          }
          if (has_c) {
            const uint32_t d0 = cd[i * 2], d1 = cd[i * 2 + 1];
            const int sx = (int16_t)d0, sy = (int16_t)(d0 >> 16), rx = (int16_t)d1, ry = (int16_t)(d1 >> 16);
            AcT acc;
            unsigned soff, base;
            int slot0;
            bool in = src_pos(w, sx, sy, soff);
            in = ring_pos(w, rx, ry, base, slot0) && in;
            if (in) {
              const bool reuse = gsrc_ok && sx == gsx && sy == gsy;
              if (__all(reuse)) {
                if constexpr (kKeepSrc) {
  #pragma unroll
                  for (int k = 0; k < G::kUnitsPerLane; ++k) acc = sad_unit(gsrc[k], ring_unit(base, slot0, k), acc);
                }
              } else {
  #pragma unroll kGenUnroll
                for (int k = 0; k < G::kUnitsPerLane; ++k)
                  acc = sad_unit(lds_unit<G::kUnitBytes>(sbuf, soff + unit_soff(k)), ring_unit(base, slot0, k), acc);
              }
            } else {
              acc = generic_ref(sx, sy, rx, ry);
            }
            const ResT rc = reduce(acc);
            if (lane_in_cand == 0) store_cand((int64_t)f_rel * n_cands + c0 + i, rc);
          }
        }
      };

      auto batch_of = [&](int cy) {  // what step cy adds on top of step cy - 1 (nothing past the last cell)
        const bool real = cy < a.cell_rows;
        const int cyc = real ? cy : a.cell_rows - 1;  // (addresses of an empty batch stay inside the planes)
        const Seg sg = seg_of(cyc);
        const int4 bv = bats[cyc];  // {first new row, new rows, source cell rows, ring byte offset of the first new row}
        Batch b;
        b.ya = uni(bv.x);
        b.yb = real ? b.ya + uni(bv.y) : b.ya;
        b.sy0 = cyc * a.sb_h; b.ns = real ? uni(bv.z) : 0;
        b.first = (unsigned)uni(bv.w);
        b.g0 = sg.g0; b.ng = real ? min(sg.g1 - sg.g0, a.gcap) : 0;
        b.c0 = sg.c0; b.nc = real ? min(sg.c1 - sg.c0, a.ccap) : 0;
        b.buf = cy & 1;
        if ((dbg & 2) && cy > 0) { b.yb = b.ya; b.ns = 0; }  // (timing ablation: list slices only)
        if (dbg & 8192) { b.ng = 0; b.nc = 0; }              // (timing ablation: no list slices, profiles/r04_sad_strip.md)
        return b;
      };
      // Per-step records, worked out once per strip (one lane per step) instead of by every wavefront in every step: the scalar
      // arithmetic of a step -- window rows, ring slot (a modulo), bucket bounds -- was ~330 clock ticks on every evaluating wavefront's
      // chain and more on the loaders', much of it reloads of spilled scalars.
      int crowded = 0;  // does any cell of this strip hold more entries than a slice buffer (the loaders' overflow() path)?
      for (int cy = tid; cy < a.cell_rows; cy += kAll) {
        const int b = cy * a.cells_per_row + cx;
        const int4 sg = make_int4(groups ? group_off[b] : 0, groups ? group_off[b + 1] : 0, cands ? cand_off[b] : 0, cands ? cand_off[b + 1] : 0);
        reinterpret_cast<int4 *>(lds + a.seg_off)[cy] = sg;
        crowded |= (sg.y - sg.x > a.gcap) | (sg.w - sg.z > a.ccap);
        const int wy0 = win_y0(cy), wy1 = win_y1(cy);
        const int sh = min(cy * a.sb_h + a.sb_h, a.s_ymax) - cy * a.sb_h;
        const bool ok = strip_ok && wy1 - wy0 >= H && sh >= H;
        const unsigned wh_ok = strip_ok && wy1 - wy0 >= H ? (unsigned)(wy1 - wy0 - H) : 0u;
        wins[cy] = make_int4(wy0, (wy0 - a.ymin) % a.R, (int)wh_ok, (int)(((unsigned)(sh - H) & 0x7fffffffu) | (ok ? 0x80000000u : 0u)));
        const int ya = cy > 0 ? win_y1(cy - 1) : win_y0(0);
        bats[cy] = make_int4(ya, wy1 - ya, sh, ((ya - a.ymin) % a.R) * a.pitch);
      }
      // (an OR over the workgroup through 16 words of the flag area: __syncthreads_or() brings its own static LDS, and the widest cells
      // leave none)
      if (lane == 0) misc[16 + wave] = __ballot(crowded != 0) != 0 ? 1 : 0;
      __syncthreads();
      bool any_crowded;  // (uniform: the loaders skip overflow()'s per-step look-up otherwise)
      {
        const int4 *f = reinterpret_cast<const int4 *>(misc + 16);
        const int4 f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
        any_crowded = uni((f0.x | f0.y | f0.z | f0.w | f1.x | f1.y | f1.z | f1.w | f2.x | f2.y | f2.z | f2.w | f3.x | f3.y | f3.z | f3.w)) != 0;
      }
      // ---- prologue of the strip: the whole first window in passes of what the staging registers hold, the first source
      // cell and list slices (every wavefront of the workgroup takes part: nothing to evaluate yet)
      SB_T(p0);
      {
        Stage st0;
        const int rows_per_pass = (kRingN * kAll) / a.cpr;
        const Batch b = batch_of(0);
        for (int ya = win_y0(0); ya < win_y1(0); ya += rows_per_pass) {
          Batch p = b;
          p.ya = ya; p.yb = min(ya + rows_per_pass, win_y1(0));
          if (ya != win_y0(0)) { p.ns = 0; p.ng = 0; p.nc = 0; }
          int me = tid;
          asm volatile("" : "+v"(me));  // (as in overflow() below: nothing of this loop is worth keeping across the strip)
          request_gen(p, st0, me, kAll);
          commit_gen(p, st0, me, kAll);
        }
      }
      // The two roles run their own step loops with the same sequence of workgroup barriers.  The loader's loop body is
      // straight-line around the registers that are in flight across the barrier (commit, then request): any join
      // with a live staged register makes the compiler copy it -- and wait for it -- on the spot.
      // Steady-state transport of the loader lanes: a batch always has the same shape (sb_h rows below the previous one),
      // so each lane's chunk offsets are computed once per strip and a staged load costs a compare, a select and the load
      // (scalar base + 32-bit lane offset).  The general forms above cost ~25 instructions per chunk, and with ~14 chunks
      // per lane that made the loaders, not the memory system or the evaluation, the longest thing in a step.
      unsigned r_goff[kRingN], r_loff[kRingN], s_goff[kSrcN], s_loff[kSrcN];
      int lt_s = lt;  // (opaque: the tables below must not be computed -- and kept alive -- across the prologue above)
      asm volatile("" : "+v"(lt_s));
#pragma unroll
      for (int i = 0; i < kRingN; ++i) {
        const unsigned q = (unsigned)(lt_s + i * kLT), row = __umulhi(q, a.magic_cpr), col = q - row * (unsigned)a.cpr;
        r_goff[i] = row * (unsigned)gpitch + (unsigned)min((int)col, colmax) * 16u;
        r_loff[i] = row * (unsigned)a.pitch + col * 16u;
      }
  #pragma unroll
      for (int i = 0; i < kSrcN; ++i) {
        const unsigned q = (unsigned)(lt_s + i * kLT), row = __umulhi(q, a.magic_scpr), col = q - row * (unsigned)a.scpr;
        s_goff[i] = row * (unsigned)sgpitch + (unsigned)min((int)col, scolmax) * 16u;
        s_loff[i] = row * (unsigned)a.spitch + col * 16u;
      }
      const unsigned ring_bytes = (unsigned)(a.R * a.pitch);
      auto request = [&](const Batch &b, Stage &st) {
        // (an empty batch still issues its loads -- at row ymax - 1 / the last source row, inside the planes)
        const char *rb = ref_frame + (int64_t)min(b.ya, a.ymax - 1) * gpitch + (int64_t)wx0 * kES;
        const char *sb_ = src_frame + (int64_t)min(b.sy0, a.s_ymax - 1) * sgpitch + (int64_t)sx0 * kES;
        const unsigned rlim = (unsigned)((b.yb - b.ya) * gpitch), slim = (unsigned)(b.ns * sgpitch);
  #pragma unroll
        for (int i = 0; i < kRingN; ++i) st.ring[i] = *reinterpret_cast<const V4 *>(rb + (r_goff[i] < rlim ? r_goff[i] : 0u));
  #pragma unroll
        for (int i = 0; i < kSrcN; ++i) st.srcv[i] = *reinterpret_cast<const V4 *>(sb_ + (s_goff[i] < slim ? s_goff[i] : 0u));
  #pragma unroll
        for (int i = 0; i < kGN; ++i) st.g[i] = gwords[(unsigned)max(min(b.g0 * 5 + min(lt + i * kLT, b.ng * 5 - 1), gwords_n - 1), 0)];
  #pragma unroll
        for (int i = 0; i < kCN; ++i) st.c[i] = cwords[(unsigned)max(min(b.c0 * 2 + min(lt + i * kLT, b.nc * 2 - 1), cwords_n - 1), 0)];
      };
      auto commit = [&](const Batch &b, const Stage &st) {
        const unsigned rlim = (unsigned)((b.yb - b.ya) * gpitch), slim = (unsigned)(b.ns * sgpitch);
        const unsigned first = b.first;
        // Only a batch that starts inside the mirrored slots or wraps around the end of the ring has rows to store twice (a uniform
        // test per step; with R a multiple of the step height it is one step in R / sb_h): the others skip the per-chunk mirror test.
        const bool mirrored = kMirror > 0 && (first < (unsigned)(kMirror * a.pitch) || first + (unsigned)(b.yb - b.ya) * (unsigned)a.pitch > ring_bytes);
        if (mirrored) {
  #pragma unroll
          for (int i = 0; i < kRingN; ++i)
            if (r_goff[i] < rlim) {
              unsigned t = first + r_loff[i];
              t = min(t, t - ring_bytes);  // wrap: t - ring_bytes underflows to a huge value unless t >= ring_bytes
              const V4 v = st.ring[i];
              *reinterpret_cast<V4 *>(lds + a.ring_off + t) = v;
              if (t < (unsigned)(kMirror * a.pitch)) *reinterpret_cast<V4 *>(lds + a.ring_off + t + ring_bytes) = v;
            }
        } else {
  #pragma unroll
          for (int i = 0; i < kRingN; ++i)
            if (r_goff[i] < rlim) *reinterpret_cast<V4 *>(lds + a.ring_off + first + r_loff[i]) = st.ring[i];
        }
  #pragma unroll
        for (int i = 0; i < kSrcN; ++i)
          if (s_goff[i] < slim)
            *reinterpret_cast<V4 *>(lds + src_off_of(b.buf) + s_loff[i]) = st.srcv[i];
  #pragma unroll
        for (int i = 0; i < kGN; ++i)
          if (lt + i * kLT < b.ng * 5) *reinterpret_cast<uint32_t *>(lds + gdesc_off_of(b.buf) + (lt + i * kLT) * 4) = st.g[i];
  #pragma unroll
        for (int i = 0; i < kCN; ++i)
          if (lt + i * kLT < b.nc * 2) *reinterpret_cast<uint32_t *>(lds + cdesc_off_of(b.buf) + (lt + i * kLT) * 4) = st.c[i];
      };
      const int steps = (dbg & 256) ? 0 : (a.cell_rows + 1) & ~1;  // both roles run an even number of steps (the odd one out only meets the barriers); (dbg 256: timing ablation, prologue only)
      if constexpr (kLoader) {
        // Two loader groups take alternate steps: during step cy the group of that parity writes batch cy + 1 (which it requested
        // during step cy - 1) to LDS while the other group requests batch cy + 2, so a batch -- ~30 KB per CU -- has a whole step to
        // arrive and each wavefront only ever waits for its own loads (with two batches in one wavefront the compiler's vmcnt
        // bookkeeping drained the younger batch too).  Each group's loop is straight-line around the staged registers.
        Stage st;
        auto overflow = [&](int cy, bool mine) {  // a crowded bucket: further slices through the same buffers
          if (cy >= a.cell_rows) return;
          const Seg cur = seg_of(cy);
          int g = cur.g0 + min(cur.g1 - cur.g0, a.gcap), c = cur.c0 + min(cur.c1 - cur.c0, a.ccap);
          while (g < cur.g1 || c < cur.c1) {
            Batch o;
            o.ya = o.yb = win_y0(0); o.sy0 = 0; o.ns = 0; o.buf = cy & 1;
            o.g0 = g; o.ng = min(cur.g1 - g, a.gcap); o.c0 = c; o.nc = min(cur.c1 - c, a.ccap);
            __syncthreads();  // the evaluating wavefronts are done with the previous slice
            if (mine) {
              // (the lane index goes through an opaque statement: otherwise every per-lane quantity of this rarely taken path is
              // hoisted out of the step loop and stays in registers -- or in scratch -- for the whole strip)
              int me = lt;
              asm volatile("" : "+v"(me));
              Stage so;
              request_gen(o, so, me, kLT);
              commit_gen(o, so, me, kLT);
            }
            __syncthreads();
            g += o.ng; c += o.nc;
          }
        };
        // A group's two steps: in its ACTIVE step it writes the batch it holds (batch cy + 1) to LDS; in its PASSIVE step -- while the other
        // group writes -- it works out and requests the batch it will write next (cy + 2 seen from the passive step).  Doing both in the
        // active step (the first form) made one wavefront's commit -> bookkeeping -> request sequence, ~4400 clock ticks, the longest
        // thing in a step, with the other group idle and the evaluating wavefronts (~3700) waiting at the barrier for it.
        Batch held = batch_of(0);
        auto active = [&](int cy) {
          if (dbg & 64) { __syncthreads(); return; }  // (timing ablation: barriers only)
          SB_T(l0);
          const Batch bc = held;  // the batch this group requested in its passive step (its description rides along in scalars)
          SB_T(l1);
#ifdef AOMHIP_SB_PROF
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          SB_T(l1w);
          SB_ACC(7, l1w, l1);
#endif
          commit(bc, st);  // requested in the previous step
          SB_T(l2);
          if (any_crowded) overflow(cy, true);
          SB_T(l3);
          __syncthreads();  // step cy + 1 is in LDS; nobody reads step cy's rows / cell / slices any more
          SB_T(l5);
          SB_ACC(0, l1, l0); SB_ACC(1, l2, l1); SB_ACC(2, l3, l2); SB_ACC(4, l5, l3); SB_ACC(5, 1, 0); SB_WAIT(l5, l3);
        };
        auto passive = [&](int cy) {
          if (dbg & 64) { __syncthreads(); return; }
          SB_T(q0);
          held = batch_of(cy + 2);
          request(held, st);  // in flight across this step's barrier
          SB_T(q1);
          if (any_crowded) overflow(cy, false);
          __syncthreads();
          SB_T(q2);
          SB_ACC(3, q1, q0); SB_ACC(6, q2, q1); SB_WAIT(q2, q1);
        };
        if (grp == 0) {
          held = batch_of(1);
          request(held, st);
          __syncthreads();
          for (int cy = 0; cy < steps; cy += 2) {
            active(cy);
            passive(cy + 1);
          }
        } else {
          __syncthreads();
          for (int cy = 0; cy < steps; cy += 2) {
            passive(cy);
            active(cy + 1);
          }
        }
      } else {
        __syncthreads();
        SB_T(p1);
        SB_ACC(5, p1, p0); SB_ACC(6, 1, 0);
        for (int cy = 0; cy < steps; ++cy) {
          if (cy >= a.cell_rows) {  // (the padding step of an odd walk)
            __syncthreads();
            break;
          }
          if (dbg & (64 | 2048)) { __syncthreads(); continue; }  // (2048: timing ablation, the evaluators skip their bookkeeping too)
          const int buf = cy & 1;
          SB_T(t1);
          const Seg cur = seg_of(cy);
          const int4 wv = wins[cy];
          Win w;
          w.wy0 = uni(wv.x);
          w.s0 = uni(wv.y);
          w.wh_ok = (unsigned)uni(wv.z);
          w.sy0 = (unsigned)(cy * a.sb_h);
          w.sh_ok = (unsigned)uni(wv.w) & 0x7fffffffu;
          const bool step_ok = (unsigned)uni(wv.w) >> 31;
          int g = cur.g0, c = cur.c0;
          bool first = true;
          SB_T(t1b);
          SB_ACC(0, t1b, t1);
          do {
            const int ng = min(cur.g1 - g, a.gcap), nc = min(cur.c1 - c, a.ccap);
            if (!first) {
              __syncthreads();
              __syncthreads();  // the loaders have put the next slices in place
            }
            if (!(dbg & 1)) {
              if (step_ok) {
                eval(w, buf, g, ng, c, nc);
              } else {  // a window / cell lower or narrower than the block: nothing can be served from LDS
                for (int gi = slot; gi < ng; gi += kPerWg) {
                  const uint32_t *e = reinterpret_cast<const uint32_t *>(glist + g + gi);
                  const uint32_t e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3], e4 = e[4];
                  const int sx = (int16_t)e0, sy = (int16_t)(e0 >> 16);
                  const ResT r0 = reduce(generic_ref(sx, sy, (int16_t)e1, (int16_t)e3));
                  const ResT r1 = reduce(generic_ref(sx, sy, (int16_t)(e1 >> 16), (int16_t)(e3 >> 16)));
                  const ResT r2 = reduce(generic_ref(sx, sy, (int16_t)e2, (int16_t)e4));
                  const ResT r3 = reduce(generic_ref(sx, sy, (int16_t)(e2 >> 16), (int16_t)(e4 >> 16)));
                  if (lane_in_cand == 0) store_group((int64_t)f_rel * n_groups + g + gi, r0, r1, r2, r3);
                }
                for (int ci = slot; ci < nc; ci += kPerWg) {
                  const uint32_t *e = reinterpret_cast<const uint32_t *>(clist + c + ci);
                  const uint32_t e0 = e[0], e1 = e[1];
                  const ResT v = reduce(generic_ref((int16_t)e0, (int16_t)(e0 >> 16), (int16_t)e1, (int16_t)(e1 >> 16)));
                  if (lane_in_cand == 0) store_cand((int64_t)f_rel * n_cands + c + ci, v);
                }
              }
            }
            g += ng; c += nc;
            first = false;
          } while (g < cur.g1 || c < cur.c1);
          SB_T(t2);
          __syncthreads();
          SB_T(t4);
          SB_ACC(1, t2, t1); SB_ACC(3, t4, t2); SB_ACC(4, 1, 0); SB_WAIT(t4, t2);
        }
      }
    }
    SB_FLUSH;
  };
  // The younger half of the evaluating wavefronts shares its SIMDs with the older half and loses the issue arbitration by age: it
  // finished a step ~900 clock ticks later and set the step's length.  One priority level evens that out (4K 10-bit -4 %, 1080p -1.5 %).
  if (wave >= kWaves / 2 && wave < kWaves) __builtin_amdgcn_s_setprio(1);
  if (is_loader) run(std::true_type{}); else run(std::false_type{});
}

struct SbLaunch {
  hipStream_t stream;
  int grid, threads, upl;
  bool deep;  // CfgDeep
  size_t lds_bytes;
  StripArgs a;
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
  uint32_t *out4b, *out1b;  // variance form: the sse arrays (out4 / out1 hold the variances)
  bool var;
};

template <typename T, int W, int H, bool SKIP, typename C, bool VAR = false>
static int launch_nt(const SbLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r) {
  auto k = sad_strip_kernel<T, W, H, SKIP, C, VAR>;
  static thread_local size_t granted = 0;  // per instantiation
  if (l.lds_bytes > granted) {
    AOMHIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)l.lds_bytes));
    granted = l.lds_bytes;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)l.grid), dim3(C::kAll), l.lds_bytes, l.stream, s, r, l.a,
                     l.groups, l.group_off, l.n_groups, l.gfs, l.out4, l.cands, l.cand_off, l.n_cands, l.cfs, l.out1, l.out4b, l.out1b);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

// The deep configuration is an experiment (AOMHIP_SB_CFG=deep, 16x16 on 8-bit planes only): its evaluation is ~25 % shorter per step,
// but four loader wavefronts cannot carry a step's transport (15 chunks per lane: the commit sequence alone outlasts the evaluation).
constexpr bool deep_size(int es, int w, int h) { return es == 1 && w == 16 && h == 16; }
inline bool use_deep(int es, int w, int h, bool skip) {
  if (!deep_size(es, w, h) || skip) return false;
  const char *e = getenv("AOMHIP_SB_CFG");
  return e && e[0] == 'd';
}
template <typename T, int W, int H, bool SKIP>
static int launch(const SbLaunch &l, const PlaneView<T> &s, const PlaneView<T> &r) {
  if constexpr (deep_size((int)sizeof(T), W, H) && !SKIP) {
    if (l.deep) return launch_nt<T, W, H, SKIP, CfgDeep>(l, s, r);
  }
  return launch_nt<T, W, H, SKIP, CfgWide>(l, s, r);
}

#ifdef AOMHIP_SB_ONLY_16  // (codegen experiments: one block size compiles in seconds)
#define AOMHIP_FOR_BLOCK_SIZES(X) X(16, 16)
#else
#define AOMHIP_FOR_BLOCK_SIZES(X)                                                                                \
  X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(16, 32) X(32, 16) X(32, 32) X(32, 64) X(64, 32) \
  X(64, 64) X(64, 128) X(128, 64) X(128, 128) X(4, 16) X(16, 4) X(8, 32) X(32, 8) X(16, 64) X(64, 16)
#endif

#ifdef AOMHIP_SB_ONLY_16
#define AOMHIP_FOR_VAR_BLOCK_SIZES(X) X(16, 16)
#else
#define AOMHIP_FOR_VAR_BLOCK_SIZES(X) X(4, 4) X(4, 8) X(8, 4) X(8, 8) X(8, 16) X(16, 8) X(16, 16) X(4, 16) X(16, 4) X(8, 32) X(32, 8)
#endif

template <typename T>
static int dispatch(const SbLaunch &l, bool skip, const PlaneView<T> &s, const PlaneView<T> &r, int bw, int bh) {
  if (l.var) {   // the variance form: whole blocks of at most 256 pixels
#define X(W, H) \
  if (bw == W && bh == H) return launch_nt<T, W, H, false, CfgWide, true>(l, s, r);
    AOMHIP_FOR_VAR_BLOCK_SIZES(X)
#undef X
    set_error("aomhip_variance_sb_batch: blocks of at most 256 pixels (%dx%d): use aomhip_variance_batch", bw, bh);
    return AOMHIP_ERR_INVALID;
  }
#define X(W, H) \
  if (bw == W && bh == H) return skip ? launch<T, W, H, (H >= 2)>(l, s, r) : launch<T, W, H, false>(l, s, r);
  AOMHIP_FOR_BLOCK_SIZES(X)
#undef X
  set_error("unsupported block size %dx%d", bw, bh);
  return AOMHIP_ERR_INVALID;
}

static unsigned magic_of(int d) { return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); }

}  // namespace sb
}  // namespace aomhip

using namespace aomhip;

#ifdef AOMHIP_SB_PROF
extern "C" int aomhip_debug_sb_prof(unsigned long long out[32], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sb::g_sb_prof), 256) != hipSuccess) return AOMHIP_ERR_HIP;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sb::g_sb_prof), z, 256) != hipSuccess) return AOMHIP_ERR_HIP; }
  return AOMHIP_OK;
}
#endif

static int sb_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame,
                    int n_frames, int bw, int bh, int flags, int sb_w, int sb_h, int range,
                    int n_buckets, const aomhip_sad_x4d_cand *d_groups,
                    const int32_t *d_group_bucket_offsets, int n_groups, int64_t group_frame_stride,
                    uint32_t *d_out_groups, const aomhip_sad_cand *d_cands,
                    const int32_t *d_cand_bucket_offsets, int n_cands, int64_t cand_frame_stride,
                    uint32_t *d_out_cands, bool var, uint32_t *d_out_groups_b, uint32_t *d_out_cands_b) {
  if (!ctx || !src || !ref || !src->base || !ref->base) {
    set_error("null argument");
    return AOMHIP_ERR_INVALID;
  }
  if ((d_groups && (!d_group_bucket_offsets || !d_out_groups || (var && !d_out_groups_b))) ||
      (d_cands && (!d_cand_bucket_offsets || !d_out_cands || (var && !d_out_cands_b))) || (!d_groups && !d_cands) || (var && flags != 0)) {
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
  if (cells_per_row > 256) {
    set_error("at most 256 cells per row (%d of width %d)", cells_per_row, sb_w);
    return AOMHIP_ERR_INVALID;
  }
  if (n_buckets == 0 || n_frames == 0) return AOMHIP_OK;
  const int es = ref->bit_depth > 8 ? 2 : 1, epc = 16 / es;
  sb::SbLaunch l;
  sb::StripArgs &a = l.a;
  memset(&a, 0, sizeof(a));
  // Rows travel as 16-byte chunks (one spare chunk when the window start is not chunk aligned by construction).  The LDS
  // row pitches are odd multiples of 16 bytes (ring) / not multiples of 64 (source cell), so that the rows a block's
  // lanes read at the same column fall into distinct LDS banks.
  const bool aligned = (sb_w % epc) == 0 && (range % epc) == 0 && (ref->border % epc) == 0;
  a.cpr = ((sb_w + 2 * range) * es + 15) / 16 + (aligned ? 0 : 1);
  a.pitch = (a.cpr | 1) * 16;
  a.R = 2 * sb_h + 2 * range;
  a.scpr = (sb_w * es + 15) / 16 + (((sb_w % epc) == 0 && (src->border % epc) == 0) ? 0 : 1);
  a.spitch = ((a.scpr & 3) == 0 ? a.scpr + 1 : a.scpr) * 16;
  l.deep = sb::use_deep(es, bw, bh, (flags & AOMHIP_SAD_SKIP_ROWS) != 0);
  const int ring_chunks = l.deep ? sb::CfgDeep::kRingN * sb::CfgDeep::kLT : sb::CfgWide::kRingN * sb::CfgWide::kLT;
  const int src_chunks = l.deep ? sb::CfgDeep::kSrcN * sb::CfgDeep::kLT : sb::CfgWide::kSrcN * sb::CfgWide::kLT;
  const int g_words = l.deep ? sb::CfgDeep::kGN * sb::CfgDeep::kLT : sb::CfgWide::kGN * sb::CfgWide::kLT;
  const int c_words = l.deep ? sb::CfgDeep::kCN * sb::CfgDeep::kLT : sb::CfgWide::kCN * sb::CfgWide::kLT;
  if (sb_h * a.cpr > ring_chunks || sb_h * a.scpr > src_chunks) {
    set_error("a step of %d rows x (%d + %d) bytes exceeds what the loader wavefronts keep in flight (%d + %d KB): use a lower cell",
              sb_h, a.cpr * 16, a.scpr * 16, ring_chunks / 64, src_chunks / 64);
    return AOMHIP_ERR_INVALID;
  }
  const size_t ring_bytes = (size_t)(a.R + sb::mirror_rows(bh)) * a.pitch, cell_bytes = (size_t)sb_h * a.spitch;
  const size_t kLds = 160 * 1024, misc_bytes = 72 * 4 + 256 + (size_t)cell_rows * 48 + 16;  // flags, active strips, 3 records per step
  if (ring_bytes + 2 * cell_bytes + misc_bytes + 4 * 256 > kLds) {
    set_error("LDS ring of %d rows x %d bytes + two %d x %d source cells (%zu bytes) exceed the 160 KB LDS of a CU: "
              "use a lower cell (sb_h) or a narrower one", a.R, a.pitch, sb_w, sb_h, ring_bytes + 2 * cell_bytes);
    return AOMHIP_ERR_INVALID;
  }
  {  // the rest of the LDS holds the list slices: two buffers, groups (20 B) and candidates (8 B) in equal numbers
    size_t per_buf = (kLds - ring_bytes - 2 * cell_bytes - misc_bytes) / 2;
    if (per_buf > 16 * 1024) per_buf = 16 * 1024;
    int cap = (int)((per_buf - 32) / 28);
    if (cap > g_words / 5) cap = g_words / 5;  // what the loader lanes hold per step
    if (cap > c_words / 2) cap = c_words / 2;
    if (const char *e = getenv("AOMHIP_SB_DESC_CAP")) cap = atoi(e) < cap && atoi(e) > 0 ? atoi(e) : cap;  // tests: force the crowded-bucket path
    a.gcap = a.ccap = cap;
    size_t off = 0;
    a.ring_off = 0; off += ring_bytes;
    for (int i = 0; i < 2; ++i) { a.src_off[i] = (int)off; off += cell_bytes; }
    for (int i = 0; i < 2; ++i) {
      a.gdesc_off[i] = (int)off; off += ((size_t)cap * 20 + 15) & ~(size_t)15;
      a.cdesc_off[i] = (int)off; off += ((size_t)cap * 8 + 15) & ~(size_t)15;
    }
    a.misc_off = (int)off; a.seg_off = (int)((off + 72 * 4 + 256 + 15) & ~(size_t)15); off += misc_bytes;
    l.lds_bytes = (off + 15) & ~(size_t)15;
  }
  a.magic_cpr = sb::magic_of(a.cpr); a.magic_scpr = sb::magic_of(a.scpr); a.magic_R = sb::magic_of(a.R);
  a.first_frame = first_frame; a.n_frames = n_frames;
  a.sb_w = sb_w; a.sb_h = sb_h; a.range = range; a.cells_per_row = cells_per_row; a.cell_rows = cell_rows;
  a.xmin = -ref->border; a.xmax = ref->width + ref->border; a.ymin = -ref->border; a.ymax = ref->height + ref->border;
  a.row_end = ref->stride - ref->border; a.border = ref->border;
  a.s_xmax = src->width + src->border; a.s_ymax = src->height + src->border; a.s_row_end = src->stride - src->border;
  a.s_border = src->border;
  a.shift = src->bit_depth == 10 ? 2 : src->bit_depth == 12 ? 4 : 0;
  if (const char *e = getenv("AOMHIP_SB_DBG")) a.dbg = atoi(e);
  l.stream = ctx->stream;
  {  // persistent grid: what the chip holds at once, a multiple of 8 so that blockIdx % 8 keeps naming one XCD
    static thread_local int cus = 0;
    if (!cus) {
      hipDeviceProp_t prop;
      AOMHIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
      cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int per_cu = (int)(kLds / l.lds_bytes) < 1 ? 1 : (int)(kLds / l.lds_bytes);
    l.threads = per_cu >= 2 ? 512 : 1024;
    int grid = cus * (per_cu > 2 ? 2 : per_cu);
    if (const char *e = getenv("AOMHIP_SB_THREADS")) l.threads = atoi(e);
    if (const char *e = getenv("AOMHIP_SB_GRID")) grid = atoi(e);
    grid &= ~7;
    if (grid < 8) grid = 8;
    const int items = n_frames * cells_per_row;
    if (n_frames < 8 && grid > items) grid = items;
    l.grid = grid;
  }
  l.groups = n_groups > 0 ? d_groups : nullptr; l.group_off = d_group_bucket_offsets; l.n_groups = n_groups; l.gfs = group_frame_stride;
  l.out4 = d_out_groups;
  l.cands = n_cands > 0 ? d_cands : nullptr; l.cand_off = d_cand_bucket_offsets; l.n_cands = n_cands; l.cfs = cand_frame_stride;
  l.out1 = d_out_cands;
  l.var = var; l.out4b = d_out_groups_b; l.out1b = d_out_cands_b;
  const bool skip = (flags & AOMHIP_SAD_SKIP_ROWS) != 0;
  if (src->bit_depth == 8) return sb::dispatch<uint8_t>(l, skip, view_of<uint8_t>(*src), view_of<uint8_t>(*ref), bw, bh);
  return sb::dispatch<uint16_t>(l, skip, view_of<uint16_t>(*src), view_of<uint16_t>(*ref), bw, bh);
}

extern "C" int aomhip_sad_sb_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame,
                                   int n_frames, int bw, int bh, int flags, int sb_w, int sb_h, int range,
                                   int n_buckets, const aomhip_sad_x4d_cand *d_groups,
                                   const int32_t *d_group_bucket_offsets, int n_groups, int64_t group_frame_stride,
                                   uint32_t *d_out_groups, const aomhip_sad_cand *d_cands,
                                   const int32_t *d_cand_bucket_offsets, int n_cands, int64_t cand_frame_stride,
                                   uint32_t *d_out_cands) {
  return sb_batch(ctx, src, ref, first_frame, n_frames, bw, bh, flags, sb_w, sb_h, range, n_buckets, d_groups, d_group_bucket_offsets, n_groups,
                  group_frame_stride, d_out_groups, d_cands, d_cand_bucket_offsets, n_cands, cand_frame_stride, d_out_cands, false, nullptr, nullptr);
}

extern "C" int aomhip_variance_sb_batch(aomhip_ctx *ctx, const aomhip_planes *src, const aomhip_planes *ref, int first_frame, int n_frames, int bw,
                                        int bh, int sb_w, int sb_h, int range, int n_buckets, const aomhip_sad_x4d_cand *d_groups,
                                        const int32_t *d_group_bucket_offsets, int n_groups, int64_t group_frame_stride, uint32_t *d_var_groups,
                                        uint32_t *d_sse_groups, const aomhip_sad_cand *d_cands, const int32_t *d_cand_bucket_offsets, int n_cands,
                                        int64_t cand_frame_stride, uint32_t *d_var_cands, uint32_t *d_sse_cands) {
  return sb_batch(ctx, src, ref, first_frame, n_frames, bw, bh, 0, sb_w, sb_h, range, n_buckets, d_groups, d_group_bucket_offsets, n_groups,
                  group_frame_stride, d_var_groups, d_cands, d_cand_bucket_offsets, n_cands, cand_frame_stride, d_var_cands, true, d_sse_groups,
                  d_sse_cands);
}
