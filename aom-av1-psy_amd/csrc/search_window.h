// A reference window shared by the blocks of one workgroup ("cell") of a motion-search kernel.  Not part of the ABI.
//
// The search kernels give one wavefront to one block for its whole greedy search (mcomp.c's searches are sequential per
// block); a search is ~30 dependent rounds of "evaluate 8 sites, pick one", and with the sites read through L1 / L2 every
// round is a global-memory round trip (~2 us at the occupancy the kernels reach, profiles/r03_search.md).  Here a workgroup
// owns a CELL of neighbouring blocks, stages once the part of the reference plane that their searches normally stay in
// -- the bounding box of their start positions grown by R pixels on every side -- and every round whose sites lie inside
// it reads LDS.  A round that leaves the window reads the plane exactly as before: the window is a speed contract,
// never a validity one.
//
// Cells are consecutive list entries; the window is computed from the block RECORDS the cell really holds, so a list in any order
// costs coverage, never correctness.
#ifndef AOMHIP_CSRC_SEARCH_WINDOW_H_
#define AOMHIP_CSRC_SEARCH_WINDOW_H_

#include "common.h"
#include "search_device.h"

namespace aomhip {

#ifdef AOMHIP_CELL_PROF
extern __device__ unsigned int g_cell_prof[40960 * 16];
#endif

struct CellMap {
  int n_cells;
  int win_r;       // pixels of reach wanted around the cell's start positions; < 0: no window
  int lds_bytes;   // window budget (the launch's dynamic LDS)
  int xmin, ymin, xmax, ymax;  // pixels of the bordered plane that may be read: [xmin, xmax) x [ymin, ymax)
};

// Blocks (= wavefronts) per cell and the reach of the window, measured on the 4K 10-bit 16x16 DIAMOND search (profiles/r04_search_cell.md):
// the fine rounds (radius <= 8: 22 of the 28 rounds of a step_param-4 search) are the ones worth serving from LDS, and the smaller the
// workgroup the better the CU stays filled when blocks of one cell take different numbers of rounds: 2 blocks x reach 16 = 0.141 ms per
// frame, 4 x 16 = 0.149, 8 x 16 = 0.163, 16 x 64 (the whole first step inside the window, 79 KB, one or two workgroups per CU) = 0.18 - 0.20,
// no window = 0.231.  Cells are consecutive list entries: for raster-ordered lists that is a strip of horizontal neighbours, whose
// windows share all their rows (2 x 2 cells from a grid hint measured 3 % slower and were dropped).
#ifndef AOMHIP_CELL_WAVES
#define AOMHIP_CELL_WAVES 2   // (kernel experiments: tools/fps_build_exp.sh name "-DAOMHIP_CELL_WAVES=4")
#endif
constexpr int kCellWaves = AOMHIP_CELL_WAVES;
#ifndef AOMHIP_BIG_CELL_WAVES
#define AOMHIP_BIG_CELL_WAVES 2
#endif
#ifndef AOMHIP_BIG_CELL_PIXELS
#define AOMHIP_BIG_CELL_PIXELS 1024
#endif
constexpr int kCellReach = 16;
constexpr int kCellReachBig = 8;   // blocks of >= 1024 pixels
// The general search kernel (fullpel_search.inc) on blocks below 1 024 pixels: FOUR blocks per cell and a reach of 32.  Its n-step searches wander -- the
// centre moves in half of the steps -- and the steps that leave the window read the plane at 2.2 x the cost of a window step (profiles/r06_fps_nstep.md);
// a window shared by four neighbours costs (64 + 2 R) x (16 + 2 R) pixels where two cost (32 + 2 R) x (16 + 2 R), so the reach that fits the CU's LDS at
// six wavefronts per SIMD doubles.  NSTEP step_param 3, 4K 10-bit 16x16, ms per frame (same box): 2 x 16 0.794, 2 x 32 0.806, 3 x 32 0.694, 4 x 24 0.705,
// **4 x 32 0.638**, 4 x 36 0.724, 5 x 36 0.696, 6 x 48 0.680, 8 x 40 0.647; blocks of 32x32 keep 2 x 8 (0.513; 4 x 16 0.511).
#ifndef AOMHIP_FPS_CELL_WAVES_SMALL
#define AOMHIP_FPS_CELL_WAVES_SMALL 4
#endif
constexpr int kFpsCellReach = 32;

// Host side: how a search call cuts its list into cells.  waves = 0: no window (the kernels' plain form).
// AOMHIP_SEARCH_CELL=0 switches the window off, AOMHIP_SEARCH_CELL_R=<pixels> overrides the reach (kernel A/B only; results never depend
// on either).
struct CellPlan {
  int waves;
  int lds;       // dynamic LDS bytes of the launch
  CellMap map;
};
// blocks per cell (= wavefronts per workgroup) of the general search kernel, by block size
constexpr int cell_waves_for(int bw, int bh) { return bw * bh >= AOMHIP_BIG_CELL_PIXELS ? AOMHIP_BIG_CELL_WAVES : AOMHIP_FPS_CELL_WAVES_SMALL; }
// static_lds: the LDS the kernel declares by itself (site table, per-wavefront source slices): it comes off every budget below, so the
// workgroups-per-CU figure the window is rounded to is the one the launch really gets and the cap never exceeds the CU's 160 KB
inline CellPlan plan_cells(const aomhip_planes *ref, int bw, int bh, int n_blocks, int reach, int waves = kCellWaves, int static_lds = 2048,
                            int cap_small = kCellReach) {
  static const int env_on = [] { const char *e = getenv("AOMHIP_SEARCH_CELL"); return e ? atoi(e) : 1; }();
  static const int env_r = [] { const char *e = getenv("AOMHIP_SEARCH_CELL_R"); return e ? atoi(e) : -1; }();
  CellPlan p{};
  if (!env_on || reach < 0) return p;   // (reach < 0: the caller wants the form without a window)
  const int es = ref->bit_depth == 8 ? 1 : 2;
  static const int env_rb = [] { const char *e = getenv("AOMHIP_SEARCH_CELL_RB"); return e ? atoi(e) : -1; }();   // (blocks of >= 1024 pixels)
  // Blocks of 32x32 and larger take a reach of 8: their window is 13 KB per two-block cell at 16 (16-bit planes) and the workgroups per CU
  // it costs outweigh the radius 9 .. 16 rounds it serves -- temporal filter, 4K 10-bit: 7.04 -> 6.55 ms (R x RB sweep in profiles/r04_search_cell.md)
  const bool big = bw * bh >= 1024;
  const int cap = big ? (env_rb >= 0 ? env_rb : (env_r >= 0 ? env_r : kCellReachBig)) : (env_r >= 0 ? env_r : cap_small);
  const int r = (env_r >= 0 || (big && env_rb >= 0)) ? cap : (reach < cap ? reach : cap);
  // the window of a cell whose blocks are horizontal neighbours and start at the same MV; rounded up so that the workgroups of a CU
  // fill its 160 KB without a remainder (the slack serves cells whose start MVs differ; a window that still does not fit shrinks its
  // reach, stage_cell_window)
  int64_t pitch = ((int64_t)waves * bw + 2 * r) * es + 16;
  pitch += ((7 - (pitch >> 2)) & 31) << 2;
  int64_t want = pitch * (bh + 2 * r);
  for (int k = 32 / waves; k >= 1; --k) {   // (16 two-wavefront workgroups = the CU's 32 wavefronts)
    const int64_t budget = (160 * 1024 - 512 * k) / k - 256 - static_lds;
    if (want <= budget) { want = budget; break; }
  }
  if (want > 156 * 1024 - static_lds) want = 156 * 1024 - static_lds;
  p.waves = waves;
  p.lds = (int)want;
  // readable rows end at height + border: true of every valid plane (aomhip_planes_alloc rounds the height up to 8 first, a caller-built
  // plane need not), so the window's copy never reads past the allocation
  p.map = CellMap{ (n_blocks + waves - 1) / waves, r, (int)want, -ref->border, -ref->border, ref->stride - ref->border,
                   ref->height + ref->border };
  return p;
}

// list index of wave `wave` of cell `cell`, or -1
template <int WAVES>
__device__ __forceinline__ int cell_block_index(int cell, int wave, int n_blocks) {
  const int bi = cell * WAVES + wave;
  return bi < n_blocks ? bi : -1;
}

// The staged window: pixels [x0, x1) x [y0, y1) of the frame at LDS byte offsets (y - y0) * pitch + (x - x0) * sizeof(T).
// All members are wave-uniform (SGPRs).
struct CellWin {
  int x0, y0, x1, y1, pitch;
  __device__ __forceinline__ bool covers(int px0, int py0, int px1, int py1) const {  // [px0, px1) x [py0, py1)
    return px0 >= x0 && px1 <= x1 && py0 >= y0 && py1 <= y1;
  }
};

// Called by every thread of the workgroup (NT threads).  `have`, (px, py): this wave's block and the pixel position of its
// (clamped) start MV's top-left corner.  Returns the window (empty: x1 <= x0) with its pixels in `lds_win`; ends in a
// workgroup barrier.
template <typename T, int W, int H, int WAVES>
__device__ __forceinline__ CellWin stage_cell_window(const CellMap &m, const T *ref_frame, int rstride, bool have, int px, int py,
                                                     int wave, uint32_t *lds_win, [[maybe_unused]] unsigned long long t_begin = 0, [[maybe_unused]] int prof_bi = -1) {
  constexpr int NT = WAVES * 64;
  constexpr int ES = (int)sizeof(T);
  __shared__ int bb[WAVES][2];
#ifdef AOMHIP_CELL_PROF
  const unsigned long long tp0 = __builtin_readcyclecounter();
#endif
  if ((threadIdx.x & 63) == 0) {
    bb[wave][0] = have ? px : INT_MAX;
    bb[wave][1] = have ? py : INT_MAX;
  }
  __syncthreads();
  // bounding box of the start positions: lane w of every wavefront takes wave w's entry, one DPP row (16 lanes) reduces them
  static_assert(WAVES <= 16, "one DPP row");
  const int ln = threadIdx.x & 63;
  const int ex = ln < WAVES ? bb[ln][0] : INT_MAX, ey = ln < WAVES ? bb[ln][1] : INT_MAX;
  auto row16 = [](int v, bool take_min) {
    auto op = [take_min](int a, int b) { return take_min ? min(a, b) : max(a, b); };
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    return v;
  };
  int bx0 = row16(ex, true), by0 = row16(ey, true);
  int bx1 = row16(ex == INT_MAX ? INT_MIN : ex, false), by1 = row16(ey == INT_MAX ? INT_MIN : ey, false);
  bx0 = __builtin_amdgcn_readfirstlane(bx0); by0 = __builtin_amdgcn_readfirstlane(by0);
  bx1 = __builtin_amdgcn_readfirstlane(bx1); by1 = __builtin_amdgcn_readfirstlane(by1);
  CellWin cw{ 0, 0, 0, 0, 0 };
  if (bx0 == INT_MAX) return cw;   // (uniform over the workgroup: no barrier is skipped by part of it)
  // the largest reach <= win_r whose window fits the budget
  int r = m.win_r;
  for (;;) {
    // left edge rounded down so that a row's first byte is 16-byte aligned in the plane when the plane's rows are
    int x0 = max(bx0 - r, m.xmin), x1 = min(bx1 + W + r, m.xmax);
    const int y0 = max(by0 - r, m.ymin), y1 = min(by1 + H + r, m.ymax);
    const uintptr_t a = reinterpret_cast<uintptr_t>(ref_frame + (int64_t)y0 * rstride + x0);
    x0 -= min((int)((a & 15) / ES), x0 - m.xmin);
    // whole 16-byte chunks per row, never past the readable pixels
    const int wb = (x1 - x0) * ES, up = (wb + 15) & ~15;
    x1 = x0 + up / ES <= m.xmax ? x0 + up / ES : x0 + (wb & ~15) / ES;
    // + 16: a unit's realignment dword behind the last pixel; then the smallest pitch whose dword count is 7 mod 32: the 8-lane
    // groups read 16-byte units of consecutive rows, and with the rows of the 4 sites of a half-wavefront a power-of-two number of
    // rows / columns apart any even dword pitch puts them on the same banks (measured: pitch 400 = 8-way conflicts at r >= 8)
    int pitch = (x1 - x0) * ES + 16;
    pitch += ((7 - (pitch >> 2)) & 31) << 2;
    if (x1 > x0 && y1 > y0 && (int64_t)pitch * (y1 - y0) <= m.lds_bytes) {
      cw = CellWin{ x0, y0, x1, y1, pitch };
      break;
    }
    if (r <= 0) return cw;
    r = r > 8 ? r - 8 : 0;
  }
  // copy: 16-byte chunks, consecutive threads on consecutive chunks of a row (the rows of the window are only dword aligned: four
  // ds_write_b32 per chunk)
  const int cpr = (cw.x1 - cw.x0) * ES / 16, total = cpr * (cw.y1 - cw.y0);
  const char *g0 = reinterpret_cast<const char *>(ref_frame + (int64_t)cw.y0 * rstride + cw.x0);
#ifdef AOMHIP_CELL_PROF
  const unsigned long long tp1 = __builtin_readcyclecounter();
#endif
  // four loads in flight per thread before the first LDS write: the copy costs one memory round trip per batch, not one per chunk
  constexpr int kBatch = 4;
  for (int q0 = threadIdx.x; q0 < total; q0 += kBatch * NT) {
    MU128 v[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int q = min(q0 + j * NT, total - 1);
      const int rr = q / cpr, c = q - rr * cpr;
      v[j] = *reinterpret_cast<const MU128 *>(g0 + (int64_t)rr * rstride * ES + c * 16);
    }
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
      const int q = q0 + j * NT;
      if (q < total) {
        const int rr = q / cpr, c = q - rr * cpr;
        uint32_t *d = lds_win + ((rr * cw.pitch + c * 16) >> 2);
        d[0] = v[j].v[0]; d[1] = v[j].v[1]; d[2] = v[j].v[2]; d[3] = v[j].v[3];
      }
    }
  }
#ifdef AOMHIP_CELL_PROF
  const unsigned long long tp2 = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef AOMHIP_CELL_PROF
  if ((threadIdx.x & 63) == 0 && prof_bi >= 0 && prof_bi < 40960) {
    const unsigned long long tp3 = __builtin_readcyclecounter();
    unsigned int *pr = g_cell_prof + prof_bi * 16;
    pr[9] = (unsigned int)(tp0 - t_begin); pr[10] = (unsigned int)(tp1 - tp0); pr[11] = (unsigned int)(tp2 - tp1); pr[12] = (unsigned int)(tp3 - tp2);
  }
#endif
  return cw;
}

// Variance of the W x H block at LDS byte offset `off` of the window (row pitch `pitch`) against the source units held in
// registers by the 8 lanes of group 0 (G8 geometry): get_mvpred_var_cost's vf(src, ref) without a global-memory round trip.
// sum = S(src) - S(ref), sse = S(src^2) + S(ref^2) - 2 S(src * ref) in 32-bit modular arithmetic (exact: the true sse of a KEEP-sized
// block is < 2^32 at every bit depth); the packed dot products do 2 (16-bit) or 4 (8-bit) pixels per instruction.
template <typename T> __device__ __forceinline__ uint32_t dotw(uint32_t a, uint32_t b, uint32_t acc) {
  if constexpr (sizeof(T) == 1) return __builtin_amdgcn_udot4(a, b, acc, false);
  else return __builtin_amdgcn_udot2(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned short)))) unsigned short, a),
                                     __builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned short)))) unsigned short, b), acc, false);
}

template <typename T, int W, int H>
__device__ __forceinline__ uint32_t group8_variance_lds(const uint32_t *win, unsigned off, int pitch, int l, bool active, int bit_depth,
                                                        const typename G8<T, W, H>::L (&s)[G8<T, W, H>::KEEP ? G8<T, W, H>::PER_LANE : 1]) {
  using G = G8<T, W, H>;
  static_assert(G::KEEP, "LDS path is only instantiated for blocks whose source units stay in registers");
  static_assert(W * H <= (sizeof(T) == 1 ? 16384 : 256), "32-bit sse");
  uint32_t ssum = 0, rsum = 0, ss = 0, rr = 0, sr = 0;
  if (active) {
#pragma unroll
    for (int k = 0; k < G::PER_LANE; ++k) {
      const int u = l + 8 * k;
      if (u < G::U) {
        const int row = u / G::UPR, colb = (u % G::UPR) * G::UB;
        const unsigned o = off + (unsigned)(row * pitch + colb);
        const uint32_t *p = win + (o >> 2);
        const unsigned sh = o & 3;
        uint32_t d[G::UB / 4 + 1];
#pragma unroll
        for (int i = 0; i <= G::UB / 4; ++i) d[i] = p[i];
#pragma unroll
        for (int i = 0; i < G::UB / 4; ++i) {
          const uint32_t rv = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh), sv = s[k].v[i];
          ssum = sadw<T>(sv, 0u, ssum);
          rsum = sadw<T>(rv, 0u, rsum);
          ss = dotw<T>(sv, sv, ss);
          rr = dotw<T>(rv, rv, rr);
          sr = dotw<T>(sv, rv, sr);
        }
      }
    }
  }
  int32_t sum = (int32_t)ssum - (int32_t)rsum;
  uint32_t sse = ss + rr - 2u * sr;
  sum += __builtin_amdgcn_update_dpp(0, sum, 0xB1, 0xf, 0xf, false);
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x4E, 0xf, 0xf, false);
  sum += __builtin_amdgcn_update_dpp(0, sum, 0x141, 0xf, 0xf, false);
  sse += __builtin_amdgcn_update_dpp(0u, sse, 0xB1, 0xf, 0xf, false);
  sse += __builtin_amdgcn_update_dpp(0u, sse, 0x4E, 0xf, 0xf, false);
  sse += __builtin_amdgcn_update_dpp(0u, sse, 0x141, 0xf, 0xf, false);
  // variance.c:141-148 / :383-420 (the same finish as group16_variance)
  int32_t sfin;
  uint32_t q;
  if (bit_depth == 10) {
    q = (uint32_t)(((uint64_t)sse + 8) >> 4);
    sfin = (sum + 2) >> 2;
  } else if (bit_depth == 12) {
    q = (uint32_t)(((uint64_t)sse + 128) >> 8);
    sfin = (sum + 8) >> 4;
  } else {
    q = sse;
    sfin = sum;
  }
  constexpr int LOG2N = __builtin_ctz(W * H);
  const int64_t sq = ((int64_t)sfin * sfin) >> LOG2N;
  if (bit_depth == 8) return q - (uint32_t)sq;
  const int64_t v = (int64_t)q - sq;
  return v >= 0 ? (uint32_t)v : 0;
}

}  // namespace aomhip

#endif  // AOMHIP_CSRC_SEARCH_WINDOW_H_
