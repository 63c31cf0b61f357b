// AV1 deblocking filter over a whole plane on gfx950.
// Reference: taps aom_dsp/loopfilter.c (8-bit :20-511, highbd :515-997); thresholds
// av1/common/av1_loopfilter.c:47-66,118-120; edge driver av1_loopfilter.c:223-328,1304-1352,1905-
// and its order thread_common.c:251-322,375-395 (per superblock row: vertical edges, then horizontal).
//
// The host (or an upstream kernel) reduces the reference's mode-info walk (set_lpf_parameters) to one
// 4-byte record per 4x4 unit of the plane: {len_v, lvl_v, len_h, lvl_h} = filter length (0/4/6/8/14) and
// filter level of the vertical edge on the unit's left side and of the horizontal edge on its top side.
// Because the length is derived from the smaller of the two adjacent transform sizes, the pixels an edge
// reads and writes lie strictly inside its own half-transform zone: all vertical edges of a plane are
// mutually independent, and so are all horizontal edges.  The plane is therefore filtered by exactly two
// launches -- every vertical edge, then every horizontal edge -- which tests/ prove equal to the
// reference's superblock-row order on random transform partitions.
//
//   vertical pass  : one lane per (edge unit, pixel row): one 16-byte window load (2 for 16-bit pixels),
//                    taps in registers, stores restricted to the edge's own zone (2 / 8 / 16 bytes)
//   horizontal pass: one lane per (pixel column, edge unit): lanes of a wavefront cover 64 adjacent
//                    columns, so each of the up-to-14 row accesses is one coalesced 64-pixel segment
// Algorithmic bytes: each pixel is read and written once per pass.
#include <algorithm>

#include "common.h"

namespace aomhip {

struct __attribute__((packed, aligned(1))) DU128 { uint32_t v[4]; };
struct __attribute__((packed, aligned(4))) DU128A4 { uint32_t v[4]; };   // 16 bytes at a 4-byte boundary
struct __attribute__((packed, aligned(1))) DU64 { uint32_t v[2]; };

__device__ __forceinline__ int iabsd(int v) { return v < 0 ? -v : v; }
__device__ __forceinline__ int clampd(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// x[7 + i] holds tap i (i = -7 .. 6: p6 .. p0, q0 .. q6); filters in place.  lim8 / blim8 / thr8: the 8-bit thresholds
// the reference's functions take as *limit / *blimit / *thresh (scaled by bd - 8 inside, loopfilter.c:521-522,568,596).
__device__ __forceinline__ void lpf_window_thr(int (&x)[14], int len, int lim8, int blim8, int thr8, int bd) {
  const int sh = bd - 8;
  const int lim = lim8 << sh, blim = blim8 << sh, thr = thr8 << sh, one = 1 << sh;
  const int p3 = x[3], p2 = x[4], p1 = x[5], p0 = x[6], q0 = x[7], q1 = x[8], q2 = x[9], q3 = x[10];

  // filter_mask* / flat_mask* (loopfilter.c:26-102,521-600) as MAXIMA: "no difference of the set exceeds the limit" is one comparison of the
  // largest one.  Written as a chain of `||` every comparison became a v_cmp into a scalar register pair plus an s_or -- ~25 scalar
  // instructions per line, four lines per lane: the kernels issued 531 scalar against 898 vector instructions per wavefront and waited on
  // the CU's one scalar unit (profiles/r05z inner loop PMC).  |a - b| of two pixel values is one v_sad_u32.
  auto ad = [](int a, int b) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return (int)d;
  };
  const int d_p1p0 = ad(p1, p0), d_q1q0 = ad(q1, q0);
  int m = max(d_p1p0, d_q1q0);
  const int m6 = max(m, max(ad(p2, p1), ad(q2, q1))), m8 = max(m6, max(ad(p3, p2), ad(q3, q2)));
  const int f6 = max(m, max(ad(p2, p0), ad(q2, q0))), f8 = max(f6, max(ad(p3, p0), ad(q3, q0)));
  m = len >= 8 ? m8 : (len >= 6 ? m6 : m);
  const bool mask = m <= lim && ad(p0, q0) * 2 + (ad(p1, q1) >> 1) <= blim;
  const bool flat = len >= 6 && (len >= 8 ? f8 : f6) <= one;
  bool flat2 = false;
  if (len == 14) {
    flat2 = max(max(ad(x[2], p0), ad(x[11], q0)), max(max(ad(x[1], p0), ad(x[12], q0)), max(ad(x[0], p0), ad(x[13], q0)))) <= one;
  }

  if (len == 14 && flat2 && flat && mask) {
    // 13 taps [1 1 1 1 1 2 2 2 1 1 1 1 1] over the edge-replicated window p6..q6, >> 4 (loopfilter.c:378-424)
    int o[12];
#pragma unroll
    for (int i = 1; i <= 12; ++i) {  // output index in x[]: p5 .. q5
      int s = 8;
#pragma unroll
      for (int k = -6; k <= 6; ++k) {
        int j = i + k;
        j = j < 0 ? 0 : (j > 13 ? 13 : j);
        s += ((k >= -1 && k <= 1) ? 2 : 1) * x[j];
      }
      o[i - 1] = s >> 4;
    }
#pragma unroll
    for (int i = 1; i <= 12; ++i) x[i] = o[i - 1];
  } else if (len >= 8 && flat && mask) {
    // 7 taps [1 1 1 2 1 1 1] over p3..q3, >> 3 (loopfilter.c:216-237)
    int o[6];
#pragma unroll
    for (int i = 4; i <= 9; ++i) {  // p2 .. q2
      int s = 4 + x[i];
#pragma unroll
      for (int k = -3; k <= 3; ++k) {
        int j = i + k;
        j = j < 3 ? 3 : (j > 10 ? 10 : j);
        s += x[j];
      }
      o[i - 4] = s >> 3;
    }
#pragma unroll
    for (int i = 4; i <= 9; ++i) x[i] = o[i - 4];
  } else if (len == 6 && flat && mask) {
    // 5 taps [1 2 2 2 1] over p2..q2, >> 3 (loopfilter.c:202-214)
    int o[4];
#pragma unroll
    for (int i = 5; i <= 8; ++i) {  // p1 .. q1
      int s = 4;
#pragma unroll
      for (int k = -2; k <= 2; ++k) {
        int j = i + k;
        j = j < 4 ? 4 : (j > 9 ? 9 : j);
        s += ((k >= -1 && k <= 1) ? 2 : 1) * x[j];
      }
      o[i - 5] = s >> 3;
    }
#pragma unroll
    for (int i = 5; i <= 8; ++i) x[i] = o[i - 5];
  } else {
    // filter4 / highbd_filter4 (loopfilter.c:104-134,602-638) in the signed offset domain
    const int off = 0x80 << sh, lo = -(128 << sh), hi = (128 << sh) - 1;
    const int ps1 = p1 - off, ps0 = p0 - off, qs0 = q0 - off, qs1 = q1 - off;
    const bool hev = max(d_p1p0, d_q1q0) > thr;
    int f = hev ? clampd(ps1 - qs1, lo, hi) : 0;
    f = mask ? clampd(f + 3 * (qs0 - ps0), lo, hi) : 0;
    const int f1 = clampd(f + 4, lo, hi) >> 3;
    const int f2 = clampd(f + 3, lo, hi) >> 3;
    const int f3 = hev ? 0 : ((f1 + 1) >> 1);
    x[7] = clampd(qs0 - f1, lo, hi) + off;
    x[6] = clampd(ps0 + f2, lo, hi) + off;
    x[8] = clampd(qs1 - f3, lo, hi) + off;
    x[5] = clampd(ps1 + f3, lo, hi) + off;
  }
}

__device__ __forceinline__ void lpf_window(int (&x)[14], int len, int level, int sharpness, int bd) {
  // update_sharpness / av1_loop_filter_frame_init (av1_loopfilter.c:47-66,118-120)
  int inside = level >> ((sharpness > 0) + (sharpness > 4));
  if (sharpness > 0 && inside > 9 - sharpness) inside = 9 - sharpness;
  if (inside < 1) inside = 1;
  lpf_window_thr(x, len, inside, 2 * (level + 2) + inside, level >> 4, bd);
}

// One call of aom_[highbd_]lpf_{horizontal,vertical}_{4,6,8,14}[_dual,_quad] (loopfilter.c:136-511,602-997) on a staged
// patch: `count` pixels along the edge (4 / 8 / 16), thresholds set 0 for pixels below `second_at`, set 1 from there.
// patch layout: horizontal edge: rows -reach .. reach - 1 of `count` columns (row pitch = count); vertical edge: `count`
// rows of 2 * reach columns.  q0 is row / column `reach`.
template <typename PIX>
__global__ __launch_bounds__(64) void lpf_edge_kernel(PIX *patch, int horizontal, int len, int count, int second_at, int lim0,
                                                      int blim0, int thr0, int lim1, int blim1, int thr1, int bd) {
  const int i = threadIdx.x;
  if (i >= count) return;
  const int reach = len == 14 ? 7 : (len == 8 ? 4 : (len == 6 ? 3 : 2));
  int x[14];
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    const int t = k - 7;
    x[k] = (t >= -reach && t < reach) ? (int)(horizontal ? patch[(t + reach) * count + i] : patch[i * 2 * reach + t + reach]) : 0;
  }
  const bool g1 = i >= second_at;
  lpf_window_thr(x, len, g1 ? lim1 : lim0, g1 ? blim1 : blim0, g1 ? thr1 : thr0, bd);
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    const int t = k - 7;
    if (t >= -reach && t < reach) {
      if (horizontal) patch[(t + reach) * count + i] = (PIX)x[k];
      else patch[i * 2 * reach + t + reach] = (PIX)x[k];
    }
  }
}

constexpr int kDbThreads = 256;

// Vertical edges: lane = (unit column ux, pixel row y).
template <typename PIX>
__global__ __launch_bounds__(kDbThreads) void deblock_vert_kernel(PIX *origin, int stride, int width, int height,
                                                                  const uint8_t *__restrict__ params, int units_stride,
                                                                  int sharpness, int bd) {
  const int ux = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * (kDbThreads / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ucols = (width + 3) >> 2;
  if (ux <= 0 || ux >= ucols || y >= height) return;
  const uint8_t *e = params + ((size_t)(y >> 2) * units_stride + ux) * 4;
  const int len = e[0], level = e[1];
  if (len == 0 || level == 0) return;
  PIX *s = origin + (int64_t)y * stride + 4 * ux;  // q0
  int x[14];
  if constexpr (sizeof(PIX) == 1) {
    const DU128 w = *reinterpret_cast<const DU128 *>(s - 8);  // pixels x-8 .. x+7
#pragma unroll
    for (int i = 0; i < 14; ++i) x[i] = (w.v[(i + 1) / 4] >> (8 * ((i + 1) % 4))) & 0xFF;
  } else {
    const DU128 w0 = *reinterpret_cast<const DU128 *>(s - 8), w1 = *reinterpret_cast<const DU128 *>(s);
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int px = i + 1;  // index within the 16-pixel window
      const uint32_t d = px < 8 ? w0.v[px / 2] : w1.v[(px - 8) / 2];
      x[i] = (d >> (16 * (px % 2))) & 0xFFFF;
    }
  }
  lpf_window(x, len, level, sharpness, bd);
  // write back only this edge's own zone
  if (len == 14) {
#pragma unroll
    for (int i = 1; i <= 12; ++i) s[i - 7] = (PIX)x[i];
  } else if (len == 8) {
#pragma unroll
    for (int i = 4; i <= 9; ++i) s[i - 7] = (PIX)x[i];
  } else {
#pragma unroll
    for (int i = 5; i <= 8; ++i) s[i - 7] = (PIX)x[i];
  }
}

// Horizontal edges: lane = (pixel column xcol, unit row uy).
template <typename PIX>
__global__ __launch_bounds__(kDbThreads) void deblock_horz_kernel(PIX *origin, int stride, int width, int height,
                                                                  const uint8_t *__restrict__ params, int units_stride,
                                                                  int sharpness, int bd) {
  const int xcol = blockIdx.x * 64 + (threadIdx.x & 63);
  const int uy = blockIdx.y * (kDbThreads / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int urows = (height + 3) >> 2;
  if (xcol >= width || uy <= 0 || uy >= urows) return;
  const uint8_t *e = params + ((size_t)uy * units_stride + (xcol >> 2)) * 4;
  const int len = e[2], level = e[3];
  if (len == 0 || level == 0) return;
  PIX *s = origin + (int64_t)(4 * uy) * stride + xcol;  // q0
  const int reach = len == 14 ? 7 : (len == 8 ? 4 : (len == 6 ? 3 : 2));
  int x[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int k = i - 7;
    x[i] = (k >= -reach && k < reach) ? (int)s[(int64_t)k * stride] : 0;
  }
  lpf_window(x, len, level, sharpness, bd);
  const int wr = len == 14 ? 6 : (len == 8 ? 3 : 2);
#pragma unroll
  for (int i = 1; i <= 12; ++i) {
    const int k = i - 7;
    if (k >= -wr && k < wr) s[(int64_t)k * stride] = (PIX)x[i];
  }
}

// ---- Four lines of an edge per lane (width and height multiples of 4, rows that keep 4-pixel groups naturally aligned).  The one-line kernels above are
// 32 400 wavefronts of ~330 instructions on a 4K plane: a wavefront lives for one memory round trip and little else (PMC: VALU floor 12.0 +
// 5.6 us against 30 us for the pair of launches, profiles/r04_inner_loop_pmc.json).  Here a lane owns the four pixel rows (vertical edges) /
// four pixel columns (horizontal edges) of ONE 4x4 unit's edge -- one parameter record, four independent lines whose loads are all in
// flight before the first filter runs: a quarter of the wavefronts, the same latency each.  The arithmetic per line is lpf_window's, unchanged.
template <typename PIX>
__global__ __launch_bounds__(kDbThreads) void deblock_vert4_kernel(PIX *origin, int stride, int width, int height, const uint8_t *__restrict__ params,
                                                                   int units_stride, int sharpness, int bd) {
  const int ux = blockIdx.x * 64 + (threadIdx.x & 63);
  const int uy = blockIdx.y * (kDbThreads / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ucols = width >> 2;
  if (ux <= 0 || ux >= ucols || 4 * uy >= height) return;
  const uint8_t *e = params + ((size_t)uy * units_stride + ux) * 4;
  const int len = e[0], level = e[1];
  if (len == 0 || level == 0) return;
  PIX *s = origin + (int64_t)(4 * uy) * stride + 4 * ux;  // q0 of the unit's first row
  if constexpr (sizeof(PIX) == 1) {
    // 8-bit planes: the 16-pixel window of a line is ONE 16-byte load (4-byte aligned: the edge sits at a multiple of 4 pixels)
    DU128A4 w[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const DU128A4 *>(s + (int64_t)r * stride - 8);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t (&wv)[4] = w[r].v;
      int x[14];
#pragma unroll
      for (int i = 0; i < 14; ++i) x[i] = (int)((wv[(i + 1) / 4] >> (8 * ((i + 1) % 4))) & 0xFF);
      lpf_window(x, len, level, sharpness, bd);
      // this edge's own zone only, cut at its natural alignments: q0 is a dword boundary, x - 6 and x - 2 are halfword boundaries
      uint8_t *sr = s + (int64_t)r * stride;
      auto p2 = [&](int i) { return (uint16_t)((uint32_t)x[i] | ((uint32_t)x[i + 1] << 8)); };   // pixels (i - 7, i - 6) relative to q0
      auto p4 = [&](int i) { return (uint32_t)x[i] | ((uint32_t)x[i + 1] << 8) | ((uint32_t)x[i + 2] << 16) | ((uint32_t)x[i + 3] << 24); };
      if (len == 14) {
        *reinterpret_cast<uint16_t *>(sr - 6) = p2(1);
        *reinterpret_cast<uint32_t *>(sr - 4) = p4(3);
        *reinterpret_cast<uint32_t *>(sr) = p4(7);
        *reinterpret_cast<uint16_t *>(sr + 4) = p2(11);
      } else if (len == 8) {
        sr[-3] = (uint8_t)x[4];
        *reinterpret_cast<uint16_t *>(sr - 2) = p2(5);
        *reinterpret_cast<uint16_t *>(sr) = p2(7);
        sr[2] = (uint8_t)x[9];
      } else {
        *reinterpret_cast<uint16_t *>(sr - 2) = p2(5);
        *reinterpret_cast<uint16_t *>(sr) = p2(7);
      }
    }
  } else {
  DU128 w0[4], w1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    w0[r] = *reinterpret_cast<const DU128 *>(s + (int64_t)r * stride - 8);
    w1[r] = *reinterpret_cast<const DU128 *>(s + (int64_t)r * stride);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    int x[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int px = i + 1;  // index within the 16-pixel window
      const uint32_t d = px < 8 ? w0[r].v[px / 2] : w1[r].v[(px - 8) / 2];
      x[i] = (d >> (16 * (px % 2))) & 0xFFFF;
    }
    lpf_window(x, len, level, sharpness, bd);
    // this edge's own zone only (the neighbours' zones are theirs to write), as few stores as its 4-byte alignment allows: the edge sits at
    // a multiple of 4 pixels, so x - 6 and x - 2 are dword boundaries (the one-line kernel issues a 2-byte store per pixel)
    uint16_t *sr = s + (int64_t)r * stride;
    auto pk = [&](int i) { return (uint32_t)x[i] | ((uint32_t)x[i + 1] << 16); };   // pixels (i - 7, i - 6) relative to q0
    if (len == 14) {
      *reinterpret_cast<DU128 *>(sr - 6) = DU128{ { pk(1), pk(3), pk(5), pk(7) } };
      *reinterpret_cast<DU64 *>(sr + 2) = DU64{ { pk(9), pk(11) } };
    } else if (len == 8) {
      sr[-3] = (uint16_t)x[4];
      *reinterpret_cast<DU64 *>(sr - 2) = DU64{ { pk(5), pk(7) } };
      sr[2] = (uint16_t)x[9];
    } else {
      *reinterpret_cast<DU64 *>(sr - 2) = DU64{ { pk(5), pk(7) } };
    }
  }
  }
}

template <typename PIX>
__global__ __launch_bounds__(kDbThreads) void deblock_horz4_kernel(PIX *origin, int stride, int width, int height, const uint8_t *__restrict__ params,
                                                                   int units_stride, int sharpness, int bd) {
  const int ux = blockIdx.x * 64 + (threadIdx.x & 63);
  const int uy = blockIdx.y * (kDbThreads / 64) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (4 * ux >= width || uy <= 0 || 4 * uy >= height) return;
  const uint8_t *e = params + ((size_t)uy * units_stride + ux) * 4;
  const int len = e[2], level = e[3];
  if (len == 0 || level == 0) return;
  PIX *s = origin + (int64_t)(4 * uy) * stride + 4 * ux;  // q0 of the unit's first column
  const int reach = len == 14 ? 7 : (len == 8 ? 4 : (len == 6 ? 3 : 2));
  const int wr = len == 14 ? 6 : (len == 8 ? 3 : 2);
  int out[4][14];
  if constexpr (sizeof(PIX) == 1) {   // 8-bit planes: the unit's four columns of a row are one dword
    uint32_t rows[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int k = i - 7;
      rows[i] = (k >= -reach && k < reach) ? *reinterpret_cast<const uint32_t *>(s + (int64_t)k * stride) : 0u;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int i = 0; i < 14; ++i) out[c][i] = (int)((rows[i] >> (8 * c)) & 0xFF);
      lpf_window(out[c], len, level, sharpness, bd);
    }
#pragma unroll
    for (int i = 1; i <= 12; ++i) {
      const int k = i - 7;
      if (k >= -wr && k < wr)
        *reinterpret_cast<uint32_t *>(s + (int64_t)k * stride) =
            (uint32_t)out[0][i] | ((uint32_t)out[1][i] << 8) | ((uint32_t)out[2][i] << 16) | ((uint32_t)out[3][i] << 24);
    }
  } else {
  uint2 rows[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) {
    const int k = i - 7;
    rows[i] = (k >= -reach && k < reach) ? *reinterpret_cast<const uint2 *>(s + (int64_t)k * stride) : make_uint2(0, 0);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int i = 0; i < 14; ++i) out[c][i] = (int)(((c < 2 ? rows[i].x : rows[i].y) >> (16 * (c & 1))) & 0xFFFF);
    lpf_window(out[c], len, level, sharpness, bd);
  }
#pragma unroll
  for (int i = 1; i <= 12; ++i) {
    const int k = i - 7;
    if (k >= -wr && k < wr)
      *reinterpret_cast<uint2 *>(s + (int64_t)k * stride) =
          make_uint2((uint32_t)out[0][i] | ((uint32_t)out[1][i] << 16), (uint32_t)out[2][i] | ((uint32_t)out[3][i] << 16));
  }
  }
}

// ---- Both passes in ONE launch, out of place (aomhip_deblock_plane_fused).  A workgroup produces a kFW x kFH tile of the output:
// it stages the tile + an 8-pixel halo on every side in LDS (as 16-bit pixels), filters every vertical edge that touches the staged
// region's tile columns -- for the halo ROWS too: the horizontal edges at the tile's top and bottom read vertically filtered pixels
// 7 rows outside it -- then every horizontal edge of the tile (inclusive of both boundary rows), and stores the tile.  The plane is
// read 1.4 x (the halos come out of L2: neighbouring tiles run at the same time) and written once, instead of read twice and
// written twice; the halo's vertical edges are filtered twice (1.25 x the vertical-edge arithmetic).  In place this would race with
// the neighbours' stores -- hence source and destination planes.  Edge zones are disjoint (see the header of this file), so the
// order of the edges inside a pass is free and every intermediate equals the two-launch form's.
constexpr int kFW = 128, kFH = 64, kFHalo = 8;
constexpr int kFRows = kFH + 2 * kFHalo, kFCols = kFW + 2 * kFHalo;   // 80 x 144 staged pixels
constexpr int kFPitch = 148;                                            // LDS row pitch in pixels: 74 dwords = 2 * (5 r mod 32) banks over 32 rows
constexpr int kFUCols = kFW / 4 + 1, kFURows = kFRows / 4;              // 33 x 20 edge-parameter records

template <typename PIX>
__global__ __launch_bounds__(kDbThreads) void deblock_fused_kernel(const PIX *__restrict__ src_origin, int src_stride, PIX *__restrict__ dst_origin,
                                                                   int dst_stride, int width, int height, int border,
                                                                   const uint8_t *__restrict__ params, int units_stride, int sharpness, int bd) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[kFRows * kFPitch];
  __shared__ uint32_t sp[kFURows * kFUCols];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * kFW, y0 = blockIdx.y * kFH;
  const int ucols = (width + 3) >> 2, urows = (height + 3) >> 2;
  // 1. stage pixels (8 per step; coordinates clamped into the bordered plane: what lies outside the frame is never used by an edge
  //    that is filtered) and the edge-parameter records of the region
  for (int q = tid; q < kFRows * (kFCols / 8); q += kDbThreads) {
    const int r = q / (kFCols / 8), c8 = q - r * (kFCols / 8);
    const int y = min(max(y0 - kFHalo + r, -border), height + border - 1);
    const int x = x0 - kFHalo + c8 * 8;
    uint16_t v[8];
    if (x >= -border && x + 8 <= width + border) {
      const PIX *p = src_origin + (int64_t)y * src_stride + x;
      if constexpr (sizeof(PIX) == 2) {
        const DU128 w = *reinterpret_cast<const DU128 *>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (uint16_t)(w.v[i / 2] >> (16 * (i % 2)));
      } else {
        const uint32_t w0 = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p)), w1 = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(p) + 4);
        // (byte loads would do as well: 8-bit planes are not the measured path; the two dword loads need p 4-byte aligned, else scalar)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (uint16_t)(((i < 4 ? w0 : w1) >> (8 * (i % 4))) & 0xFF);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int xc = min(max(x + i, -border), width + border - 1);
        v[i] = (uint16_t)src_origin[(int64_t)y * src_stride + xc];
      }
    }
    uint32_t *t = reinterpret_cast<uint32_t *>(&tile[r * kFPitch + c8 * 8]);
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = (uint32_t)v[2 * i] | ((uint32_t)v[2 * i + 1] << 16);
  }
  for (int q = tid; q < kFURows * kFUCols; q += kDbThreads) {
    const int ur = q / kFUCols, uc = q - ur * kFUCols;
    const int uy = (y0 - kFHalo) / 4 + ur, ux = x0 / 4 + uc;   // (y0 - 8 is a multiple of 4: the division is exact, also at y0 = 0)
    uint32_t rec = 0;
    if (uy >= 0 && uy < urows && ux >= 0 && ux < ucols) rec = *reinterpret_cast<const uint32_t *>(params + ((size_t)uy * units_stride + ux) * 4);
    sp[q] = rec;
  }
  __syncthreads();
  // 2. vertical edges: item = (edge column e, staged row r); consecutive lanes take consecutive rows (conflict-free at this pitch)
  for (int q = tid; q < kFUCols * kFRows; q += kDbThreads) {
    const int e = q / kFRows, r = q - e * kFRows;
    const int y = y0 - kFHalo + r, ux = x0 / 4 + e;
    if (ux <= 0 || ux >= ucols || y < 0 || y >= height) continue;
    const uint32_t rec = sp[(r >> 2) * kFUCols + e];
    const int len = rec & 0xFF, level = (rec >> 8) & 0xFF;
    if (len == 0 || level == 0) continue;
    uint16_t *row = &tile[r * kFPitch + 4 * e];   // pixels x - 8 .. x + 7 of the edge at x = x0 + 4 e (q0 = row[8])
    uint32_t w[8];
    {
      const uint2 a = *reinterpret_cast<const uint2 *>(row), b = *reinterpret_cast<const uint2 *>(row + 4), c = *reinterpret_cast<const uint2 *>(row + 8),
                  d = *reinterpret_cast<const uint2 *>(row + 12);
      w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; w[4] = c.x; w[5] = c.y; w[6] = d.x; w[7] = d.y;
    }
    int x[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) x[i] = (w[(i + 1) / 2] >> (16 * ((i + 1) % 2))) & 0xFFFF;
    lpf_window(x, len, level, sharpness, bd);
    // the edge's own zone: p5 .. q5 (len 14), p2 .. q2 (8), p1 .. q1 (4, 6) -- pairs of pixels at dword-aligned positions
    if (len == 14) {
#pragma unroll
      for (int i = 1; i <= 11; i += 2) *reinterpret_cast<uint32_t *>(row + i + 1) = (uint32_t)x[i] | ((uint32_t)x[i + 1] << 16);
    } else if (len == 8) {
      row[5] = (uint16_t)x[4];
      *reinterpret_cast<uint32_t *>(row + 6) = (uint32_t)x[5] | ((uint32_t)x[6] << 16);
      *reinterpret_cast<uint32_t *>(row + 8) = (uint32_t)x[7] | ((uint32_t)x[8] << 16);
      row[10] = (uint16_t)x[9];
    } else {
      *reinterpret_cast<uint32_t *>(row + 6) = (uint32_t)x[5] | ((uint32_t)x[6] << 16);
      *reinterpret_cast<uint32_t *>(row + 8) = (uint32_t)x[7] | ((uint32_t)x[8] << 16);
    }
  }
  __syncthreads();
  // 3. horizontal edges of the tile: item = (edge row k, pixel column pair); the two columns of a pair share their 4x4 unit's record
  for (int q = tid; q < (kFH / 4 + 1) * (kFW / 2); q += kDbThreads) {
    const int k = q / (kFW / 2), c2 = q - k * (kFW / 2);
    const int uy = y0 / 4 + k, xg = x0 + 2 * c2;
    if (uy <= 0 || uy >= urows || xg >= width) continue;
    const uint32_t rec = sp[(k + kFHalo / 4) * kFUCols + (c2 >> 1)];
    const int len = (rec >> 16) & 0xFF, level = rec >> 24;
    if (len == 0 || level == 0) continue;
    const int reach = len == 14 ? 7 : (len == 8 ? 4 : (len == 6 ? 3 : 2));
    uint32_t *col = reinterpret_cast<uint32_t *>(&tile[(kFHalo + 4 * k) * kFPitch + kFHalo + 2 * c2]);   // q0 of both columns
    constexpr int kPd = kFPitch / 2;
    int xa[14], xb[14];
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int t = i - 7;
      const uint32_t v = (t >= -reach && t < reach) ? col[t * kPd] : 0u;
      xa[i] = v & 0xFFFF; xb[i] = v >> 16;
    }
    lpf_window(xa, len, level, sharpness, bd);
    if (xg + 1 < width) lpf_window(xb, len, level, sharpness, bd);
    const int wr = len == 14 ? 6 : (len == 8 ? 3 : 2);
#pragma unroll
    for (int i = 1; i <= 12; ++i) {
      const int t = i - 7;
      if (t >= -wr && t < wr) col[t * kPd] = (uint32_t)xa[i] | ((uint32_t)xb[i] << 16);
    }
  }
  __syncthreads();
  // 4. store the tile
  for (int q = tid; q < kFH * (kFW / 8); q += kDbThreads) {
    const int r = q / (kFW / 8), c8 = q - r * (kFW / 8);
    const int y = y0 + r, x = x0 + c8 * 8;
    if (y >= height || x >= width) continue;
    const uint32_t *t = reinterpret_cast<const uint32_t *>(&tile[(r + kFHalo) * kFPitch + kFHalo + c8 * 8]);
    PIX *o = dst_origin + (int64_t)y * dst_stride + x;
    if (x + 8 <= width) {
      if constexpr (sizeof(PIX) == 2) {
        DU128 w;
#pragma unroll
        for (int i = 0; i < 4; ++i) w.v[i] = t[i];
        *reinterpret_cast<DU128 *>(o) = w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = (PIX)(t[i] & 0xFF); o[2 * i + 1] = (PIX)((t[i] >> 16) & 0xFF); }
      }
    } else {
      for (int i = 0; i < 8 && x + i < width; ++i) o[i] = (PIX)((t[i / 2] >> (16 * (i % 2))) & 0xFFFF);
    }
  }
}

// aom_get_sse_plane -> get_sse / highbd_get_sse (aom_dsp/psnr.c:84-198): the sum of squared differences of two whole planes, one
// atomic per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void plane_sse_kernel(const T *__restrict__ a, int a_stride, const T *__restrict__ b, int b_stride, int width,
                                                        int height, unsigned long long *__restrict__ out) {
  // A wavefront per row (four rows per workgroup), 16 bytes per lane and step, four steps of both planes requested together: the planes are read once at
  // the width of the memory path and a 4K plane ends in 540 atomic adds.  (Round 1's form read ONE pixel per lane and step -- 32 400 workgroups on a 4K plane,
  // each ending in an atomic add to the ONE result word: 392 us, all of it the serialised atomics, against 11 us now; a filter-level trial of
  // aomhip_lpf_search_sse -- copy + deblock + this -- went from 420 to 40 us.  tools/plane_sse_time.py)
  constexpr int VEC = 16 / (int)sizeof(T), kBatch = 4;
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  __shared__ unsigned long long part[4];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int y = blockIdx.x * 4 + wave;
  unsigned long long acc = 0;
  if (y < height) {
    const T *ra = a + (int64_t)y * a_stride, *rb = b + (int64_t)y * b_stride;
    const int nvec = width / VEC;   // whole 16-byte pieces of the row (any byte alignment: the loads are dword-addressable)
    for (int v0 = lane; v0 < nvec; v0 += 64 * kBatch) {
      DU128 va[kBatch], vb[kBatch];
#pragma unroll
      for (int t = 0; t < kBatch; ++t) {
        const int v = min(v0 + 64 * t, nvec - 1);
        va[t] = *reinterpret_cast<const DU128 *>(ra + (int64_t)v * VEC);
        vb[t] = *reinterpret_cast<const DU128 *>(rb + (int64_t)v * VEC);
      }
      uint32_t q = 0;   // <= kBatch * 16 * 4095^2 < 2^32
#pragma unroll
      for (int t = 0; t < kBatch; ++t) {
        if (v0 + 64 * t < nvec) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (sizeof(T) == 2) {
              const s16x2_t d = __builtin_bit_cast(s16x2_t, va[t].v[i]) - __builtin_bit_cast(s16x2_t, vb[t].v[i]);
              q = (uint32_t)__builtin_amdgcn_sdot2(d, d, (int)q, false);
            } else {
#pragma unroll
              for (int h = 0; h < 2; ++h) {   // bytes (2 h, 2 h + 1) of the dword as 16-bit halves
                const uint32_t sel = h ? 0x0c030c02u : 0x0c010c00u;
                const s16x2_t d = __builtin_bit_cast(s16x2_t, __builtin_amdgcn_perm(0u, va[t].v[i], sel)) -
                                  __builtin_bit_cast(s16x2_t, __builtin_amdgcn_perm(0u, vb[t].v[i], sel));
                q = (uint32_t)__builtin_amdgcn_sdot2(d, d, (int)q, false);
              }
            }
          }
        }
      }
      acc += q;
    }
    for (int x = nvec * VEC + lane; x < width; x += 64) {   // the row's last width % VEC pixels
      const int d = (int)ra[x] - (int)rb[x];
      acc += (unsigned)__mul24(d, d);
    }
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) acc += __shfl_xor(acc, m, 64);
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long t = part[0] + part[1] + part[2] + part[3];
    if (t) atomicAdd(out, t);
  }
}

}  // namespace aomhip

using namespace aomhip;

extern "C" {

int aomhip_deblock_plane(aomhip_ctx *ctx, const aomhip_planes *p, int frame, const uint8_t *d_edge_params,
                         int units_stride, int sharpness, int passes) {
  if (!ctx || !p || !p->base || !d_edge_params || frame < 0 || frame >= p->n_frames || sharpness < 0 ||
      sharpness > 7 || units_stride < (p->width + 3) / 4) {
    set_error("aomhip_deblock_plane: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if ((passes & 1) && p->border < 4) {  // the vertical pass loads the 16-pixel window x - 8 .. x + 7 of every edge (x = 4: from -4)
    set_error("aomhip_deblock_plane: the vertical pass needs a border of >= 4 pixels (plane has %d)", p->border);
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = p->bit_depth == 8 ? 1 : 2;
  char *origin = static_cast<char *>(p->base) +
                 ((size_t)frame * p->frame_stride + (size_t)p->border * p->stride + p->border) * esz;
  const int ucols = (p->width + 3) / 4, urows = (p->height + 3) / 4;
  const int rows_per_wg = kDbThreads / 64;
  const dim3 gv((ucols + 63) / 64, (p->height + rows_per_wg - 1) / rows_per_wg);
  const dim3 gh((p->width + 63) / 64, (urows + rows_per_wg - 1) / rows_per_wg);
  // planes whose rows keep 4-pixel groups naturally aligned (8 bytes at 16 bits, 4 at 8): four lines of an edge per lane (AOMHIP_DEBLOCK_LINES=1: the one-line kernels, A/B)
  const char *one = getenv("AOMHIP_DEBLOCK_LINES");
  const bool four = (p->width & 3) == 0 && (p->height & 3) == 0 && (p->stride & 3) == 0 && (p->border & 3) == 0 && p->border >= 8 &&
                    (reinterpret_cast<uintptr_t>(origin) & (4 * esz - 1)) == 0 && !(one && atoi(one) == 1);
  if (four) {
    const dim3 g4((ucols + 63) / 64, (urows + rows_per_wg - 1) / rows_per_wg);
    if (passes & 1) {
      if (esz == 1)
        hipLaunchKernelGGL(deblock_vert4_kernel<uint8_t>, g4, dim3(kDbThreads), 0, ctx->stream, reinterpret_cast<uint8_t *>(origin), p->stride, p->width,
                           p->height, d_edge_params, units_stride, sharpness, p->bit_depth);
      else
        hipLaunchKernelGGL(deblock_vert4_kernel<uint16_t>, g4, dim3(kDbThreads), 0, ctx->stream, reinterpret_cast<uint16_t *>(origin), p->stride, p->width,
                           p->height, d_edge_params, units_stride, sharpness, p->bit_depth);
      AOMHIP_LAUNCH_CHECK();
    }
    if (passes & 2) {
      if (esz == 1)
        hipLaunchKernelGGL(deblock_horz4_kernel<uint8_t>, g4, dim3(kDbThreads), 0, ctx->stream, reinterpret_cast<uint8_t *>(origin), p->stride, p->width,
                           p->height, d_edge_params, units_stride, sharpness, p->bit_depth);
      else
        hipLaunchKernelGGL(deblock_horz4_kernel<uint16_t>, g4, dim3(kDbThreads), 0, ctx->stream, reinterpret_cast<uint16_t *>(origin), p->stride, p->width,
                           p->height, d_edge_params, units_stride, sharpness, p->bit_depth);
      AOMHIP_LAUNCH_CHECK();
    }
    return AOMHIP_OK;
  }
  if (passes & 1) {
    if (esz == 1)
      hipLaunchKernelGGL(deblock_vert_kernel<uint8_t>, gv, dim3(kDbThreads), 0, ctx->stream,
                         reinterpret_cast<uint8_t *>(origin), p->stride, p->width, p->height, d_edge_params,
                         units_stride, sharpness, p->bit_depth);
    else
      hipLaunchKernelGGL(deblock_vert_kernel<uint16_t>, gv, dim3(kDbThreads), 0, ctx->stream,
                         reinterpret_cast<uint16_t *>(origin), p->stride, p->width, p->height, d_edge_params,
                         units_stride, sharpness, p->bit_depth);
    AOMHIP_LAUNCH_CHECK();
  }
  if (passes & 2) {
    if (esz == 1)
      hipLaunchKernelGGL(deblock_horz_kernel<uint8_t>, gh, dim3(kDbThreads), 0, ctx->stream,
                         reinterpret_cast<uint8_t *>(origin), p->stride, p->width, p->height, d_edge_params,
                         units_stride, sharpness, p->bit_depth);
    else
      hipLaunchKernelGGL(deblock_horz_kernel<uint16_t>, gh, dim3(kDbThreads), 0, ctx->stream,
                         reinterpret_cast<uint16_t *>(origin), p->stride, p->width, p->height, d_edge_params,
                         units_stride, sharpness, p->bit_depth);
    AOMHIP_LAUNCH_CHECK();
  }
  return AOMHIP_OK;
}

int aomhip_deblock_plane_fused(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *dst, int dst_frame,
                               const uint8_t *d_edge_params, int units_stride, int sharpness) {
  if (!ctx || !src || !dst || !src->base || !dst->base || !d_edge_params || src_frame < 0 || src_frame >= src->n_frames || dst_frame < 0 ||
      dst_frame >= dst->n_frames || sharpness < 0 || sharpness > 7 || units_stride < (src->width + 3) / 4 || src->width != dst->width ||
      src->height != dst->height || src->bit_depth != dst->bit_depth) {
    set_error("aomhip_deblock_plane_fused: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  if (src->base == dst->base && src_frame == dst_frame) {
    set_error("aomhip_deblock_plane_fused: source and destination must be different frames (tiles read their neighbours' unfiltered halo)");
    return AOMHIP_ERR_INVALID;
  }
  if (src->border < 8) {
    set_error("aomhip_deblock_plane_fused: the source plane needs a border of >= 8 pixels (has %d)", src->border);
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = src->bit_depth == 8 ? 1 : 2;
  // the kernel moves tile rows as 16-byte (16-bit planes) / 4-byte (8-bit) vectors: origins, strides and frame strides of both planes must
  // keep that alignment (aomhip_planes_alloc does; a caller-built aomhip_planes may not)
  {
    const size_t al = esz == 2 ? 16 : 4;
    auto aligned = [&](const aomhip_planes *q, int f) {
      const uintptr_t o = reinterpret_cast<uintptr_t>(q->base) + ((size_t)f * q->frame_stride + (size_t)q->border * q->stride + q->border) * esz;
      return o % al == 0 && ((size_t)q->stride * esz) % al == 0;
    };
    if (!aligned(src, src_frame) || !aligned(dst, dst_frame)) {
      set_error("aomhip_deblock_plane_fused: plane origin / stride not %zu-byte aligned (use aomhip_deblock_plane for such planes)", al);
      return AOMHIP_ERR_INVALID;
    }
  }
  const char *so = static_cast<const char *>(src->base) + ((size_t)src_frame * src->frame_stride + (size_t)src->border * src->stride + src->border) * esz;
  char *dor = static_cast<char *>(dst->base) + ((size_t)dst_frame * dst->frame_stride + (size_t)dst->border * dst->stride + dst->border) * esz;
  const dim3 grid((src->width + kFW - 1) / kFW, (src->height + kFH - 1) / kFH);
  if (esz == 1)
    hipLaunchKernelGGL(deblock_fused_kernel<uint8_t>, grid, dim3(kDbThreads), 0, ctx->stream, reinterpret_cast<const uint8_t *>(so), src->stride,
                       reinterpret_cast<uint8_t *>(dor), dst->stride, src->width, src->height, src->border, d_edge_params, units_stride, sharpness,
                       src->bit_depth);
  else
    hipLaunchKernelGGL(deblock_fused_kernel<uint16_t>, grid, dim3(kDbThreads), 0, ctx->stream, reinterpret_cast<const uint16_t *>(so), src->stride,
                       reinterpret_cast<uint16_t *>(dor), dst->stride, src->width, src->height, src->border, d_edge_params, units_stride, sharpness,
                       src->bit_depth);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_plane_sse(aomhip_ctx *ctx, const aomhip_planes *a, int a_frame, const aomhip_planes *b, int b_frame, uint64_t *d_sse) {
  if (!ctx || !a || !b || !a->base || !b->base || !d_sse || a_frame < 0 || a_frame >= a->n_frames || b_frame < 0 || b_frame >= b->n_frames ||
      a->width != b->width || a->height != b->height || a->bit_depth != b->bit_depth) {
    set_error("aomhip_plane_sse: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = a->bit_depth == 8 ? 1 : 2;
  const char *pa = static_cast<const char *>(a->base) + ((size_t)a_frame * a->frame_stride + (size_t)a->border * a->stride + a->border) * esz;
  const char *pb = static_cast<const char *>(b->base) + ((size_t)b_frame * b->frame_stride + (size_t)b->border * b->stride + b->border) * esz;
  if (hipMemsetAsync(d_sse, 0, sizeof(uint64_t), ctx->stream) != hipSuccess) {
    set_error("aomhip_plane_sse: memset failed");
    return AOMHIP_ERR_HIP;
  }
  const dim3 grid((a->height + 3) / 4);   // a wavefront per row
  if (esz == 1)
    hipLaunchKernelGGL(plane_sse_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const uint8_t *>(pa), a->stride,
                       reinterpret_cast<const uint8_t *>(pb), b->stride, a->width, a->height, reinterpret_cast<unsigned long long *>(d_sse));
  else
    hipLaunchKernelGGL(plane_sse_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const uint16_t *>(pa), a->stride,
                       reinterpret_cast<const uint16_t *>(pb), b->stride, a->width, a->height, reinterpret_cast<unsigned long long *>(d_sse));
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

int aomhip_lpf_search_sse(aomhip_ctx *ctx, const aomhip_planes *recon, int recon_frame, const aomhip_planes *scratch, int scratch_frame,
                          const aomhip_planes *source, int source_frame, const uint8_t *d_edge_params, int64_t trial_stride, int n_trials,
                          int units_stride, int sharpness, int passes, uint64_t *d_sse) {
  if (!ctx || !recon || !scratch || !source || !recon->base || !scratch->base || !source->base || !d_edge_params || n_trials <= 0 || !d_sse ||
      recon_frame < 0 || recon_frame >= recon->n_frames || scratch_frame < 0 || scratch_frame >= scratch->n_frames || source_frame < 0 ||
      source_frame >= source->n_frames || recon->width != scratch->width || recon->height != scratch->height ||
      recon->stride != scratch->stride || recon->border != scratch->border || recon->frame_stride != scratch->frame_stride ||
      recon->bit_depth != scratch->bit_depth || recon->width != source->width || recon->height != source->height ||
      recon->bit_depth != source->bit_depth || (recon->base == scratch->base && recon_frame == scratch_frame) ||
      trial_stride < (int64_t)units_stride * ((recon->height + 3) / 4) * 4) {
    set_error("aomhip_lpf_search_sse: invalid argument (scratch must have the reconstruction's geometry and be another frame)");
    return AOMHIP_ERR_INVALID;
  }
  const size_t esz = recon->bit_depth == 8 ? 1 : 2;
  const char *rframe = static_cast<const char *>(recon->base) + (size_t)recon_frame * recon->frame_stride * esz;
  char *sframe = static_cast<char *>(scratch->base) + (size_t)scratch_frame * scratch->frame_stride * esz;
  const char *sorigin = sframe + ((size_t)scratch->border * scratch->stride + scratch->border) * esz;
  const char *oorigin = static_cast<const char *>(source->base) +
                        ((size_t)source_frame * source->frame_stride + (size_t)source->border * source->stride + source->border) * esz;
  if (hipMemsetAsync(d_sse, 0, sizeof(uint64_t) * (size_t)n_trials, ctx->stream) != hipSuccess) {
    set_error("aomhip_lpf_search_sse: memset failed");
    return AOMHIP_ERR_HIP;
  }
  const dim3 grid((recon->height + 3) / 4);   // (plane_sse_kernel: a wavefront per row)
  for (int t = 0; t < n_trials; ++t) {
    // try_filter_frame (av1/encoder/picklpf.c:49-86): filter a copy, measure, and leave the unfiltered frame as it was
    if (hipMemcpyAsync(sframe, rframe, (size_t)recon->frame_stride * esz, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
      set_error("aomhip_lpf_search_sse: frame copy failed");
      return AOMHIP_ERR_HIP;
    }
    const int rc = aomhip_deblock_plane(ctx, scratch, scratch_frame, d_edge_params + (size_t)t * trial_stride, units_stride, sharpness, passes);
    if (rc != AOMHIP_OK) return rc;
    if (esz == 1)
      hipLaunchKernelGGL(plane_sse_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const uint8_t *>(sorigin), scratch->stride,
                         reinterpret_cast<const uint8_t *>(oorigin), source->stride, recon->width, recon->height,
                         reinterpret_cast<unsigned long long *>(d_sse + t));
    else
      hipLaunchKernelGGL(plane_sse_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const uint16_t *>(sorigin), scratch->stride,
                         reinterpret_cast<const uint16_t *>(oorigin), source->stride, recon->width, recon->height,
                         reinterpret_cast<unsigned long long *>(d_sse + t));
    AOMHIP_LAUNCH_CHECK();
  }
  return AOMHIP_OK;
}


// aom_lpf_* / aom_highbd_lpf_* (aom_dsp/aom_dsp_rtcd_defs.pl:474-594) on host pointers: the pixels the reference function
// touches (reach = 2 / 3 / 4 / 7 each side of the edge, `count` pixels along it) are staged, filtered by the plane
// kernels' own tap code (lpf_window_thr) and copied back.  is_hbd: `s` is a uint16_t plane.  count 4 (single), 8 (dual:
// the second four pixels use blimit1 / limit1 / thresh1) or 16 (quad: one threshold set).
void aomhip_lpf_any(void *s, int pitch, int horizontal, int len, int count, const uint8_t *blimit0, const uint8_t *limit0,
                    const uint8_t *thresh0, const uint8_t *blimit1, const uint8_t *limit1, const uint8_t *thresh1, int bd,
                    int is_hbd) {
  aomhip_ctx *ctx = default_ctx();
  if (!ctx) return;
  if ((len != 4 && len != 6 && len != 8 && len != 14) || (count != 4 && count != 8 && count != 16) || (bd != 8 && bd != 10 && bd != 12)) {
    set_error("aomhip_lpf: unsupported length %d / count %d / bit depth %d", len, count, bd);
    return note_failure("aomhip_lpf", AOMHIP_ERR_INVALID);
  }
  const int reach = len == 14 ? 7 : (len == 8 ? 4 : (len == 6 ? 3 : 2));
  const size_t esz = is_hbd ? 2 : 1, bytes = (size_t)2 * reach * count * esz;
  char *h = static_cast<char *>(pinned(ctx, bytes)), *d = static_cast<char *>(scratch(ctx, bytes));
  if (!h || !d) return note_failure("aomhip_lpf scratch", AOMHIP_ERR_NOMEM);
  char *base = static_cast<char *>(s);
  // horizontal edge: 2*reach rows of `count` pixels starting at row -reach; vertical: `count` rows of 2*reach pixels from column -reach
  const int rows = horizontal ? 2 * reach : count, cols = horizontal ? count : 2 * reach;
  char *origin = base + (horizontal ? -(ptrdiff_t)reach * pitch : -(ptrdiff_t)reach) * (ptrdiff_t)esz;
  for (int r = 0; r < rows; ++r) memcpy(h + (size_t)r * cols * esz, origin + (ptrdiff_t)r * pitch * (ptrdiff_t)esz, (size_t)cols * esz);
  if (hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return note_failure("aomhip_lpf H2D");
  const int l1 = limit1 ? *limit1 : *limit0, b1 = blimit1 ? *blimit1 : *blimit0, t1 = thresh1 ? *thresh1 : *thresh0;
  const int second_at = blimit1 ? 4 : count;
  if (is_hbd)
    hipLaunchKernelGGL(lpf_edge_kernel<uint16_t>, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<uint16_t *>(d), horizontal, len, count,
                       second_at, (int)*limit0, (int)*blimit0, (int)*thresh0, l1, b1, t1, bd);
  else
    hipLaunchKernelGGL(lpf_edge_kernel<uint8_t>, dim3(1), dim3(64), 0, ctx->stream, reinterpret_cast<uint8_t *>(d), horizontal, len, count,
                       second_at, (int)*limit0, (int)*blimit0, (int)*thresh0, l1, b1, t1, 8);
  if (hipGetLastError() != hipSuccess) return note_failure("aomhip_lpf launch");
  if (hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
    return note_failure("aomhip_lpf D2H");
  for (int r = 0; r < rows; ++r) memcpy(origin + (ptrdiff_t)r * pitch * (ptrdiff_t)esz, h + (size_t)r * cols * esz, (size_t)cols * esz);
}

}  // extern "C"
