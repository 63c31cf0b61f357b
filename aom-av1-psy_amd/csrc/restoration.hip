// libaomhip -- the Wiener-filter statistics of the loop-restoration search on gfx950: av1_compute_stats /
// av1_compute_stats_highbd (av1/encoder/pickrst.c:948-1083, find_average pickrst.h:32-56) for a list of
// restoration units in one launch.
//
// Per unit: avg = floor(mean of the degraded pixels), then over the unit's pixels M[k] = sum Y[k] X and
// H[k][l] = sum Y[k] Y[l] with Y = the win x win window of (degraded - avg) (k = column offset major, row offset
// minor) and X = source - avg.  These are exact integer sums, so the order of summation is free.
//
// Mapping (second version; the first one -- a thread per (k, l) entry reading both factors from LDS for every term --
// ran at 2 % of the integer rate, profiles/r01_wiener_stats.md): a WAVEFRONT owns one pair of window COLUMNS (dxa <= dxb;
// 28 pairs for win 7, plus win tasks that pair a column with X for M), i.e. a win x win block of H, and keeps its
// win^2 sums in registers.  Its LANES are 64 adjacent pixel columns; each lane walks down the rows of a 16-row band
// with the two columns' last `win` values in registers (one LDS read per column per pixel, conflict-free because
// adjacent lanes read adjacent elements) and issues win^2 v_mad_i32_i24 per pixel.  32-bit band sums (4095^2 x 16
// rows x 4 column chunks < 2^31) are folded into 64-bit totals per band, and the 64 lanes are reduced once per unit.
// A workgroup = 4 wavefronts = 4 tasks of one unit; ceil(tasks / 4) workgroups share a unit and each stages the
// band's (pixel - avg) tile for itself.
// The 8-bit function's down-sampled mode (every 4th row weighted by 4, the last one by what is left,
// pickrst.c:988-1013) is a per-row weight here.
#include "common.h"

namespace aomhip {

constexpr int kBandRows = 16;
constexpr int kMaxUnit = 256;                      // restoration units are at most 256 pixels wide (RESTORATION_UNITSIZE_MAX)
constexpr int kTileW = kMaxUnit + 6 + 2;           // + the window margin, padded to an even count

template <typename T, int WIN>
__global__ __launch_bounds__(256) void wiener_stats_kernel(const T *__restrict__ dgd, int dgd_stride, const T *__restrict__ src, int src_stride,
                                                           const aomhip_rect *__restrict__ units, int downsample, int divider,
                                                           int64_t *__restrict__ M_out, int64_t *__restrict__ H_out) {
  constexpr int HALF = WIN / 2, WIN2 = WIN * WIN;
  constexpr int NP = WIN * (WIN + 1) / 2;          // column pairs dxa <= dxb
  constexpr int NT = NP + WIN;                     // + the M tasks (column dxa against X)
  constexpr int TG = (NT + 3) / 4;                 // workgroups per unit
  __shared__ int16_t ytile[(kBandRows + 2 * HALF) * kTileW];
  __shared__ int16_t xtile[kBandRows * kTileW];
  __shared__ unsigned long long red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int unit = blockIdx.x / TG, task = (blockIdx.x % TG) * 4 + wave;
  const aomhip_rect u = units[unit];
  const int uw = u.h_end - u.h_start, uh = u.v_end - u.v_start;
  if (uw <= 0 || uh <= 0 || uw > kMaxUnit) return;  // (checked on the host when the caller passes the list there too)

  // find_average: floor(sum / count)
  unsigned long long s = 0;
  for (int i = tid; i < uw * uh; i += 256) {
    const int r = i / uw, c = i - r * uw;
    s += dgd[(int64_t)(u.v_start + r) * dgd_stride + u.h_start + c];
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const int avg = (int)((red[0] + red[1] + red[2] + red[3]) / (unsigned long long)(uw * uh));

  // this wavefront's task: columns (dxa, dxb) of the window, or (dxa, X)
  const bool live = task < NT, is_m = task >= NP;
  int dxa = 0, dxb = 0;
  if (live && !is_m) {
    int rem = task;
    while (rem >= WIN - dxa) {
      rem -= WIN - dxa;
      ++dxa;
    }
    dxb = dxa + rem;
  } else if (live) {
    dxa = task - NP;
  }
  int64_t tot[WIN2];
#pragma unroll
  for (int i = 0; i < WIN2; ++i) tot[i] = 0;

  for (int band = 0; band < uh; band += kBandRows) {
    const int rows = min(kBandRows, uh - band);
    __syncthreads();
    // stage (pixel - avg): rows band - HALF .. band + rows - 1 + HALF, columns -HALF .. uw - 1 + HALF
    for (int i = tid; i < (rows + 2 * HALF) * (uw + 2 * HALF); i += 256) {
      const int r = i / (uw + 2 * HALF), c = i - r * (uw + 2 * HALF);
      ytile[r * kTileW + c] = (int16_t)((int)dgd[(int64_t)(u.v_start + band + r - HALF) * dgd_stride + u.h_start + c - HALF] - avg);
    }
    for (int i = tid; i < rows * uw; i += 256) {
      const int r = i / uw, c = i - r * uw;
      xtile[r * kTileW + c] = (int16_t)((int)src[(int64_t)(u.v_start + band + r) * src_stride + u.h_start + c] - avg);
    }
    __syncthreads();
    if (!live) continue;
    int part[WIN2];
#pragma unroll
    for (int i = 0; i < WIN2; ++i) part[i] = 0;
    for (int j = lane; j < uw; j += 64) {
      // window entry (dx, dy) of pixel (r, j) is tile element (r + dy, j + dx)
      const int16_t *ca = ytile + j + dxa, *cb = ytile + j + dxb;
      // the two columns' last WIN values live in ROTATING slots: tile row t sits in slot t % WIN, so the row loop is
      // unrolled by WIN and every register index below is static (no per-row shifting of the windows)
      int wa[WIN], wb[WIN];
#pragma unroll
      for (int i = 0; i < WIN - 1; ++i) {
        wa[i] = ca[i * kTileW];
        wb[i] = cb[i * kTileW];
      }
      for (int r0 = 0; r0 < rows; r0 += WIN) {
#pragma unroll
        for (int q = 0; q < WIN; ++q) {
          const int r = r0 + q;
          if (r < rows) {
            wa[(q + WIN - 1) % WIN] = ca[(r + WIN - 1) * kTileW];
            wb[(q + WIN - 1) % WIN] = cb[(r + WIN - 1) * kTileW];
            int weight = 1;
            bool use = true;
            if (downsample) {
              const int row = band + r;
              use = (row & 3) == 0;
              weight = min(4, uh - row);
            }
            if (use) {
              if (is_m) {
                const int xw = (int)xtile[r * kTileW + j] * weight;
#pragma unroll
                for (int ya = 0; ya < WIN; ++ya) part[ya] += __mul24(wa[(q + ya) % WIN], xw);
              } else {
#pragma unroll
                for (int ya = 0; ya < WIN; ++ya) {
                  const int aw = downsample ? wa[(q + ya) % WIN] * weight : wa[(q + ya) % WIN];  // <= 255 * 4
#pragma unroll
                  for (int yb = 0; yb < WIN; ++yb) part[ya * WIN + yb] += __mul24(aw, wb[(q + yb) % WIN]);
                }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WIN2; ++i) tot[i] += part[i];
  }
  if (!live) return;
  int64_t *M = M_out + (int64_t)unit * WIN2;
  int64_t *H = H_out + (int64_t)unit * WIN2 * WIN2;
#pragma unroll
  for (int i = 0; i < WIN2; ++i) {
    int64_t v = tot[i];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor((long long)v, m, 64);
    if (lane == 0 && !(is_m && i >= WIN)) {
      v /= divider;  // bit_depth_divider, C division (pickrst.c:1041-1045,1073-1081)
      if (is_m) {
        M[dxa * WIN + i] = v;
      } else {
        const int k = dxa * WIN + i / WIN, l = dxb * WIN + i % WIN;   // window index = column offset major (pickrst.c:957-963)
        H[k * WIN2 + l] = v;
        H[l * WIN2 + k] = v;
      }
    }
  }
}

template <typename T>
static void launch_stats(hipStream_t st, int win, int n_units, const void *d, int dstride, const void *s, int sstride, const aomhip_rect *units,
                         int downsample, int divider, int64_t *M, int64_t *H) {
  if (win == 7)
    hipLaunchKernelGGL((wiener_stats_kernel<T, 7>), dim3(n_units * 9), dim3(256), 0, st, static_cast<const T *>(d), dstride,
                       static_cast<const T *>(s), sstride, units, downsample, divider, M, H);
  else
    hipLaunchKernelGGL((wiener_stats_kernel<T, 5>), dim3(n_units * 5), dim3(256), 0, st, static_cast<const T *>(d), dstride,
                       static_cast<const T *>(s), sstride, units, downsample, divider, M, H);
}


// ---- the self-guided restoration filter: av1_selfguided_restoration (av1/common/restoration.c:871-915; calculate_intermediate_result :672-764,
// selfguided_restoration_fast_internal :766-823 for r[0] = 2 on every other row, selfguided_restoration_internal :825-869 for r[1] = 1; AV1 spec
// 7.17.3) -- apply_sgr of the encoder's search_sgrproj runs it per restoration unit and parameter set.  The reference's running box sums are
// truncated only where they are never used (tests/test_golden_sgr.py checks that property), so a unit's output does not depend on how it is cut: one 256-lane workgroup
// per 32 x 32 tile of a unit (tile rows start on even rows of the unit: the r[0] filter's row parity), the tile's 38 x 38 footprint staged in LDS
// once for both filters, A[] / B[] of the 34 x 34 positions the tile's outputs read in LDS, then the weighted 3 x 3 sums.
#include "sgr_table.inc"
__device__ const int kSgrParams[16][4] = AOMHIP_SGR_PARAMS;
__device__ const int32_t kXByXplus1[256] = AOMHIP_X_BY_XPLUS1;
__device__ const int32_t kOneByX[25] = AOMHIP_ONE_BY_X;

constexpr int kSgrTile = 32, kSgrFoot = kSgrTile + 6, kSgrAB = kSgrTile + 2;

// A unit as the kernels use it: clipped to the plane and to the stated maximum size.  Identity for every unit the entry points accept -- they can
// only check the optional HOST copy of the list -- and what keeps a bad device-side rectangle from writing past the caller's flt0 / flt1 rows or
// outside the destination plane (it then filters the clipped rectangle).
__device__ __forceinline__ aomhip_rect clip_unit(aomhip_rect u, int plane_w, int plane_h, int max_w, int max_h) {
  u.h_start = min(max(u.h_start, 0), plane_w); u.v_start = min(max(u.v_start, 0), plane_h);
  u.h_end = min(min(u.h_end, plane_w), u.h_start + max_w); u.v_end = min(min(u.v_end, plane_h), u.v_start + max_h);
  return u;
}

template <typename T>
__global__ __launch_bounds__(256) void selfguided_kernel(const T *__restrict__ dgd, int dgd_stride, const aomhip_rect *__restrict__ units,
                                                          const int32_t *__restrict__ sgr_idx, int bit_depth, int32_t *__restrict__ flt0,
                                                          int32_t *__restrict__ flt1, int flt_stride, int64_t flt_pitch, int tiles_x, int plane_w,
                                                          int plane_h, int max_w, int max_h) {
  __shared__ int32_t s_d[kSgrFoot * kSgrFoot];
  __shared__ int32_t s_A[kSgrAB * kSgrAB], s_B[kSgrAB * kSgrAB];
  const int ui = blockIdx.x;
  const aomhip_rect u = clip_unit(units[ui], plane_w, plane_h, max_w, max_h);
  const int w = u.h_end - u.h_start, h = u.v_end - u.v_start;
  const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
  const int ox = tx * kSgrTile, oy = ty * kSgrTile;   // the tile inside the unit
  if (ox >= w || oy >= h) return;
  const int idx = sgr_idx[ui];
  // footprint: unit coordinates ox - 3 .. ox + 34, clamped to the unit's own 3-pixel surround (what lies beyond cannot reach an output)
  for (int t = threadIdx.x; t < kSgrFoot * kSgrFoot; t += 256) {
    const int fy = t / kSgrFoot, fx = t - fy * kSgrFoot;
    const int y = min(max(oy + fy - 3, -3), h + 2), x = min(max(ox + fx - 3, -3), w + 2);
    s_d[t] = (int32_t)dgd[(int64_t)(u.v_start + y) * dgd_stride + u.h_start + x];
  }
  __syncthreads();
  const int sh = bit_depth - 8;
  for (int pass = 0; pass < 2; ++pass) {
    const int r = kSgrParams[idx][pass], sv = kSgrParams[idx][2 + pass];
    if (r <= 0) continue;   // (uniform)
    int32_t *dst = (pass ? flt1 : flt0) + (int64_t)ui * flt_pitch;
    const int n = (2 * r + 1) * (2 * r + 1);
    // A, B at tile positions (i, j) in [-1, 32]: s_A[(i + 1) * 34 + j + 1]; the r[0] filter needs the odd rows only
    for (int t = threadIdx.x; t < kSgrAB * kSgrAB; t += 256) {
      const int ai = t / kSgrAB, aj = t - ai * kSgrAB;   // position (ai - 1, aj - 1)
      if (pass == 0 && (ai & 1)) continue;                // (ai - 1 even: not computed, not read)
      uint32_t sum = 0, sq = 0;
      for (int y = -r; y <= r; ++y)
        for (int x = -r; x <= r; ++x) {
          const uint32_t v = (uint32_t)s_d[(ai + 2 + y) * kSgrFoot + (aj + 2 + x)];   // footprint row of position i is i + 3 = ai + 2
          sum += v; sq += v * v;
        }
      const uint32_t a = (sq + ((1u << (2 * sh)) >> 1)) >> (2 * sh), b = (sum + ((1u << sh) >> 1)) >> sh;
      const uint32_t p = (a * n < b * b) ? 0u : a * n - b * b;
      const uint32_t z = (p * (uint32_t)sv + (1u << 19)) >> 20;                        // SGRPROJ_MTABLE_BITS
      const int32_t A = kXByXplus1[z < 255u ? z : 255u];
      s_A[t] = A;
      s_B[t] = (int32_t)(((uint32_t)(256 - A) * sum * (uint32_t)kOneByX[n - 1] + (1u << 11)) >> 12);   // SGRPROJ_SGR, SGRPROJ_RECIP_BITS
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kSgrTile * kSgrTile; t += 256) {
      const int i = t >> 5, j = t & 31;
      if (oy + i >= h || ox + j >= w) continue;
      const int k = (i + 1) * kSgrAB + (j + 1);
      int32_t a, b;
      int nb;
      if (pass == 0) {
        if (!(i & 1)) {   // (the tile starts on an even row of the unit)
          nb = 5;
          a = (s_A[k - kSgrAB] + s_A[k + kSgrAB]) * 6 + (s_A[k - 1 - kSgrAB] + s_A[k - 1 + kSgrAB] + s_A[k + 1 - kSgrAB] + s_A[k + 1 + kSgrAB]) * 5;
          b = (s_B[k - kSgrAB] + s_B[k + kSgrAB]) * 6 + (s_B[k - 1 - kSgrAB] + s_B[k - 1 + kSgrAB] + s_B[k + 1 - kSgrAB] + s_B[k + 1 + kSgrAB]) * 5;
        } else {
          nb = 4;
          a = s_A[k] * 6 + (s_A[k - 1] + s_A[k + 1]) * 5;
          b = s_B[k] * 6 + (s_B[k - 1] + s_B[k + 1]) * 5;
        }
      } else {
        nb = 5;
        a = (s_A[k] + s_A[k - 1] + s_A[k + 1] + s_A[k - kSgrAB] + s_A[k + kSgrAB]) * 4 +
            (s_A[k - 1 - kSgrAB] + s_A[k - 1 + kSgrAB] + s_A[k + 1 - kSgrAB] + s_A[k + 1 + kSgrAB]) * 3;
        b = (s_B[k] + s_B[k - 1] + s_B[k + 1] + s_B[k - kSgrAB] + s_B[k + kSgrAB]) * 4 +
            (s_B[k - 1 - kSgrAB] + s_B[k - 1 + kSgrAB] + s_B[k + 1 - kSgrAB] + s_B[k + 1 + kSgrAB]) * 3;
      }
      const int32_t v = a * s_d[(i + 3) * kSgrFoot + (j + 3)] + b;
      const int rs = 8 + nb - 4;   // SGRPROJ_SGR_BITS + nb - SGRPROJ_RST_BITS
      dst[(int64_t)(oy + i) * flt_stride + ox + j] = (v + ((1 << rs) >> 1)) >> rs;
    }
    __syncthreads();
  }
}

// av1_apply_selfguided_restoration (restoration.c:917-956) after selfguided_kernel: per pixel u = dat << 4, v = (u << 7) + xq0 (flt0 - u) + xq1 (flt1 - u)
// over the radii in use, (xq0, xq1) = av1_decode_xq(xqd) (:631-643), w = (int16_t)ROUND_POWER_OF_TWO(v, 11), clipped.  One lane per pixel.
template <typename T>
__global__ __launch_bounds__(256) void selfguided_apply_kernel(const T *__restrict__ dat, int dat_stride, T *__restrict__ dst, int dst_stride,
                                                                const aomhip_rect *__restrict__ units, const int32_t *__restrict__ sgr_idx,
                                                                const int32_t *__restrict__ xqd, int bit_depth, const int32_t *__restrict__ flt0,
                                                                const int32_t *__restrict__ flt1, int flt_stride, int64_t flt_pitch, int plane_w, int plane_h,
                                                                int max_w, int max_h) {
  const int ui = blockIdx.x;
  const aomhip_rect u = clip_unit(units[ui], plane_w, plane_h, max_w, max_h);
  const int w = u.h_end - u.h_start, h = u.v_end - u.v_start;
  const int idx = sgr_idx[ui];
  const int r0 = kSgrParams[idx][0], r1 = kSgrParams[idx][1];
  int xq0, xq1;
  if (r0 == 0) { xq0 = 0; xq1 = 128 - xqd[2 * ui + 1]; }
  else if (r1 == 0) { xq0 = xqd[2 * ui]; xq1 = 0; }
  else { xq0 = xqd[2 * ui]; xq1 = 128 - xq0 - xqd[2 * ui + 1]; }
  const int32_t *f0 = flt0 + (int64_t)ui * flt_pitch, *f1 = flt1 + (int64_t)ui * flt_pitch;
  const int mx = (1 << bit_depth) - 1;
  for (int t = blockIdx.y * 256 + threadIdx.x; t < w * h; t += gridDim.y * 256) {
    const int i = t / w, j = t - i * w;
    const int uu = (int)dat[(int64_t)(u.v_start + i) * dat_stride + u.h_start + j] << 4;
    int v = uu << 7;
    if (r0 > 0) v += xq0 * (f0[(int64_t)i * flt_stride + j] - uu);
    if (r1 > 0) v += xq1 * (f1[(int64_t)i * flt_stride + j] - uu);
    const int wv = (int)(int16_t)((v + (1 << 10)) >> 11);
    dst[(int64_t)(u.v_start + i) * dst_stride + u.h_start + j] = (T)min(max(wv, 0), mx);
  }
}

// av1_[highbd_]wiener_convolve_add_src (av1/common/convolve.c:1093-1257) as wiener_filter_stripe calls it (steps of 16): the unit's two 7-tap
// filters (8 stored taps, centre reduced by 128: the passes add the source back).  One workgroup per 32 x 32 tile: footprint rows -3 .. +4 and
// columns -3 .. +4 staged in LDS, horizontal pass (+ src << 7 + offset, rounded by round_0, clamped to WIENER_CLAMP_LIMIT) into 40 x 32 uint16,
// vertical pass (offset removed, rounded by round_1, clipped).  round_0 / round_1 = get_conv_params_wiener(bd): 3 / 11, at 12 bits 5 / 9.
constexpr int kWienerTile = 32, kWienerFoot = kWienerTile + 8;
template <typename T>
__global__ __launch_bounds__(256) void wiener_kernel(const T *__restrict__ dat, int dat_stride, T *__restrict__ dst, int dst_stride,
                                                      const aomhip_rect *__restrict__ units, const int16_t *__restrict__ filters, int bd, int tiles_x, int plane_w,
                                                      int plane_h, int max_w, int max_h) {
  __shared__ uint16_t s_src[kWienerFoot * kWienerFoot];
  __shared__ uint16_t s_tmp[kWienerFoot * kWienerTile];
  const int ui = blockIdx.x;
  const aomhip_rect u = clip_unit(units[ui], plane_w, plane_h, max_w, max_h);
  const int w = u.h_end - u.h_start, h = u.v_end - u.v_start;
  const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
  const int ox = tx * kWienerTile, oy = ty * kWienerTile;
  if (ox >= w || oy >= h) return;
  int fx[8], fy[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { fx[k] = filters[16 * ui + k]; fy[k] = filters[16 * ui + 8 + k]; }
  for (int t = threadIdx.x; t < kWienerFoot * kWienerFoot; t += 256) {   // unit coordinates ox - 3 .. ox + 36, clamped to the unit's 3 (4) pixel surround
    const int sy = t / kWienerFoot, sx = t - sy * kWienerFoot;
    const int y = min(max(oy + sy - 3, -3), h + 2), x = min(max(ox + sx - 3, -3), w + 2);   // (what tap 7 reads is multiplied by 0: any staged value will do)
    s_src[t] = (uint16_t)dat[(int64_t)(u.v_start + y) * dat_stride + u.h_start + x];
  }
  __syncthreads();
  const int round_0 = bd == 12 ? 5 : 3, round_1 = 14 - round_0;
  const int limit = (1 << (bd + 8 - round_0)) - 1;
  for (int t = threadIdx.x; t < kWienerFoot * kWienerTile; t += 256) {   // temp row r = footprint row r, columns of the tile
    const int r = t >> 5, x = t & 31;
    const uint16_t *p = s_src + r * kWienerFoot + x;   // footprint column x = source column x - 3
    int sum = ((int)p[3] << 7) + (1 << (bd + 6));
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)p[k] * fx[k];
    s_tmp[t] = (uint16_t)min(max((sum + ((1 << round_0) >> 1)) >> round_0, 0), limit);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kWienerTile * kWienerTile; t += 256) {
    const int y = t >> 5, x = t & 31;
    if (oy + y >= h || ox + x >= w) continue;
    const uint16_t *p = s_tmp + y * kWienerTile + x;   // temp row y = source row y - 3
    int sum = ((int)p[3 * kWienerTile] << 7) - (1 << (bd + round_1 - 1));
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += (int)p[k * kWienerTile] * fy[k];
    const int v = (sum + ((1 << round_1) >> 1)) >> round_1;
    dst[(int64_t)(u.v_start + oy + y) * dst_stride + u.h_start + ox + x] = (T)min(max(v, 0), (1 << bd) - 1);
  }
}

// ---- the self-guided filter's projection statistics: av1_calc_proj_params[_high_bd] (av1/encoder/pickrst.c:470-657: H[2][2], C[2] of
// get_proj_subspace) and av1_[lowbd|highbd]_pixel_proj_error (:226-370: get_pixel_proj_error, evaluated once per xq that finer_search tries).
// One 256-lane workgroup per (unit [, xq]): lanes stride the unit's pixels row-major, the sums are exact 64-bit integers (products of
// two ~17-bit values, up to 2^16 pixels), reduced by shuffles and one LDS step.  Streaming: 2 pixels + 2 int32 per pixel.
__device__ __forceinline__ long long wg_sum_i64(long long v, long long *scratch /* 4 */) {
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

template <typename T>
__global__ __launch_bounds__(256) void proj_params_kernel(const T *__restrict__ src, int src_stride, const T *__restrict__ dat, int dat_stride,
                                                           const aomhip_rect *__restrict__ units, const int32_t *__restrict__ flt0, const int32_t *__restrict__ flt1,
                                                           int flt_stride, int64_t flt_pitch, const int32_t *__restrict__ radii, int64_t *__restrict__ H,
                                                           int64_t *__restrict__ C) {
  __shared__ long long scratch[4];
  const int ui = blockIdx.x;
  const aomhip_rect u = units[ui];
  const int w = u.h_end - u.h_start, h = u.v_end - u.v_start, n = w * h;
  const int r0 = radii[2 * ui], r1 = radii[2 * ui + 1];
  const int32_t *f0 = flt0 + (int64_t)ui * flt_pitch, *f1 = flt1 + (int64_t)ui * flt_pitch;
  long long h00 = 0, h01 = 0, h11 = 0, c0 = 0, c1 = 0;
  for (int t = threadIdx.x; t < n; t += 256) {
    const int i = t / w, j = t - i * w;
    const int64_t po = (int64_t)(u.v_start + i) * dat_stride + u.h_start + j, so = (int64_t)(u.v_start + i) * src_stride + u.h_start + j;
    const int uu = (int)dat[po] << 4;                    // SGRPROJ_RST_BITS
    const int sv = ((int)src[so] << 4) - uu;
    const int a = r0 > 0 ? f0[(int64_t)i * flt_stride + j] - uu : 0, b = r1 > 0 ? f1[(int64_t)i * flt_stride + j] - uu : 0;
    h00 += (long long)a * a; h11 += (long long)b * b; h01 += (long long)a * b;
    c0 += (long long)a * sv; c1 += (long long)b * sv;
  }
  h00 = wg_sum_i64(h00, scratch); h01 = wg_sum_i64(h01, scratch); h11 = wg_sum_i64(h11, scratch);
  c0 = wg_sum_i64(c0, scratch); c1 = wg_sum_i64(c1, scratch);
  if (threadIdx.x == 0) {
    const long long size = n;   // (C's integer division: towards zero, as `H[0][0] /= size` in the reference)
    int64_t *Ho = H + 4 * (int64_t)ui, *Co = C + 2 * (int64_t)ui;
    Ho[0] = r0 > 0 ? h00 / size : 0; Ho[3] = r1 > 0 ? h11 / size : 0;
    Ho[1] = Ho[2] = (r0 > 0 && r1 > 0) ? h01 / size : 0;
    Co[0] = r0 > 0 ? c0 / size : 0; Co[1] = r1 > 0 ? c1 / size : 0;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void proj_error_kernel(const T *__restrict__ src, int src_stride, const T *__restrict__ dat, int dat_stride,
                                                          const aomhip_rect *__restrict__ units, const int32_t *__restrict__ flt0, const int32_t *__restrict__ flt1,
                                                          int flt_stride, int64_t flt_pitch, const int32_t *__restrict__ radii, const int32_t *__restrict__ xq,
                                                          int n_xq, int64_t *__restrict__ err_out) {
  __shared__ long long scratch[4];
  const int ui = blockIdx.x, qi = blockIdx.y;
  const aomhip_rect u = units[ui];
  const int w = u.h_end - u.h_start, h = u.v_end - u.v_start, n = w * h;
  const int r0 = radii[2 * ui], r1 = radii[2 * ui + 1];
  const int xq0 = xq[2 * ((int64_t)ui * n_xq + qi)], xq1 = xq[2 * ((int64_t)ui * n_xq + qi) + 1];
  const int32_t *f0 = flt0 + (int64_t)ui * flt_pitch, *f1 = flt1 + (int64_t)ui * flt_pitch;
  long long err = 0;
  for (int t = threadIdx.x; t < n; t += 256) {
    const int i = t / w, j = t - i * w;
    const int d = (int)dat[(int64_t)(u.v_start + i) * dat_stride + u.h_start + j], sv = (int)src[(int64_t)(u.v_start + i) * src_stride + u.h_start + j];
    const int uu = d << 4;
    int v = 1 << 10;                                     // half of 1 << (SGRPROJ_RST_BITS + SGRPROJ_PRJ_BITS)
    if (r0 > 0) v += xq0 * (f0[(int64_t)i * flt_stride + j] - uu);
    if (r1 > 0) v += xq1 * (f1[(int64_t)i * flt_stride + j] - uu);
    const int e = ((r0 > 0 || r1 > 0) ? (v >> 11) : 0) + d - sv;
    err += (long long)e * e;
  }
  err = wg_sum_i64(err, scratch);
  if (threadIdx.x == 0) err_out[(int64_t)ui * n_xq + qi] = err;
}

}  // namespace aomhip

using namespace aomhip;

extern "C" int aomhip_compute_stats_batch(aomhip_ctx *ctx, const aomhip_planes *dgd, int dgd_frame, const aomhip_planes *src, int src_frame,
                                          int wiener_win, const aomhip_rect *d_units, const aomhip_rect *h_units, int n_units,
                                          int use_downsampled_wiener_stats, int64_t *d_M, int64_t *d_H) {
  if (!ctx || !dgd || !src || !dgd->base || !src->base || (wiener_win != 7 && wiener_win != 5) || (n_units > 0 && !d_units) ||
      n_units < 0 || !d_M || !d_H || dgd_frame < 0 || dgd_frame >= dgd->n_frames || src_frame < 0 || src_frame >= src->n_frames ||
      dgd->width != src->width || dgd->height != src->height || dgd->bit_depth != src->bit_depth || dgd->border < 3 ||
      (use_downsampled_wiener_stats && dgd->bit_depth != 8)) {
    set_error("aomhip_compute_stats_batch: invalid argument (window 7 or 5; border >= 3; the down-sampled mode exists for 8-bit only)");
    return AOMHIP_ERR_INVALID;
  }
  for (int i = 0; h_units && i < n_units; ++i) {
    const aomhip_rect &r = h_units[i];
    if (r.h_start < 0 || r.v_start < 0 || r.h_end > dgd->width || r.v_end > dgd->height || r.h_end <= r.h_start || r.v_end <= r.v_start ||
        r.h_end - r.h_start > kMaxUnit) {
      set_error("aomhip_compute_stats_batch: unit %d is empty, outside the plane or wider than %d", i, kMaxUnit);
      return AOMHIP_ERR_INVALID;
    }
  }
  if (n_units == 0) return AOMHIP_OK;
  const size_t esz = dgd->bit_depth == 8 ? 1 : 2;
  const char *d = static_cast<const char *>(dgd->base) +
                  ((size_t)dgd_frame * dgd->frame_stride + (size_t)dgd->border * dgd->stride + dgd->border) * esz;
  const char *s = static_cast<const char *>(src->base) +
                  ((size_t)src_frame * src->frame_stride + (size_t)src->border * src->stride + src->border) * esz;
  const int divider = dgd->bit_depth == 12 ? 16 : dgd->bit_depth == 10 ? 4 : 1;
  if (esz == 1)
    launch_stats<uint8_t>(ctx->stream, wiener_win, n_units, d, dgd->stride, s, src->stride, d_units, use_downsampled_wiener_stats != 0, divider, d_M, d_H);
  else
    launch_stats<uint16_t>(ctx->stream, wiener_win, n_units, d, dgd->stride, s, src->stride, d_units, 0, divider, d_M, d_H);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

static int check_proj(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *dat, int dat_frame, const aomhip_rect *d_units, int n_units,
                      const int32_t *d_flt0, const int32_t *d_flt1, int flt_stride, int64_t flt_pitch, const int32_t *d_radii, const char *who) {
  if (!ctx || !src || !dat || !src->base || !dat->base || n_units < 0 || (n_units > 0 && (!d_units || !d_flt0 || !d_flt1 || !d_radii)) || src_frame < 0 ||
      src_frame >= src->n_frames || dat_frame < 0 || dat_frame >= dat->n_frames || src->bit_depth != dat->bit_depth || flt_stride <= 0 || flt_pitch <= 0) {
    set_error("%s: invalid argument", who);
    return AOMHIP_ERR_INVALID;
  }
  return AOMHIP_OK;
}

extern "C" int aomhip_calc_proj_params_batch(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *dat, int dat_frame,
                                             const aomhip_rect *d_units, int n_units, const int32_t *d_flt0, const int32_t *d_flt1, int flt_stride, int64_t flt_pitch,
                                             const int32_t *d_radii, int64_t *d_H, int64_t *d_C) {
  int rc = check_proj(ctx, src, src_frame, dat, dat_frame, d_units, n_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, "aomhip_calc_proj_params_batch");
  if (rc != AOMHIP_OK) return rc;
  if (n_units > 0 && (!d_H || !d_C)) { set_error("aomhip_calc_proj_params_batch: invalid argument"); return AOMHIP_ERR_INVALID; }
  if (n_units == 0) return AOMHIP_OK;
  const int64_t so = (int64_t)src_frame * src->frame_stride + (int64_t)src->border * src->stride + src->border;
  const int64_t po = (int64_t)dat_frame * dat->frame_stride + (int64_t)dat->border * dat->stride + dat->border;
  if (src->bit_depth == 8)
    hipLaunchKernelGGL(proj_params_kernel<uint8_t>, dim3(n_units), dim3(256), 0, ctx->stream, static_cast<const uint8_t *>(src->base) + so, src->stride,
                       static_cast<const uint8_t *>(dat->base) + po, dat->stride, d_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, d_H, d_C);
  else
    hipLaunchKernelGGL(proj_params_kernel<uint16_t>, dim3(n_units), dim3(256), 0, ctx->stream, static_cast<const uint16_t *>(src->base) + so, src->stride,
                       static_cast<const uint16_t *>(dat->base) + po, dat->stride, d_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, d_H, d_C);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_pixel_proj_error_batch(aomhip_ctx *ctx, const aomhip_planes *src, int src_frame, const aomhip_planes *dat, int dat_frame,
                                             const aomhip_rect *d_units, int n_units, const int32_t *d_flt0, const int32_t *d_flt1, int flt_stride, int64_t flt_pitch,
                                             const int32_t *d_radii, const int32_t *d_xq, int n_xq, int64_t *d_err) {
  int rc = check_proj(ctx, src, src_frame, dat, dat_frame, d_units, n_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, "aomhip_pixel_proj_error_batch");
  if (rc != AOMHIP_OK) return rc;
  if (n_xq < 0 || n_xq > 65535 || (n_units > 0 && n_xq > 0 && (!d_xq || !d_err))) { set_error("aomhip_pixel_proj_error_batch: invalid argument"); return AOMHIP_ERR_INVALID; }
  if (n_units == 0 || n_xq == 0) return AOMHIP_OK;
  const int64_t so = (int64_t)src_frame * src->frame_stride + (int64_t)src->border * src->stride + src->border;
  const int64_t po = (int64_t)dat_frame * dat->frame_stride + (int64_t)dat->border * dat->stride + dat->border;
  if (src->bit_depth == 8)
    hipLaunchKernelGGL(proj_error_kernel<uint8_t>, dim3(n_units, n_xq), dim3(256), 0, ctx->stream, static_cast<const uint8_t *>(src->base) + so, src->stride,
                       static_cast<const uint8_t *>(dat->base) + po, dat->stride, d_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, d_xq, n_xq, d_err);
  else
    hipLaunchKernelGGL(proj_error_kernel<uint16_t>, dim3(n_units, n_xq), dim3(256), 0, ctx->stream, static_cast<const uint16_t *>(src->base) + so, src->stride,
                       static_cast<const uint16_t *>(dat->base) + po, dat->stride, d_units, d_flt0, d_flt1, flt_stride, flt_pitch, d_radii, d_xq, n_xq, d_err);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_selfguided_restoration_batch(aomhip_ctx *ctx, const aomhip_planes *dgd, int dgd_frame, const aomhip_rect *d_units, const aomhip_rect *h_units,
                                                   int n_units, const int32_t *d_sgr_params_idx, int max_unit_width, int max_unit_height, int32_t *d_flt0,
                                                   int32_t *d_flt1, int flt_stride, int64_t flt_pitch) {
  if (!ctx || !dgd || !dgd->base || n_units < 0 || (n_units > 0 && (!d_units || !d_sgr_params_idx || !d_flt0 || !d_flt1)) || dgd_frame < 0 ||
      dgd_frame >= dgd->n_frames || dgd->border < 3 || max_unit_width < 1 || max_unit_height < 1 || flt_stride < max_unit_width ||
      flt_pitch < (int64_t)flt_stride * max_unit_height) {
    set_error("aomhip_selfguided_restoration_batch: invalid argument (border >= 3; flt_stride / flt_pitch hold the largest unit)");
    return AOMHIP_ERR_INVALID;
  }
  for (int i = 0; h_units && i < n_units; ++i) {
    const aomhip_rect &r = h_units[i];
    if (r.h_start < 0 || r.v_start < 0 || r.h_end > dgd->width || r.v_end > dgd->height || r.h_end <= r.h_start || r.v_end <= r.v_start ||
        r.h_end - r.h_start > max_unit_width || r.v_end - r.v_start > max_unit_height) {
      set_error("aomhip_selfguided_restoration_batch: unit %d is empty, outside the plane or larger than the stated maximum", i);
      return AOMHIP_ERR_INVALID;
    }
  }
  if (n_units == 0) return AOMHIP_OK;
  const int tiles_x = (max_unit_width + kSgrTile - 1) / kSgrTile, tiles_y = (max_unit_height + kSgrTile - 1) / kSgrTile;
  const int64_t po = (int64_t)dgd_frame * dgd->frame_stride + (int64_t)dgd->border * dgd->stride + dgd->border;
  const dim3 grid((unsigned)n_units, (unsigned)(tiles_x * tiles_y));
  if (dgd->bit_depth == 8)
    hipLaunchKernelGGL(selfguided_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, static_cast<const uint8_t *>(dgd->base) + po, dgd->stride, d_units,
                       d_sgr_params_idx, 8, d_flt0, d_flt1, flt_stride, flt_pitch, tiles_x, dgd->width, dgd->height, max_unit_width, max_unit_height);
  else
    hipLaunchKernelGGL(selfguided_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, static_cast<const uint16_t *>(dgd->base) + po, dgd->stride, d_units,
                       d_sgr_params_idx, dgd->bit_depth, d_flt0, d_flt1, flt_stride, flt_pitch, tiles_x, dgd->width, dgd->height, max_unit_width, max_unit_height);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

static int check_lr_units(const aomhip_planes *dat, const aomhip_rect *h_units, int n_units, int max_w, int max_h, const char *who) {
  for (int i = 0; h_units && i < n_units; ++i) {
    const aomhip_rect &r = h_units[i];
    if (r.h_start < 0 || r.v_start < 0 || r.h_end > dat->width || r.v_end > dat->height || r.h_end <= r.h_start || r.v_end <= r.v_start ||
        r.h_end - r.h_start > max_w || r.v_end - r.v_start > max_h) {
      set_error("%s: unit %d is empty, outside the plane or larger than the stated maximum", who, i);
      return AOMHIP_ERR_INVALID;
    }
  }
  return AOMHIP_OK;
}

extern "C" int aomhip_apply_selfguided_restoration_batch(aomhip_ctx *ctx, const aomhip_planes *dat, int dat_frame, const aomhip_planes *dst, int dst_frame,
                                                         const aomhip_rect *d_units, const aomhip_rect *h_units, int n_units, const int32_t *d_sgr_params_idx,
                                                         const int32_t *d_xqd, int max_unit_width, int max_unit_height, int32_t *d_flt0, int32_t *d_flt1,
                                                         int flt_stride, int64_t flt_pitch) {
  if (!dst || !dst->base || !dat || dst_frame < 0 || dst_frame >= dst->n_frames || dst->bit_depth != dat->bit_depth || dst->width != dat->width ||
      dst->height != dat->height || (n_units > 0 && !d_xqd)) {
    set_error("aomhip_apply_selfguided_restoration_batch: invalid argument");
    return AOMHIP_ERR_INVALID;
  }
  int rc = aomhip_selfguided_restoration_batch(ctx, dat, dat_frame, d_units, h_units, n_units, d_sgr_params_idx, max_unit_width, max_unit_height, d_flt0, d_flt1,
                                               flt_stride, flt_pitch);
  if (rc != AOMHIP_OK || n_units == 0) return rc;
  const int64_t po = (int64_t)dat_frame * dat->frame_stride + (int64_t)dat->border * dat->stride + dat->border;
  const int64_t qo = (int64_t)dst_frame * dst->frame_stride + (int64_t)dst->border * dst->stride + dst->border;
  const int chunks = (max_unit_width * max_unit_height + 256 * 16 - 1) / (256 * 16);
  const dim3 grid((unsigned)n_units, (unsigned)(chunks < 1 ? 1 : chunks));
  if (dat->bit_depth == 8)
    hipLaunchKernelGGL(selfguided_apply_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, static_cast<const uint8_t *>(dat->base) + po, dat->stride,
                       static_cast<uint8_t *>(dst->base) + qo, dst->stride, d_units, d_sgr_params_idx, d_xqd, 8, d_flt0, d_flt1, flt_stride, flt_pitch, dat->width, dat->height, max_unit_width, max_unit_height);
  else
    hipLaunchKernelGGL(selfguided_apply_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, static_cast<const uint16_t *>(dat->base) + po, dat->stride,
                       static_cast<uint16_t *>(dst->base) + qo, dst->stride, d_units, d_sgr_params_idx, d_xqd, dat->bit_depth, d_flt0, d_flt1, flt_stride,
                       flt_pitch, dat->width, dat->height, max_unit_width, max_unit_height);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}

extern "C" int aomhip_wiener_convolve_add_src_batch(aomhip_ctx *ctx, const aomhip_planes *dat, int dat_frame, const aomhip_planes *dst, int dst_frame,
                                                    const aomhip_rect *d_units, const aomhip_rect *h_units, int n_units, const int16_t *d_filters,
                                                    int max_unit_width, int max_unit_height) {
  if (!ctx || !dat || !dat->base || !dst || !dst->base || n_units < 0 || (n_units > 0 && (!d_units || !d_filters)) || dat_frame < 0 ||
      dat_frame >= dat->n_frames || dst_frame < 0 || dst_frame >= dst->n_frames || dst->bit_depth != dat->bit_depth || dst->width != dat->width ||
      dst->height != dat->height || dat->border < 3 || max_unit_width < 1 || max_unit_height < 1) {
    set_error("aomhip_wiener_convolve_add_src_batch: invalid argument (border >= 3)");
    return AOMHIP_ERR_INVALID;
  }
  if (int rc = check_lr_units(dat, h_units, n_units, max_unit_width, max_unit_height, "aomhip_wiener_convolve_add_src_batch")) return rc;
  if (n_units == 0) return AOMHIP_OK;
  const int tiles_x = (max_unit_width + kWienerTile - 1) / kWienerTile, tiles_y = (max_unit_height + kWienerTile - 1) / kWienerTile;
  const int64_t po = (int64_t)dat_frame * dat->frame_stride + (int64_t)dat->border * dat->stride + dat->border;
  const int64_t qo = (int64_t)dst_frame * dst->frame_stride + (int64_t)dst->border * dst->stride + dst->border;
  const dim3 grid((unsigned)n_units, (unsigned)(tiles_x * tiles_y));
  if (dat->bit_depth == 8)
    hipLaunchKernelGGL(wiener_kernel<uint8_t>, grid, dim3(256), 0, ctx->stream, static_cast<const uint8_t *>(dat->base) + po, dat->stride,
                       static_cast<uint8_t *>(dst->base) + qo, dst->stride, d_units, d_filters, 8, tiles_x, dat->width, dat->height, max_unit_width, max_unit_height);
  else
    hipLaunchKernelGGL(wiener_kernel<uint16_t>, grid, dim3(256), 0, ctx->stream, static_cast<const uint16_t *>(dat->base) + po, dat->stride,
                       static_cast<uint16_t *>(dst->base) + qo, dst->stride, d_units, d_filters, dat->bit_depth, tiles_x, dat->width, dat->height, max_unit_width, max_unit_height);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
