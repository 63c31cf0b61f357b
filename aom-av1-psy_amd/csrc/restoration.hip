// libaomhip -- the Wiener-filter statistics of the loop-restoration search on gfx950: av1_compute_stats /
// av1_compute_stats_highbd (av1/encoder/pickrst.c:948-1083, find_average pickrst.h:32-56) for a list of
// restoration units in one launch.
//
// Per unit: avg = floor(mean of the degraded pixels), then over the unit's pixels M[k] = sum Y[k] X and
// H[k][l] = sum Y[k] Y[l] with Y = the win x win window of (degraded - avg) (k = column offset major, row offset
// minor) and X = source - avg.  These are exact integer sums, so the order of summation is free: one workgroup owns
// one unit, every thread owns up to five (k, l) entries of the upper triangle (1225 + 49 M entries for win 7) and
// keeps their 64-bit sums in registers, and the unit streams through LDS in bands of 16 rows as (pixel - avg) int16
// tiles -- each staged pixel is reused by ~1274 multiply-adds.  Inner loop: two LDS reads + one v_mad_i32_i24 per
// term, 32-bit partial sums over 64-pixel runs (4095^2 * 64 < 2^31) folded into the 64-bit totals.
// The 8-bit function's down-sampled mode (every 4th row weighted by 4, the last one by what is left,
// pickrst.c:988-1013) is a per-row weight here.
#include "common.h"

namespace aomhip {

constexpr int kBandRows = 16;
constexpr int kMaxUnit = 256;                      // restoration units are at most 256 pixels wide (RESTORATION_UNITSIZE_MAX)
constexpr int kTileW = kMaxUnit + 6 + 2;           // + the window margin, padded to an even count
constexpr int kMaxTerms = 5;                       // ceil((1225 + 49) / 256)

template <typename T>
__global__ __launch_bounds__(256) void wiener_stats_kernel(const T *__restrict__ dgd, int dgd_stride, const T *__restrict__ src, int src_stride,
                                                           const aomhip_rect *__restrict__ units, int win, int downsample, int divider,
                                                           int64_t *__restrict__ M_out, int64_t *__restrict__ H_out) {
  __shared__ int16_t ytile[(kBandRows + 6) * kTileW];
  __shared__ int16_t xtile[kBandRows * kTileW];
  __shared__ unsigned long long red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const aomhip_rect u = units[blockIdx.x];
  const int uw = u.h_end - u.h_start, uh = u.v_end - u.v_start;
  if (uw <= 0 || uh <= 0 || uw > kMaxUnit) return;  // (checked on the host when the caller passes the list there too)
  const int half = win >> 1, win2 = win * win, nH = win2 * (win2 + 1) / 2;

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

  // this thread's terms: entry p of [upper triangle of H, row-major | M]
  int oa[kMaxTerms], ob[kMaxTerms], kk[kMaxTerms], ll[kMaxTerms];
  int64_t acc[kMaxTerms];
  int n_terms = 0;
  for (int p = tid; p < nH + win2; p += 256) {
    int k, l;
    if (p < nH) {
      k = 0;
      int rem = p;
      while (rem >= win2 - k) {
        rem -= win2 - k;
        ++k;
      }
      l = k + rem;
    } else {
      k = p - nH;
      l = -1;  // M entry: the second factor is X
    }
    kk[n_terms] = k;
    ll[n_terms] = l;
    // window index -> (column offset, row offset): idx = (dx + half) * win + (dy + half) (pickrst.c:957-963)
    oa[n_terms] = (k % win) * kTileW + k / win;
    ob[n_terms] = l < 0 ? half : (l % win) * kTileW + l / win;   // X tile: same column origin as the window centre
    acc[n_terms] = 0;
    ++n_terms;
  }

  for (int band = 0; band < uh; band += kBandRows) {
    const int rows = min(kBandRows, uh - band);
    __syncthreads();
    // stage (pixel - avg): window rows band - half .. band + rows - 1 + half, columns -half .. uw - 1 + half
    for (int i = tid; i < (rows + 2 * half) * (uw + 2 * half); i += 256) {
      const int r = i / (uw + 2 * half), c = i - r * (uw + 2 * half);
      ytile[r * kTileW + c] = (int16_t)((int)dgd[(int64_t)(u.v_start + band + r - half) * dgd_stride + u.h_start + c - half] - avg);
    }
    for (int i = tid; i < rows * uw; i += 256) {
      const int r = i / uw, c = i - r * uw;
      xtile[r * kTileW + c + half] = (int16_t)((int)src[(int64_t)(u.v_start + band + r) * src_stride + u.h_start + c] - avg);
    }
    __syncthreads();
    for (int t = 0; t < n_terms; ++t) {
      const bool is_m = ll[t] < 0;
      for (int r = 0; r < rows; ++r) {
        const int row = band + r;
        int weight = 1;
        if (downsample) {
          if (row & 3) continue;
          weight = min(4, uh - row);
        }
        const int16_t *pa = ytile + r * kTileW + oa[t];
        const int16_t *pb = is_m ? xtile + r * kTileW + ob[t] - half : ytile + r * kTileW + ob[t];
        // (M: X of pixel j sits at column j + half of its tile row, the window entry k of pixel j at column j + k / win)
        int64_t rowsum = 0;
        for (int j0 = 0; j0 < uw; j0 += 64) {
          const int je = min(uw, j0 + 64);
          int part = 0;
          for (int j = j0; j < je; ++j) part += __mul24((int)pa[j], (int)(is_m ? pb[j + half] : pb[j]));
          rowsum += part;
        }
        acc[t] += rowsum * weight;
      }
    }
  }
  int64_t *M = M_out + (int64_t)blockIdx.x * win2;
  int64_t *H = H_out + (int64_t)blockIdx.x * win2 * win2;
  for (int t = 0; t < n_terms; ++t) {
    const int64_t v = acc[t] / divider;  // bit_depth_divider, C division (pickrst.c:1041-1045,1073-1081)
    if (ll[t] < 0) {
      M[kk[t]] = v;
    } else {
      H[kk[t] * win2 + ll[t]] = v;
      H[ll[t] * win2 + kk[t]] = v;
    }
  }
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
    hipLaunchKernelGGL(wiener_stats_kernel<uint8_t>, dim3(n_units), dim3(256), 0, ctx->stream, reinterpret_cast<const uint8_t *>(d), dgd->stride,
                       reinterpret_cast<const uint8_t *>(s), src->stride, d_units, wiener_win, use_downsampled_wiener_stats != 0, divider, d_M,
                       d_H);
  else
    hipLaunchKernelGGL(wiener_stats_kernel<uint16_t>, dim3(n_units), dim3(256), 0, ctx->stream, reinterpret_cast<const uint16_t *>(d), dgd->stride,
                       reinterpret_cast<const uint16_t *>(s), src->stride, d_units, wiener_win, 0, divider, d_M, d_H);
  AOMHIP_LAUNCH_CHECK();
  return AOMHIP_OK;
}
